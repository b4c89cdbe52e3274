"""`deepchopper predict` pipeline on one GPU: FASTQ bytes -> tokenise -> forward -> labels / intervals.

Reference call stack: cli.predict (deepchopper/cli.py:66-152) -> OnlyFqDataModule
(deepchopper/data/only_fq.py:204-239) -> tokenizer / collator (deepchopper/models/llm/tokenizer.py)
-> TokenClassificationLit.predict_step (basic_module.py:197-207) -> CustomWriter (callbacks.py:12-25),
then `chop` smooths the per-base labels into intervals (src/bin/predict.rs:130-192).

Here every batch stays on the GPU between the three stages; the host only parses FASTQ text and
receives labels / intervals (and, when asked for, the reference's `.pt` wire format).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

from . import ops
from .tokenizer import DEFAULT_MAX_LENGTH, check_read_status, encode_ids_int8


@dataclass
class SmoothParams:
    """Defaults of `deepchopper chop` (cli.py:158-164, src/bin/predict.rs:39-64)."""

    window: int = 21
    min_interval: int = 13
    approved: int = 20
    max_process_intervals: int = 4
    min_read_length_after_chop: int = 20
    min_read_len: int = 150


class PredictPipeline:
    """tokenise -> forward -> smooth on device-resident batches."""

    def __init__(self, model, max_length: int = DEFAULT_MAX_LENGTH, smooth: SmoothParams | None = None):
        self.model = model
        self.max_length = max_length
        self.smooth = smooth or SmoothParams()

    # ---- stage 1 ---------------------------------------------------------------------
    def tokenize(self, seq: torch.Tensor, qual: torch.Tensor, offsets: torch.Tensor, max_len_bases: int,
                 *, pad_left: bool = True, ids_i64: bool = False, want_labels: bool = True):
        row_stride = min(max_len_bases, self.max_length - 1) + 1
        return ops.tokenize_batch(seq, qual, offsets, max_length=self.max_length, row_stride=row_stride,
                                  pad_left=pad_left, ids_i64=ids_i64, want_labels=want_labels)

    # ---- whole hot path on one device batch ---------------------------------------------
    def run_device_batch(self, seq: torch.Tensor, qual: torch.Tensor, offsets: torch.Tensor, lens_host: np.ndarray,
                         *, reference_padding: bool = False, want_labels: bool = True, ids_i64: bool = False):
        """seq/qual: concatenated raw bytes on the device; offsets int64[n+1] on the device;
        lens_host: read lengths (host copy, known from parsing).

        reference_padding=False (native): every read is run at its own length (== the reference
        at --batch-size 1).  reference_padding=True: the batch is left-padded to its longest read
        and the pad tokens run through the model exactly as in the reference."""
        n = int(lens_host.shape[0])
        max_len = int(lens_host.max())
        tok = self.tokenize(seq, qual, offsets, max_len, pad_left=True, ids_i64=ids_i64, want_labels=want_labels)
        row_stride = tok["input_ids"].shape[1]
        ntok = np.minimum(lens_host, self.max_length - 1).astype(np.int32) + 1
        first = (row_stride - ntok).astype(np.int32)
        if reference_padding or bool(np.all(ntok == row_stride)):
            logits = self.model.net.forward(tok["input_ids"], tok["input_quals"])
        else:
            logits = self.model.net.forward_ragged(tok["input_ids"], tok["input_quals"], first, ntok)
        d_first = ops.to_device(first)
        d_len = ops.to_device((ntok - 1).astype(np.int32))
        out_offsets = None
        if want_labels:
            oo = np.zeros(n + 1, np.int64)
            np.cumsum(ntok - 1, out=oo[1:])
            out_offsets = ops.to_device(oo)
        sp = self.smooth
        sm = ops.smooth_intervals_from_logits(logits, d_first, d_len, window=sp.window, min_interval=sp.min_interval,
                                              approved=sp.approved, want_labels=want_labels, out_offsets=out_offsets)
        return {"tok": tok, "logits": logits, "smooth": sm, "first": first, "n_tokens": ntok}

    # ---- host-facing convenience -----------------------------------------------------------
    def predict_reads(self, ids: list[str], seqs: list[bytes], quals: list[bytes], *, reference_padding: bool = False):
        """Returns one dict per read: id, labels (int8), smoothed labels, intervals, keep segments."""
        for i, s, q in zip(ids, seqs, quals):
            if not i or not s or not q:
                raise ValueError("Incomplete FASTQ record")  # only_fq.py:38-40
            if len(s) != len(q):
                raise ValueError(f"Sequence/quality length mismatch in record '{i}': seq={len(s)}, qual={len(q)}")
        sb, offs = ops.pack_reads(seqs)
        qb, _ = ops.pack_reads(quals)
        lens = np.diff(offs)
        d_seq, d_qual, d_offs = ops.to_device(sb), ops.to_device(qb), ops.to_device(offs)
        res = self.run_device_batch(d_seq, d_qual, d_offs, lens, reference_padding=reference_padding)
        torch.cuda.synchronize()
        net = self.model.net
        if net.gemm_mode == "fp16x3" and net.overflow() > 0:
            # an activation left the fp16 range: redo the batch with the bf16 split (fp32 exponent range)
            net.set_gemm_mode("bf16x6")
            try:
                res = self.run_device_batch(d_seq, d_qual, d_offs, lens, reference_padding=reference_padding)
                torch.cuda.synchronize()
            finally:
                net.set_gemm_mode("fp16x3")
        check_read_status(res["tok"]["status"].cpu().numpy(), ids)
        sm = res["smooth"]
        o = sm["offsets"].cpu().numpy()
        pred, smooth = sm["pred"].cpu().numpy(), sm["smooth"].cpu().numpy()
        iv, niv = sm["intervals"].cpu().numpy(), sm["n_intervals"].cpu().numpy()
        kp, nkp = sm["keep"].cpu().numpy(), sm["n_keep"].cpu().numpy()
        out = []
        for r, rid in enumerate(ids):
            out.append({
                "id": rid,
                "prediction": pred[o[r]: o[r + 1]],
                "smooth": smooth[o[r]: o[r + 1]],
                "intervals": [tuple(x) for x in iv[r, : niv[r]].tolist()],
                "keep": [tuple(x) for x in kp[r, : nkp[r]].tolist()],
                "is_truncated": bool(res["tok"]["status"][r].item() & 1),
            })
        return out, res


    # ---- streaming: FASTQ chunks -> labels / intervals, host I/O overlapped with GPU work ---------------
    def _collect(self, names, lens, seq_h, qual_h, offs_h, d_in, res):
        """D2H of one finished batch (blocks until the GPU is done with it)."""
        net = self.model.net
        torch.cuda.current_stream().synchronize()
        if net.gemm_mode == "fp16x3" and net.overflow() > 0:
            net.set_gemm_mode("bf16x6")  # an activation left the fp16 range: redo with the fp32-range split
            try:
                res = self.run_device_batch(*d_in, lens)
                torch.cuda.current_stream().synchronize()
            finally:
                net.set_gemm_mode("fp16x3")
        status = res["tok"]["status"].cpu().numpy()
        check_read_status(status, names)
        sm = res["smooth"]
        return {
            "names": names, "lengths": lens, "seq": seq_h, "qual": qual_h, "in_offsets": offs_h,
            "offsets": sm["offsets"].cpu().numpy(), "pred": sm["pred"].cpu().numpy(), "smooth": sm["smooth"].cpu().numpy(),
            "intervals": sm["intervals"].cpu().numpy(), "n_intervals": sm["n_intervals"].cpu().numpy(),
            "keep": sm["keep"].cpu().numpy(), "n_keep": sm["n_keep"].cpu().numpy(), "truncated": (status & 1) != 0,
            "device": res,  # device-resident tensors of the batch (logits, token matrices) for write_pt_batch
        }

    def predict_stream(self, chunks, *, prefetch: int = 2):
        """`chunks`: an iterator of fastq.FastqChunk (e.g. `FastqReader.chunks(max_reads, max_tokens)`).
        Yields one result dict per chunk, in input order: names, lengths, packed per-base labels
        (`pred`, `smooth` + `offsets`), `intervals` / `n_intervals`, `keep` / `n_keep`, `truncated`.

        Three things overlap: a helper thread parses (and inflates) the next chunks; the GPU runs batch k
        (all launches are asynchronous); the host collects and hands out batch k-1.  Every read runs at
        its own length (native mode, == the reference at --batch-size 1)."""
        import queue
        import threading

        q: queue.Queue = queue.Queue(maxsize=max(1, prefetch))
        stop = threading.Event()

        def produce():
            try:
                for c in chunks:
                    item = (c.names(), c.lengths.copy(), c.seq.copy(), c.qual.copy(), c.offsets.copy())
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:  # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        pending = None
        try:
            while True:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                if item is not None:
                    names, lens, seq_h, qual_h, offs_h = item
                    d_in = (ops.to_device(seq_h), ops.to_device(qual_h), ops.to_device(offs_h))
                    launched = (names, lens, seq_h, qual_h, offs_h, d_in, self.run_device_batch(*d_in, lens))
                else:
                    launched = None
                if pending is not None:
                    yield self._collect(*pending)
                pending = launched
                if item is None:
                    break
        finally:
            stop.set()
            th.join(timeout=5)


def write_pt_batch(path: str, res: dict, ids: list[str]) -> None:
    """CustomWriter.write_on_batch_end (callbacks.py:12-25): the reference's predict -> chop wire
    format, `{prediction f32[B,L,2], target i64[B,L], seq i64[B,L], qual f32[B,L], id i64[B,256]}`."""
    tok = res["tok"]
    truncated = (tok["status"].cpu().numpy() & 1) != 0
    idm = np.stack([encode_ids_int8(i, bool(t)) for i, t in zip(ids, truncated)]).astype(np.int64)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({
        "prediction": res["logits"].cpu(),
        "target": tok["labels"].cpu().to(torch.int64),
        "seq": tok["input_ids"].cpu().to(torch.int64),
        "qual": tok["input_quals"].cpu(),
        "id": torch.from_numpy(idm),
    }, path)
