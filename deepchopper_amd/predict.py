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
import queue
import threading
import time
from dataclasses import dataclass

import numpy as np
import torch

from . import ops
from .fastq import decode_names
from .tokenizer import DEFAULT_MAX_LENGTH, PAD_TOKEN_ID, check_read_status, encode_ids_int8


@dataclass
class SmoothParams:
    """Defaults of `deepchopper chop` (cli.py:158-164, src/bin/predict.rs:39-64)."""

    window: int = 21
    min_interval: int = 13
    approved: int = 20
    max_process_intervals: int = 4
    min_read_length_after_chop: int = 20
    min_read_len: int = 150


class _Pinned:
    """Page-locked host buffers that grow on demand, one per key: H2D / D2H copies from / into them are
    asynchronous, so no copy of the pipeline blocks the host on GPU work."""

    def __init__(self):
        self._buf: dict[str, torch.Tensor] = {}

    def get(self, key: str, nbytes: int) -> torch.Tensor:
        b = self._buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8, pin_memory=True)
            self._buf[key] = b
        return b[:nbytes]

    def array(self, key: str, shape, dtype) -> np.ndarray:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.get(key, n).numpy().view(dtype).reshape(shape)


class _Slot:
    """Host side of one in-flight batch: pinned staging for its inputs and its results."""

    def __init__(self):
        self.pin = _Pinned()
        self.n = 0


def wait_event(ev) -> None:
    """Wait for a CUDA/HIP event WITHOUT burning a core.  `event.synchronize()` spins on this stack -- with
    `Event(blocking=True)` as well: measured 100 % of one core in the wait for the whole run, plus a runtime thread beside it
    (tools/spin_probe.py, profiles/r06_spin_probe.jsonl) -- which at one rank per GPU is two cores per rank spent on waiting,
    more than the rank's parse + staging + BGZF work together.  Polling `query()` between short sleeps costs a few hundred
    wake-ups per batch of tens of milliseconds and at most half a millisecond of latency.  DC_EVENT_WAIT=sync keeps
    synchronize() (A/B)."""
    if _EVENT_WAIT_SYNC:
        ev.synchronize()
        return
    pause = 50e-6
    while not ev.query():
        time.sleep(pause)
        if pause < 500e-6:
            pause *= 2


_EVENT_WAIT_SYNC = os.environ.get("DC_EVENT_WAIT", "poll") == "sync"
_COPY_STREAM = os.environ.get("DC_COPY_STREAM", "0") == "1"


class PredictPipeline:
    """tokenise -> forward -> smooth on device-resident batches."""

    def __init__(self, model, max_length: int = DEFAULT_MAX_LENGTH, smooth: SmoothParams | None = None):
        self.model = model
        self.max_length = max_length
        self.smooth = smooth or SmoothParams()
        self._copy_stream = None
        # main-thread seconds of predict_stream: enqueueing a batch (wall / CPU) and waiting for a batch's results (wall / CPU --
        # a wait that SLEEPS costs no CPU; tools/spin_probe.py)
        self.stream_timers = {"launch_s": 0.0, "launch_cpu_s": 0.0, "wait_s": 0.0, "wait_cpu_s": 0.0}
        # reference_padding: rows share their [PAD] prefix (dc_forward_shared_prefix) instead of running it once per row;
        # DC_SHARED_PREFIX=0 keeps the padded launches (A/B, parity tests)
        self.shared_prefix = os.environ.get("DC_SHARED_PREFIX", "1") != "0"
        # ... in the linear form by default (y = (k * (z - z_pad) + F) * gate: the long convolution covers a read's own tokens,
        # not its batch's longest read; equal to the padded launch to fp32 rounding, tests/test_gpu_forward.py);
        # DC_PREFIX_MODE=exact keeps the form that is bit-identical to the padded launch on the block-Toeplitz kernels
        net = getattr(model, "net", None)
        if net is not None and hasattr(net, "set_prefix_mode"):
            net.set_prefix_mode("exact" if os.environ.get("DC_PREFIX_MODE", "linear") in ("exact", "1") else "linear")

    # ---- stage 1 ---------------------------------------------------------------------
    def tokenize(self, seq: torch.Tensor, qual: torch.Tensor, offsets: torch.Tensor, max_len_bases: int,
                 *, pad_left: bool = True, ids_i64: bool = False, want_labels: bool = True):
        row_stride = min(max_len_bases, self.max_length - 1) + 1
        return ops.tokenize_batch(seq, qual, offsets, max_length=self.max_length, row_stride=row_stride,
                                  pad_left=pad_left, ids_i64=ids_i64, want_labels=want_labels)

    # ---- whole hot path on one device batch ---------------------------------------------
    def run_device_batch(self, seq: torch.Tensor, qual: torch.Tensor, offsets: torch.Tensor, lens_host: np.ndarray,
                         *, reference_padding: bool = False, want_labels: bool = True, ids_i64: bool = False,
                         staging: _Pinned | None = None, sub_sizes: list[int] | None = None):
        """seq/qual: concatenated raw bytes on the device; offsets int64[n+1] on the device;
        lens_host: read lengths (host copy, known from parsing).

        reference_padding=False (native): every read is run at its own length (== the reference
        at --batch-size 1).  reference_padding=True: the batch is left-padded to its longest read
        and the pad tokens run through the model exactly as in the reference.

        sub_sizes (with reference_padding): the batch is SEVERAL reference batches fused into one launch
        (`fastq.fuse_chunks`): rows sub_sizes[0] | sub_sizes[1] | ...; every sub-batch keeps its own left padding -- a
        row of sub-batch i runs through the model with the pad tokens that make it as long as the longest read of
        sub-batch i, exactly what the reference's collator gives it -- so the logits are those of separate launches
        (`res["sub_cols"][i]` = the column its rows start at; `write_pt_batch(..., rows, col0)` cuts a sub-batch out).

        Every launch is asynchronous: the per-read index arrays go up in one copy BEFORE the forward is
        launched (from `staging`, pinned, when given), and nothing is read back here."""
        n = int(lens_host.shape[0])
        max_len = int(lens_host.max())
        row_stride = min(max_len, self.max_length - 1) + 1
        ntok = np.minimum(lens_host, self.max_length - 1).astype(np.int32) + 1
        first = (row_stride - ntok).astype(np.int32)
        # one small H2D for smoothing's per-read arrays: int64 out_offsets[n+1] | int32 first[n] | int32 len[n]
        nb = 8 * (n + 1) + 8 * n
        host = staging.get("aux", nb).numpy() if staging is not None else np.empty(nb, np.uint8)
        oo = host[: 8 * (n + 1)].view(np.int64)
        oo[0] = 0
        np.cumsum(ntok - 1, out=oo[1:])
        total = int(oo[n])
        host[8 * (n + 1): 8 * (n + 1) + 4 * n].view(np.int32)[:] = first
        host[8 * (n + 1) + 4 * n:].view(np.int32)[:] = ntok - 1
        d_aux = torch.empty(nb, dtype=torch.uint8, device=seq.device)
        d_aux.copy_(torch.from_numpy(host) if staging is None else staging.get("aux", nb), non_blocking=staging is not None)
        out_offsets = d_aux[: 8 * (n + 1)].view(torch.int64)
        d_first = d_aux[8 * (n + 1): 8 * (n + 1) + 4 * n].view(torch.int32)
        d_len = d_aux[8 * (n + 1) + 4 * n:].view(torch.int32)

        tok = self.tokenize(seq, qual, offsets, max_len, pad_left=True, ids_i64=ids_i64, want_labels=want_labels)
        assert tok["input_ids"].shape[1] == row_stride
        sub_cols = None
        net = self.model.net
        if reference_padding and not bool(np.all(ntok == row_stride)):
            # tokens of a row as the reference's collator pads it: up to the longest read of ITS (sub-)batch
            sub_tok = np.empty(n, np.int32)
            if sub_sizes is not None and len(sub_sizes) > 1:
                sub_cols = []
                r0 = 0
                for k in sub_sizes:
                    lb = int(ntok[r0: r0 + k].max())
                    sub_tok[r0: r0 + k] = lb
                    sub_cols.append(row_stride - lb)
                    r0 += k
                assert r0 == n, "sub_sizes must add up to the batch"
            else:
                sub_tok[:] = row_stride
            sub_first = (row_stride - sub_tok).astype(np.int32)
            if self.shared_prefix:
                # the pad runs of all rows go through the model once per launch (HyenaDNA is causal): same logits
                logits = net.forward_shared_prefix(tok["input_ids"], tok["input_quals"], sub_first, sub_tok, sub_tok - ntok,
                                                   pad_token_id=PAD_TOKEN_ID)
            else:
                logits = net.forward_ragged(tok["input_ids"], tok["input_quals"], sub_first, sub_tok)
        elif reference_padding or bool(np.all(ntok == row_stride)):
            logits = net.forward(tok["input_ids"], tok["input_quals"])
        else:
            logits = net.forward_ragged(tok["input_ids"], tok["input_quals"], first, ntok)
        sp = self.smooth
        sm = ops.smooth_intervals_from_logits(logits, d_first, d_len, window=sp.window, min_interval=sp.min_interval,
                                              approved=sp.approved, want_labels=want_labels,
                                              out_offsets=out_offsets if want_labels else None, total=total)
        if not want_labels:
            sm["offsets"] = out_offsets
        return {"tok": tok, "logits": logits, "smooth": sm, "first": first, "n_tokens": ntok, "offsets_host": oo.copy(),
                "sub_sizes": sub_sizes, "sub_cols": sub_cols}

    def run_checked(self, seq, qual, offsets, lens_host, **kw):
        """run_device_batch with the fp16x3 range guard consulted (synchronises): a batch in which an activation
        left the fp16 range is re-run with the bf16 split (fp32 exponent range)."""
        net = self.model.net
        if net.gemm_mode != "fp16x3":
            res = self.run_device_batch(seq, qual, offsets, lens_host, **kw)
            torch.cuda.current_stream().synchronize()
            return res
        counter = torch.zeros(1, dtype=torch.int32, device=seq.device)
        net.set_overflow_target(counter)
        try:
            res = self.run_device_batch(seq, qual, offsets, lens_host, **kw)
        finally:
            net.set_overflow_target(None)
        if int(counter.item()) > 0:
            res = self._rerun_bf16x6(seq, qual, offsets, lens_host, **kw)
        return res

    def _rerun_bf16x6(self, seq, qual, offsets, lens_host, **kw):
        net = self.model.net
        net.set_gemm_mode("bf16x6")
        try:
            res = self.run_device_batch(seq, qual, offsets, lens_host, **kw)
            torch.cuda.current_stream().synchronize()
        finally:
            net.set_gemm_mode("fp16x3")
        res["range_guard_rerun"] = True
        return res

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
        res = self.run_checked(d_seq, d_qual, d_offs, lens, reference_padding=reference_padding)
        status = res["tok"]["status"].cpu().numpy()
        check_read_status(status, ids)
        sm = res["smooth"]
        o = res["offsets_host"]
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
                "is_truncated": bool(status[r] & 1),
            })
        return out, res

    # ---- streaming: FASTQ chunks -> labels / intervals, host I/O overlapped with GPU work ---------------
    def predict_stream(self, chunks, *, prefetch: int = 2, want_labels: bool = True, reference_padding: bool = False,
                       ids_i64: bool = False, borrow: bool = False, want_names: bool = True, serialize: bool = False):
        """`chunks`: an iterator of fastq.FastqChunk (e.g. `FastqReader.chunks(max_reads, max_tokens)`).
        Yields one result dict per chunk, in input order: `names`, `lengths`, packed per-base labels
        (`pred`, `smooth` + `offsets`; only with want_labels), `intervals` / `n_intervals`, `keep` / `n_keep`,
        `truncated`, `pred_len`, the chunk's host buffers (`seq`, `qual`, `in_offsets`, `headers`,
        `header_offsets`, `name_len`) and `device` (the batch's device tensors, for `write_pt_batch`).

        Four things overlap: a helper thread parses (and inflates) the next chunks straight into pinned
        staging; the compute stream runs batch k (tokenise, forward, smoothing: all launches asynchronous,
        inputs uploaded by non-blocking copies); a copy stream brings batch k-1's results into pinned host
        memory; the host hands out batch k-2... i.e. the only wait is for a batch's own copy event.
        The fp16x3 range guard is attributed per batch: every batch has its own device counter, read back with
        its results, so a batch launched while an earlier one is collected cannot lose (or lend) its overflow.

        serialize=True (A/B measurements only): wait for every batch before the next one is launched -- the
        behaviour of a pipeline whose per-batch copies block the host.

        borrow=True: the yielded host arrays are views into pinned staging that is reused once the consumer
        asks for the next result (the fused chop driver consumes them at once); otherwise they are copies."""
        net = self.model.net
        dev = torch.device("cuda", torch.cuda.current_device())
        compute = torch.cuda.current_stream()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        copy_stream = self._copy_stream
        n_slots = max(1, prefetch) + 3
        free: queue.Queue = queue.Queue()
        for _ in range(n_slots):
            free.put(_Slot())
        q: queue.Queue = queue.Queue(maxsize=max(1, prefetch))
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(dev)  # pinned allocations of this thread belong to the pipeline's device
                for c in chunks:
                    slot = None
                    while slot is None and not stop.is_set():
                        try:
                            slot = free.get(timeout=0.1)
                        except queue.Empty:
                            continue
                    if stop.is_set():
                        return
                    n, nb = c.n, int(c.offsets[c.n])
                    slot.n = n
                    slot.seq = slot.pin.array("seq", (nb,), np.uint8)
                    slot.qual = slot.pin.array("qual", (nb,), np.uint8)
                    slot.offs = slot.pin.array("offs", (n + 1,), np.int64)
                    slot.seq[:] = c.seq
                    slot.qual[:] = c.qual
                    slot.offs[:] = c.offsets
                    slot.headers = np.array(c._names, copy=True)
                    slot.header_offsets = np.array(c._name_offsets, copy=True)
                    slot.name_len = None if c._name_len is None else np.array(c._name_len, copy=True)
                    slot.lens = np.diff(slot.offs)
                    slot.sub_sizes = getattr(c, "sub_sizes", None)
                    while not stop.is_set():
                        try:
                            q.put(slot, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:  # surfaces in the consumer
                q.put(e)

        def launch(slot: _Slot):
            n, nb = slot.n, int(slot.offs[slot.n])
            d_seq = torch.empty(nb, dtype=torch.uint8, device=dev)
            d_qual = torch.empty(nb, dtype=torch.uint8, device=dev)
            d_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
            d_seq.copy_(slot.pin.get("seq", nb), non_blocking=True)
            d_qual.copy_(slot.pin.get("qual", nb), non_blocking=True)
            d_offs.copy_(slot.pin.get("offs", 8 * (n + 1)).view(torch.int64), non_blocking=True)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            guard = net.gemm_mode == "fp16x3"
            if guard:
                net.set_overflow_target(counter)
            try:
                res = self.run_device_batch(d_seq, d_qual, d_offs, slot.lens, reference_padding=reference_padding,
                                            want_labels=want_labels, ids_i64=ids_i64, staging=slot.pin,
                                            sub_sizes=slot.sub_sizes if reference_padding else None)
            finally:
                if guard:
                    net.set_overflow_target(None)
            # results -> pinned host memory, on the COMPUTE stream behind the batch's own kernels.  (Rounds 1-5 used a second
            # stream behind an event of the compute stream: on this stack a cross-stream event wait is resolved by a runtime
            # thread that busy-waits on the host for as long as the awaited work runs -- one core per rank at ~95 % for the
            # whole run (tools/spin_probe_torch.py: same-stream copy 0.01 s of runtime-thread CPU, cross-stream 0.26 s per
            # 0.8 s; the pipeline 2.6 s of 2.8 s).  The copies are a few MB per launch of tens of milliseconds: in line they
            # cost < 0.5 % of the step, and batch k + 1 is enqueued before the host waits for batch k either way.
            # DC_COPY_STREAM=1 keeps the second stream (A/B).)
            sm = res["smooth"]
            outs = {"status": res["tok"]["status"], "guard": counter, "intervals": sm["intervals"],
                    "n_intervals": sm["n_intervals"], "keep": sm["keep"], "n_keep": sm["n_keep"]}
            if want_labels:
                outs["pred"], outs["smooth"] = sm["pred"], sm["smooth"]
            host = {}
            side = copy_stream if _COPY_STREAM else compute
            if _COPY_STREAM:
                done = torch.cuda.Event()
                done.record(compute)
            with torch.cuda.stream(side):
                if _COPY_STREAM:
                    copy_stream.wait_event(done)
                for k, t in outs.items():
                    hb = slot.pin.get("out_" + k, t.numel() * t.element_size())
                    hb.view(t.dtype).view(t.shape).copy_(t, non_blocking=True) if t.numel() else None
                    host[k] = hb.numpy().view(_NP[t.dtype]).reshape(tuple(t.shape))
                copied = torch.cuda.Event()
                copied.record(side)
            return {"slot": slot, "res": res, "d_in": (d_seq, d_qual, d_offs), "host": host, "copied": copied}

        def collect(b):
            slot, res, host = b["slot"], b["res"], b["host"]
            t_w = time.perf_counter()
            c_w = time.thread_time()
            wait_event(b["copied"])
            self.stream_timers["wait_s"] += time.perf_counter() - t_w
            self.stream_timers["wait_cpu_s"] += time.thread_time() - c_w
            if int(host["guard"][0]) > 0:  # this batch (and only this one) left the fp16 range: redo it in bf16x6
                res = self._rerun_bf16x6(*b["d_in"], slot.lens, reference_padding=reference_padding,
                                         want_labels=want_labels, ids_i64=ids_i64,
                                         sub_sizes=slot.sub_sizes if reference_padding else None)
                sm = res["smooth"]
                host = {"status": res["tok"]["status"].cpu().numpy(), "intervals": sm["intervals"].cpu().numpy(),
                        "n_intervals": sm["n_intervals"].cpu().numpy(), "keep": sm["keep"].cpu().numpy(),
                        "n_keep": sm["n_keep"].cpu().numpy()}
                if want_labels:
                    host["pred"], host["smooth"] = sm["pred"].cpu().numpy(), sm["smooth"].cpu().numpy()
            own = (lambda a: a) if borrow else (lambda a: np.array(a, copy=True))
            status = host["status"]
            names = decode_names(slot.headers, slot.header_offsets, slot.name_len, slot.n) if want_names else None
            if (status & 2).any():
                check_read_status(status, names or decode_names(slot.headers, slot.header_offsets, slot.name_len, slot.n))
            out = {
                "names": names, "lengths": slot.lens, "seq": own(slot.seq), "qual": own(slot.qual), "in_offsets": own(slot.offs),
                "headers": slot.headers, "header_offsets": slot.header_offsets, "name_len": slot.name_len,
                "offsets": res["offsets_host"], "pred_len": (res["n_tokens"] - 1).astype(np.int32),
                "intervals": own(host["intervals"]), "n_intervals": own(host["n_intervals"]),
                "keep": own(host["keep"]), "n_keep": own(host["n_keep"]), "truncated": (status & 1) != 0,
                "range_guard_rerun": bool(res.get("range_guard_rerun", False)),
                "sub_sizes": res.get("sub_sizes"), "sub_cols": res.get("sub_cols"),
                "device": res,  # device-resident tensors of the batch (logits, token matrices) for write_pt_batch
                "ready_event": b["copied"],  # ... final behind this event (recorded on the copy stream after the batch's compute)
            }
            if want_labels:
                out["pred"], out["smooth"] = own(host["pred"]), own(host["smooth"])
            return out

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        pending = None
        try:
            while True:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                t_l, c_l = time.perf_counter(), time.thread_time()
                launched = launch(item) if item is not None else None
                self.stream_timers["launch_s"] += time.perf_counter() - t_l
                self.stream_timers["launch_cpu_s"] += time.thread_time() - c_l
                if serialize and launched is not None:
                    launched["copied"].synchronize()
                if pending is not None:
                    yield collect(pending)
                    free.put(pending["slot"])  # the consumer is done with the borrowed views
                pending = launched
                if item is None:
                    break
        finally:
            stop.set()
            th.join(timeout=5)


_NP = {torch.int8: np.int8, torch.uint8: np.uint8, torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32}


def write_pt_batch(path: str, res: dict, ids: list[str], rows: slice | None = None, col0: int = 0) -> None:
    """CustomWriter.write_on_batch_end (callbacks.py:12-25): the reference's predict -> chop wire
    format, `{prediction f32[B,L,2], target i64[B,L], seq i64[B,L], qual f32[B,L], id i64[B,256]}`.
    `rows` / `col0` cut one reference batch out of a fused launch (`run_device_batch(..., sub_sizes=...)`): its rows, from the
    column where its own left padding starts; `ids` are the names of those rows."""
    tok = res["tok"]
    rows = slice(None) if rows is None else rows
    truncated = (tok["status"][rows].cpu().numpy() & 1) != 0
    idm = np.stack([encode_ids_int8(i, bool(t)) for i, t in zip(ids, truncated)]).astype(np.int64)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({
        "prediction": res["logits"][rows, col0:].contiguous().cpu(),
        "target": tok["labels"][rows, col0:].contiguous().cpu().to(torch.int64),
        "seq": tok["input_ids"][rows, col0:].contiguous().cpu().to(torch.int64),
        "qual": tok["input_quals"][rows, col0:].contiguous().cpu(),
        "id": torch.from_numpy(idm),
    }, path)


def write_pt_batches(dir_path: str, rank: int, first_index: int, res: dict, ids: list[str], pool=None, pending: list | None = None) -> int:
    """Every reference batch of a (possibly fused) launch as its own `{rank}_{index}.pt` (CustomWriter's naming,
    callbacks.py:25).  The device tensors come to the host ONCE per launch; a batch is cut out of the host copies (its
    rows, from the column where its own left padding starts) and copied, so a file holds its own storage only.
    With `pool` (a ThreadPoolExecutor) the pickling / writing of the files runs on its threads (futures appended to
    `pending`; the caller waits for them and bounds how many are in flight).  Returns the number of files."""
    tok = res["tok"]
    subs = res.get("sub_sizes") or [len(ids)]
    cols = res.get("sub_cols") or [0] * len(subs)
    host = {"prediction": res["logits"].cpu(), "target": tok["labels"].cpu(), "seq": tok["input_ids"].cpu(),
            "qual": tok["input_quals"].cpu()}
    truncated = (tok["status"].cpu().numpy() & 1) != 0
    os.makedirs(dir_path, exist_ok=True)

    def save(i: int, r0: int, k: int, c0: int) -> None:
        idm = np.stack([encode_ids_int8(n, bool(t)) for n, t in zip(ids[r0: r0 + k], truncated[r0: r0 + k])]).astype(np.int64)
        torch.save({
            "prediction": host["prediction"][r0: r0 + k, c0:].clone(),
            "target": host["target"][r0: r0 + k, c0:].to(torch.int64, copy=True),  # (copies: a view would drag the fused
            "seq": host["seq"][r0: r0 + k, c0:].to(torch.int64, copy=True),        #  launch's whole storage into the file)
            "qual": host["qual"][r0: r0 + k, c0:].clone(),
            "id": torch.from_numpy(idm),
        }, os.path.join(dir_path, f"{rank}_{first_index + i}.pt"))

    def save_range(lo: int, hi: int, starts: list[int]) -> None:
        for i in range(lo, hi):
            save(i, starts[i], subs[i], cols[i])

    starts = [0] * len(subs)
    for i in range(1, len(subs)):
        starts[i] = starts[i - 1] + subs[i - 1]
    if pool is None:
        save_range(0, len(subs), starts)
    else:
        step = max(1, -(-len(subs) // 4))  # a launch's files as up to four jobs
        for lo in range(0, len(subs), step):
            pending.append(pool.submit(save_range, lo, min(len(subs), lo + step), starts))
    return len(subs)


class PtWriter:
    """CustomWriter's `.pt` files (callbacks.py:12-25) written by libdeepchopper_io's thread pool (include/deepchopper_io.h:
    dc_pt_*): the same container torch.save writes -- `torch.load` and the reference's `chop` read it -- without torch.save's
    byte-wise CRC-32 and per-file Python.  `submit` hands over every reference batch of one (possibly fused) launch and
    returns.
    Device tensors (the pipeline's): the files' storages are cut out, converted and laid out ON THE GPU (dc_pt_pack), come to
    pinned host memory in one asynchronous copy on the writer's own stream and go into the files without another pass; the
    launch's device tensors and the pinned copy stay referenced here until the files are on disk (at most `max_inflight`
    launches in flight).  Host tensors: the writer's threads cut the batches out of the dense arrays themselves."""

    def __init__(self, threads: int = 4, max_inflight: int = 3):
        import ctypes as C

        from . import _iolib

        self._io = _iolib
        self._h = C.c_void_p()
        _iolib.raise_for(_iolib.lib().dc_pt_open(int(max(1, threads)), C.byref(self._h)))
        self._staged: list = []   # packed launches whose copy to the host is still running: (event, submit arguments, keep-alive)
        self._pending: list = []  # (ticket, keep-alive, pinned buffer or None)
        self._max = max(1, max_inflight)
        self._stream = None       # pack + device -> host copies run beside the pipeline's compute stream
        self._d_out = None        # device payload buffer (one: its producers and consumers are ordered on self._stream)
        self._pinned_free: list = []
        self.timers = {"pack_enqueue": 0.0, "pinned_alloc": 0.0, "wait_copy": 0.0, "wait_files": 0.0, "host_prep": 0.0}  # main-thread seconds

    # ---- geometry shared by both forms ----
    @staticmethod
    def _files(res: dict, n: int):
        subs = np.asarray(res.get("sub_sizes") or [n], np.int32)
        cols = np.asarray(res.get("sub_cols") or [0] * len(subs), np.int32)
        row0 = np.zeros(len(subs), np.int32)
        np.cumsum(subs[:-1], out=row0[1:])
        return row0, subs, cols

    @staticmethod
    def _names(ids: list[str]):
        enc = [i.encode() for i in ids]
        offs = np.zeros(len(ids) + 1, np.int64)
        np.cumsum([len(e) for e in enc], out=offs[1:])
        return b"".join(enc), offs

    def submit(self, dir_path: str, rank: int, first_index: int, res: dict, ids: list[str], ready_event=None, truncated=None) -> int:
        """`ready_event`: a CUDA event behind which the launch's device tensors are final (the pipeline's per-batch copy
        event); without one the writer's stream waits for the current stream.  `truncated`: the reads' truncation flags on
        the host (the pipeline's results carry them; read from the device here otherwise, which waits for the current stream)."""
        os.makedirs(dir_path, exist_ok=True)
        if truncated is None:
            truncated = (res["tok"]["status"].cpu().numpy() & 1) != 0
        truncated = np.ascontiguousarray(np.asarray(truncated).astype(np.uint8))
        if res["logits"].is_cuda:
            return self._submit_device(dir_path, rank, first_index, res, ids, ready_event, truncated)
        return self._submit_host(dir_path, rank, first_index, res, ids, truncated)

    def _submit_host(self, dir_path, rank, first_index, res, ids, truncated) -> int:
        import ctypes as C

        tok = res["tok"]
        n = len(ids)
        row0, subs, cols = self._files(res, n)
        ids_t = tok["input_ids"]
        if ids_t.dtype != torch.uint8:
            ids_t = ids_t.to(torch.uint8)  # (token ids are < 16; the file's int64 is made by the writer)
        host = (res["logits"].contiguous(), tok["labels"].contiguous(), ids_t.contiguous(), tok["input_quals"].contiguous())
        assert host[0].dtype == torch.float32 and host[1].dtype == torch.int8 and host[3].dtype == torch.float32
        stride = int(host[1].shape[1])
        assert host[0].shape == (n, stride, 2) and host[2].shape == (n, stride) and host[3].shape == (n, stride)
        names, offs = self._names(ids)
        ticket = C.c_int64(0)
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        rc = self._io.lib().dc_pt_submit(self._h, os.fsencode(dir_path), int(rank), int(first_index), len(subs), ptr(row0), ptr(subs),
                                         ptr(cols), C.c_void_p(host[0].data_ptr()), C.c_void_p(host[1].data_ptr()),
                                         C.c_void_p(host[2].data_ptr()), C.c_void_p(host[3].data_ptr()), n, stride,
                                         C.c_char_p(names), ptr(offs), ptr(truncated), C.byref(ticket))
        self._io.raise_for(rc)
        self._pending.append((ticket.value, host, None))
        self._bound()
        return len(subs)

    def _submit_device(self, dir_path, rank, first_index, res, ids, ready_event, truncated) -> int:
        from . import ops

        tok = res["tok"]
        n = len(ids)
        row0, subs, cols = self._files(res, n)
        logits, labels, ids_t, quals = res["logits"], tok["labels"], tok["input_ids"], tok["input_quals"]
        stride = int(ids_t.shape[1])
        n_cols = (stride - cols).astype(np.int32)
        sizes = subs.astype(np.int64) * n_cols * 28
        pay_off = np.zeros(len(subs), np.int64)
        np.cumsum(((sizes + 63) // 64 * 64)[:-1], out=pay_off[1:])
        total = int(pay_off[-1] + sizes[-1])
        table = np.stack([row0.astype(np.int64), subs.astype(np.int64), cols.astype(np.int64), pay_off], axis=1)
        dev = logits.device
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        t0 = time.perf_counter()
        pinned = self._take_pinned(total)
        t1 = time.perf_counter()
        self.timers["pinned_alloc"] += t1 - t0
        with torch.cuda.stream(self._stream):
            if self._d_out is None or self._d_out.numel() < total:  # (allocated under the writer's stream: the caching allocator
                self._d_out = None                                  #  hands a freed block back to that stream only)
                self._d_out = torch.empty(int(total * 1.25) + 4096, dtype=torch.uint8, device=dev)
            if ready_event is not None:
                self._stream.wait_event(ready_event)
            else:
                self._stream.wait_stream(torch.cuda.current_stream(dev))
            ops.pt_pack(logits, labels, ids_t, quals, table, self._d_out)
            pinned[:total].copy_(self._d_out[:total], non_blocking=True)
            copied = torch.cuda.Event()  # (waited for by polling: wait_event)
            copied.record(self._stream)
        t2 = time.perf_counter()
        self.timers["pack_enqueue"] += t2 - t1
        names, offs = self._names(ids)
        self.timers["host_prep"] += time.perf_counter() - t2
        args = (os.fsencode(dir_path), int(rank), int(first_index), row0, subs, n_cols, pay_off, total, n, names, offs, truncated)
        self._staged.append((copied, args, (logits, labels, ids_t, quals), pinned))  # (the device tensors: until the copy is done)
        self._flush(block=False)
        self._bound()
        return len(subs)

    def _take_pinned(self, nbytes: int) -> torch.Tensor:
        for i, t in enumerate(self._pinned_free):
            if t.numel() >= nbytes:
                return self._pinned_free.pop(i)
        if len(self._pinned_free) >= 2:  # the ring has grown past what is in flight: let the smallest go
            self._pinned_free.sort(key=lambda t: t.numel())
            self._pinned_free.pop(0)
        return torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)

    def _flush(self, block: bool) -> None:
        """Launches whose payload has arrived in host memory go to the writer threads (in order)."""
        import ctypes as C

        oldest = True  # (block=True waits for ONE launch, the oldest; the rest go only if their copy has landed)
        while self._staged and ((block and oldest) or self._staged[0][0].query()):
            oldest = False
            copied, args, _dev_keep, pinned = self._staged.pop(0)
            t0 = time.perf_counter()
            wait_event(copied)
            self.timers["wait_copy"] += time.perf_counter() - t0
            d, rank, first_index, row0, subs, n_cols, pay_off, total, n, names, offs, truncated = args
            ticket = C.c_int64(0)
            ptr = lambda a: C.c_void_p(a.ctypes.data)
            rc = self._io.lib().dc_pt_submit_packed(self._h, d, rank, first_index, len(subs), ptr(row0), ptr(subs), ptr(n_cols), ptr(pay_off),
                                                    C.c_void_p(pinned.data_ptr()), total, n, C.c_char_p(names), ptr(offs),
                                                    ptr(truncated), C.byref(ticket))
            if rc:
                self._pinned_free.append(pinned)
            self._io.raise_for(rc)
            self._pending.append((ticket.value, None, pinned))

    def _bound(self) -> None:
        while len(self._staged) + len(self._pending) > self._max:
            if self._pending:
                self._wait_oldest()
            else:
                self._flush(block=True)

    def _wait_oldest(self) -> None:
        ticket, _keep, pinned = self._pending.pop(0)
        t0 = time.perf_counter()
        try:
            self._io.raise_for(self._io.lib().dc_pt_wait(self._h, ticket))
        finally:
            self.timers["wait_files"] += time.perf_counter() - t0
            if pinned is not None:
                self._pinned_free.append(pinned)

    def close(self) -> None:
        """Waits for every file; a failed write raises here."""
        if self._h:
            try:
                while self._staged:
                    self._flush(block=True)
            finally:
                h, self._h = self._h, None
                try:
                    rc = self._io.lib().dc_pt_close(h)  # returns only when nothing reads the host copies any more
                finally:
                    self._pending.clear()
                    self._staged.clear()
                    self._pinned_free.clear()
            self._io.raise_for(rc)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # a writer dropped without close(): finish what was queued, quietly
        try:
            self.close()
        except Exception:
            pass
