"""Host-side mirror of deepchopper/models/llm/tokenizer.py for the predict path.

The per-read Python work of the reference (HF tokenizer call, `vectorize_targets`,
`F.normalize`, list padding in the collator: tokenizer.py:34-93,145-178) is one fused GPU launch
here (`dc_tokenize_batch`); this module keeps the reference's names and return structure on top
of it, plus the small host-only pieces (id encoding, error checks).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops

# tokenizer.max_len_single_sentence of the Hub tokenizer (model_max_length 32770 - 1 special
# token); unverifiable offline and irrelevant below 32 kb -- override if needed.
DEFAULT_MAX_LENGTH = 32769
MAX_ID_LENGTH = 256
IGNORE_INDEX = -100
VOCAB = {"[CLS]": 0, "[SEP]": 1, "[BOS]": 2, "[MASK]": 3, "[PAD]": 4, "[RESERVED]": 5, "[UNK]": 6,
         "A": 7, "C": 8, "G": 9, "T": 10, "N": 11}
PAD_TOKEN_ID = VOCAB["[PAD]"]  # what the collator pads `input_ids` with (tokenizer.py:71-87; src/smooth/utils.rs:6-25)


def encode_ids_int8(read_id: str, truncated: bool, max_id_length: int = MAX_ID_LENGTH) -> np.ndarray:
    """tokenizer.py:170-175 + the collator's int8 cast (tokenizer.py:91): [len, truncated,
    ord(c)...] clipped / zero padded to 256.  Like `torch.tensor(..., dtype=torch.int8)` in the
    reference, values outside int8 (ids longer than 127 characters, non-ASCII) raise."""
    row = [len(read_id), int(truncated)] + [ord(c) for c in read_id]
    row = row[:max_id_length] + [0] * max(0, max_id_length - len(row))
    a = np.asarray(row, dtype=np.int64)
    if a.max(initial=0) > 127:
        raise RuntimeError("value cannot be converted to type int8 without overflow")
    return a.astype(np.int8)


def check_read_status(status: np.ndarray, ids) -> None:
    """Reads whose sequence lost characters in normalize_seq (whitespace) end up with mismatched
    id / quality lengths in the reference and fail in its collator; fail loudly here too."""
    bad = np.flatnonzero(status & 2)
    if bad.size:
        raise ValueError(f"read '{ids[int(bad[0])]}': sequence contains whitespace; sequence and quality lengths "
                         f"differ after normalisation ({bad.size} such reads)")


def tokenize_and_align_labels_and_quals_ids(data: dict, tokenizer=None, max_length: int = DEFAULT_MAX_LENGTH,
                                            pad_qual: int = 0, pad_label: int = IGNORE_INDEX,
                                            max_id_length: int = MAX_ID_LENGTH) -> dict:
    """tokenizer.py:145-178 for one record `{"id", "seq", "qual"}` (seq: str/bytes of bases, qual:
    raw FASTQ quality string).  Returns input_ids / labels / input_quals / id as the reference's
    dataset row would hold them (torch CPU tensors)."""
    seq = data["seq"].encode() if isinstance(data["seq"], str) else bytes(data["seq"])
    qual = data["qual"].encode() if isinstance(data["qual"], str) else bytes(data["qual"])
    if len(seq) != len(qual):
        raise ValueError(f"Sequence/quality length mismatch in record '{data['id']}': seq={len(seq)}, qual={len(qual)}")
    s = ops.to_device(np.frombuffer(seq, np.uint8))
    q = ops.to_device(np.frombuffer(qual, np.uint8))
    offs = ops.to_device(np.array([0, len(seq)], np.int64))
    stride = min(len(seq), max_length - 1) + 1
    out = ops.tokenize_batch(s, q, offs, max_length=max_length, row_stride=stride, pad_left=True, ids_i64=True)
    st = out["status"].cpu().numpy()
    check_read_status(st, [data["id"]])
    return {
        "input_ids": out["input_ids"][0].cpu(),
        "labels": out["labels"][0].cpu().to(torch.int64),
        "input_quals": out["input_quals"][0].cpu(),
        "id": encode_ids_int8(data["id"], bool(st[0] & 1), max_id_length).astype(np.int64).tolist(),
    }


class DataCollatorForTokenClassificationWithQual:
    """tokenizer.py:34-93 over raw records: `torch_call(records)` pads on the LEFT to the longest
    read of the batch and returns input_ids i64, input_quals f32, labels i8, id i8[B,256]."""

    def __init__(self, tokenizer=None, max_length: int = DEFAULT_MAX_LENGTH, padding_side: str = "left"):
        self.max_length = max_length
        self.padding_side = padding_side

    def torch_call(self, records: list[dict]) -> dict:
        seqs = [r["seq"].encode() if isinstance(r["seq"], str) else bytes(r["seq"]) for r in records]
        quals = [r["qual"].encode() if isinstance(r["qual"], str) else bytes(r["qual"]) for r in records]
        ids = [r["id"] for r in records]
        sb, offs = ops.pack_reads(seqs)
        qb, _ = ops.pack_reads(quals)
        stride = min(max(len(s) for s in seqs), self.max_length - 1) + 1
        out = ops.tokenize_batch(ops.to_device(sb), ops.to_device(qb), ops.to_device(offs), max_length=self.max_length,
                                 row_stride=stride, pad_left=self.padding_side == "left", ids_i64=True)
        st = out["status"].cpu().numpy()
        check_read_status(st, ids)
        idm = np.stack([encode_ids_int8(i, bool(s & 1)) for i, s in zip(ids, st)])
        return {"input_ids": out["input_ids"], "input_quals": out["input_quals"], "labels": out["labels"],
                "id": torch.from_numpy(idm)}

    __call__ = torch_call
