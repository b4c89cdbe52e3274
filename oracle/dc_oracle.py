"""CPU oracle (numpy) for the integer / byte half of the `deepchopper predict` hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Each function restates one reference
function and cites it as `path:line` under /root/reference.  Where the reference's behaviour
comes from a third-party crate that is absent from the tree, the crate and version are named and
its published algorithm is restated.

Conventions
-----------
* integer overflow follows the reference's *release* profile (`Cargo.toml:99-103`, no
  `overflow-checks`): `u8` arithmetic wraps modulo 256.
* "literal" functions (`*_literal`) are line-by-line pure-Python loops for small cases; the
  vectorised numpy versions are what tests use at scale and are cross-checked against them.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# constants (src/default.rs:1-7)
# --------------------------------------------------------------------------------------
QUAL_OFFSET = 33
BASES = b"ATCGN"
KMER_SIZE = 3
VECTORIZED_TARGET = False
MIN_READ_LEN = 150
MIN_CHOPED_SEQ_LEN = 20
IGNORE_LABEL = -100

# HyenaDNA CharacterTokenizer vocabulary as pinned by the reference itself:
# src/smooth/utils.rs:6-25 (7..11 = A,C,G,T,N), fixtures tests/data/eval/*/*.pt ({1,4,7,8,9,10}),
# comment deepchopper/data/only_fq.py:59-61 ([UNK]).
TOK_CLS, TOK_SEP, TOK_BOS, TOK_MASK, TOK_PAD, TOK_RESERVED, TOK_UNK = 0, 1, 2, 3, 4, 5, 6
TOK_A, TOK_C, TOK_G, TOK_T, TOK_N = 7, 8, 9, 10, 11
# tokenizer.max_len_single_sentence of the Hub tokenizer (model_max_length 32770 minus one
# special token) -- believed, unverified here.  Irrelevant below 32 kb.
DEFAULT_MAX_LENGTH = 32769
MAX_ID_LENGTH = 256


class EncodingError(Exception):
    """src/error.rs:4-26 -- messages are the `#[error]` strings."""


ERR_SEQ_SHORTER = "The sequence is shorter than the k-mer size"
ERR_TARGET_INVALID = "The target region is invalid"
ERR_INVALID_KMER_ID = "The k-mer id is invalid"


# --------------------------------------------------------------------------------------
# T1  encode_qual                                    src/python.rs:25-35, triat.rs:111-117
# --------------------------------------------------------------------------------------
def encode_qual(qual: bytes | str, qual_offset: int = QUAL_OFFSET) -> np.ndarray:
    """phred[i] = qual_ascii[i] - offset, as u8 (wrapping), returned as int64 like the
    reference's Vec<usize>."""
    q = np.frombuffer(qual.encode() if isinstance(qual, str) else bytes(qual), dtype=np.uint8)
    return (q - np.uint8(qual_offset)).astype(np.int64)  # uint8 arithmetic wraps


# --------------------------------------------------------------------------------------
# T2  normalize_seq                 src/python.rs:272-275 -> needletail 0.7.3 Sequence::normalize
# --------------------------------------------------------------------------------------
def _build_normalize_lut(iupac: bool) -> np.ndarray:
    """Published needletail algorithm (`needletail::sequence::normalize`): ACGTN- kept; acg ->
    upper; t,u,U -> T; '.', '~' -> '-'; with iupac the ten ambiguity codes are kept / upper-cased;
    space, tab, CR, LF removed; everything else (incl. lower-case 'n') -> 'N'.
    LUT value 0 means "drop this byte".  Third-party: **parity unpinned** by any reference test."""
    lut = np.full(256, ord("N"), dtype=np.uint8)
    for c in b"ACGTN-":
        lut[c] = c
    for lo, up in zip(b"acg", b"ACG"):
        lut[lo] = up
    for c in b"tuU":
        lut[c] = ord("T")
    for c in b".~":
        lut[c] = ord("-")
    if iupac:
        for c in b"BDHVRYSWKM":
            lut[c] = c
            lut[c + 32] = c
    for c in b" \t\r\n":
        lut[c] = 0
    return lut


_NORM_LUT = {False: _build_normalize_lut(False), True: _build_normalize_lut(True)}


def normalize_seq(seq: bytes | str, iupac: bool = False) -> bytes:
    s = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    out = _NORM_LUT[bool(iupac)][s]
    return out[out != 0].tobytes()


# --------------------------------------------------------------------------------------
# T3  HyenaDNA CharacterTokenizer (third-party remote code; vocabulary pinned as above)
#     call site deepchopper/models/llm/tokenizer.py:148
# --------------------------------------------------------------------------------------
_TOKEN_LUT = np.full(256, TOK_UNK, dtype=np.int64)
for _c, _t in zip(b"ACGTN", (TOK_A, TOK_C, TOK_G, TOK_T, TOK_N)):
    _TOKEN_LUT[_c] = _t


def tokenize_chars(seq: bytes | str, max_length: int = DEFAULT_MAX_LENGTH) -> np.ndarray:
    """One token per character + trailing [SEP]; `truncation=True` keeps `max_length-1`
    characters when `len(seq) >= max_length` so that the total is `max_length`."""
    s = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    if len(s) >= max_length:
        s = s[: max_length - 1]
    return np.concatenate([_TOKEN_LUT[s], np.array([TOK_SEP], dtype=np.int64)])


# --------------------------------------------------------------------------------------
# kmer.rs helpers used by T4
# --------------------------------------------------------------------------------------
def vectorize_target(start: int, end: int, length: int) -> np.ndarray:
    """src/kmer.rs:33-46."""
    if start > end or end > length:
        raise EncodingError(ERR_TARGET_INVALID)
    r = np.zeros(length, dtype=np.int64)
    r[start:end] = 1
    return r


def vectorize_targets(targets: Sequence[int], length: int) -> np.ndarray:
    """src/kmer.rs:51-72."""
    if len(targets) == 0:
        return np.zeros(length, dtype=np.int64)
    assert len(targets) % 2 == 0
    r = np.zeros(length, dtype=np.int64)
    for s, e in zip(targets[0::2], targets[1::2]):
        if s > e or e > length:
            raise EncodingError(ERR_TARGET_INVALID)
        r[s:e] = 1
    return r


# --------------------------------------------------------------------------------------
# T4  tokenize_and_align_labels_and_quals_ids   deepchopper/models/llm/tokenizer.py:145-178
# --------------------------------------------------------------------------------------
def l2_normalize_quals(quals_f32: np.ndarray) -> np.ndarray:
    """`torch.nn.functional.normalize(quals, dim=0)` = x / max(||x||_2, 1e-12) in fp32.
    Canonical evaluation order used by this build (oracle and HIP alike): the sum of squares is
    accumulated exactly in integers (phred values are small integers), converted to fp32,
    square-rooted (correctly rounded) and each element divided in fp32.  torch's own fp32
    reduction order is unspecified; it agrees with this to ~1 ulp (checked in tests)."""
    q = quals_f32.astype(np.float32)
    sumsq = int(np.sum(q.astype(np.int64) ** 2)) if np.all(q == np.floor(q)) else None
    if sumsq is None:
        nrm = np.float32(np.sqrt(np.sum(q.astype(np.float64) ** 2)))
    else:
        nrm = np.sqrt(np.float32(sumsq))
    nrm = np.maximum(nrm, np.float32(1e-12))
    return (q / nrm).astype(np.float32)


def encode_id(read_id: str | bytes, truncated: bool, max_id_length: int = MAX_ID_LENGTH) -> np.ndarray:
    """tokenizer.py:170-175: [len(id), int(truncated), ord(c)...] clipped / zero-padded to 256.
    Returned as int64 *before* the collator's int8 cast."""
    rid = read_id.decode() if isinstance(read_id, bytes) else read_id
    new_id = [len(rid), int(truncated)] + [ord(c) for c in rid]
    if len(new_id) > max_id_length:
        new_id = new_id[:max_id_length]
    else:
        new_id = new_id + [0] * (max_id_length - len(new_id))
    return np.array(new_id, dtype=np.int64)


def tokenize_and_align_labels_and_quals_ids(
    read_id: str,
    seq: bytes | str,
    qual_phred: np.ndarray,
    target: Sequence[int] = (0, 0),
    max_length: int = DEFAULT_MAX_LENGTH,
    pad_qual: int = 0,
    pad_label: int = IGNORE_LABEL,
) -> dict:
    seq_b = seq.encode() if isinstance(seq, str) else bytes(seq)
    input_ids = tokenize_chars(seq_b, max_length)
    truncation = False
    qual_phred = np.asarray(qual_phred)
    if len(seq_b) >= max_length:
        truncation = True
        if target[1] + 2 > max_length:
            labels = np.concatenate([vectorize_target(0, 0, max_length - 1), [pad_label]])
        else:
            labels = np.concatenate([vectorize_targets(list(target), max_length - 1), [pad_label]])
        quals = np.concatenate([qual_phred[: max_length - 1], [pad_qual]]).astype(np.float32)
    else:
        labels = np.concatenate([vectorize_targets(list(target), len(seq_b)), [pad_label]])
        quals = np.concatenate([qual_phred, [pad_qual]]).astype(np.float32)
    return {
        "input_ids": input_ids,
        "labels": labels.astype(np.int64),
        "input_quals": l2_normalize_quals(quals),
        "id": encode_id(read_id, truncation),
    }


def parse_fastq_record(name: str, seq: str, qual: str) -> dict:
    """Per-record body of deepchopper/data/only_fq.py:36-72 with `has_targets=False` and
    pyfastx `uppercase=True` (only_fq.py:34)."""
    if not name or not seq or not qual:
        raise ValueError("Incomplete FASTQ record")
    if len(seq) != len(qual):
        raise ValueError(
            f"Sequence/quality length mismatch in record '{name}': seq={len(seq)}, qual={len(qual)}"
        )
    return {
        "id": name,
        "seq": normalize_seq(seq.upper(), False),
        "qual": encode_qual(qual, QUAL_OFFSET),
        "target": [0, 0],
    }


# --------------------------------------------------------------------------------------
# T5  DataCollatorForTokenClassificationWithQual.torch_call  tokenizer.py:34-93
# --------------------------------------------------------------------------------------
def collate(features: Sequence[dict], padding_side: str = "left") -> dict:
    """Pad to the longest read of the batch (tokenizer pads on the LEFT per the fixtures):
    ids with [PAD]=4, labels -100 (int8), quals 0.0 (fp32), id int8[B,256] (values > 127 wrap)."""
    lmax = max(len(f["input_ids"]) for f in features)
    B = len(features)
    ids = np.full((B, lmax), TOK_PAD, dtype=np.int64)
    labels = np.full((B, lmax), IGNORE_LABEL, dtype=np.int8)
    quals = np.zeros((B, lmax), dtype=np.float32)
    idm = np.zeros((B, MAX_ID_LENGTH), dtype=np.int8)
    for b, f in enumerate(features):
        n = len(f["input_ids"])
        sl = slice(lmax - n, lmax) if padding_side == "left" else slice(0, n)
        ids[b, sl] = f["input_ids"]
        labels[b, sl] = f["labels"].astype(np.int8)
        quals[b, sl] = f["input_quals"]
        if np.any(f["id"] > 127) or np.any(f["id"] < -128):
            # torch.tensor(list_of_python_ints, dtype=torch.int8) refuses out-of-range values
            # (tokenizer.py:91): ids longer than 127 characters make the reference raise.
            raise RuntimeError("value cannot be converted to type int8 without overflow")
        idm[b] = f["id"].astype(np.int8)
    return {"input_ids": ids, "input_quals": quals, "labels": labels, "id": idm}


# --------------------------------------------------------------------------------------
# K1  generate_kmers / generate_kmers_table                       src/kmer.rs:213-228
# --------------------------------------------------------------------------------------
def generate_kmers(bases: bytes | str, k: int) -> list[bytes]:
    b = bases.encode() if isinstance(bases, str) else bytes(bases)
    return [bytes(c) for c in itertools.product(b, repeat=k)]


def generate_kmers_table(bases: bytes | str, k: int) -> dict[bytes, int]:
    return {km: i for i, km in enumerate(generate_kmers(bases, k))}


# --------------------------------------------------------------------------------------
# kmer.rs: windows / offsets / reconstruction / region maths          src/kmer.rs:79-211
# --------------------------------------------------------------------------------------
def seq_to_kmers(seq: bytes, k: int, overlap: bool) -> list[bytes]:
    """src/kmer.rs:144-150 (`par_windows` / `par_chunks`; chunks keep the short tail)."""
    if overlap:
        return [seq[i : i + k] for i in range(0, max(len(seq) - k + 1, 0))]
    return [seq[i : i + k] for i in range(0, len(seq), k)]


def py_seq_to_kmers(seq: str, k: int, overlap: bool) -> list[str]:
    """src/python.rs:243-249: the PyO3 wrapper normalises first."""
    return [x.decode() for x in seq_to_kmers(normalize_seq(seq, False), k, overlap)]


def seq_to_kmers_and_offset(seq: bytes, k: int, overlap: bool):
    """src/kmer.rs:175-211."""
    if k == 0 or k > len(seq):
        raise EncodingError(ERR_SEQ_SHORTER)
    if overlap:
        return [(seq[i : i + k], (i, i + k)) for i in range(len(seq) - k + 1)]
    return [(seq[i : i + k], (i, i + k)) for i in range(0, len(seq), k) if len(seq[i : i + k]) == k]


def kmers_to_seq(kmers: Sequence[bytes]) -> bytes:
    """src/kmer.rs:152-172."""
    if not kmers:
        return b""
    out = bytearray(kmers[0])
    for km in kmers[1:]:
        if len(km) == 0:
            raise ValueError("Invalid kmer")
        out.append(km[-1])
    return bytes(out)


def kmerids_to_seq(kmer_ids: Sequence[int], id2kmer: dict[int, bytes]) -> bytes:
    """src/kmer.rs:79-91."""
    kms = []
    for i in kmer_ids:
        if i not in id2kmer:
            raise EncodingError(ERR_INVALID_KMER_ID)
        kms.append(id2kmer[i])
    return kmers_to_seq(kms)


def to_original_targtet_region(start: int, end: int, k: int) -> tuple[int, int]:
    """src/kmer.rs:93-105."""
    return (start, end + k - 1) if end > start else (start, end)


def to_kmer_target_region(start: int, end: int, k: int, seq_len: int | None = None) -> tuple[int, int]:
    """src/kmer.rs:107-142."""
    if start >= end or k == 0:
        raise EncodingError(ERR_TARGET_INVALID)
    if seq_len is not None and end > seq_len:
        raise EncodingError(ERR_TARGET_INVALID)
    n = end - start - k + 1 if end - start >= k else 0
    return (start, start + n) if n > 0 else (start, end)


def splite_qual_by_offsets(target: Sequence[int], offsets: Sequence[tuple[int, int]]) -> list[int]:
    """src/kmer.rs:14-28."""
    return [0 if s == e else sum(target[s:e]) // (e - s) for s, e in offsets]


# --------------------------------------------------------------------------------------
# K4  parse_target_from_id                                src/fq_encode/triat.rs:14-48
# --------------------------------------------------------------------------------------
def _lexical_usize(b: bytes) -> int:
    """`lexical::parse::<usize>`: non-empty, ASCII digits only, no sign / whitespace; u64 range."""
    if len(b) == 0 or not all(48 <= c <= 57 for c in b):
        raise ValueError("parse number error")
    v = int(b)
    if v >= 1 << 64:
        raise ValueError("overflow")
    return v


def parse_target_from_id(src: bytes | str) -> list[tuple[int, int]]:
    s = src.encode() if isinstance(src, str) else bytes(src)
    if len(s) == 0:
        return []
    if b"|" not in s:
        return [(0, 0)]
    number_part = s.split(b"|")[-1]
    out = []
    try:
        for tgt in number_part.split(b"-"):
            parts = tgt.split(b":")
            if len(parts) < 2:
                raise ValueError("parse number error")
            out.append((_lexical_usize(parts[0]), _lexical_usize(parts[1])))
    except ValueError:
        return [(0, 0)]
    return out


def py_parse_target_from_id(src: bytes | str) -> list[int]:
    """src/python.rs:277-286: flattened [s, e, s, e, ...]."""
    return [v for se in parse_target_from_id(src) for v in se]


# --------------------------------------------------------------------------------------
# K2/K3/K5  TensorEncoder                      src/fq_encode/tensor_encoder.rs:127-327
# --------------------------------------------------------------------------------------
def kmer_ids_of(seq: bytes, k: int, bases: bytes) -> np.ndarray | None:
    """Overlapping windows -> id = sum_j idx(b_j) * nb^(k-1-j); None if any byte is not in
    `bases` (the reference's table lookup fails and the read is dropped,
    tensor_encoder.rs:242-250,294-299)."""
    s = np.frombuffer(seq, dtype=np.uint8)
    lut = np.full(256, -1, dtype=np.int64)
    for i, c in enumerate(bases):
        lut[c] = i  # later duplicates win, like HashMap insertion order would
    d = lut[s]
    if len(s) < k:
        return np.zeros(0, dtype=np.int32)
    if np.any(d < 0):
        return None
    nb = len(bases)
    w = len(s) - k + 1
    ids = np.zeros(w, dtype=np.int64)
    for j in range(k):
        ids = ids * nb + d[j : j + w]
    return ids.astype(np.int32)


def kmer_mean_qual(qual: bytes, k: int, qual_offset: int) -> tuple[np.ndarray, np.ndarray]:
    """src/fq_encode/triat.rs:102-134.  phred = q - offset (u8, wrapping); k-mer quality =
    (u8-wrapping window sum) / k (integer).  needletail `.kmers(k)` yields nothing when
    len < k.  (The reference pipes the windows through rayon `par_bridge`, which does not
    guarantee order; the parity target is the in-order sequence -- SURVEY 8a/K3.)"""
    q = np.frombuffer(qual, dtype=np.uint8)
    ph = (q - np.uint8(qual_offset)).astype(np.uint8)
    if len(ph) < k or k == 0:
        return ph.astype(np.int32), np.zeros(0, dtype=np.int32)
    w = len(ph) - k + 1
    acc = np.zeros(w, dtype=np.uint8)
    for j in range(k):
        acc = (acc + ph[j : j + w]).astype(np.uint8)  # wraps mod 256
    return ph.astype(np.int32), (acc // np.uint8(k)).astype(np.int32)


def read_fastq_records(path) -> list[tuple[bytes, bytes, bytes]]:
    """Minimal 4-line FASTQ reader; name = definition up to the first blank, as noodles does."""
    recs = []
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines):
        if not lines[i].startswith(b"@"):
            i += 1
            continue
        name = lines[i][1:].split(b" ")[0].split(b"\t")[0]
        seq = lines[i + 1].rstrip(b"\r")
        qual = lines[i + 3].rstrip(b"\r") if i + 3 < len(lines) else b""
        recs.append((name, seq, qual))
        i += 4
    return recs


def tensor_encode(
    records: Sequence[tuple[bytes, bytes, bytes]],
    k: int = KMER_SIZE,
    bases: bytes = BASES,
    qual_offset: int = QUAL_OFFSET,
    vectorized_target: bool = VECTORIZED_TARGET,
    max_width: int = 0,
    max_seq_len: int = 0,
):
    """TensorEncoder::encode over already-parsed records.
    Returns (input i32[N,2,W], target i32[N,T,2|W], qual i32[N,Lmax])."""
    # fetch_records: tensor_encoder.rs:127-175
    kept = []
    for rid, seq, qual in records:
        nlen = len(normalize_seq(seq, False))
        if nlen < k:
            continue
        if nlen != len(qual):
            raise ValueError(f"record: id {rid.decode(errors='replace')} seq_len != qual_len")
        max_seq_len = max(max_seq_len, nlen)
        kept.append((rid, seq, qual))
    if max_seq_len < k:
        raise EncodingError(ERR_SEQ_SHORTER)
    max_width = max(max_width, max_seq_len - k + 1)

    inputs, targets, quals = [], [], []
    for rid, seq, qual in kept:  # encode_record: tensor_encoder.rs:221-282
        cur_w = max(len(seq) - k, 0) + 1
        if cur_w > max_width:
            continue
        ids = kmer_ids_of(seq, k, bases)
        if ids is None or len(ids) != cur_w:
            continue
        row_ids = np.full(max_width, -1, dtype=np.int32)
        row_ids[:cur_w] = ids
        ph, kq = kmer_mean_qual(qual, k, qual_offset)
        row_kq = np.full(max_width, -1, dtype=np.int32)
        row_kq[: min(len(kq), max_width)] = kq[:max_width]
        row_q = np.full(max_seq_len, -1, dtype=np.int32)
        row_q[: min(len(ph), max_seq_len)] = ph[:max_seq_len]
        # encode_target: tensor_encoder.rs:177-219
        try:
            kt = [to_kmer_target_region(s, e, k, None) for s, e in parse_target_from_id(rid)]
        except EncodingError:
            continue
        if vectorized_target:
            t = np.zeros((len(kt), max_width), dtype=np.int32)
            for r, (s, e) in enumerate(kt):
                if s < e and e <= max_width:
                    t[r, s:e] = 1
        else:
            t = np.array(kt, dtype=np.int32).reshape(len(kt), 2)
        inputs.append(np.stack([row_ids, row_kq]))
        targets.append(t)
        quals.append(row_q)
    if not inputs:
        raise ValueError("Failed to stack inputs")
    nt = {t.shape for t in targets}
    if len(nt) != 1:
        raise ValueError("Failed to stack targets")
    return np.stack(inputs), np.stack(targets), np.stack(quals)


# --------------------------------------------------------------------------------------
# S0  argmax + mask + id decode                       src/smooth/predict.rs:264-317
# --------------------------------------------------------------------------------------
_ID2BASE = np.full(256, ord("N"), dtype=np.uint8)
for _t, _c in zip((TOK_A, TOK_C, TOK_G, TOK_T, TOK_N), b"ACGTN"):
    _ID2BASE[_t] = _c


def id_list2seq(ids: Iterable[int]) -> str:
    """src/smooth/utils.rs:34-46: 7..11 -> ACGTN, anything else -> N."""
    a = np.asarray(list(ids), dtype=np.int64)
    out = np.where((a >= 0) & (a < 256), _ID2BASE[np.clip(a, 0, 255)], ord("N")).astype(np.uint8)
    return out.tobytes().decode()


def summary_predict(predictions, labels, ignore_label=IGNORE_LABEL):
    """src/utils.rs:9-55."""
    tp, tl = [], []
    for p, l in zip(predictions, labels):
        p = np.asarray(p)
        l = np.asarray(l)
        m = l != ignore_label
        n = min(len(p), len(l))
        tp.append(p[:n][m[:n]])
        tl.append(l[:n][m[:n]])
    return tp, tl


@dataclass
class Predict:
    """src/smooth/predict.rs:33-46."""

    prediction: np.ndarray
    seq: str
    id: str
    is_truncated: bool
    qual: str | None = None

    def prediction_region(self):
        return get_label_region(self.prediction)

    def smooth_prediction(self, window_size):
        return get_label_region(majority_voting(self.prediction, window_size))

    def smooth_label(self, window_size):
        return majority_voting(self.prediction, window_size)

    def smooth_and_select_intervals(self, smooth_window_size, min_interval_size, approved_interval_number):
        return smooth_label_region(self.prediction, smooth_window_size, min_interval_size, approved_interval_number)

    def seq_len(self):
        return len(self.seq)


def argmax_first(logits: np.ndarray) -> np.ndarray:
    """argmax over the last (class) axis; exactly equal logits -> lowest index ("first max
    wins"; candle-core's tie-break is third-party and unverified -- SURVEY 8a/S0)."""
    return np.argmax(logits, axis=-1)


def load_predicts_from_batch(batch: dict, ignore_label: int = IGNORE_LABEL) -> dict[str, Predict]:
    """load_predicts_from_batch_pt on an in-memory dict of arrays
    (keys prediction f32[B,L,2], target, seq, id)."""
    pred = argmax_first(np.asarray(batch["prediction"])).astype(np.int64)
    target = np.asarray(batch["target"]).astype(np.int64)
    seq = np.asarray(batch["seq"]).astype(np.int64)
    idm = np.asarray(batch["id"]).astype(np.int64)
    tp, _ = summary_predict(pred, target, ignore_label)
    ts, _ = summary_predict(seq, target, ignore_label)
    out = {}
    for i in range(len(tp)):
        n = int(idm[i][0])
        rid = "".join(chr(int(c)) for c in idm[i][2 : 2 + n])
        out[rid] = Predict(
            prediction=tp[i].astype(np.int8),
            seq=id_list2seq(ts[i]),
            id=rid,
            is_truncated=bool(idm[i][1] != 0),
        )
    return out


# --------------------------------------------------------------------------------------
# S1  majority_voting                                     src/smooth/utils.rs:48-97
# --------------------------------------------------------------------------------------
def majority_voting_literal(labels: Sequence[int], window_size: int) -> list[int]:
    """Line-by-line restatement (pure Python, small inputs).  For more than two distinct labels
    with equal counts the reference's HashMap iteration makes the result arbitrary; this oracle
    (and the build) define the path for binary labels only."""
    if window_size % 2 == 0:
        window_size += 1
    half = window_size // 2
    n = len(labels)
    out = []
    for i in range(n):
        start = max(i - half, 0)
        end = min(n, i + half + 1)
        if end == n and (end - start) < window_size:
            start = max(end - window_size, 0)
        window = list(labels[start:end])
        counts: dict[int, int] = {}
        for lab in window:
            counts[lab] = counts.get(lab, 0) + 1
        if len(counts) == 2:
            v = list(counts.values())
            if v[0] == v[1]:
                out.append(labels[i])
                continue
        out.append(max(counts.items(), key=lambda kv: kv[1])[0])
    return out


def majority_voting(labels, window_size: int) -> np.ndarray:
    """Vectorised binary form: ones = P[end]-P[start]; 2*ones > len -> 1, < -> 0, == -> labels[i]."""
    lab = np.asarray(labels).astype(np.int8)
    n = len(lab)
    if n == 0:
        return lab.copy()
    if not np.all((lab == 0) | (lab == 1)):
        return np.asarray(majority_voting_literal(lab.tolist(), window_size), dtype=np.int8)
    w = window_size + 1 if window_size % 2 == 0 else window_size
    h = w // 2
    i = np.arange(n, dtype=np.int64)
    start = np.maximum(i - h, 0)
    end = np.minimum(n, i + h + 1)
    pull = (end == n) & ((end - start) < w)
    start = np.where(pull, np.maximum(end - w, 0), start)
    P = np.concatenate([[0], np.cumsum(lab.astype(np.int64))])
    ones = P[end] - P[start]
    ln = end - start
    out = np.where(2 * ones > ln, 1, np.where(2 * ones < ln, 0, lab)).astype(np.int8)
    return out


# --------------------------------------------------------------------------------------
# S2  get_label_region                                         src/utils.rs:671-695
# --------------------------------------------------------------------------------------
def get_label_region_literal(labels: Sequence[int]) -> list[tuple[int, int]]:
    regions = []
    start = 0
    end = 0
    for i, lab in enumerate(labels):
        if lab == 1:
            if start == 0:
                start = i
            end = i
        elif start != 0:
            regions.append((start, end + 1))
            start = 0
            end = 0
    if start != 0:
        regions.append((start, end + 1))
    return regions


def get_label_region(labels) -> list[tuple[int, int]]:
    """Maximal runs of 1 with the reference's `start == 0` sentinel quirk: a run that begins at
    index 0 loses its first element; a lone 1 at index 0 is dropped."""
    lab = (np.asarray(labels) == 1).astype(np.int8)
    n = len(lab)
    if n == 0:
        return []
    d = np.diff(np.concatenate([[0], lab, [0]]))
    starts = np.flatnonzero(d == 1)
    ends = np.flatnonzero(d == -1)
    out = []
    for s, e in zip(starts.tolist(), ends.tolist()):
        if s == 0:
            s = 1
        if s < e:
            out.append((s, e))
    return out


# --------------------------------------------------------------------------------------
# S3  smooth_label_region == Predict::smooth_and_select_intervals
#     src/utils.rs:699-721, src/smooth/predict.rs:186-209
# --------------------------------------------------------------------------------------
def smooth_label_region(labels, smooth_window_size=21, min_interval_size=13, approved_interval_number=20):
    regs = get_label_region(majority_voting(labels, smooth_window_size))
    res = [(s, e) for s, e in regs if e - s >= min_interval_size]
    if len(res) > approved_interval_number:
        return []
    return res


# --------------------------------------------------------------------------------------
# S5  generate_unmaped_intervals / remove_intervals_and_keep_left  src/output/split.rs:260-320
# --------------------------------------------------------------------------------------
def generate_unmaped_intervals(intervals: Sequence[tuple[int, int]], total_length: int):
    if len(intervals) == 0:
        return [(0, total_length)]
    res = []
    cur = 0
    for s, e in intervals:
        if cur < s:
            res.append((cur, s))
        cur = e
    # usize arithmetic: total_length - 1 underflows for total_length == 0 (debug panic);
    # unreachable on the hot path (reads shorter than 150 are passed through).
    if cur < total_length - 1:
        res.append((cur, total_length - 1))
    return res


def remove_intervals_and_keep_left(seq: bytes, intervals: Sequence[tuple[int, int]]):
    ivs = sorted(intervals, key=lambda r: r[0])  # stable, like par_sort_by
    sel = generate_unmaped_intervals(ivs, len(seq))
    parts = []
    for s, e in sel:
        if s < len(seq):
            parts.append(seq[s:e])
        else:
            raise EncodingError(f"The interval is invalid: {s}..{e}")
    return parts, sel


# --------------------------------------------------------------------------------------
# S6  split_noodle_records_by_remove_intervals   src/output/split.rs:60-136,171-226
# --------------------------------------------------------------------------------------
def split_records_by_remove_intervals(
    seq: bytes,
    rid: bytes,
    qual: bytes,
    target: Sequence[tuple[int, int]],
    min_chop_read_length: int = MIN_CHOPED_SEQ_LEN,
    id_annotation: bool = True,
    chop_type: str = "all",
) -> list[tuple[bytes, bytes, bytes]]:
    seqs, sel = remove_intervals_and_keep_left(seq, target)
    quals, _ = remove_intervals_and_keep_left(qual, target)
    if len(seqs) != len(quals) or any(len(a) != len(b) for a, b in zip(seqs, quals)):
        raise EncodingError("The sequence and quality scores have different lengths")
    ids = [rid + b"|%d:%d" % (s, e) for s, e in sel]
    n_before = len(seqs)
    keep = [i for i in range(n_before) if len(seqs[i]) >= min_chop_read_length]
    ids = [ids[i] for i in keep]
    seqs = [seqs[i] for i in keep]
    quals = [quals[i] for i in keep]
    cur = "terminal" if n_before == 1 else "internal"
    if (
        (chop_type == "terminal" and cur == "internal")
        or (chop_type == "internal" and cur == "terminal")
        or (len(seqs) > 0 and len(seqs[0]) == len(seq))
    ):
        return [(rid, seq, qual)]
    out = []
    for i, s, q in zip(ids, seqs, quals):
        if id_annotation:
            i = i + (b"|T" if cur == "terminal" else b"|I")
        out.append((i, s, q))
    return out


def split_records_by_intervals(seq: bytes, rid: bytes, qual: bytes, target: Sequence[tuple[int, int]]):
    """split_noodle_records_by_intervals (src/output/split.rs:138-169), the `--ocq` form: one record per
    interval, id `{id}|{start}:{end}`, no annotation, no length filter."""
    return [(rid + b"|%d:%d" % (s, e), seq[s:e], qual[s:e]) for s, e in target]


# --------------------------------------------------------------------------------------
# S4  chop gating                                        src/bin/predict.rs:130-192
# --------------------------------------------------------------------------------------
def chop_read(
    predict: Predict,
    fq_id: bytes,
    fq_seq: bytes,
    fq_qual: bytes,
    smooth_window_size=21,
    min_interval_size=13,
    approved_interval_number=20,
    max_process_intervals=4,
    min_read_length_after_chop=20,
    chop_type="all",
    output_chopped_seqs=False,
    fq_header: bytes | None = None,
):
    """One iteration of `process_chunk`.  Returns the output records as (definition line, seq, qual).

    `fq_header`: the record's whole definition line (name + description).  Records that pass through are
    `fq_record.clone()` (predict.rs:146-164) and the noodles FASTQ writer emits `@name description`; records
    built by the split functions carry `Definition::new(id, "")`, i.e. the bare name."""
    orig = [(fq_header if fq_header is not None else fq_id, fq_seq, fq_qual)]
    if predict.seq_len() < MIN_READ_LEN:
        return orig
    ivs = predict.smooth_and_select_intervals(smooth_window_size, min_interval_size, approved_interval_number)
    if len(ivs) > max_process_intervals or len(ivs) == 0:
        return orig
    if predict.seq_len() != len(fq_qual):
        return orig
    if output_chopped_seqs:
        return split_records_by_intervals(predict.seq.encode(), fq_id, fq_qual, ivs)
    return split_records_by_remove_intervals(
        predict.seq.encode(), fq_id, fq_qual, ivs, min_read_length_after_chop, True, chop_type
    )
