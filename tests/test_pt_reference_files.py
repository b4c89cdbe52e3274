"""The native `.pt` reader (libdeepchopper_io: dc_pt_load) on the REFERENCE's own prediction files.

`tests/golden/pt/chunk{0,1}/{0,1,2}.pt` are the reference's committed `CustomWriter` outputs (its `tests/data/eval/`; data, kept
byte for byte).  The reference pins its loader on chunk0/0.pt: `test_load_predict`, src/smooth/predict.rs:323-330, `len == 12`.
chunk1 was written by an older writer that stored `target` and `id` as the collator's int8 (`torch CharStorage`,
deepchopper/models/llm/tokenizer.py:89-91) instead of CustomWriter's int64 (deepchopper/models/callbacks.py:21-22); candle's
pickle reader knows no CharStorage, so the reference's own chop could not read those three -- this reader parses `data.pkl`
(dtype, shape, stride, storage offset per key) and reads both layouts.
"""
import os
import shutil
import zipfile

import numpy as np
import pytest
import torch

from deepchopper_amd import rs_api

PT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pt")
FILES = [(c, i) for c in (0, 1) for i in (0, 1, 2)]


def _expect(path):
    """torch.load -> first-max argmax over the unmasked positions, ids, truncation flags (src/smooth/predict.rs:264-317)."""
    d = torch.load(path, map_location="cpu", weights_only=True)
    lg, tg, idm = d["prediction"].numpy(), d["target"].numpy().astype(np.int64), d["id"].numpy().astype(np.int64)
    out = {}
    for r in range(lg.shape[0]):
        n = int(idm[r, 0])
        rid = "".join(chr(c) for c in idm[r, 2: 2 + n])
        out[rid] = ((lg[r, :, 1] > lg[r, :, 0]).astype(np.int8)[tg[r] != -100], bool(idm[r, 1] != 0))
    return d, out


@pytest.mark.parametrize("chunk,i", FILES)
def test_native_reader_on_the_reference_files(chunk, i):
    path = os.path.join(PT, f"chunk{chunk}", f"{i}.pt")
    d, want = _expect(path)
    assert d["target"].dtype == (torch.int64 if chunk == 0 else torch.int8)  # the two writer generations
    store = rs_api.LabelStore.load([path], threads=1)
    assert store.fallback_files == [], "the native reader must recognise the reference's own files"
    assert len(store) == 12 == len(want)  # src/smooth/predict.rs:328
    for rid, (lab, trunc) in want.items():
        k = store.index[rid]
        assert np.array_equal(store.labels[store.offsets[k]: store.offsets[k + 1]], lab), rid
        assert bool(store.truncated[k]) == trunc


def test_native_reader_on_the_reference_directories():
    """load_predicts_from_batch_pts over the two directories as `deepchopper chop <dir>` walks them: 36 reads each, all native."""
    for chunk in (0, 1):
        store = rs_api.LabelStore.load(os.path.join(PT, f"chunk{chunk}"), threads=3)
        assert len(store) == 36 and store.fallback_files == []


def _rewrite(src, dst, edit):
    """Copy a `.pt` with one entry's bytes edited (sizes unchanged; CRCs recomputed unless the edit says otherwise)."""
    with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, "w", zipfile.ZIP_STORED) as zout:
        for info in zin.infolist():
            zout.writestr(info.filename, edit(info.filename, zin.read(info.filename)))


def test_native_reader_refuses_what_it_does_not_understand(tmp_path):
    """nothing is decoded from byte sizes alone.  Same sizes, different meaning -> not recognised (the torch.load
    route takes the file): a float `target` storage, a transposed (non-contiguous) `prediction` is READ through its strides,
    a payload whose CRC does not match the directory is refused, an unknown global in data.pkl is refused."""
    src = os.path.join(PT, "chunk0", "2.pt")
    d, want = _expect(src)
    # (a) non-contiguous prediction written by torch.save of a permuted view: same bytes count, different strides
    perm = dict(d)
    perm["prediction"] = d["prediction"].permute(2, 0, 1).contiguous().permute(1, 2, 0)
    assert not perm["prediction"].is_contiguous()
    torch.save(perm, tmp_path / "strided.pt")
    store = rs_api.LabelStore.load([str(tmp_path / "strided.pt")], threads=1)
    assert store.fallback_files == []
    for rid, (lab, _t) in want.items():
        k = store.index[rid]
        assert np.array_equal(store.labels[store.offsets[k]: store.offsets[k + 1]], lab), rid
    # (b) target stored as float64: the byte size of an int64 target, refused natively, read by the fallback
    flt = dict(d)
    flt["target"] = d["target"].to(torch.float64)
    torch.save(flt, tmp_path / "float_target.pt")
    store = rs_api.LabelStore.load([str(tmp_path / "float_target.pt")], threads=1)
    assert [os.path.basename(f) for f in store.fallback_files] == ["float_target.pt"]
    # (c) a flipped payload byte under an unchanged directory CRC
    raw = bytearray(open(src, "rb").read())
    with zipfile.ZipFile(src) as z:
        info = next(i for i in z.infolist() if i.filename.endswith("/data/0"))
    raw[info.header_offset + 4096] ^= 0x40
    (tmp_path / "crc.pt").write_bytes(bytes(raw))
    store = rs_api.LabelStore.load([str(tmp_path / "crc.pt")], threads=1)
    assert [os.path.basename(f) for f in store.fallback_files] == ["crc.pt"] and len(store) == 0
    # (d) data.pkl naming a class this reader does not know
    _rewrite(src, tmp_path / "glob.pt", lambda name, b: b.replace(b"torch\nLongStorage", b"torch\nHalfStorage") if name.endswith("data.pkl") else b)
    store = rs_api.LabelStore.load([str(tmp_path / "glob.pt")], threads=1)
    assert [os.path.basename(f) for f in store.fallback_files] == ["glob.pt"]


def test_hostile_pickles_are_refused_not_followed(tmp_path):
    """A `data.pkl` of hundreds of thousands of nested tuples, or of megabytes, is not torch.save's ~500-byte stream: refused
    before it can recurse through the parser's value destructors or grow its stack without bound."""
    src = os.path.join(PT, "chunk0", "2.pt")
    for name, pk in (("deep", b"\x80\x02" + b"(" * 200000 + b"t" * 200000 + b"."), ("huge", b"\x80\x02" + b"N0" * 900000 + b".")):
        dst = tmp_path / f"{name}.pt"
        _rewrite(src, dst, lambda n, b, pk=pk: pk if n.endswith("data.pkl") else b)
        store = rs_api.LabelStore.load([str(dst)], threads=1)
        assert len(store) == 0 and [os.path.basename(f) for f in store.fallback_files] == [f"{name}.pt"]


def test_storage_offset_and_shared_storage(tmp_path):
    """Tensors that are views into one larger storage (non-zero storage offset): read where the pickle says they are."""
    src = os.path.join(PT, "chunk1", "1.pt")
    d, want = _expect(src)
    big = torch.cat([torch.full((7,), 3, dtype=d["target"].dtype), d["target"].reshape(-1)])
    v = dict(d)
    v["target"] = big[7:].view(d["target"].shape)
    assert v["target"].storage_offset() == 7
    torch.save(v, tmp_path / "offset.pt")
    store = rs_api.LabelStore.load([str(tmp_path / "offset.pt")], threads=1)
    assert store.fallback_files == []
    for rid, (lab, _t) in want.items():
        k = store.index[rid]
        assert np.array_equal(store.labels[store.offsets[k]: store.offsets[k + 1]], lab), rid


def test_native_reader_agrees_with_torch_load_or_refuses(tmp_path):
    """Property: for any dict torch.save writes under the five keys -- float / double predictions, int64 / int32 / int16 / int8 /
    uint8 / bool targets and ids, contiguous, transposed, sliced or offset views, extra keys, other key orders -- the native
    reader either gives exactly the labels / ids / flags of `torch.load` -> first-max argmax + mask, or reports the file as
    not recognised.  It never decodes something else."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    int_types = [torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8]
    counter = {"native": 0, "refused": 0}

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 6), st.integers(3, 40), st.sampled_from([torch.float32, torch.float64, torch.float16]),
           st.sampled_from(int_types), st.sampled_from(int_types + [torch.float32]), st.sampled_from(["c", "t", "slice", "offset"]),
           st.booleans(), st.integers(0, 2 ** 31 - 1))
    def run(B, L, pdt, tdt, idt, view, extra, seed):
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(B, L, 2, generator=g)
        logits[:, ::5, 1] = logits[:, ::5, 0]  # exact ties -> label 0
        target = torch.zeros(B, L, dtype=torch.int64)
        target[:, : L // 3] = -100
        target[:, -1] = -100
        ids = torch.zeros(B, 256, dtype=torch.int64)
        names = []
        for r in range(B):
            name = f"r{seed % 1000}_{r}"
            names.append(name)
            ids[r, 0], ids[r, 1] = len(name), r % 2
            ids[r, 2: 2 + len(name)] = torch.tensor([ord(c) for c in name])
        p = logits.to(pdt)
        if view == "t":
            p = p.permute(2, 0, 1).contiguous().permute(1, 2, 0)
        elif view == "slice":
            p = torch.cat([p, p], dim=1)[:, :L]
        elif view == "offset":
            p = torch.cat([torch.zeros(5, dtype=pdt), p.reshape(-1)])[5:].view(B, L, 2)
        tt = target.to(tdt) if tdt != torch.uint8 else target.clamp(min=0).to(tdt)  # (uint8 cannot hold -100: nothing is masked)
        d = {"prediction": p, "target": tt, "seq": torch.zeros(B, L, dtype=torch.int64), "qual": torch.zeros(B, L), "id": ids.to(idt)}
        if extra:
            d = {"note": torch.arange(3), **d}
        path = tmp_path / f"case_{seed}.pt"
        torch.save(d, path)
        store = rs_api.LabelStore.load([str(path)], threads=1)
        if store.fallback_files:
            counter["refused"] += 1
            assert len(store) == 0 or True  # (the torch.load route may or may not take it here: no GPU for its summary step)
            os.remove(path)
            return
        counter["native"] += 1
        back = torch.load(path, map_location="cpu", weights_only=True)
        lg, tg = back["prediction"].float().numpy(), back["target"].numpy().astype(np.int64)
        assert len(store) == B
        for r, name in enumerate(names):
            k = store.index[name]
            want = (lg[r, :, 1] > lg[r, :, 0]).astype(np.int8)[tg[r] != -100]
            assert np.array_equal(store.labels[store.offsets[k]: store.offsets[k + 1]], want), (name, view, pdt, tdt)
            assert bool(store.truncated[k]) == bool(r % 2)
        os.remove(path)

    run()
    assert counter["native"] > 5 and counter["refused"] > 5, counter  # both outcomes were exercised
