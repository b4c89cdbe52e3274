"""CPU tests: the C-ABI library loads and exports every declared symbol, fails loudly without a
GPU, and the host-side mirrors of the PyO3 helpers reproduce the reference's known answers."""
import ctypes
import os

import numpy as np
import pytest

from deepchopper_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol(hip_lib):
    syms = _lib.header_symbols()
    assert len(syms) >= 20 and "dc_forward" in syms and "dc_tokenize_batch" in syms
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in include/deepchopper_hip.h but not exported"
    # and every bound signature is declared in the header
    assert set(_lib._SIGNATURES) <= set(syms)
    assert hip_lib.dc_version() >= 100


def test_no_gpu_fails_loudly(hip_lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.DeepChopperHipError):
        _lib.require_gpu()
    cfg = _lib.ModelConfig(d_model=256, d_inner=1024, n_layer=4, vocab=16, filter_order=64, emb_dim=5,
                           short_filter_order=3, max_seq_len=32770, head_hidden=1024, num_class=2, gelu_tanh=1,
                           ln_eps=1e-5, filter_shift=0.05)
    w = _lib.ModelWeights()
    h = ctypes.c_void_p()
    rc = hip_lib.dc_model_create(ctypes.byref(cfg), ctypes.byref(w), ctypes.byref(h))
    assert rc == 3 and b"no CPU fallback" in hip_lib.dc_last_error()


def test_argument_validation_without_gpu(hip_lib):
    rc = hip_lib.dc_tokenize_batch(None, None, None, None, 4, 100, 33, 10, 1, 1, None, None, None, None, None)
    assert rc == 1 and b"null pointer" in hip_lib.dc_last_error()
    assert hip_lib.dc_tokenize_batch(None, None, None, None, 0, 100, 33, 10, 1, 1, None, None, None, None, None) == 0
    rc = hip_lib.dc_kmer_validate(None, 1, 1, 3, 40, b"ACGT", 4, 1)
    assert rc == 1 and b"k must be" in hip_lib.dc_last_error()


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "deepchopper_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_host_helpers_match_reference_kats(kats):
    from deepchopper_amd import rs_api as r

    for c in kats["generate_kmers"]["cases"]:
        assert r.generate_kmers(c["bases"], c["k"]) == c["expected"]
    for c in kats["generate_kmers_table"]["cases"]:
        assert r.generate_kmers_table(c["bases"], c["k"]) == c["expected"]
    for c in kats["to_kmer_target_region"]["cases"]:
        if "error" in c:
            with pytest.raises(Exception, match=c["error"]):
                r.to_kmer_target_region(c["start"], c["end"], c["k"], c["seq_len"])
        else:
            assert list(r.to_kmer_target_region(c["start"], c["end"], c["k"], c["seq_len"])) == c["expected"]
    for c in kats["to_original_targtet_region"]["cases"]:
        assert list(r.to_original_targtet_region(c["start"], c["end"], c["k"])) == c["expected"]
    for c in kats["seq_to_kmers_and_offset"]["cases"]:
        got = r.seq_to_kmers_and_offset(c["seq"], c["k"], c["overlap"])
        assert len(got) == c["expected_len"] and [got[0][0], list(got[0][1])] == c["first"]
        assert [got[1][0], list(got[1][1])] == c["second"]
    with pytest.raises(Exception, match="shorter than the k-mer size"):
        r.seq_to_kmers_and_offset("AT", 3, True)
    for c in kats["vectorize_target"]["cases"]:
        if "error" in c:
            with pytest.raises(Exception, match=c["error"]):
                r.vectorize_target(c["start"], c["end"], c["length"])
        else:
            assert r.vectorize_target(c["start"], c["end"], c["length"]) == c["expected"]
    for c in kats["vectorize_targets"]["cases"]:
        if "error" in c:
            with pytest.raises(Exception, match=c["error"]):
                r.vectorize_targets(c["targets"], c["length"])
        else:
            assert r.vectorize_targets(c["targets"], c["length"]) == c["expected"]
    for c in kats["parse_target_from_id"]["cases"]:
        assert r.parse_target_from_id(c["id"]) == [v for se in c["expected"] for v in se]
    for c in kats["id_list2seq"]["cases"]:
        assert r.id_list2seq(c["ids"]) == c["expected"]
    for c in kats["remove_intervals_and_keep_left"]["cases"]:
        parts, _ = r.remove_intervals_and_keep_left(c["seq"], [tuple(x) for x in c["intervals"]])
        assert parts == c["expected"]
    for c in kats["generate_unmaped_intervals"]["cases"]:
        assert r.generate_unmaped_intervals([tuple(x) for x in c["intervals"]], c["total_length"]) == [tuple(x) for x in c["expected"]]
    table = {v: k for k, v in r.generate_kmers_table("ATCGN", 3).items()}
    assert r.kmerids_to_seq([table_id for table_id in (0, 1)], table) == "AAAT"
    with pytest.raises(Exception, match="k-mer id is invalid"):
        r.kmerids_to_seq([999], table)
    assert r.kmers_to_seq(["ACG", "CGT", "GTA"]) == "ACGTA"
    assert r.splite_qual_by_offsets([10, 20, 30, 41], [(0, 2), (2, 2), (1, 4)]) == [15, 0, 30]
    assert r.default.QUAL_OFFSET == 33 and r.default.KMER_SIZE == 3 and r.default.BASES == "ATCGN"


def test_id_encoding_and_sharding():
    from deepchopper_amd.sharding import merge_by_id, shard_range
    from deepchopper_amd.tokenizer import encode_ids_int8
    from oracle import dc_oracle as o

    row = encode_ids_int8("read_000000001", False)
    assert row.dtype == np.int8 and row.shape == (256,)
    assert np.array_equal(row.astype(np.int64), o.encode_id("read_000000001", False))
    with pytest.raises(RuntimeError, match="int8"):
        encode_ids_int8("x" * 128, False)
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert merge_by_id([{"a": 1}, {"b": 2}, {"a": 1}]) == {"a": 1, "b": 2}
    with pytest.raises(ValueError):
        merge_by_id([{"a": 1}, {"a": 2}])


def test_fastq_reader(tmp_path, golden_dir):
    from deepchopper_amd.fastq import iter_fastq
    from deepchopper_amd import synth

    recs = list(iter_fastq(os.path.join(golden_dir, "twenty_five_records.fq")))
    assert len(recs) == 25 and all(len(s) == len(q) for _, s, q in recs)
    seq, qual, ids = synth.synth_reads_fixed(5, 60, seed=42)
    p = tmp_path / "x.fq.gz"
    import gzip
    with gzip.open(p, "wb") as fh:
        fh.write(synth.to_fastq(seq, qual, ids))
    got = list(iter_fastq(str(p)))
    assert [g[0] for g in got] == ids and got[2][1] == seq[2].tobytes()
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r1\nACGT\n+\nII\n")
    with pytest.raises(ValueError, match="length mismatch"):
        list(iter_fastq(str(bad)))
    empty = tmp_path / "e.fq"
    empty.write_bytes(b"")
    with pytest.raises(ValueError, match="No valid records"):
        list(iter_fastq(str(empty)))


def test_synthetic_generators_are_deterministic():
    from deepchopper_amd import synth

    a = synth.synth_reads_fixed(4, 200, seed=42)
    b = synth.synth_reads_fixed(4, 200, seed=42)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert set(np.unique(a[0]).tolist()) <= set(b"ACGTN") and a[1].min() >= 34 and a[1].max() <= 73
    assert synth.PLANT in a[0][0].tobytes() and synth.PLANT not in a[0][1].tobytes()
    lab = synth.synth_label_stream(5000, seed=43)
    assert lab.dtype == np.int8 and 0 < lab.mean() < 0.5
    assert np.array_equal(lab, synth.synth_label_stream(5000, seed=43))


def test_pip_install_builds_and_ships_the_native_libraries(tmp_path):
    """The installable artefact (reference: Cargo.toml:16,113-117 + pyproject.toml:58 -- the wheel carries the
    native module and the console entry): `pip install .` into an empty prefix runs the build hook (make: hipcc for gfx950 +
    g++), ships both shared objects AND the C-ABI headers inside the package; from there -- source tree NOT on the path --
    `deepchopper --version` works and both libraries export every symbol their header declares."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    target = tmp_path / "site"
    proc = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--quiet",
                           "--target", str(target), root], capture_output=True, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    pkg = target / "deepchopper_amd"
    for rel in ("lib/libdeepchopper_hip.so", "lib/libdeepchopper_io.so", "include/deepchopper_hip.h", "include/deepchopper_io.h"):
        assert (pkg / rel).exists(), rel
    assert (target / "deepchopper" / "__init__.py").exists() and (target / "bin" / "deepchopper").exists()
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "DC_HIP_LIB", "DC_IO_LIB")}
    env["PYTHONPATH"] = str(target)
    script = ("import deepchopper_amd, os; from deepchopper_amd import _lib, _iolib; "
              "assert os.path.dirname(deepchopper_amd.__file__).startswith(%r), deepchopper_amd.__file__; "
              "assert _lib.HEADER_PATH.startswith(%r) and _iolib.HEADER_PATH.startswith(%r); "
              "h = _lib.lib(); io = _iolib.lib(); "
              "m = [s for s in _lib.header_symbols() if not hasattr(h, s)] + [s for s in _iolib.header_symbols() if not hasattr(io, s)]; "
              "assert not m, m; print(len(_lib.header_symbols()), len(_iolib.header_symbols()))") % (str(target), str(target), str(target))
    out = subprocess.run([sys.executable, "-c", script], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    n_hip, n_io = (int(x) for x in out.stdout.split())
    assert n_hip >= 30 and n_io >= 17
    ver = subprocess.run([sys.executable, str(target / "bin" / "deepchopper"), "--version"], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=300)
    assert ver.returncode == 0 and ver.stdout.startswith("DeepChopper Version:"), ver.stderr[-2000:]


def test_write_pt_batches_cuts_every_reference_batch_out_of_a_fused_launch(tmp_path):
    """predict.write_pt_batches on host tensors (no GPU): a fused launch of 3 reference batches (5 + 5 + 2 rows, left padding
    per batch) becomes three `{rank}_{index}.pt` files in the CustomWriter layout (callbacks.py:17-25): rows of the batch,
    columns from the batch's own padding on, dtypes int64 / float32, id rows [len, truncated, ascii...], and every file
    holds its own storage only (a view would drag the whole launch into each file)."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from deepchopper_amd.predict import write_pt_batches

    n, stride = 12, 2000
    subs, cols = [5, 5, 2], [0, 600, 1250]
    g = torch.Generator().manual_seed(0)
    res = {"logits": torch.randn(n, stride, 2, generator=g),
           "tok": {"labels": torch.randint(-100, 1, (n, stride), dtype=torch.int8, generator=g),
                   "input_ids": torch.randint(0, 12, (n, stride), dtype=torch.int64, generator=g),
                   "input_quals": torch.rand(n, stride, generator=g),
                   "status": torch.tensor([0] * 11 + [1], dtype=torch.int32)},
           "sub_sizes": subs, "sub_cols": cols}
    ids = [f"read_{i}" for i in range(n)]
    for pool in (None, ThreadPoolExecutor(2)):
        out = tmp_path / ("serial" if pool is None else "threads")
        pending = []
        assert write_pt_batches(str(out), 3, 7, res, ids, pool, pending) == 3
        for f in pending:
            f.result()
        assert sorted(os.listdir(out)) == ["3_7.pt", "3_8.pt", "3_9.pt"]
        r0 = 0
        for i, (k, c0) in enumerate(zip(subs, cols)):
            path = out / f"3_{7 + i}.pt"
            d = torch.load(path, weights_only=True)
            assert d["prediction"].shape == (k, stride - c0, 2) and d["prediction"].dtype == torch.float32
            assert torch.equal(d["prediction"], res["logits"][r0: r0 + k, c0:])
            assert d["target"].dtype == torch.int64 and torch.equal(d["target"], res["tok"]["labels"][r0: r0 + k, c0:].to(torch.int64))
            assert d["seq"].dtype == torch.int64 and torch.equal(d["seq"], res["tok"]["input_ids"][r0: r0 + k, c0:])
            assert torch.equal(d["qual"], res["tok"]["input_quals"][r0: r0 + k, c0:])
            assert d["id"].shape == (k, 256) and d["id"][0, 0] == len(ids[r0]) and d["id"][0, 2] == ord("r")
            assert int(d["id"][-1, 1]) == (1 if r0 + k == n else 0)  # the truncated flag of the last read
            assert os.path.getsize(path) < 1.2 * (k * (stride - c0) * 28) + 8192  # its own storage, not the launch's
            r0 += k


def _fake_launch(n, stride, subs, cols, seed=0, ids_dtype="uint8"):
    import torch

    g = torch.Generator().manual_seed(seed)
    return {"logits": torch.randn(n, stride, 2, generator=g),
            "tok": {"labels": torch.randint(-100, 2, (n, stride), dtype=torch.int8, generator=g),
                    "input_ids": torch.randint(0, 12, (n, stride), dtype=torch.int64, generator=g).to(getattr(torch, ids_dtype)),
                    "input_quals": torch.rand(n, stride, generator=g),
                    "status": torch.tensor([0] * (n - 1) + [1], dtype=torch.int32)},
            "sub_sizes": subs, "sub_cols": cols}


def test_native_pt_writer_writes_what_torch_save_writes(tmp_path):
    """predict.PtWriter (libdeepchopper_io: dc_pt_*) against predict.write_pt_batches (torch.save) on the same fused launch:
    the same file names; `torch.load` gives the same keys in the same order, dtypes, shapes and values; the pickle stream inside
    (`data.pkl`: what candle's pickle::read_all parses in the reference's chop, src/smooth/predict.rs:264-317) is byte for byte
    torch.save's; every entry's CRC-32 checks out and every storage starts on a 64-byte boundary; a second launch continues
    the numbering.  Batches of one row, of one column and a 70 000-column row (sizes pickled as BININT1 / BININT2 / BININT)."""
    import zipfile

    import torch

    from deepchopper_amd.predict import PtWriter, write_pt_batches

    launches = [(12, 2000, [5, 5, 2], [0, 600, 1250]), (4, 70001, [1, 3], [0, 70000]), (3, 300, [3], [0])]
    first = 7
    with PtWriter(threads=3, max_inflight=1) as w:
        for li, (n, stride, subs, cols) in enumerate(launches):
            res = _fake_launch(n, stride, subs, cols, seed=li, ids_dtype="uint8" if li != 2 else "int64")
            ids = [f"read_{li}_{i}/x" for i in range(n)]
            assert w.submit(str(tmp_path / "native"), 3, first, res, ids) == len(subs)
            assert write_pt_batches(str(tmp_path / "torch"), 3, first, res, ids) == len(subs)
            first += len(subs)
    names = sorted(os.listdir(tmp_path / "torch"))
    assert names == sorted(os.listdir(tmp_path / "native")) and len(names) == 6
    for name in names:
        a = torch.load(tmp_path / "torch" / name, weights_only=True)
        b = torch.load(tmp_path / "native" / name, weights_only=True)
        assert list(a) == list(b) == ["prediction", "target", "seq", "qual", "id"]
        for k in a:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].stride() == b[k].stride(), (name, k)
            assert torch.equal(a[k], b[k]), (name, k)
        with zipfile.ZipFile(tmp_path / "torch" / name) as za, zipfile.ZipFile(tmp_path / "native" / name) as zb:
            assert zb.testzip() is None
            stem = name[:-3]
            assert za.read(f"{stem}/data.pkl") == zb.read(f"{stem}/data.pkl")
            assert [i.filename for i in za.infolist()] == [i.filename for i in zb.infolist()]
            assert zb.read(f"{stem}/byteorder") == za.read(f"{stem}/byteorder") and zb.read(f"{stem}/version") == za.read(f"{stem}/version")
            raw = open(tmp_path / "native" / name, "rb").read()
            for i in zb.infolist():
                assert i.compress_type == zipfile.ZIP_STORED
                nlen, xlen = int.from_bytes(raw[i.header_offset + 26: i.header_offset + 28], "little"), int.from_bytes(raw[i.header_offset + 28: i.header_offset + 30], "little")
                assert (i.header_offset + 30 + nlen + xlen) % 64 == 0, i.filename


def test_native_pt_writer_errors(tmp_path):
    """An id the reference's int8 cast cannot hold (tokenizer.py:91: longer than 127 characters, or a non-ASCII character)
    fails with the reference's message at submit; an unwritable directory fails at the wait / close."""
    import pytest

    from deepchopper_amd.predict import PtWriter

    res = _fake_launch(2, 50, [2], [0])
    with PtWriter(threads=1) as w:
        with pytest.raises(RuntimeError, match="cannot be converted to type int8"):
            w.submit(str(tmp_path / "a"), 0, 0, res, ["ok", "x" * 128])
        with pytest.raises(RuntimeError, match="cannot be converted to type int8"):
            w.submit(str(tmp_path / "a"), 0, 0, res, ["ok", "naïve"])
        assert w.submit(str(tmp_path / "a"), 0, 0, res, ["ok", "x" * 127]) == 1
    w = PtWriter(threads=1)
    (tmp_path / "b").mkdir()
    (tmp_path / "b" / "0_0.pt").mkdir()  # the file's name is taken by a directory
    w.submit(str(tmp_path / "b"), 0, 0, res, ["r0", "r1"])
    with pytest.raises(RuntimeError, match="0_0.pt"):
        w.close()


def test_native_pt_loader_reads_torch_save_and_native_files(tmp_path):
    """rs_api.LabelStore.load (libdeepchopper_io: dc_pt_load) = load_predicts_from_batch_pts' argmax + mask
    (src/smooth/predict.rs:211-317) on files torch.save wrote and on files the native writer wrote: ids, truncation flags and
    labels equal numpy's argmax over the unmasked positions (ties -> label 0, candle's first maximum); general masks, not
    only left padding; a file that is no `.pt` is reported in the reference's words and skipped."""
    import numpy as np
    import torch

    from deepchopper_amd import rs_api
    from deepchopper_amd.predict import PtWriter

    rng = np.random.default_rng(3)
    want = {}
    d = tmp_path / "pred" / "0"
    d.mkdir(parents=True)
    for fi, (B, L) in enumerate([(12, 300), (1, 17), (5, 2049)]):
        logits = rng.standard_normal((B, L, 2)).astype(np.float32)
        logits[:, ::7, 1] = logits[:, ::7, 0]  # exact ties
        target = np.zeros((B, L), np.int64)
        for r in range(B):
            p = int(rng.integers(0, L - 2))
            target[r, :p] = -100
            target[r, -1] = -100
            target[r, p + int(rng.integers(0, L - p - 1))] = -100  # a masked position in the middle
        ids = [f"read{fi}_{r}|x" for r in range(B)]
        idm = np.zeros((B, 256), np.int64)
        for r, s in enumerate(ids):
            idm[r, 0], idm[r, 1] = len(s), int(r == B - 1)
            idm[r, 2: 2 + len(s)] = [ord(c) for c in s]
            want[s] = ((logits[r, :, 1] > logits[r, :, 0]).astype(np.int8)[target[r] != -100], r == B - 1)
        torch.save({"prediction": torch.from_numpy(logits), "target": torch.from_numpy(target),
                    "seq": torch.zeros(B, L, dtype=torch.int64), "qual": torch.zeros(B, L), "id": torch.from_numpy(idm)}, d / f"0_{fi}.pt")
    # the same through the native writer (left padding only: what the pipeline produces)
    res = _fake_launch(12, 500, [5, 7], [100, 0], seed=9)
    res["tok"]["labels"][:] = 0
    res["tok"]["labels"][:5, :130] = -100
    res["tok"]["labels"][:, -1] = -100
    names = [f"native_{i}" for i in range(12)]
    with PtWriter(threads=2) as w:
        w.submit(str(d), 1, 0, res, names, truncated=np.zeros(12, bool))
    lg = res["logits"].numpy()
    for r, s in enumerate(names):
        c0 = 100 if r < 5 else 0
        keep = res["tok"]["labels"][r, c0:].numpy() != -100
        want[s] = ((lg[r, c0:, 1] > lg[r, c0:, 0]).astype(np.int8)[keep], False)
    (d / "broken.pt").write_bytes(b"not a zip at all" * 10)
    store = rs_api.LabelStore.load([str(tmp_path / "pred")], threads=3)
    assert len(store) == len(want) == 30
    for rid, (lab, trunc) in want.items():
        i = store.index[rid]
        assert np.array_equal(store.labels[store.offsets[i]: store.offsets[i + 1]], lab), rid
        assert bool(store.truncated[i]) == trunc
        p = store.predict(rid)
        assert p.prediction == lab.tolist() and p.is_truncated == trunc


def test_native_pt_loader_survives_damaged_files(tmp_path):
    """dc_pt_load on damaged `.pt` files -- truncated anywhere, bytes flipped in the ZIP structures, sizes that do not agree --
    neither crashes nor invents reads: a file either loads with every label in {0, 1} and a plausible id, or is reported as
    failed (and then goes to torch.load, which fails too: the reference's "load pt ... fail" line)."""
    import numpy as np

    from deepchopper_amd import rs_api
    from deepchopper_amd.predict import PtWriter

    res = _fake_launch(6, 120, [6], [0], seed=4)
    res["tok"]["labels"][:] = 0
    good = tmp_path / "good"
    with PtWriter(threads=1) as w:
        w.submit(str(good), 0, 0, res, [f"r{i}" for i in range(6)], truncated=np.zeros(6, bool))
    raw = (good / "0_0.pt").read_bytes()
    rng = np.random.default_rng(0)
    bad = tmp_path / "bad"
    bad.mkdir()
    cases = 0
    for cut in (0, 10, 21, 22, 100, len(raw) // 2, len(raw) - 30, len(raw) - 1):
        (bad / f"cut{cut}.pt").write_bytes(raw[:cut])
        cases += 1
    tail = len(raw) - 1400  # the central directory and the end record live here
    for k in range(60):
        b = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(tail, len(raw))) if k % 2 else int(rng.integers(0, len(raw)))
            b[pos] = int(rng.integers(0, 256))
        (bad / f"flip{k}.pt").write_bytes(bytes(b))
        cases += 1
    store = rs_api.LabelStore.load([str(bad)], threads=3)
    assert len(store.ids) <= 6 * cases
    assert store.labels.size == 0 or (set(np.unique(store.labels).tolist()) <= {0, 1})
    assert all(len(i) <= 254 for i in store.ids)


def test_pt_writer_flush_block_waits_for_one_launch_only():
    """PtWriter._flush(block=True) (what `_bound` calls when too much is in flight) synchronises on the OLDEST staged launch and
    hands on only those whose copy has already landed; round 5's loop reused one name for its flag and for the launch's first
    file index, so with a non-zero index it blocked on every staged launch.  `_bound()` brings the total back to `max_inflight`
    with one wait."""
    import numpy as np

    from deepchopper_amd.predict import PtWriter

    class Event:
        """A copy that lands when it is WAITED for (polled a second time: predict.wait_event polls query() between sleeps;
        the loop's own "has it landed?" look is one poll)."""

        def __init__(self):
            self.done, self.waited, self.polls = False, 0, 0

        def query(self):
            if self.done:
                return True
            self.polls += 1
            if self.polls >= 2:
                self.done, self.waited = True, self.waited + 1
            return self.done

        def synchronize(self):
            self.waited += 1
            self.done = True

    class Pinned:
        def data_ptr(self):
            return 0

    class Lib:
        def __init__(self):
            self.calls = []

        def dc_pt_submit_packed(self, h, d, rank, first_index, n_files, *rest):
            self.calls.append((rank, first_index, n_files))
            return 0

        def dc_pt_wait(self, h, ticket):
            return 0

        def dc_pt_close(self, h):
            return 0

    class Io:
        def __init__(self):
            self._lib = Lib()

        def lib(self):
            return self._lib

        def raise_for(self, rc):
            assert rc == 0

    w = PtWriter(threads=1, max_inflight=3)
    real_io, real_h = w._io, w._h
    w._io = Io()
    try:
        i32, i64 = np.zeros(1, np.int32), np.zeros(1, np.int64)
        events = [Event() for _ in range(5)]
        for k, ev in enumerate(events):  # first file index 7, 8, ... : never zero
            args = (b"/nowhere", 0, 7 + k, i32, i32, i32, i64, 0, 0, b"", i64, np.zeros(1, np.uint8))
            w._staged.append((ev, args, None, Pinned()))
        w._flush(block=True)
        assert [e.waited for e in events] == [1, 0, 0, 0, 0] and len(w._staged) == 4
        assert w._io.lib().calls == [(0, 7, 1)]
        events[2].done = True  # its copy has landed: goes along with the oldest (unwaited), and the loop stops at the fourth
        w._flush(block=True)
        assert [e.waited for e in events] == [1, 1, 0, 0, 0] and len(w._staged) == 2
        assert w._io.lib().calls == [(0, 7, 1), (0, 8, 1), (0, 9, 1)]
        w._pending.clear()
        w._staged.extend((ev, (b"/nowhere", 0, 20 + k, i32, i32, i32, i64, 0, 0, b"", i64, np.zeros(1, np.uint8)), None, Pinned())
                         for k, ev in enumerate([Event(), Event()]))
        n_calls = len(w._io.lib().calls)
        w._bound()  # 4 staged, nothing pending, max_inflight 3 -> exactly one launch handed on (and its files waited for)
        assert len(w._staged) == 3 and len(w._pending) == 0 and len(w._io.lib().calls) == n_calls + 1
        assert [e.waited for e in events[3:]] == [1, 0]
    finally:
        w._staged.clear()
        w._pending.clear()
        w._io, w._h = real_io, real_h
        w.close()


def test_makefile_knows_every_header_the_host_library_includes():
    """`make` must rebuild libdeepchopper_io.so when any header it includes changes (round 5: fast_deflate.h was not a
    prerequisite, so editing the DEFLATE encoder alone left a stale library that build(force=False) would ship)."""
    import re
    import subprocess

    csrc = os.path.join(ROOT, "deepchopper_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j4"], check=True, capture_output=True)
    assert subprocess.run(["make", "-C", csrc, "-q"], capture_output=True).returncode == 0
    included = set()
    for src in ("fastq_io.cpp", "pt_io.cpp"):
        included |= {os.path.basename(h) for h in re.findall(r'#include "([^"]+)"', open(os.path.join(csrc, src)).read())}
    assert {"fast_deflate.h", "fast_inflate.h", "deepchopper_io.h"} <= included
    for h in sorted(included):
        path = os.path.join(csrc, h) if os.path.exists(os.path.join(csrc, h)) else os.path.join(ROOT, "include", h)
        st = os.stat(path)
        try:
            os.utime(path, None)
            assert subprocess.run(["make", "-C", csrc, "-q", "../lib/libdeepchopper_io.so"], capture_output=True).returncode == 1, h
        finally:
            os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns))
    assert subprocess.run(["make", "-C", csrc, "-q"], capture_output=True).returncode == 0
