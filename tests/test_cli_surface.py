"""The reference's command surface, flag by flag (CPU: argument parsing only).

`deepchopper predict` flags and defaults: deepchopper/cli.py:66-90; `deepchopper chop`: cli.py:155-169; the chop binary
`deepchopper-chop` (what the reference's `chop` shells out to): src/bin/predict.rs:19-78; console entry
`deepchopper = "deepchopper.cli:app"`: pyproject.toml:58."""
import os

import pytest

from deepchopper_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _help(argv, fn=cli.main):
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as e:
        fn(argv)
    assert e.value.code == 0
    return buf.getvalue()


def test_predict_flags_and_defaults():
    h = _help(["predict", "--help"])
    for flag in ("--gpus", "-g", "--output", "-o", "--batch-size", "-b", "--workers", "-w", "--model", "-m", "--limit-batches",
                 "--max-sample", "--verbose", "-v"):
        assert flag in h, flag
    import argparse

    # defaults of the reference: -b 12, -w 0, -m rna002 (cli.py:70-84).  Its `-g 0` (CPU) does not exist in this build: an
    # omitted --gpus parses as None = "one GPU, with a notice" (test_default_invocation_takes_one_gpu_with_a_notice)
    ns = {}
    orig = argparse.ArgumentParser.parse_args

    def grab(self, args=None, namespace=None):
        r = orig(self, args, namespace)
        ns.update(vars(r))
        raise SystemExit(0)

    argparse.ArgumentParser.parse_args = grab
    try:
        with pytest.raises(SystemExit):
            cli.main(["predict", "reads.fq"])
    finally:
        argparse.ArgumentParser.parse_args = orig
    assert (ns["gpus"], ns["batch_size"], ns["workers"], ns["model"], ns["output"]) == (None, 12, 0, "rna002", None)
    assert ns["limit_batches"] is None and ns["max_sample"] is None and ns["threads"] == 4  # --chop's measured default


def test_chop_flags_and_defaults():
    h = _help(["chop", "--help"])
    for flag in ("--smooth-window", "--min-interval-size", "--approved-intervals", "--max-process-intervals", "--min-read-length",
                 "--output-chopped", "--chop-type", "--threads", "-t", "--verbose", "-v", "--output", "-o", "--max-batch"):
        assert flag in h, flag
    p = cli.argparse.ArgumentParser()
    cli._add_chop_options(p, prefix_output=True)
    d = vars(p.parse_args([]))
    assert (d["smooth_window"], d["min_interval_size"], d["approved_intervals"], d["max_process_intervals"], d["min_read_length"],
            d["chop_type"], d["threads"], d["ocq"]) == (21, 13, 20, 4, 20, "all", 2, False)  # cli.py:158-165
    assert vars(p.parse_args(["--output-chopped"]))["ocq"] and vars(p.parse_args(["--ocq"]))["ocq"]


def test_chop_binary_flags():
    h = _help(["--help"], fn=cli.chop_main)
    for flag in ("--pdt", "--fq", "-t", "--threads", "-m", "--max-batch-size", "-s", "--smooth-window-size", "--mis", "-a",
                 "--approved-interval-number", "--mpi", "--mcr", "--ocq", "--ct", "-o", "--output-prefix", "--chunk-size", "-d"):
        assert flag in h, flag


def test_version_and_console_entries():
    assert "DeepChopper Version" in _help(["--version"])
    import tomli

    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
        scripts = tomli.load(fh)["project"]["scripts"]
    assert scripts["deepchopper"] == "deepchopper_amd.cli:main" and scripts["deepchopper-chop"] == "deepchopper_amd.cli:chop_main"
    mod, fn = scripts["deepchopper-chop"].split(":")
    assert callable(getattr(__import__(mod, fromlist=[fn]), fn))


def test_no_cpu_mode_is_said_out_loud(tmp_path, capsys):
    fq = tmp_path / "r.fq"
    fq.write_text("@r\nACGT\n+\nIIII\n")
    assert cli.main(["predict", str(fq), "-g", "0"]) == 1
    assert "no CPU fallback" in capsys.readouterr().err


def test_default_invocation_takes_one_gpu_with_a_notice(tmp_path, capsys, monkeypatch):
    """`deepchopper predict reads.fq` -- the reference's documented command (deepchopper/cli.py:66-90: `--gpus` defaults to 0 =
    CPU, :126-138) -- must not be an error here: the omitted flag means one GPU, said in one line on stderr.  (No GPU in
    the CPU suite: the run is cut where the first HIP call would be made, behind the GPU-count gate.)"""
    from deepchopper_amd import _lib

    class Reached(Exception):
        pass

    def stop():
        raise Reached

    monkeypatch.setattr(_lib, "require_gpu", stop)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    fq = tmp_path / "r.fq"
    fq.write_text("@r\nACGT\n+\nIIII\n")
    with pytest.raises(Reached):
        cli.main(["predict", str(fq)])
    err = capsys.readouterr().err
    assert "no CPU mode in this build; using GPU 0" in err and "Error" not in err


def test_import_name_of_the_reference_resolves_to_this_package():
    """`import deepchopper` / `deepchopper.deepchopper` (the PyO3 module, src/python.rs:879-958) / `deepchopper.models`
    / `python -m deepchopper`: the names a caller of the reference uses for this path."""
    import subprocess
    import sys

    import deepchopper
    import deepchopper_amd
    from deepchopper.deepchopper import Predict, default, encode_qual, majority_voting  # noqa: F401
    from deepchopper.models import DeepChopper

    assert DeepChopper is deepchopper_amd.DeepChopper is deepchopper.DeepChopper
    assert deepchopper.cli is cli and deepchopper.deepchopper is deepchopper_amd.rs_api
    # `from .deepchopper import *` (deepchopper/__init__.py:4): the compiled module's names at top level
    for name in ("encode_qual", "majority_voting", "smooth_label_region", "load_predicts_from_batch_pts", "Predict",
                 "encode_fq_path_to_tensor", "write_fq", "write_fq_parallel", "RecordData"):
        assert getattr(deepchopper, name) is getattr(deepchopper_amd.rs_api, name), name
    assert (default.QUAL_OFFSET, default.BASES, default.KMER_SIZE, default.VECTORIZED_TARGET) == (33, "ATCGN", 3, False)
    assert not hasattr(deepchopper, "train")  # outside this build: the name is simply absent
    out = subprocess.run([sys.executable, "-m", "deepchopper", "--version"], cwd=ROOT, capture_output=True, text=True, check=True)
    assert out.stdout.strip() == f"DeepChopper Version: {deepchopper_amd.__version__}"


def test_write_fq_and_write_fq_parallel(tmp_path):
    """src/output/writefq.rs:17-37 (plain, "@id" without description) and :222-250 (multithreaded BGZF)."""
    import gzip

    from deepchopper_amd import rs_api
    from deepchopper_amd.fastq import FastqReader

    recs = [rs_api.RecordData("r1", "ACGT", "IIII"), ("r2|0:2|T", "AC", "#5"), rs_api.RecordData("r3", "N" * 70000, "!" * 70000)]
    want = b"".join(b"@%s\n%s\n+\n%s\n" % tuple(x.encode() for x in ((r.id, r.seq, r.qual) if isinstance(r, rs_api.RecordData) else r))
                    for r in recs)
    plain = tmp_path / "a.fq"
    rs_api.write_fq(recs, plain)
    assert plain.read_bytes() == want
    bg = tmp_path / "a.fq.gz"
    rs_api.write_fq_parallel(recs, bg, 3)
    assert gzip.open(bg, "rb").read() == want
    raw = bg.read_bytes()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"  # BGZF extra field
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # EOF block
    with FastqReader(str(bg)) as rd:
        names = [n for c in rd.chunks(10, 1 << 20) for n in c.names()]
    assert names == ["r1", "r2|0:2|T", "r3"]
    r = recs[0]
    r.seq = "TTTT"
    assert (r.id, r.seq, r.qual) == ("r1", "TTTT", "IIII") and "TTTT" in repr(r)


def test_highlight_targets_like_the_pyo3_export():
    """src/vis.rs:5-23: bold magenta over the target ranges (colored's `.magenta().bold()`), wrapped at `text_width` VISIBLE
    characters (textwrap breaks the one long word a read is; escape sequences take no room).  The reference's own test only
    prints: parity unpinned, behaviour restated from the two crates."""
    import re

    from deepchopper_amd import rs_api

    seq = "ATGCACTGACTGACATGCACTGACTGAC"
    assert rs_api.highlight_targets(seq, [(0, 3), (10, 13)]) == "\x1b[1;35mATG\x1b[0mCACTGAC\x1b[1;35mTGA\x1b[0mCATGCACTGACTGAC"
    strip = lambda x: re.sub(r"\x1b\[[0-9;]*m", "", x)  # noqa: E731
    for w in (1, 5, 7, 27, 28, 80):
        h = rs_api.highlight_targets(seq, [(0, 3), (10, 13), (20, 28)], w)
        lines = h.split("\n")
        assert strip(h).replace("\n", "") == seq
        assert all(len(strip(l)) == w for l in lines[:-1]) and 0 < len(strip(lines[-1])) <= w
        assert h.count("\x1b[1;35m") == 3 == h.count("\x1b[0m")
    assert rs_api.highlight_targets("", []) == "" and rs_api.highlight_targets("ACGT", [(2, 2)]) == "ACGT"
    with pytest.raises(IndexError):
        rs_api.highlight_targets("ACGT", [(2, 9)])
    import deepchopper

    assert deepchopper.highlight_targets is rs_api.highlight_targets  # exported under the reference's import name too
