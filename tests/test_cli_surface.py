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

    # defaults of the reference: -g 0, -b 12, -w 0, -m rna002 (cli.py:70-84)
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
    assert (ns["gpus"], ns["batch_size"], ns["workers"], ns["model"], ns["output"]) == (0, 12, 0, "rna002", None)
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
