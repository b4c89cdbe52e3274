"""Single-read API behind the reference's web demo: `predict(text)` of
deepchopper/ui/main.py:53-106 without the gradio front end.

The reference builds a one-record HF dataset, runs Lightning's `trainer.predict` on it, then smooths the labels
and cuts the sequence through its PyO3 helpers.  Here the record goes through the same GPU pipeline as every
other batch (`PredictPipeline.predict_reads`): labels, smoothed intervals and kept segments come back from
one fused launch each.
"""
from __future__ import annotations

from .predict import PredictPipeline, SmoothParams
from .rs_api import remove_intervals_and_keep_left


def parse_fq_record(text: str):
    """ui/main.py:21-37: records of four lines; the whole first line (with '@') is the id, as in the reference."""
    lines = text.strip().split("\n")
    for i in range(0, len(lines), 4):
        content = lines[i: i + 4]
        if len(content) != 4:
            raise ValueError("not enough values to unpack (a FASTQ record has four lines)")
        record_id, seq, _, qual = (c.strip("\r") for c in content)
        assert len(seq) == len(qual)  # noqa: S101  (the reference asserts the same)
        yield {"id": record_id, "seq": seq, "qual": qual}


def predict(text: str, smooth_window_size: int = 21, min_interval_size: int = 13, approved_interval_number: int = 20,
            max_process_intervals: int = 8, model=None):
    """Returns (smooth_interval_json, highlighted_text) exactly as ui/main.py:53-106 does:
    [{"start": s, "end": e}, ...] for the predicted adapter intervals and the read cut into
    (segment, "ada" | None) pieces.  `model`: a deepchopper_amd.DeepChopper (default: from_pretrained)."""
    if model is None:
        from .model import DeepChopper

        model = DeepChopper.from_pretrained("yangliz5/deepchopper")
    recs = list(parse_fq_record(text))
    assert len(recs) == 1  # noqa: S101  (ui/main.py:78)
    pipe = PredictPipeline(model, smooth=SmoothParams(window=smooth_window_size, min_interval=min_interval_size,
                                                      approved=approved_interval_number))
    rec = recs[0]
    out, _ = pipe.predict_reads([rec["id"]], [rec["seq"].upper().encode()], [rec["qual"].encode()])
    targets = out[0]["intervals"]
    smooth_interval_json: list[dict[str, int]] = []
    highlighted_text: list[tuple[str, str | None]] = []
    if not targets or len(targets) > max_process_intervals:
        return smooth_interval_json, highlighted_text
    seq = rec["seq"]
    _sel, selected_intervals = remove_intervals_and_keep_left(seq, targets)
    total = sorted([tuple(x) for x in selected_intervals] + [tuple(x) for x in targets])
    smooth_interval_json.extend({"start": s, "end": e} for s, e in targets)
    tset = set(tuple(x) for x in targets)
    highlighted_text.extend((seq[s:e], "ada" if (s, e) in tset else None) for s, e in total)
    return smooth_interval_json, highlighted_text
