"""CPU oracle for the `deepchopper predict` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is imported, linked or executed by the
product package `deepchopper_amd/`; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may use it, and there only as the checker / the timed
CPU baseline, never as the thing shipped.

Contents
--------
dc_oracle.py    numpy restatement of the integer / byte paths (tokenise T1-T5, k-mer encode
                K1-K5, smoothing + interval arithmetic S0-S6).  Pinned by the reference's own
                known-answer tests (tests/golden/reference_kats.json, transcribed from the
                `#[test]` vectors cited there) and by the reference's committed `.pt` fixtures.
forward_ref.py  torch-fp32 CPU restatement of the HyenaDNA token-classification forward
                (F1-F3).  The head (F3) is pinned by golden vectors generated from the
                reference's own `head.py` (tests/golden/make_head_golden.py).  The backbone
                (F2) is third-party code that the reference pulls from the HF Hub and that could
                not be loaded in this pipeline: **parity unpinned** for the backbone logits.
"""
