"""Dynamic batching for mixed-length reads (BASELINE config 4).

The forward runs ragged batches natively (every read at its own length, reads of similar length
grouped 32 at a time inside the long-convolution kernel), so batching only has to bound the work
per launch: contiguous runs of reads are cut by a token budget (activations cost ~13.3 KB per
token) and a read-count cap.  Input order is preserved, which keeps outputs in FASTQ order.
"""
from __future__ import annotations

import numpy as np


def token_budget_batches(lengths, max_tokens: int = 2_000_000, max_reads: int = 8192, max_length: int = 32769):
    """Yield (start, stop) index ranges.  `lengths` are base counts; a read costs
    ceil((min(len, max_length-1) + 1) / 32) * 32 token slots in the packed layout."""
    lengths = np.asarray(lengths, dtype=np.int64)
    tok = (np.minimum(lengths, max_length - 1) + 1 + 31) // 32 * 32
    start, acc = 0, 0
    for i, t in enumerate(tok.tolist()):
        if i > start and (acc + t > max_tokens or i - start >= max_reads):
            yield start, i
            start, acc = i, 0
        acc += t
    if start < len(tok):
        yield start, len(tok)
