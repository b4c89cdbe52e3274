#!/bin/bash
# Throughput on the other single-GPU-sized BASELINE configs (run on the GPU box).
cd "$(dirname "$0")/.."
python bench.py --steps 8 --warmup 2 --batch 2048 --read-len 1000 --cpu-seconds 0 2>/dev/null | python tools/bench_stages.py "C2 1kb b2048" | cut -c1-90
python bench.py --steps 8 --warmup 2 --batch 4096 --read-len 1000 --cpu-seconds 0 2>/dev/null | python tools/bench_stages.py "C2 1kb b4096" | cut -c1-90
python bench.py --steps 8 --warmup 2 --batch 1024 --read-len 2000 --cpu-seconds 0 2>/dev/null | python tools/bench_stages.py "C3 2kb b1024" | cut -c1-90
python bench.py --steps 8 --warmup 2 --batch 2048 --read-len 512 --cpu-seconds 0 2>/dev/null | python tools/bench_stages.py "C1 512bp b2048" | cut -c1-90
