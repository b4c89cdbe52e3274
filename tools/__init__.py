"""Developer tooling (benchmarks, profiling drivers); `bench_legs` is imported by bench.py."""
