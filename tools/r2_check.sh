#!/bin/bash
# round-2 iteration helper (GPU box): parity suite, then C3 with the range kernel and with the cursor kernel
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2/tests.txt
cat gpurun_out/r2/tests.txt
python bench.py --no-cpu-baseline --steps 50 --cache /tmp/c3.seg > gpurun_out/r2/bench_range.json 2> gpurun_out/r2/bench_range.err
tail -3 gpurun_out/r2/bench_range.err; cat gpurun_out/r2/bench_range.json
VBM25_RANGE=0 python bench.py --no-cpu-baseline --steps 50 --cache /tmp/c3.seg > gpurun_out/r2/bench_cursor.json 2> gpurun_out/r2/bench_cursor.err
cat gpurun_out/r2/bench_cursor.json
