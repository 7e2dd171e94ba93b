#!/bin/bash
# round-2 iteration helper (GPU box): parity suite, C3 bench, phase profile
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2/tests.txt
cat gpurun_out/r2/tests.txt
python bench.py --no-cpu-baseline --steps 50 --cache /tmp/c3.seg > gpurun_out/r2/bench_range.json 2> gpurun_out/r2/bench_range.err
tail -3 gpurun_out/r2/bench_range.err; python -c "
import json;d=json.load(open('gpurun_out/r2/bench_range.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
