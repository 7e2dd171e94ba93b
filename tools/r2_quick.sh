#!/bin/bash
# quick C3 timing (GPU box); builds the cache once per box
python bench.py --no-cpu-baseline --steps 100 --cache /tmp/c3.seg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C3',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
