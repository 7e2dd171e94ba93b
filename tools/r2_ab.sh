#!/bin/bash
# A/B of library builds on C3 (GPU box): tools/time_lib.py loads csrc/<name>
python bench.py --no-cpu-baseline --steps 20 --cache /tmp/c3.seg > /dev/null 2>&1
for lib in "$@"; do python tools/time_lib.py $lib /tmp/c3.seg; python tools/time_lib.py $lib /tmp/c3.seg; done
