#!/bin/bash
# knob sweep on C3 (GPU box): prints kernel ms per setting
python - <<'PY'
import os, subprocess, json, sys
def run(env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "30", "--cache", "/tmp/c3.seg"],
                         env=e, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(env, "kernel_ms", d["roofline"]["kernel_ms"], "ms_per_step", d["ms_per_step"], flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
run({})
for g in (128, 256, 384):
    run({"VBM25_RANGE_GRID": str(g)})
for it in (512, 2048, 4096):
    run({"VBM25_CUR_ITEMS": str(it)})
PY
