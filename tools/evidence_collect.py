#!/usr/bin/env python3
"""Copies what tools/evidence.sh left under gpurun_out/<sub> into profiles/ under the round's names and enters the build's
HBM traffic (C3, C5) into profiles/pmc_traffic.json.   usage: evidence_collect.py <sub> <tag, e.g. r5>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sub, tag = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", sub), os.path.join(ROOT, "profiles")
names = {
    "bench_c3_full.json": f"{tag}_c3_bench_default_run.json",
    "bench_c3_driver_flags.json": f"{tag}_c3_bench_steps20_warmup5.json",
    "bench_c3_k100.json": f"{tag}_c3_bench_k100.json",
    "c3_kernel_stats.csv": f"{tag}_c3_kernel_stats.csv",
    "pmc_summary.csv": f"{tag}_c3_pmc_scan_win_kernel.csv",
    "c3_phase_timers.txt": f"{tag}_c3_phase_timers.txt",
    "c5_kernel_stats.csv": f"{tag}_c5_kernel_stats.csv",
    "c5_phase_timers.txt": f"{tag}_c5_phase_timers.txt",
    "pmc5_summary.csv": f"{tag}_c5_pmc_scan_dense_kernel.csv",
    "c3_win_ablation_raw.txt": f"{tag}_c3_win_ablation_raw.txt",
    "bench_c3z.json": f"{tag}_c3z_bench.json",
    "c3z_kernel_stats.csv": f"{tag}_c3z_kernel_stats.csv",
    "mixed_batch.txt": f"{tag}_mixed_batch.txt",
    "multi_enqueue.txt": f"{tag}_multi_enqueue.txt",
    "stream_host_time.txt": f"{tag}_stream_host_time.txt",
    "index_planes.txt": f"{tag}_index_planes.txt",
}
for a, b in names.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
        print(b)
    else:
        print("missing", a)
sha = open(os.path.join(src, "lib_sha16.txt")).read().strip()
tp = os.path.join(dst, "pmc_traffic.json")
table = json.load(open(tp))
for f in ("pmc_traffic_c3.json", "pmc_traffic_c5.json"):
    d = json.load(open(os.path.join(src, f)))
    assert list(d) == [sha], (list(d), sha)
    table.setdefault(sha, {}).update(d[sha])
json.dump(table, open(tp, "w"), indent=1)
print("pmc_traffic.json:", sha, sorted(table[sha]))
