#!/usr/bin/env python3
"""The derived planes as a measured choice: C3 with every plane (scan_win_kernel on post_id16 / win_off), without the window planes
(scan_range_kernel on post_rel16), and without post_rel16 either (scan_range_kernel unpacking the blob's delta streams itself), and -- round 6 -- with the window TABLES but neither post_id16 nor
post_rel16: decode_id16_kernel unpacks the batch's terms from the blob into the batch's scratch plane ahead of scan_win_kernel
(kernel ms = both kernels).  The switches are read at index creation (vbm25_tuning_set: win_planes, rel16_plane, id16_plane)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for label, tune in (("all planes (default)", "win=1"), ("no window planes", "win_planes=0"), ("no window planes, no post_rel16", "win_planes=0,rel16_plane=0"),
                    ("window tables, no post_id16, no post_rel16", "id16_plane=0,rel16_plane=0")):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "100", "--extra-budget-s", "0", "--tune", tune],
                         capture_output=True, text=True, timeout=600).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    c, r = d["config"], d["roofline"]
    print(f"{label:44s} index {c['index_hbm_bytes'] / 1e9:5.2f} GB ({c['index_hbm_bytes'] / c['postings']:.2f} B/posting)  {r['kernel']:18s} {r['kernel_ms']:.4f} ms = "
          f"{r['frac']:.3f} of 8 TB/s  {d['value']:9.0f} q/s resident, {c['host_buffer_inclusive_qps_per_gpu']:9.0f} q/s through host buffers (pipelined)  "
          f"64-query sample bit-exact vs the oracle: {bool(c.get('verified_sample'))}", flush=True)
