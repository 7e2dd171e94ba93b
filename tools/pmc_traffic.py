#!/usr/bin/env python3
"""HBM bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, both reported in KiB), as the
entry profiles/pmc_traffic.json holds -- keyed by the first 16 hex digits of the sha256 of the libvbm25.so the passes ran on
(bench.py reports roofline.traffic only when the library it loaded is that build).
FETCH_SIZE is doubled (gfx950 correction of MI355X_MICROARCH.md, HBM section: the counter tallies 128-byte requests at 64 bytes);
WRITE_SIZE is taken as reported (uncalibrated there; it is 1-2 % of the total on these kernels).
usage: pmc_traffic.py <workload> <kernel-substring> <fetch pass dir> <write pass dir>   -> one JSON object on stdout"""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def avg(d, kern, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if kern in row.get("Kernel_Name", "") and row["Counter_Name"] == counter:
                per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return (sum(per.values()) / len(per), len(per)) if per else (None, 0)


workload, kern, fetch_dir, write_dir = sys.argv[1:5]
lib = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", "libvbm25.so")
sha16 = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
f, nf = avg(fetch_dir, kern, "FETCH_SIZE")
w, nw = avg(write_dir, kern, "WRITE_SIZE")
out = {sha16: {workload: {"kernel": kern, "fetch_size_kib": f, "write_size_kib": w, "launches": [nf, nw],
                          "hbm_bytes_per_launch": None if f is None or w is None else int(2 * f * 1024 + w * 1024),
                          "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2 x FETCH_SIZE KiB + WRITE_SIZE KiB"}}}
print(json.dumps(out, indent=1))
