#!/usr/bin/env python3
"""Average rocprofv3 counter values per launch of one kernel.
usage: pmc_summary.py <kernel-name-substring> <pass-name>=<rocprofv3 output dir> ...
Prints csv rows: pass,counter,avg_per_launch,launches"""
import csv
import glob
import os
import sys
from collections import defaultdict

kern = sys.argv[1]
print("pass,counter,avg_per_launch,launches")
for arg in sys.argv[2:]:
    name, d = arg.split("=", 1)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> value
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if kern not in row.get("Kernel_Name", ""):
                    continue
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for c, per in sorted(acc.items()):
        vals = list(per.values())
        print(f"{name},{c},{sum(vals) / len(vals):.1f},{len(vals)}")
