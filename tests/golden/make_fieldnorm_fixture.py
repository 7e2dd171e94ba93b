#!/usr/bin/env python3
"""Writes tests/golden/fieldnorm_table.json from the reference's own constant table
(crates/bm25/src/bm25.rs:15-272).  Needs /root/reference; the committed JSON travels."""
import json
import os
import re

src = open("/root/reference/crates/bm25/src/bm25.rs").read()
body = src[src.index("FIELDNORM_TO_LENGTH"):]
body = body[body.index("= [") + 3:body.index("];")]
vals = [int(x.replace("_", "")) for x in re.findall(r"[0-9_]+", body)]
assert len(vals) == 256
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fieldnorm_table.json")
json.dump(vals, open(path, "w"))
print("wrote", path, vals[:5], vals[-1])
