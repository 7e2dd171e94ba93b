import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import vectorchord_bm25_amd as vb
from bench import make_queries
dbg=int(sys.argv[1]); nt=int(sys.argv[2])
seg = vb.Segment.synth(300_000, 33_000, mean_len=100, len_mode=1, seed=5)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, 33_000, 64, nt, seed=nt, zipf_s=0.0)
vb.set_tuning("fused", 0); vb.set_tuning("team_dbg", dbg)
b = vb.Batch(gix, 64, len(terms), 10); b.set_queries(terms, off); print("route", b.debug_route(), flush=True)
b.run(); h, n = b.fetch(); print("ok dbg", dbg, "nt", nt, n[:4], b.debug_counts(), flush=True)
