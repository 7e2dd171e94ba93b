"""N > 1 path on CPU: two gloo processes shard a query batch, "search" their shard (the
oracle stands in for the GPU here) and all-gather the hit records; the gathered records must
equal the single-process result.  Covers shard_bounds / shard_queries / gather_hits."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries

rank, world, nq, k = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(sys.argv[3]), 10
dist.init_process_group("gloo", rank=rank, world_size=world)
c = make_corpus(3000, 300, seed=3, length="lognormal", mean_len=40)
oix = orc.OracleIndex.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                            c["post_doc"], c["post_tf"])
terms, off = make_queries(c, nq, 4, seed=7)
t, o = vb.sharded.shard_queries(terms, off, world, rank)
hits, nh, _ = oix.search_batch(t, o, k, mode="brute", threads=1)
local = torch.from_numpy(np.frombuffer(hits.tobytes(), dtype=np.int64).copy())
allw = vb.sharded.gather_hits(local, nq, k)
if rank == 0:
    np.save(sys.argv[2], allw.numpy())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nq", [16, 17])
def test_two_rank_gather_equals_single_process(nq):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from corpus import make_corpus, make_queries
    import vectorchord_bm25_amd as vb

    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER)
        out = os.path.join(d, "out.npy")
        port = 29500 + os.getpid() % 1000 + nq
        subprocess.check_call(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
             "--master-addr", "127.0.0.1", "--master-port", str(port), script, ROOT, out, str(nq)],
            timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        got = np.load(out)
    c = make_corpus(3000, 300, seed=3, length="lognormal", mean_len=40)
    oix = orc.OracleIndex.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                                c["post_doc"], c["post_tf"])
    terms, off = make_queries(c, nq, 4, seed=7)
    ref, _, _ = oix.search_batch(terms, off, 10, mode="brute", threads=1)
    assert got.tobytes() == ref.tobytes()
    # shard arithmetic
    for world in (1, 2, 3, 8):
        b = [vb.sharded.shard_bounds(nq, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == nq and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


BENCH_WORKER = r"""
# bench.run() under gloo: the test supplies the CPU stand-in for the GPU batch object (bench.py itself knows
# nothing about the oracle outside its --verify / cpu_baseline legs)
import os, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import bench
import orc
import vectorchord_bm25_amd as vb


class OracleScorer:
    def __init__(self, seg, nq, k):
        self.oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
        self.k = k
        self.words = torch.zeros(nq * k * 3, dtype=torch.int64)

    def set_queries(self, terms, off):
        self.terms, self.off = terms, off

    def run(self, stream=None):
        self.hits, self.n_hits, _ = self.oix.search_batch(self.terms, self.off, self.k, mode="brute", threads=1)
        self.words.copy_(torch.from_numpy(np.frombuffer(self.hits.tobytes(), dtype=np.int64).copy()))

    def fetch(self):
        return self.hits, self.n_hits


bench.run(sys.argv[2:], scorer_factory=OracleScorer, backend="gloo")
"""


def test_bench_step_two_ranks_gloo():
    """bench.py's own distributed step (rank 0 makes the batches, broadcast, shard, search, gather to rank 0) under
    gloo, with an oracle-backed scorer injected from here in place of the GPU batch objects."""
    import json

    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "bench_worker.py")
        open(script, "w").write(BENCH_WORKER)
        port = 29500 + (os.getpid() + 77) % 1000
        out = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
             "--master-addr", "127.0.0.1", "--master-port", str(port), script, ROOT,
             "--gpus", "2", "--workload", "C1", "--steps", "3", "--warmup", "1", "--batches", "2", "--verify"],
            timeout=600, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["config"]["scatter_ms"] > 0 and d["config"]["gather_ms"] > 0
    assert d["config"]["verified_bit_exact_vs_oracle"]["queries"] == 128
    assert d["config"]["batches_rotated"] == 2
    assert d["config"]["workload"].startswith("C4")
    # every rank's own figures (what makes a scaling line readable): one value per rank, the slowest rank's step is the line's
    pr = d["config"]["per_rank"]
    assert sorted(pr) == ["gather_exposed_ms", "gather_ms", "kernel_ms", "ms_per_step"] and all(len(v) == 2 for v in pr.values())
    assert max(pr["ms_per_step"]) <= d["ms_per_step"] * 1.0001 + 1e-3 and min(pr["ms_per_step"]) > 0
    assert pr["gather_ms"][0] > 0 and pr["kernel_ms"] == [0.0, 0.0]  # (no GPU here: nothing timed by HIP events)
    assert "roofline" not in d and "cpu_baseline" not in d  # nothing measured on a GPU here


def test_bench_has_no_cpu_scorer():
    """The bench's product path may not reach the oracle: only --verify and cpu_baseline import it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "OracleScorer" not in src and "VBM25_BENCH_BACKEND" not in src
    assert src.count("import orc") == 1  # inside oracle_index(), used by --verify and cpu_baseline only


SUBGROUP_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import vectorchord_bm25_amd as vb

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
group = dist.new_group([1, 2])      # shard ranks 0, 1 of this group are global ranks 1, 2
nq, k = 5, 2
if rank in (1, 2):
    g = dist.get_rank(group)
    lo, hi = vb.sharded.shard_bounds(nq, 2, g)
    local = torch.arange(lo * k * 3, hi * k * 3, dtype=torch.int64)
    for dst in (0, 1):
        got = vb.sharded.gather_to_root(local, nq, k, dst=dst, group=group)
        if g == dst:
            assert torch.equal(got, torch.arange(nq * k * 3, dtype=torch.int64)), got
        else:
            assert got is None
    if rank == 1:
        open(sys.argv[2], "w").write("ok")
dist.barrier()
dist.destroy_process_group()
"""


def test_gather_to_root_in_a_subgroup():
    """dst of gather_to_root is a rank of the group; a group that does not start at global rank 0 must work."""
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(SUBGROUP_WORKER)
        out = os.path.join(d, "ok")
        port = 29500 + (os.getpid() + 191) % 1000
        r = subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
             "--master-addr", "127.0.0.1", "--master-port", str(port), script, ROOT, out],
            timeout=300, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(out).read() == "ok"
