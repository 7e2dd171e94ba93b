#!/usr/bin/env python3
"""bench.py -- queries/sec of the batched BM25 top-k path on MI355X.

A "step" is one pass of the hot path (plan -> posting scan -> merge) over one batch of
synthetic queries already resident in HBM.  Default workload = BASELINE.json configs[2]
("C3"): 10M docs / 30k vocab, 1024 five-term queries, top-10, one MI355X.  With --gpus N
every rank holds a replica of the index and its own 1024 queries (weak scaling); the only
collective is the all-gather of the per-rank top-k (RCCL).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (n_docs, vocab, mean_len, len_mode, zipf_s, queries/GPU, terms/query, k)
    "C1": (1_000, 1_000, 100, 1, 0.0, 64, 3, 10),
    "C2": (1_000_000, 30_000, 100, 1, 0.0, 1, 3, 10),
    "C3": (10_000_000, 30_000, 100, 1, 0.0, 1024, 5, 10),
    "C5": (50_000_000, 100_000, 100, 1, 1.0, 1024, 10, 100),
}


class _DevArray:
    """__cuda_array_interface__ view of a device buffer owned by libvbm25 (for torch)."""

    def __init__(self, ptr, n_int64):
        self.__cuda_array_interface__ = {"shape": (n_int64,), "typestr": "<i8", "data": (ptr, False),
                                         "version": 2}


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def make_queries(seg, vocab, nq, nterms, seed, zipf_s):
    rng = np.random.default_rng(seed)
    if zipf_s > 0:
        p = 1.0 / np.arange(1, vocab + 1) ** zipf_s
        p /= p.sum()
    rows = []
    while len(rows) < nq:
        if zipf_s <= 0:
            toks = rng.choice(vocab, nterms, replace=False)
        else:  # the first nterms DISTINCT draws, in draw order (np.unique alone would keep the lowest ranks)
            draws = rng.choice(vocab, nterms * 4, p=p)
            _, first = np.unique(draws, return_index=True)
            toks = draws[np.sort(first)[:nterms]]
        ids = seg.token_terms(toks.astype(np.uint32))
        ids = ids[ids != 0xffffffff]
        if len(ids) == nterms:  # queries whose tokens are all present in the vocab
            rows.append(np.sort(ids))
    terms = np.concatenate(rows).astype(np.uint32)
    off = (np.arange(nq + 1) * nterms).astype(np.uint32)
    return terms, off


def cpu_baseline(seg, terms, off, k, budget_s=20.0):
    """Faithful C++ restatement of the reference's Block-WAND search (oracle/), one query per
    thread, on a bounded sample of the same batch.  NOT the Rust binary (no rustc here)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    cores = usable_cpus()
    nq = len(off) - 1
    probe = min(nq, 8)
    _, _, t_probe = oix.search_batch(terms[:off[probe]], off[:probe + 1], k, mode="wand", threads=1)
    per_q = max(t_probe / probe, 1e-6)
    sample = int(min(nq, max(cores, budget_s * cores / per_q)))
    reps = 1
    if sample == nq:  # whole batch is cheap: repeat it
        reps = int(max(1, min(50, budget_s * cores / (per_q * nq))))
    t = 0.0
    for _ in range(reps):
        _, _, dt = oix.search_batch(terms[:off[sample]], off[:sample + 1], k, mode="wand", threads=cores)
        t += dt
    return {"value": round(sample * reps / t, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{sample} of the batch's {nq} queries x{reps}, Block-WAND restatement "
                      f"(oracle/), one query per thread; 1-thread rate {1.0 / per_q:.1f} q/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=0, help="override queries per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--cache", default="", help="segment file: load if present, else build and save")
    args = ap.parse_args()

    import torch

    import vectorchord_bm25_amd as vb

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    # VBM25_BENCH_FORCE_DIST=1 runs the RCCL code path even with one rank (GPU test of the N>1 path)
    use_dist = world > 1 or os.environ.get("VBM25_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[args.workload]
    if args.queries:
        nq = args.queries
    threads = args.build_threads or usable_cpus()
    t0 = time.perf_counter()
    cache = args.cache
    if use_dist and not cache:
        # one node: rank 0 builds the (deterministic) segment once, the other ranks load it
        import shutil
        need = 6 * n_docs * mean_len  # generous bound of the segment file size
        shm = "/tmp"
        try:
            if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need:
                shm = "/dev/shm"
        except OSError:
            pass
        cache = os.path.join(shm, f"vbm25_{args.workload}_{os.environ.get('MASTER_PORT', '0')}.seg")
        if rank == 0 and os.path.exists(cache):
            os.remove(cache)
        dist.barrier()
    if use_dist and rank != 0:
        dist.barrier()  # rank 0 has written the file
        seg = vb.Segment.load(cache)
    else:
        if cache and os.path.exists(cache):
            seg = vb.Segment.load(cache)
        else:
            seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s,
                                   seed=20260925, threads=threads)
            if cache:
                seg.save(cache)
        if use_dist:
            dist.barrier()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    gix = vb.GpuIndex(seg, device=local_rank)
    t_upload = time.perf_counter() - t0
    terms, off = make_queries(seg, vocab, nq, nterms, seed=1 + rank, zipf_s=zipf_s)
    algo_bytes = sum(seg.query_bytes(terms[off[q]:off[q + 1]], k) for q in range(nq))

    batch = vb.Batch(gix, nq, len(terms), k)
    batch.set_queries(terms, off)
    stream = torch.cuda.current_stream()
    local = None
    if use_dist:
        import ctypes as C
        hp, nh = C.c_void_p(), C.c_void_p()
        vb._lib.check(vb.lib().vbm25_batch_device_results(batch.h, C.byref(hp), C.byref(nh)))
        local = torch.as_tensor(_DevArray(hp.value, nq * k * 3), device=f"cuda:{local_rank}")

    def step():
        batch.run(stream.cuda_stream)
        if use_dist:  # the path's only exchange: every rank gets all top-k lists
            return vb.sharded.gather_hits(local, world * nq, k)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    batch.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_launch = batch.kernel_ms()
    batch.set_timing(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same batch through the host-buffer boundary (upload queries, run, download hits)
    t0 = time.perf_counter()
    for _ in range(5):
        batch.set_queries(terms, off)
        batch.run(stream.cuda_stream)
        hits, n_hits = batch.fetch()
    pcie_qps = 5 * nq / (time.perf_counter() - t0)

    # sanity: results exist and are sorted (full parity lives in tests/ and smoke())
    hits, n_hits = batch.fetch()
    assert (n_hits == k).all() or args.workload in ("C1",), "missing hits"
    s = hits["score"]
    assert (s[:, :-1] >= s[:, 1:]).all()

    result_line = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        out = {
            "metric": "queries/sec top-10 BM25, 10M synthetic docs; achieved HBM GB/s vs peak",
            "value": round(world * nq * args.steps / elapsed, 1),
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n_docs} docs / {vocab} vocab / "
                                   f"{nq} x {nterms}-term queries per GPU / top-{k}",
                       "doc_length": "lognormal(ln 80, 0.6) clamp [8,2000]" if len_mode == 1 else f"fixed {mean_len}",
                       "token_distribution": f"zipf({zipf_s})" if zipf_s > 0 else "uniform",
                       "k1": 1.2, "b": 0.75, "index_hbm_bytes": gix.device_bytes,
                       "postings": int(seg.arrays()["term_df"].astype(np.int64).sum()),
                       "parallelism": f"query-batch data parallel x{world}, index replicated",
                       "build_s": round(t_build, 2), "upload_s": round(t_upload, 2),
                       "host_buffer_inclusive_qps_per_gpu": round(pcie_qps, 1)},
        }
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        try:  # PMC-derived HBM bytes per launch, collected separately (profiles/)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pmc.get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": "scan_cursor_kernel", "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": int(algo_bytes),
                           "kernel_ms": round(kernel_ms, 4), "launches_timed": n_launch}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seg, terms, off, k)
        result_line = json.dumps(out)
    if use_dist:
        if rank == 0 and not args.cache and cache and os.path.exists(cache):
            os.remove(cache)
        dist.destroy_process_group()  # RCCL prints its banner here: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        print(result_line, flush=True)
    if use_dist:
        # librccl prints a version banner to stdout from its exit handlers; the contract is ONE
        # JSON line from rank 0, so leave without running them (everything is flushed and the
        # process group is already destroyed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
