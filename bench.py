#!/usr/bin/env python3
"""bench.py -- queries/sec of the batched BM25 top-k path on MI355X.

A "step" is one pass of the hot path (posting scan -> merge; the general route adds plan_kernel) over one batch of synthetic
queries already resident in HBM.  Default workload = BASELINE.json configs[2] ("C3"): 10M docs / 30k vocab, 1024 five-term
queries, top-10, one MI355X.  --workload C5 is configs[4] (50M docs / 100k vocab Zipf(1) / 10-term / top-100), C2 the single
3-term query on 1M docs (adds host-inclusive latencies), C1 the plumbing case.

Order of a run (single process, --gpus 1):
  1. corpus generated, flushed and indexed on the device; --batches (default 4) query batches made and handed over;
  2. the HOST-BUFFER figures, before the timed region (config.host_buffer_*): 400 batches through vbm25_stream_* (three in
     flight, pinned staging: queries up, records down every step), then 20 batches one at a time;
  3. W untimed warm-up steps, then EXACTLY K timed steps rotating through the batches, bracketed by synchronisations; kernel
     durations by HIP events on the launch stream (roofline.kernel_ms);
  4. outside the timed region: 64 queries of batch 0 checked bit-exact against the oracle (config.verified_sample; --verify:
     every query of every batch), the CPU baseline (oracle's Block-WAND, T = 1 and T = usable cores), and -- C3 only, under
     --extra-budget-s -- C2 and C5 in the same process (`extra`, C5 with its own CPU baseline).
The rotation is there so that a step's reads come from HBM and not from the previous step's leftovers; profiles/r5_warmup.txt
shows the kernel's duration does not depend on it (1, 4 or 16 batches: the same), and why step 2 comes first.

--gpus N WITHOUT torch.distributed's environment drives N devices from this one process through vbm25_multi_* (one host thread
per device).  Launched by torch.distributed.run (one rank per GPU) the run is BASELINE.json configs[3] ("C4"): rank 0 makes the
N x 1024 queries of every batch, broadcasts the two descriptor arrays (RCCL), every rank keeps its contiguous shard, searches it
against its replica of the index, and the hit records are gathered to rank 0 on a second stream while the next step's scan is
already running.  The broadcast is paid once (scatter_ms); the gather is inside every step (gather_ms, of which
gather_exposed_ms is not hidden).

Prints ONE JSON line (rank 0).  See DESIGN.md section 3.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
C5_CPU_NEED_S = 90.0    # extra.c5.cpu_baseline: download of the 50 M-document segment + the sample (measured: see DESIGN.md)

WORKLOADS = {
    # name: (n_docs, vocab, mean_len, len_mode, zipf_s, queries/GPU, terms/query, k)
    "C1": (1_000, 1_000, 100, 1, 0.0, 64, 3, 10),
    "C2": (1_000_000, 30_000, 100, 1, 0.0, 1, 3, 10),
    "C3": (10_000_000, 30_000, 100, 1, 0.0, 1024, 5, 10),
    "C5": (50_000_000, 100_000, 100, 1, 1.0, 1024, 10, 100),
    # not a BASELINE.json configuration: C3's shape over a Zipf(1) vocabulary -- "the realistic middle" the round-5 review asked for
    # (a batch of mixed sparse and dense queries on the general route; tools/measure.sh c3z, profiles/r6_c3z_*)
    "C3z": (10_000_000, 30_000, 100, 1, 1.0, 1024, 5, 10),
}
METRIC = "queries/sec top-10 BM25, 10M synthetic docs; achieved HBM GB/s vs peak"


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def lib_sha16(vb):
    """First 16 hex digits of the sha256 of the libvbm25.so this process has loaded."""
    import hashlib
    h = hashlib.sha256()
    with open(vb.library_path(), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def committed_traffic(sha16, workload):
    """HBM bytes per launch of the workload's dominant kernel from profiles/pmc_traffic.json -- rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes cannot run inside a timed process, so their summary is committed, keyed by the sha256 of the library
    they were taken on.  None when the loaded library is another build (the number would not be this kernel's)."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return int(table[sha16][workload]["hbm_bytes_per_launch"])
    except Exception:
        return None


def extra_workload(vb, name, budget_s, sha16, with_cpu=True):
    """The other single-GPU configurations of BASELINE.json inside the same driver-timed run, under a wall-clock budget:
    C2 (single 3-term query, 1 M documents: C-ABI latencies) and C5 (50 M documents / 100 k Zipf vocabulary / 10-term / top-100:
    scan_dense_kernel against the roofline, then -- budget permitting -- the CPU Block-WAND restatement on a sample of the
    same batch).  Fewer steps than a dedicated run."""
    import ctypes as C
    t_start = time.perf_counter()
    n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[name]
    need = {"C2": 10.0, "C5": 60.0, "C3z": 30.0}[name]  # generation + index + queries + steps on the round-4 box
    if budget_s < need:
        return {"skipped": f"{need:.0f} s needed, {max(0.0, budget_s):.0f} s of --extra-budget-s left"}
    seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
    t_build = time.perf_counter() - t_start
    gix = vb.GpuIndex(seg)
    out = {"workload": f"{name}: {n_docs} docs / {vocab} vocab / {nq} x {nterms}-term / top-{k}", "build_s": round(t_build, 2),
           "index_hbm_bytes": gix.device_bytes}
    if name == "C2":
        lt, lo = make_queries(seg, vocab, 400, nterms, seed=7, zipf_s=zipf_s)
        one = np.array([0, nterms], dtype=np.uint32)
        L = vb.lib()
        hb = np.zeros((1, k), dtype=vb.HIT_DTYPE)
        nbuf = np.zeros(1, dtype=np.uint32)
        lt = np.ascontiguousarray(lt, dtype=np.uint32)
        base, po = lt.ctypes.data, C.c_void_p(one.ctypes.data)
        ph, pn = C.c_void_p(hb.ctypes.data), C.c_void_p(nbuf.ctypes.data)
        uc = []
        for q in range(400):
            pt = C.c_void_p(base + 4 * int(lo[q]))
            t0 = time.perf_counter()
            rc = L.vbm25_search_batch(gix.h, pt, po, 1, k, ph, pn)
            if q >= 20:
                uc.append(1e6 * (time.perf_counter() - t0))
            assert rc == 0 and nbuf[0] == k
        uc.sort()
        out.update({"c_abi_nq1_us_p50": round(uc[len(uc) // 2], 1), "c_abi_nq1_us_p99": round(uc[int(len(uc) * 0.99)], 1),
                    "queries_timed": len(uc),
                    "includes": "vbm25_search_batch(nq = 1) through the bare C entry point: query hand-over, ONE launch, stream synchronisation"})
    else:
        import torch
        nb, steps, warmup = 2, 6, 2
        shards = [make_queries(seg, vocab, nq, nterms, seed=1 + bi, zipf_s=zipf_s) for bi in range(nb)]
        algo = [sum(seg.query_bytes(t[o[q]:o[q + 1]], k) for q in range(nq)) for t, o in shards]
        batches = []
        for t, o in shards:
            b = vb.Batch(gix, nq, len(t), k)
            b.set_queries(t, o)
            batches.append(b)
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(warmup):
            batches[i % nb].run(stream)
        torch.cuda.synchronize()
        for b in batches:
            b.set_timing(True)
        t0 = time.perf_counter()
        for i in range(steps):
            batches[i % nb].run(stream)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        parts = [b.kernel_ms() for b in batches]
        n_launch = sum(n for _, n in parts)
        kernel_ms = sum(ms * n for ms, n in parts) / max(1, n_launch)
        for b in batches:
            hits, n_hits = b.fetch()
            assert (n_hits == k).all()
        a = sum(algo) / len(algo)
        achieved = a / (kernel_ms * 1e-3) / 1e9
        out.update({"value": round(nq * steps / elapsed, 1), "unit": "queries/s", "steps": steps, "warmup": warmup,
                    "ms_per_step": round(1e3 * elapsed / steps, 3),
                    "roofline": {"bound": "hbm", "kernel": "scan_dense_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": committed_traffic(sha16, name),
                                 "algorithmic_bytes_per_launch": int(a), "kernel_ms": round(kernel_ms, 3), "launches_timed": n_launch}})
        left = budget_s - (time.perf_counter() - t_start)
        if not with_cpu:
            pass
        elif left < C5_CPU_NEED_S:
            out["cpu_baseline"] = {"skipped": f"{C5_CPU_NEED_S:.0f} s needed, {max(0.0, left):.0f} s of --extra-budget-s left"}
        else:  # the segment comes down from HBM for the checker; 12 s of CPU work on the first queries of batch 0
            t_cpu = time.perf_counter()
            del batches
            oix = oracle_index(seg)
            t_prep = time.perf_counter() - t_cpu
            out["cpu_baseline"] = cpu_baseline(oix, shards[0][0], shards[0][1], k, budget_s=12.0)
            out["cpu_baseline"]["prepare_s"] = round(t_prep, 1)
            out["cpu_baseline"]["seconds"] = round(time.perf_counter() - t_cpu, 1)
            del oix
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


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


def oracle_index(seg):
    """The checker (oracle/): only --verify and the cpu_baseline leg use it.  A segment that lives in HBM is downloaded for it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc

    if hasattr(seg, "download"):
        seg = seg.download()

    return orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())


def cpu_baseline(oix, terms, off, k, budget_s=20.0):
    """Faithful C++ restatement of the reference's Block-WAND search (oracle/), one query per thread
    (PostgreSQL's execution model), on a bounded sample of the same batch: T = 1 and T = usable cores,
    median of >= 5 repetitions each, plus the brute-force scorer for context.  NOT the Rust binary (there
    is no rustc in this image)."""
    cores = usable_cpus()
    nq = len(off) - 1
    probe = min(nq, 4)
    _, _, t_probe = oix.search_batch(terms[:off[probe]], off[:probe + 1], k, mode="wand", threads=1)
    per_q = max(t_probe / probe, 1e-6)
    reps = 5
    n1 = int(min(nq, max(1, 0.2 * budget_s / (reps * per_q))))
    nn = int(min(nq, max(cores, 0.6 * budget_s * cores / (reps * per_q))))

    def rate(n, threads, mode):
        ts = []
        for _ in range(reps):
            _, _, dt = oix.search_batch(terms[:off[n]], off[:n + 1], k, mode=mode, threads=threads)
            ts.append(n / dt)
        return statistics.median(ts)

    t1 = rate(n1, 1, "wand")
    tn = rate(nn, cores, "wand")
    nb = int(min(nq, max(cores, nn // 4)))
    _, _, dtb = oix.search_batch(terms[:off[nb]], off[:nb + 1], k, mode="brute", threads=cores)
    return {"value": round(tn, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "t1_qps": round(t1, 2), "tn_qps": round(tn, 2), "brute_force_tn_qps": round(nb / dtb, 2),
            "repetitions": reps, "statistic": "median", "cpu_model": cpu_model(),
            "compiler_flags": "g++ -O3 -DNDEBUG -march=x86-64-v3 -ffp-contract=off",
            "sample": f"first {nn} of the first batch's {nq} queries at T={cores} ({n1} at T=1), Block-WAND "
                      f"restatement of search.rs:28-282 over in-memory arrays (oracle/), one query per thread"}


def run(argv=None, scorer_factory=None, backend="nccl"):
    """The whole bench.  scorer_factory / backend exist for tests/test_sharded_gloo.py, which runs this very
    function under gloo with a CPU stand-in for the GPU batch objects; the product never passes them."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=0, help="override queries per GPU")
    ap.add_argument("--k", type=int, default=0, help="development aid: override the workload's k (the line then names that k; not a BASELINE.json configuration)")
    ap.add_argument("--no-verify-sample", action="store_true", help="skip the 64-query parity sample against the oracle (it downloads the segment)")
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches the timed loop rotates through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="check every query of every batch bit-exact against the oracle (outside the timed region)")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--cache", default="", help="segment file: load if present, else build and save")
    ap.add_argument("--no-host-buffer", action="store_true",
                    help="skip the host-buffer (vbm25_stream_*) figures: profiler runs, whose per-kernel averages the overlapped "
                         "launches of three batches in flight would distort")
    ap.add_argument("--extra-budget-s", type=float, default=240.0,
                    help="wall-clock budget for the C2 / C5 lines appended as `extra` (N = 1 only; 0 switches them off)")
    ap.add_argument("--tune", default="", help="development aid: library test switches, name=value[,name=value...] (vbm25_tuning_set)")
    args = ap.parse_args(argv)

    import torch

    import vectorchord_bm25_amd as vb

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and "WORLD_SIZE" not in os.environ and scorer_factory is None:
            return run_single_process(args, vb)  # one process, --gpus devices behind the C ABI (vbm25_multi_*)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    on_gpu = scorer_factory is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}" if on_gpu else "cpu"
    dist = None
    # VBM25_BENCH_FORCE_DIST=1 runs the RCCL code path even with one rank (GPU test of the N>1 path)
    use_dist = world > 1 or os.environ.get("VBM25_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if on_gpu:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    if args.tune and on_gpu:
        for kv in args.tune.split(","):
            name, value = kv.split("=")
            vb.set_tuning(name.strip(), int(value))
    n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[args.workload]
    if args.queries:
        nq = args.queries
    if args.k:
        k = args.k
    nb = max(1, args.batches if args.workload != "C2" else 1)
    threads = args.build_threads or usable_cpus()
    t0 = time.perf_counter()
    cache = args.cache
    device_build = on_gpu and not cache  # generated and sealed in HBM, every rank on its own device (deterministic: the same corpus)
    if device_build:
        seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=local_rank)
    elif use_dist and not cache:
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
    if device_build:
        pass
    elif use_dist and rank != 0:
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
    gix = (vb.GpuIndex(seg) if device_build else vb.GpuIndex(seg, device=local_rank)) if on_gpu else None
    t_upload = time.perf_counter() - t0

    # ---- the batches: rank 0 makes all world x nq queries of each, broadcasts the descriptors, every rank keeps its shard
    n_total = world * nq
    scatter_ms = 0.0
    shards = []  # per batch: (terms, off) of this rank
    for bi in range(nb):
        if use_dist:
            if rank == 0:
                terms_all, off_all = make_queries(seg, vocab, n_total, nterms, seed=1 + bi, zipf_s=zipf_s)
            else:
                terms_all = np.zeros(n_total * nterms, dtype=np.uint32)
                off_all = np.zeros(n_total + 1, dtype=np.uint32)
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            tt = torch.from_numpy(terms_all.astype(np.int32)).to(dev)
            ot = torch.from_numpy(off_all.astype(np.int32)).to(dev)
            dist.broadcast(tt, 0)
            dist.broadcast(ot, 0)
            sync()
            terms_all = tt.cpu().numpy().astype(np.uint32)
            off_all = ot.cpu().numpy().astype(np.uint32)
            shards.append(vb.sharded.shard_queries(terms_all, off_all, world, rank))
            scatter_ms += 1e3 * (time.perf_counter() - t0) / nb
        else:
            shards.append(make_queries(seg, vocab, nq, nterms, seed=1 + bi, zipf_s=zipf_s))
    nq_local = len(shards[0][1]) - 1
    algo_bytes = [sum(seg.query_bytes(t[o[q]:o[q + 1]], k) for q in range(nq_local)) for t, o in shards]

    batches, locals_ = [], []
    for t, o in shards:
        if on_gpu:
            b = vb.Batch(gix, nq_local, len(t), k)
        else:
            b = scorer_factory(seg, nq_local, k)
        b.set_queries(t, o)
        batches.append(b)
        if not use_dist:
            locals_.append(None)
        elif on_gpu:
            import ctypes as C
            hp, nh = C.c_void_p(), C.c_void_p()
            vb._lib.check(vb.lib().vbm25_batch_device_results(b.h, C.byref(hp), C.byref(nh)))
            locals_.append(torch.as_tensor(_DevArray(hp.value, nq_local * k * 3), device=dev))
        else:
            locals_.append(b.words)

    # ---- one step: the scan on the scan stream; the gather of its records on a second stream, so that the next
    # step's scan does not wait for it (every batch object has its own result buffers)
    s_scan = torch.cuda.current_stream() if on_gpu else None
    s_gather = torch.cuda.Stream() if (on_gpu and use_dist) else None
    stream_ptr = s_scan.cuda_stream if on_gpu else None
    gather_events, gather_cpu_s = [], [0.0]
    last_gathered = [None]
    gatherer = vb.sharded.RootGather(n_total, k, dev) if use_dist else None
    gathered_ev = [None] * nb  # per batch object: its records have left (the scan that rewrites them waits for this)

    def step(i):
        b = batches[i % nb]
        if on_gpu and use_dist and gathered_ev[i % nb] is not None:
            s_scan.wait_event(gathered_ev[i % nb])  # the batch's result buffer is still being gathered from
        b.run(stream_ptr)
        if not use_dist:
            return
        if not on_gpu:
            t0 = time.perf_counter()
            last_gathered[0] = gatherer(locals_[i % nb])
            gather_cpu_s[0] += time.perf_counter() - t0
            return
        done = s_scan.record_event()
        with torch.cuda.stream(s_gather):
            s_gather.wait_event(done)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            last_gathered[0] = gatherer(locals_[i % nb])
            e1.record()
            gather_events.append((e0, e1))
            gathered_ev[i % nb] = e1

    # ---- before the timed region: the boundary as the reference's caller sees it -- host buffers in, host buffers out
    # (search.rs:28-36 returns a Vec): (1) one batch at a time: upload, scan, download, each waited for; (2) PIPELINED
    # (vbm25_stream_*: three batches in flight on their own streams with pinned staging -- upload n + 1 and download n - 1
    # overlap scan n).  Taken FIRST: the chip has idled while the host drew the queries, and its clocks settle only after
    # about 30 launches of this 0.23 ms kernel (profiles/r5_warmup.txt) -- more than --warmup 5 of such steps covers; with
    # these 400 batches in front, the K timed steps below see the sustained rate whatever W is.
    pcie_qps = pcie_sync_qps = None
    pre_timed_batches = 0  # (launches of the hot path in front of the W warm-up steps: reported in config.pre_timed_batches)
    if on_gpu and not args.no_host_buffer:
        # (400 batches whatever --steps says: filling and draining the pipeline costs about two steps)
        depth, n_pipe = 3, 400 if nq_local * nterms <= 8192 else 20
        # (the twenty waited-for batches first: the device idles between them, and what runs right in front of the W warm-up steps
        # should be the dense phase -- with them last the driver's 20 timed steps were 1.5 % slower than 200)
        t0 = time.perf_counter()
        for _ in range(20):
            batches[0].set_queries(*shards[0])
            batches[0].run(stream_ptr)
            batches[0].fetch()
        pcie_sync_qps = 20 * nq_local / (time.perf_counter() - t0)
        st = vb.Stream(gix, depth, nq_local, max(len(t) for t, _ in shards), k)
        outs = [(np.zeros((nq_local, k), dtype=vb.HIT_DTYPE), np.zeros(nq_local, dtype=np.uint32)) for _ in range(depth)]
        for phase in ("warm", "timed"):
            if phase == "timed":
                t0 = time.perf_counter()
            for i in range(n_pipe if phase == "timed" else 2 * depth):
                if st.in_flight == depth:
                    st.collect(outs[i % depth])
                st.submit(*shards[i % nb])
            while st.in_flight:
                st.collect(outs[0])
        pcie_qps = n_pipe * nq_local / (time.perf_counter() - t0)
        pre_timed_batches = n_pipe + 2 * depth + 20
    sync()
    for i in range(args.warmup):
        step(i)
    sync()
    gather_events.clear()
    gather_cpu_s[0] = 0.0
    if use_dist:
        dist.barrier()
    if on_gpu:
        for b in batches:
            b.set_timing(True)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_scan_done = None
    if on_gpu and use_dist:
        s_scan.synchronize()  # every scan done; what is left now is gather time that nothing hides
        t_scan_done = time.perf_counter()
    sync()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_launch = 0.0, 0
    if on_gpu:
        parts = [b.kernel_ms() for b in batches]
        n_launch = sum(n for _, n in parts)
        kernel_ms = sum(ms * n for ms, n in parts) / max(1, n_launch)
        for b in batches:
            b.set_timing(False)
    gather_ms = gather_exposed_ms = 0.0
    if use_dist:
        if on_gpu:
            gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / max(1, len(gather_events))
            gather_exposed_ms = 1e3 * (time.perf_counter() - t_scan_done) / args.steps if t_scan_done else 0.0
        else:
            gather_ms = gather_exposed_ms = 1e3 * gather_cpu_s[0] / args.steps
        # every rank's own figures, so that a scaling line can be read: which device was slow, and whether the time went into its
        # scan (kernel_ms), into the step loop around it (ms_per_step) or into a gather that nothing hid (gather_exposed_ms)
        mine = torch.tensor([1e3 * elapsed / args.steps, kernel_ms, gather_ms, gather_exposed_ms], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = {name: [round(float(t[i].item()), 4) for t in every]
                    for i, name in enumerate(("ms_per_step", "kernel_ms", "gather_ms", "gather_exposed_ms"))}
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- outside the timed region ----
    route = None
    if on_gpu:
        route = {0: "general (plan_kernel)", 1: "one launch", 2: "plan-free scan_range_kernel", 3: "scan_win_kernel", 4: "exhaustive"}.get(
            batches[0].debug_route(), "?")
    multi_stream_qps = None
    if on_gpu and world == 1 and not use_dist and not args.no_host_buffer and args.workload in ("C3", "C1"):
        # the same resident batches with ONE STREAM PER BATCH OBJECT instead of one stream for all (outside the timed region; the
        # timed loop above stays on one stream so that a kernel's duration means something): the next batch's scan starts in the
        # tail of this one's -- what the pipelined host-buffer figure above has and the one-stream loop does not
        streams = [torch.cuda.Stream() for _ in range(nb)]
        n3 = max(40, min(4 * args.steps, 400))
        for i in range(2 * nb):
            batches[i % nb].run(streams[i % nb].cuda_stream)
        sync()
        t0 = time.perf_counter()
        for i in range(n3):
            batches[i % nb].run(streams[i % nb].cuda_stream)
        sync()
        multi_stream_qps = n3 * nq_local / (time.perf_counter() - t0)
        for i in range(nb):  # (a batch's consecutive runs stay on one stream: back to the launch stream through a synchronisation)
            batches[i].run(stream_ptr)
        sync()
    results = [b.fetch() for b in batches]
    for hits, n_hits in (results if "dbg" not in args.tune else []):  # (timing experiments switch parts of the kernel off)
        assert (n_hits == k).all() or args.workload in ("C1",), "missing hits"
        s = hits["score"]
        assert (s[:, :-1] >= s[:, 1:]).all()
    if use_dist and rank == 0 and last_gathered[0] is not None:  # rank 0's shard sits first in the gathered records
        hits_last = results[(args.steps - 1) % nb][0]
        got = last_gathered[0].cpu().numpy()[:nq_local * k * 3]
        assert got.tobytes() == np.frombuffer(hits_last.tobytes(), dtype=np.int64).tobytes(), "gathered records differ"

    oix = None
    verified = None
    verified_sample = None
    if on_gpu and world == 1 and not args.verify and not args.no_verify_sample and "dbg" not in args.tune:
        # ties the number to parity: a sample of 64 queries of the first batch, bit-exact against the oracle's brute force
        # (outside the timed region; --verify checks every query of every batch)
        t0 = time.perf_counter()
        oix = oracle_index(seg)
        t_oix = time.perf_counter() - t0
        t0 = time.perf_counter()
        ns = min(64, nq_local)
        pick = np.linspace(0, nq_local - 1, ns).astype(np.int64)
        t_all, o_all = shards[0]
        st_ = np.concatenate([t_all[o_all[q]:o_all[q + 1]] for q in pick]).astype(np.uint32)
        so_ = np.concatenate([[0], np.cumsum([o_all[q + 1] - o_all[q] for q in pick])]).astype(np.uint32)
        ob, onb, _ = oix.search_batch(st_, so_, k, mode="brute", threads=usable_cpus())
        hits0, n0 = results[0]
        assert np.array_equal(n0[pick], onb), "hit counts differ from the oracle"
        for f in ("doc_id", "payload"):
            assert np.array_equal(hits0[f][pick], ob[f]), f"{f} differs from the oracle"
        assert np.array_equal(hits0["score"][pick].view(np.uint64), ob["score"].view(np.uint64)), "score bits differ from the oracle"
        verified_sample = {"queries": int(ns), "of_batch": 0, "bit_exact_vs_oracle_brute_force": True,
                           "oracle_index_s": round(t_oix, 1), "seconds": round(time.perf_counter() - t0, 2)}
    if args.verify:  # full parity of this rank's batches, bit-exact against the oracle's brute force
        oix = oracle_index(seg)
        t0 = time.perf_counter()
        for (t, o), (hits, n_hits) in zip(shards, results):
            ob, onb, _ = oix.search_batch(t, o, k, mode="brute", threads=usable_cpus())
            assert np.array_equal(n_hits, onb), "hit counts differ from the oracle"
            assert np.array_equal(hits["doc_id"], ob["doc_id"]), "doc ids differ from the oracle"
            assert np.array_equal(hits["score"].view(np.uint64), ob["score"].view(np.uint64)), "score bits differ"
            assert np.array_equal(hits["payload"], ob["payload"]), "payloads differ"
        verified = {"queries": int(nq_local * nb), "batches": nb, "seconds": round(time.perf_counter() - t0, 2)}

    latency = None
    if args.workload == "C2" and on_gpu and world == 1:
        # host-inclusive latency of vbm25_search_batch(nq = 1): 1000 different 3-term queries
        lt, lo = make_queries(seg, vocab, 1000, nterms, seed=7, zipf_s=zipf_s)
        one = np.array([0, nterms], dtype=np.uint32)
        for q in range(20):
            vb.search_batch(gix, lt[lo[q]:lo[q + 1]], one, k)
        us = []
        for q in range(1000):
            t0 = time.perf_counter()
            vb.search_batch(gix, lt[lo[q]:lo[q + 1]], one, k)
            us.append(1e6 * (time.perf_counter() - t0))
        us.sort()
        # the same through the bare C entry point with caller-owned buffers made once (what a shim does): without the
        # Python wrapper's array allocations and conversions
        import ctypes as C
        L = vb.lib()
        hb = np.zeros((1, k), dtype=vb.HIT_DTYPE)
        nbuf = np.zeros(1, dtype=np.uint32)
        lt = np.ascontiguousarray(lt, dtype=np.uint32)
        base, po = lt.ctypes.data, C.c_void_p(one.ctypes.data)
        ph, pn = C.c_void_p(hb.ctypes.data), C.c_void_p(nbuf.ctypes.data)
        uc = []
        for q in range(1000):
            pt = C.c_void_p(base + 4 * int(lo[q]))
            t0 = time.perf_counter()
            rc = L.vbm25_search_batch(gix.h, pt, po, 1, k, ph, pn)
            uc.append(1e6 * (time.perf_counter() - t0))
            assert rc == 0 and nbuf[0] == k
        uc.sort()
        latency = {"search_batch_nq1_us_p50": round(us[500], 1), "search_batch_nq1_us_p99": round(us[990], 1),
                   "c_abi_nq1_us_p50": round(uc[500], 1), "c_abi_nq1_us_p99": round(uc[990], 1),
                   "includes": "query hand-over, ONE launch (scan_range_kernel plans, scans and merges; queries read from and hits "
                               "written to pinned host memory), stream synchronisation; search_batch_* adds the Python wrapper"}

    result_line = None
    sha16 = lib_sha16(vb) if on_gpu else None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        routes = batches[0].debug_routes() if on_gpu and hasattr(batches[0], "debug_routes") else None
        dense = zipf_s > 0 if routes is None else routes[1] > routes[0]  # (the roofline line names the kernel most of the queries go to)
        out = {
            "metric": METRIC,
            "value": round(n_total * args.steps / elapsed, 1),
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload if world == 1 else 'C4'}: {n_docs} docs / {vocab} vocab / "
                                   f"{nq_local} x {nterms}-term queries per GPU / top-{k}",
                       "batches_rotated": nb,
                       # (the host-buffer figures run BEFORE the warm-up steps so that the chip's clocks have settled whatever W is: the
                       # timed steps see this many launches of the hot path in front of them besides `warmup`; 0 with --no-host-buffer)
                       "pre_timed_batches": pre_timed_batches,
                       "doc_length": "lognormal(ln 80, 0.6) clamp [8,2000]" if len_mode == 1 else f"fixed {mean_len}",
                       "token_distribution": f"zipf({zipf_s})" if zipf_s > 0 else "uniform",
                       "k1": 1.2, "b": 0.75, "index_hbm_bytes": gix.device_bytes if on_gpu else None,
                       "postings": int(seg.n_postings if device_build else seg.arrays()["term_df"].astype(np.int64).sum()),
                       "corpus_built": "on the device (vbm25_device_segment_synth + vbm25_index_create_from_device)" if device_build
                                       else "on the host (vbm25_segment_synth) and uploaded",
                       "parallelism": f"{nb} batches of {n_total} queries, each sharded over {world} GPU(s), index replicated",
                       "build_s": round(t_build, 2), "upload_s": round(t_upload, 2),
                       "scatter_ms": round(scatter_ms, 3), "gather_ms": round(gather_ms, 4),
                       "gather_exposed_ms": round(gather_exposed_ms, 4),
                       "per_rank": per_rank if use_dist else None,
                       "route": route,
                       "routes": None if routes is None else {
                           "queries_sparse": routes[0], "queries_dense": routes[1],
                           "general_route_items_scan_range_kernel": routes[3], "general_route_items_scan_dense_kernel": routes[4],
                           "general_route_items_scan_many_kernel": routes[5],
                           "note": "batch 0; kernel_ms brackets every posting-scan kernel of a step (scan_range + scan_dense + scan_many on the general route)"},
                       "host_buffer_inclusive_qps_per_gpu": None if pcie_qps is None else round(pcie_qps, 1),
                       "host_buffer_inclusive": "vbm25_stream_*: three batches in flight on their own streams; queries uploaded from pinned staging, counts and 24-byte records written to pinned memory by merge_kernel, every step",
                       "host_buffer_one_batch_at_a_time_qps_per_gpu": None if pcie_sync_qps is None else round(pcie_sync_qps, 1),
                       "resident_one_stream_per_batch_qps_per_gpu": None if multi_stream_qps is None else round(multi_stream_qps, 1)},
        }
        if verified:
            out["config"]["verified_bit_exact_vs_oracle"] = verified
        if verified_sample:
            out["config"]["verified_sample"] = verified_sample
        if latency:
            out["config"]["latency"] = latency
        if on_gpu:
            algo = sum(algo_bytes) / len(algo_bytes)
            achieved = algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            out["roofline"] = {"bound": "hbm", "kernel": "scan_dense_kernel" if dense else ("scan_win_kernel" if route == "scan_win_kernel" else "scan_range_kernel"),
                               "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBPS, 4),
                               # HBM bytes per launch: separate rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950
                               # correction of MI355X_MICROARCH.md), committed in profiles/pmc_traffic.json under the sha256 of
                               # the library they were taken on; null when this run loaded another build
                               "traffic": committed_traffic(sha16, args.workload if world == 1 else "C3"),
                               "algorithmic_bytes_per_launch": int(algo),
                               "kernel_ms": round(kernel_ms, 4), "launches_timed": n_launch}
        if world == 1 and not args.no_cpu_baseline and on_gpu:
            oix = oix or oracle_index(seg)
            out["cpu_baseline"] = cpu_baseline(oix, shards[0][0], shards[0][1], k)
        if on_gpu:
            out["config"]["libvbm25_sha256_16"] = sha16
        if world == 1 and on_gpu and not use_dist and args.workload == "C3" and args.extra_budget_s > 0:
            # the other single-GPU configurations, driver-timed in the same run (the C3 index is released first)
            del batches, results, gix, oix, seg
            t_extra = time.perf_counter()
            extra = {"c2": extra_workload(vb, "C2", args.extra_budget_s, sha16)}
            extra["c5"] = extra_workload(vb, "C5", args.extra_budget_s - (time.perf_counter() - t_extra), sha16,
                                         with_cpu=not args.no_cpu_baseline)
            out["extra"] = extra
        result_line = json.dumps(out)
    if use_dist:
        if rank == 0 and not args.cache and cache and os.path.exists(cache):
            os.remove(cache)
        dist.destroy_process_group()  # RCCL prints its banner here: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        print(result_line, flush=True)
    return use_dist


def run_single_process(args, vb):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run: one process drives N devices through the C ABI
    (vbm25_multi_*: one host upload, replicas made GPU to GPU, contiguous shards on per-device streams, records straight into
    the caller's host arrays).  The driver's N > 1 runs come through torch.distributed.run, one rank per GPU; this is the same
    split as a Rust / C host gets it."""
    import torch
    n = args.gpus
    if not torch.cuda.is_available() or torch.cuda.device_count() < n:
        raise SystemExit(f"--gpus {n}: {torch.cuda.device_count() if torch.cuda.is_available() else 0} device(s) visible")
    n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[args.workload]
    if args.queries:
        nq = args.queries
    nb = max(1, args.batches)
    t0 = time.perf_counter()
    if args.cache and os.path.exists(args.cache):
        seg = vb.Segment.load(args.cache)
    else:
        seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925,
                               threads=args.build_threads or usable_cpus())
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    multi = vb.MultiIndex(seg, list(range(n)))
    t_upload = time.perf_counter() - t0
    n_total = n * nq
    sets = [make_queries(seg, vocab, n_total, nterms, seed=1 + bi, zipf_s=zipf_s) for bi in range(nb)]
    batches = []
    for t, o in sets:
        b = vb.MultiBatch(multi, n_total, len(t), k)
        b.set_queries(t, o)
        batches.append(b)
    for i in range(args.warmup):
        batches[i % nb].run()
    for b in batches:
        b.fetch()
    t0 = time.perf_counter()
    for i in range(args.steps):
        batches[i % nb].run()
    results = [b.fetch() for b in batches]  # waits for every device
    elapsed = time.perf_counter() - t0
    for hits, n_hits in results:
        assert (n_hits == k).all(), "missing hits"
    verified = None
    if args.verify:
        oix = oracle_index(seg)
        for (t, o), (hits, n_hits) in zip(sets, results):
            ob, onb, _ = oix.search_batch(t, o, k, mode="brute", threads=usable_cpus())
            assert np.array_equal(n_hits, onb) and hits.tobytes() == ob.tobytes(), "records differ from the oracle"
        verified = {"queries": int(n_total * nb), "batches": nb}
    out = {"metric": METRIC, "value": round(n_total * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": n,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"C4: {n_docs} docs / {vocab} vocab / {nq} x {nterms}-term queries per GPU / top-{k}",
                      "parallelism": f"ONE process, {n} devices through vbm25_multi_*: {nb} batches of {n_total} queries, contiguous "
                                     "shards, index replicated GPU to GPU, records downloaded to host arrays inside every step",
                      "batches_rotated": nb, "build_s": round(t_build, 2), "upload_and_replicate_s": round(t_upload, 2),
                      "libvbm25_sha256_16": lib_sha16(vb)}}
    if verified:
        out["config"]["verified_bit_exact_vs_oracle"] = verified
    print(json.dumps(out), flush=True)
    return False


def main():
    use_dist = run()
    if use_dist:
        # librccl prints a version banner to stdout from its exit handlers; the contract is ONE
        # JSON line from rank 0, so leave without running them (everything is flushed and the
        # process group is already destroyed)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
