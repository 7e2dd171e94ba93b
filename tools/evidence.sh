#!/bin/bash
# The round's evidence from ONE build in ONE gpurun call: default bench line (CPU baselines, extra.c2 / extra.c5), kernel trace,
# the four counter passes of C3's kernel, phase timers, the k = 100 line, C5's trace and counters.
#   gpurun -- 'bash tools/evidence.sh <out-subdir>'      then   python tools/evidence_collect.py <out-subdir> <round tag>
set -u
R=$GRAFT_REPO_ROOT; SUB=${1:-r6ev}; O=$R/gpurun_out/$SUB; mkdir -p $O; cd $R
sha256sum vectorchord-bm25_amd/csrc/libvbm25.so | cut -c1-16 > $O/lib_sha16.txt
export KERNEL=scan_win_kernel
bash tools/measure.sh $SUB c3full c3stats c3pmc
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-budget-s 0 > $O/bench_c3_driver_flags.json 2> $O/bench_c3_driver_flags.err
timeout 300 python bench.py --k 100 --steps 100 --no-cpu-baseline --extra-budget-s 0 > $O/bench_c3_k100.json 2> $O/bench_c3_k100.err
timeout 300 python tools/profile_win.py C3 > $O/c3_phase_timers.txt 2>&1; head -30 $O/c3_phase_timers.txt
timeout 300 python tools/profile_dense.py C5 > $O/c5_phase_timers.txt 2>&1; head -12 $O/c5_phase_timers.txt
bash tools/measure.sh $SUB c5stats c5pmc
# round 6: the ablation of scan_win_kernel on the development build of the same sources (timing experiments, wrong results by design),
# the Zipf(1) variant of C3 (general route: scan_dense_kernel + scan_range_kernel), mixed query lengths, the host's share of a
# multi-device step and of a pipelined step, the index planes table
TUNES="dbg=0 dbg=1 dbg=3 dbg=7 dbg=39 dbg=103 dbg=135" bash tools/win_dbg.sh $SUB > $O/c3_win_ablation_raw.txt 2>&1; cat $O/c3_win_ablation_raw.txt
bash tools/measure.sh $SUB c3z
timeout 300 python tools/mixed_batch.py > $O/mixed_batch.txt 2>&1; cat $O/mixed_batch.txt
timeout 300 python tools/multi_enqueue.py 8 > $O/multi_enqueue.txt 2>&1; cat $O/multi_enqueue.txt
timeout 300 python tools/stream_host_time.py > $O/stream_host_time.txt 2>&1; tail -3 $O/stream_host_time.txt
timeout 600 python tools/index_planes.py > $O/index_planes.txt 2>&1; tail -8 $O/index_planes.txt
