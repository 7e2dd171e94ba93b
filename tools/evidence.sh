#!/bin/bash
# The round's evidence from ONE build in ONE gpurun call: default bench line (CPU baselines, extra.c2 / extra.c5), kernel trace,
# the four counter passes of C3's kernel, phase timers, the k = 100 line, C5's trace and counters.
#   gpurun -- 'bash tools/evidence.sh <out-subdir>'      then   python tools/evidence_collect.py <out-subdir> <round tag>
set -u
R=$GRAFT_REPO_ROOT; SUB=${1:-r5ev}; O=$R/gpurun_out/$SUB; mkdir -p $O; cd $R
sha256sum vectorchord-bm25_amd/csrc/libvbm25.so | cut -c1-16 > $O/lib_sha16.txt
export KERNEL=scan_win_kernel
bash tools/measure.sh $SUB c3full c3stats c3pmc
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-budget-s 0 > $O/bench_c3_driver_flags.json 2> $O/bench_c3_driver_flags.err
timeout 300 python bench.py --k 100 --steps 100 --no-cpu-baseline --extra-budget-s 0 > $O/bench_c3_k100.json 2> $O/bench_c3_k100.err
timeout 300 python tools/profile_win.py C3 > $O/c3_phase_timers.txt 2>&1; head -30 $O/c3_phase_timers.txt
timeout 300 python tools/profile_dense.py C5 > $O/c5_phase_timers.txt 2>&1; head -12 $O/c5_phase_timers.txt
bash tools/measure.sh $SUB c5stats c5pmc
