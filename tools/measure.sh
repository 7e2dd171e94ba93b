#!/bin/bash
# One parametrised measurement script (replaces the round-3 tools/r3_*.sh one-offs).  Run through gpurun:
#   gpurun -- 'bash tools/measure.sh <out-subdir> <step> [<step> ...]'
# steps: tests (GPU search tests)  alltests (whole -m gpu suite)  c3 (verified bench line)  c3quick  c3stats (rocprofv3 kernel
#        stats)  c3pmc (four counter passes + the pmc_traffic.json entry)  c5  c5stats  c5pmc  c2  ubench  ubenchpmc
#        sweep_items  sweep_grid  sweep_c2  sweep_bigk (the parameter sweeps of round 4)
# TUNE="name=value,..." is passed to bench.py --tune.
set -u
R=$GRAFT_REPO_ROOT; SUB=${1:-r4}; shift; O=$R/gpurun_out/$SUB; mkdir -p $O; cd $R
TUNE_ARG=""; [ -n "${TUNE:-}" ] && TUNE_ARG="--tune $TUNE"
seg() { :; }  # (the corpora are generated on the device in a second or two: nothing to cache any more)
show() { python -c "
import json,sys
d=json.load(open('$1')); print('$1', d['value'], 'q/s', d['ms_per_step'], 'ms/step', d.get('roofline'), d['config'].get('verified_bit_exact_vs_oracle'))" || tail -5 ${1%.json}.err; }
for step in "$@"; do case $step in
  tests) timeout 1200 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "not c5_full and not bench_distributed" > $O/pytest_search.log 2>&1; tail -5 $O/pytest_search.log;;
  alltests) timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log;;
  c3) seg C3; timeout 600 python bench.py --no-cpu-baseline --verify --steps 100 --extra-budget-s 0 $TUNE_ARG > $O/bench_c3.json 2> $O/bench_c3.err; show $O/bench_c3.json;;
  c3quick) seg C3; timeout 600 python bench.py --no-cpu-baseline --no-verify-sample --steps 100 --extra-budget-s 0 $TUNE_ARG > $O/bench_c3q.json 2> $O/bench_c3q.err; show $O/bench_c3q.json;;
  c3full) seg C3; timeout 900 python bench.py $TUNE_ARG > $O/bench_c3_full.json 2> $O/bench_c3_full.err; show $O/bench_c3_full.json;;
  c5) seg C5; timeout 900 python bench.py --workload C5 --no-cpu-baseline --steps 10 --warmup 2 $TUNE_ARG > $O/bench_c5.json 2> $O/bench_c5.err; show $O/bench_c5.json;;
  c3z) timeout 900 python bench.py --workload C3z --steps 20 --warmup 3 --extra-budget-s 0 --no-host-buffer $TUNE_ARG > $O/bench_c3z.json 2> $O/bench_c3z.err; show $O/bench_c3z.json
     (cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3z -- python $R/bench.py --workload C3z --steps 10 --warmup 2 --no-cpu-baseline --no-verify-sample --no-host-buffer --extra-budget-s 0 $TUNE_ARG > $O/stats3z.log 2>&1); f=$(find $O/stats3z -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3z_kernel_stats.csv && head -8 $O/c3z_kernel_stats.csv; find $O/stats3z -name "*.csv" -size +5M -delete;;
  c2) seg C2; timeout 600 python bench.py --workload C2 --no-cpu-baseline --steps 200 $TUNE_ARG > $O/bench_c2.json 2> $O/bench_c2.err; show $O/bench_c2.json;;
  c3stats) seg C3; (cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-verify-sample --no-host-buffer --extra-budget-s 0 $TUNE_ARG > $O/stats.log 2>&1); f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv && head -8 $O/c3_kernel_stats.csv; find $O/stats -name "*.csv" -size +5M -delete;;
  c3pmc) seg C3; B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify-sample --no-host-buffer --extra-budget-s 0 $TUNE_ARG"
     (cd /tmp; export TMPDIR=/tmp
      timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
      timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_IFETCH --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
      timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
      timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1)
     python tools/pmc_summary.py ${KERNEL:-scan_win_kernel} sq1=$O/pmc_sq1 sq2=$O/pmc_sq2 fetch=$O/pmc_fetch write=$O/pmc_write > $O/pmc_summary.csv 2> $O/pmc_summary.err; cat $O/pmc_summary.csv
     python tools/pmc_traffic.py C3 ${KERNEL:-scan_win_kernel} $O/pmc_fetch $O/pmc_write > $O/pmc_traffic_c3.json; cat $O/pmc_traffic_c3.json
     find $O -name "*.csv" -size +5M -delete;;
  c5stats) (cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -- python $R/bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline --no-verify-sample --no-host-buffer --extra-budget-s 0 $TUNE_ARG > $O/stats5.log 2>&1); f=$(find $O/stats5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv && head -8 $O/c5_kernel_stats.csv; find $O/stats5 -name "*.csv" -size +5M -delete;;
  c5pmc) B="python $R/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-verify-sample --no-host-buffer --extra-budget-s 0 $TUNE_ARG"
     (cd /tmp; export TMPDIR=/tmp
      timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc5_sq1 -- $B > $O/pmc5_sq1.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc5_sq2 -- $B > $O/pmc5_sq2.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- $B > $O/pmc5_fetch.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- $B > $O/pmc5_write.log 2>&1)
     python tools/pmc_summary.py scan_dense_kernel sq1=$O/pmc5_sq1 sq2=$O/pmc5_sq2 fetch=$O/pmc5_fetch write=$O/pmc5_write > $O/pmc5_summary.csv 2> $O/pmc5_summary.err; cat $O/pmc5_summary.csv
     python tools/pmc_traffic.py C5 scan_dense_kernel $O/pmc5_fetch $O/pmc5_write > $O/pmc_traffic_c5.json; cat $O/pmc_traffic_c5.json
     find $O -name "*.csv" -size +5M -delete;;
  ubenchpmc) bash tools/ubench/pmc.sh > $O/mark_ceiling_pmc.txt 2>&1; tail -40 $O/mark_ceiling_pmc.txt;;
  ubench) (cd tools/ubench && { [ -x mark_ceiling ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o mark_ceiling mark_ceiling.hip; } && ./mark_ceiling > $O/mark_ceiling.txt 2>&1; cat $O/mark_ceiling.txt);;
  # ---- parameter sweeps of round 4 (DESIGN.md section 9 quotes their results)
  sweep_items) for t in range_items=2048 range_items=1536 range_items=2048,range_min_chunk=8192; do TUNE=$t bash $0 ${SUB}_$(echo $t | tr '=,' '__') c3quick; done;;
  sweep_grid) for t in range_grid=256 range_grid=384 range_grid=512; do TUNE=$t bash $0 ${SUB}_$(echo $t | tr '=,' '__') c3quick; done;;  # resident workgroups per CU: 1, 1.5, 2
  sweep_c2) for t in range_min_chunk=16384 range_min_chunk=8192 range_min_chunk=4096 range_min_chunk=2048; do
      timeout 300 python bench.py --workload C2 --no-cpu-baseline --steps 200 --tune $t 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['config']['latency']; print('$t', 'kernel_ms', d['roofline']['kernel_ms'], 'p50', l['c_abi_nq1_us_p50'], 'p99', l['c_abi_nq1_us_p99'])"; done;;
  sweep_bigk)  # scan_range_kernel<128> / <256> (sparse queries, 64 < k <= 256): LIBS="libvbm25.so libvbm25_bigk.so" compares builds (-DVBM25_RWPS_BIGK=2)
    for lib in ${LIBS:-libvbm25.so}; do for k in 100 200; do
      VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/$lib timeout 300 python bench.py --no-cpu-baseline --k $k --steps 50 --extra-budget-s 0 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'k=$k', d['value'], 'q/s', d['roofline']['kernel_ms'], 'ms')"; done; done;;
  *) echo "unknown step $step";;
esac; done
