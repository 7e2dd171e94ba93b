#!/bin/bash
# round-3 GPU call 1: (a) scan_dense_kernel stress on the round-2 failing build, the k <= 256 instantiation of HEAD and
# its assertion build; (b) baseline PMC table of scan_range_kernel on C3 (separate --pmc passes, --kernel-trace only)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c1; rm -rf $O; mkdir -p $O
cd $R
CS=$R/vectorchord-bm25_amd/csrc
for v in "old68186a1:$R/build/variants/68186a1/vectorchord-bm25_amd/csrc/libvbm25.so" "chk:$CS/libvbm25_chk.so" "d256:$CS/libvbm25_d256.so"; do
  name=${v%%:*}; so=${v#*:}
  VBM25_LIBRARY=$so DS_REPS=${DS_REPS:-30} DS_ITEMS=1024,4096,16384 DS_K=200,256 timeout 400 python tools/dense_stress.py > $O/stress_$name.log 2>&1
  echo "== $name exit $?"; tail -8 $O/stress_$name.log
done
python bench.py --no-cpu-baseline --cache /tmp/c3.seg > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cat $O/bench_c3.json
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel sq1=$O/pmc_sq1 sq2=$O/pmc_sq2 fetch=$O/pmc_fetch > $O/pmc_summary.csv 2> $O/pmc_summary.err
cat $O/pmc_summary.csv; tail -3 $O/pmc_summary.err $O/pmc_sq1.log $O/pmc_sq2.log
find $O -name "*.csv" -size +5M -delete
