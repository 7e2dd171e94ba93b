#!/bin/bash
# last call of the round on the final build: rocprofv3 kernel stats of the C3 bench (+ the SQ counter passes if time allows)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3end2; rm -rf $O; mkdir -p $O
cd $R
echo "build: $(sha256sum vectorchord-bm25_amd/csrc/libvbm25.so | cut -c1-16)  $(date -u +%FT%TZ)" > $O/build.txt
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS['C3']
vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/c3.seg')"
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- $B > $O/c3_stats.log 2>&1
find $O/c3_stats -name "*kernel_stats.csv" -exec cp {} $O/c3_kernel_stats.csv \; ; head -3 $O/c3_kernel_stats.csv
timeout 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/c3_pmc1 -- $B > $O/c3_pmc1.log 2>&1
cd $R; python tools/pmc_summary.py scan_range_kernel sq1=$O/c3_pmc1 > $O/c3_pmc_scan_range_kernel.csv 2>/dev/null; cat $O/c3_pmc_scan_range_kernel.csv
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
