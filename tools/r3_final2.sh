#!/bin/bash
# Second half of the round-3 measurement run (same build as tools/r3_final.sh): the memory-traffic counters, one per pass
# (FETCH_SIZE and WRITE_SIZE together exceed what one pass can collect), and the C5 bench line per number of rotated batches.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final2; rm -rf $O; mkdir -p $O
cd $R
echo "build: $(sha256sum vectorchord-bm25_amd/csrc/libvbm25.so | cut -c1-16)  $(date -u +%FT%TZ)" > $O/build.txt
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
for w in ('C3', 'C5'):
    n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[w]
    vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/%s.seg' % w.lower())"
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/c3_fetch -- $B > $O/c3_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- $B > $O/c3_write.log 2>&1
B5="python $R/bench.py --workload C5 --steps 3 --warmup 1 --batches 2 --no-cpu-baseline --cache /tmp/c5.seg"
timeout 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/c5_fetch -- $B5 > $O/c5_fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c5_write -- $B5 > $O/c5_write.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/c5_sq -- $B5 > $O/c5_sq.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel fetch=$O/c3_fetch write=$O/c3_write > $O/c3_pmc_mem.csv 2> $O/c3_pmc.err; cat $O/c3_pmc_mem.csv
python tools/pmc_summary.py scan_dense_kernel fetch=$O/c5_fetch write=$O/c5_write sq=$O/c5_sq > $O/c5_pmc_scan_dense_kernel.csv 2> $O/c5_pmc.err; cat $O/c5_pmc_scan_dense_kernel.csv
for nb in 4 2 4; do
  timeout 600 python bench.py --workload C5 --no-cpu-baseline --cache /tmp/c5.seg --steps 8 --warmup 2 --batches $nb > $O/c5_bench_b$nb.json 2> $O/c5_bench_b$nb.err
  python -c "
import json; d=json.load(open('$O/c5_bench_b$nb.json')); print('batches $nb:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
