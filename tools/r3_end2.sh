#!/bin/bash
# the remaining counter passes of the final build (C3)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3end3; rm -rf $O; mkdir -p $O
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
timeout 40 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/c3_pmc2 -- $B > $O/c3_pmc2.log 2>&1
timeout 40 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/c3_fetch -- $B > $O/c3_fetch.log 2>&1
timeout 40 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_write -- $B > $O/c3_write.log 2>&1
cd $R; python tools/pmc_summary.py scan_range_kernel sq2=$O/c3_pmc2 fetch=$O/c3_fetch write=$O/c3_write > $O/c3_pmc_rest.csv 2>/dev/null; cat $O/c3_pmc_rest.csv
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
