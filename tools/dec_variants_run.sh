#!/bin/bash
# On the GPU box: C3 through an index without post_id16 / post_rel16 for every libvbm25_<name>.so given: kernel ms = decode_id16_kernel + scan_win_kernel.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for v in "$@"; do
  lib=$R/vectorchord-bm25_amd/csrc/libvbm25_$v.so; [ "$v" = product ] && lib=$R/vectorchord-bm25_amd/csrc/libvbm25.so
  VBM25_LIBRARY=$lib timeout 200 python bench.py --no-cpu-baseline --no-host-buffer --steps 100 --extra-budget-s 0 --tune id16_plane=0,rel16_plane=0 2>$O/$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'], 'verified', d['config'].get('verified_sample',{}).get('bit_exact_vs_oracle_brute_force'))" | tee -a $O/variants.txt
done
