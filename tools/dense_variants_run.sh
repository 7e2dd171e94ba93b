#!/bin/bash
# On the GPU box: C5 (and C3z) for every libvbm25_<name>.so given (tools/search_variant.sh builds them): scan_dense_kernel's time.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for v in "$@"; do
  lib=$R/vectorchord-bm25_amd/csrc/libvbm25_$v.so; [ "$v" = product ] && lib=$R/vectorchord-bm25_amd/csrc/libvbm25.so
  for w in C5 C3z; do
  VBM25_LIBRARY=$lib timeout 400 python bench.py --workload $w --no-cpu-baseline --no-host-buffer --steps 6 --warmup 2 --extra-budget-s 0 2>$O/$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', '$w', 'kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'verified', d['config'].get('verified_sample',{}).get('bit_exact_vs_oracle_brute_force'))" | tee -a $O/variants.txt
  done
done
