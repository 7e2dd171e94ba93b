#!/bin/bash
# On the GPU box: time C3's dominant kernel for every libvbm25_<name>.so given (tools/win_variant.sh builds them).
#   gpurun -- 'bash tools/win_variants_run.sh <out-subdir> <name> [<name> ...]'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for v in "$@"; do
  lib=$R/vectorchord-bm25_amd/csrc/libvbm25_$v.so; [ "$v" = product ] && lib=$R/vectorchord-bm25_amd/csrc/libvbm25.so
  VBM25_LIBRARY=$lib timeout 200 python bench.py --no-cpu-baseline --no-host-buffer --steps 200 --extra-budget-s 0 2>$O/$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'verified', d['config'].get('verified_sample',{}).get('bit_exact_vs_oracle_brute_force'))" | tee -a $O/variants.txt
done
