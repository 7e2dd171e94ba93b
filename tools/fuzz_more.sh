#!/bin/bash
# More shapes through the differential fuzz test of the window kernel (tests/test_gpu_win.py) than the suite's fixed sixteen:
#   gpurun -- 'bash tools/fuzz_more.sh <out-subdir> <seed> [<seed> ...]'      (40 cases per seed and routing mode)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for seed in "$@"; do
  VBM25_FUZZ_SEED=$seed VBM25_FUZZ_CASES=40 timeout 1500 python -m pytest tests/test_gpu_win.py -x -q -m gpu -k random_shapes > $O/fuzz_$seed.log 2>&1
  echo "seed $seed: $(tail -1 $O/fuzz_$seed.log)"
done
