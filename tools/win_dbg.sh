# timing experiments on scan_win_kernel (wrong results by design): which part costs what.  Needs the development build
# (make -C vectorchord-bm25_amd/csrc libvbm25_dev.so): the product library has no `dbg` switch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5c}; mkdir -p $O; cd $R
export VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_dev.so
for t in ${TUNES:-dbg=0 dbg=1 dbg=3 dbg=7 win=0}; do
  timeout 200 python bench.py --no-cpu-baseline --no-verify-sample --no-host-buffer --steps 100 --extra-budget-s 0 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', 'kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'])"
done
