R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6m
for t in win_cut1=383,win_cut2=717 win_cut1=388,win_cut2=722 win_cut1=392,win_cut2=725 win_cut1=392,win_cut2=730 win_cut1=396,win_cut2=730 win_cut1=400,win_cut2=735 win_cut1=386,win_cut2=728; do
  python bench.py --no-cpu-baseline --no-host-buffer --no-verify-sample --steps 200 --extra-budget-s 0 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', 'kernel_ms', d['roofline']['kernel_ms'], 'step', d['ms_per_step'])" | tee -a gpurun_out/r6m/skew.txt
done
