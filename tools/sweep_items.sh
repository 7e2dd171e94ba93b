set -u
cd $GRAFT_REPO_ROOT
bash tools/measure.sh r4g alltests c3quick
for t in range_items=2048 range_items=1536 range_items=2048,range_min_chunk=8192; do
  TUNE=$t bash tools/measure.sh r4g_$(echo $t | tr '=,' '__') c3quick
done
bash tools/measure.sh r4g c5
