set -u
cd $GRAFT_REPO_ROOT
for t in range_grid=256 range_grid=384 range_grid=512; do
  TUNE=$t bash tools/measure.sh r4u_$(echo $t | tr '=,' '__') c3quick
done
