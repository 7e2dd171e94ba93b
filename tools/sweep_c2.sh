set -u
cd $GRAFT_REPO_ROOT
for t in range_min_chunk=16384 range_min_chunk=8192 range_min_chunk=4096 range_min_chunk=2048; do
  timeout 300 python bench.py --workload C2 --no-cpu-baseline --steps 200 --tune $t 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['config']['latency']; print('$t', 'kernel_ms', d['roofline']['kernel_ms'], 'p50', l['c_abi_nq1_us_p50'], 'p99', l['c_abi_nq1_us_p99'])"
done
