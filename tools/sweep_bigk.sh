# scan_range_kernel<128> / <256> (sparse queries, 64 < k <= 256) at two workgroups per CU with spills vs one without:
#   VBM25_LIBRARY=<other build> selects the library built with -DVBM25_RWPS_BIGK=2
set -u
cd $GRAFT_REPO_ROOT
for lib in libvbm25.so libvbm25_bigk.so; do for k in 100 200; do
  VBM25_LIBRARY=$PWD/vectorchord-bm25_amd/csrc/$lib timeout 300 python bench.py --no-cpu-baseline --k $k --steps 50 --extra-budget-s 0 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'k=$k', d['value'], 'q/s', d['roofline']['kernel_ms'], 'ms')"
done; done
