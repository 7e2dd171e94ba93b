set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/tests.txt
timeout 240 python bench.py --steps 50 --warmup 5 --build-threads 16 --cache /tmp/c3.seg > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
find $O -name "*.csv" -size +20M -delete
du -sh $O | tail -1
