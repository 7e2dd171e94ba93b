set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
timeout 240 python bench.py --steps 50 --warmup 5 --build-threads 16 --cache /tmp/c3.seg > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
cd $R
echo v9 >> $O/sweep.txt; VBM25_NO_CURSOR=1 timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg 2>&1 | tail -1 | grep -o "ms_per_step[^,]*\|kernel_ms[^,]*" >> $O/sweep.txt
timeout 200 python tools/check_shapes.py > $O/shapes.txt 2>&1
find $O -name "*.csv" -size +20M -delete
du -sh $O | tail -1
