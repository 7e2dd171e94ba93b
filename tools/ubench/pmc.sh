#!/bin/bash
# PMC passes over the ubench's team kernels (separate passes; --kernel-trace only)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ceil/pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="$R/tools/ubench/mark_ceiling team"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_IFETCH --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/sq4 -- $B > $O/sq4.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r4ceil/pmc'
for d in sorted(glob.glob(O+'/sq*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        seen=collections.Counter()
        for r in csv.DictReader(open(f)):
            pass
        # dispatches per kernel
        disp=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            disp[r['Kernel_Name'][:60]].add(r['Dispatch_Id'])
        for k in acc:
            print(os.path.basename(d), k, {c: round(v/len(disp[k])) for c,v in acc[k].items()})
PY
tail -3 $O/sq4.log
find $O -name "*.csv" -size +5M -delete
