#!/bin/bash
# rocprofv3 passes for the dominant kernel (scratch tool; summaries get copied to profiles/)
# usage: tools_pmc.sh <tag>
tag=${1:-x}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- $B > $O.sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O.write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- $B > $O.sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("sq","fetch","write","sq2"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        acc=collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "stress_grad" in row["Kernel_Name"]:
                a=acc[row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(sub, k, "avg_per_dispatch", v/max(n,1), "dispatches", n)
for f in glob.glob("$O/trace/*/*kernel_stats.csv"):
    print(open(f).read())
PY
