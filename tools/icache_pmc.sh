#!/bin/bash
# Instruction-cache counters of the sweep kernel at one size (is the slow first unit of a
# launch a cold instruction cache?  DESIGN.md 4.9).   usage: tools/icache_pmc.sh [bins=24926]
bins=${1:-24926}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_icache_n$bins; export TMPDIR=/tmp
mkdir -p $O
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --converge-steps 0 --reps 0 --bins $bins"
rocprofv3 --list-avail > $O.avail.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_WAIT_INST_ANY\|SQ_INST_LEVEL[A-Z_]*" $O.avail.txt | sort -u > $O.names.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $O/ic -- $B > $O.ic.log 2>&1 || { tail -5 $O.ic.log; exit 1; }
rocprofv3 --pmc SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/if -- $B > $O.if.log 2>&1 || tail -5 $O.if.log
python3 - <<PY
import csv, glob, collections
for sub in ("ic","if"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        acc=collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "stress_grad" in row["Kernel_Name"]:
                a=acc[row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
        for k,(v,n) in sorted(acc.items()):
            print("N=$bins", k, "avg per dispatch %.0f" % (v/max(n,1)), "dispatches", n)
PY
cat $O.names.txt | tr '\n' ' '
