#!/bin/bash
# rocprofv3 passes for the dominant kernel at one problem size (scratch tool; the
# summaries it writes under gpurun_out/ get copied to profiles/).
#   usage: tools_pmc.sh <tag> [bins=50000] [dtype=float32] [workload=dense]   (genome10kb: bins 309568)
# Kernel trace + stats in one pass; every PMC group in a pass of its own (--kernel-trace
# only, as MI355X_MICROARCH.md's HBM section prescribes).  Writes
#   gpurun_out/pmc_<tag>_n<bins>.txt        per-dispatch averages + the kernel stats table
#   gpurun_out/pmc_entry_<tag>_n<bins>.json the entry for profiles/pmc_latest.json
tag=${1:-x}; bins=${2:-50000}; dtype=${3:-float32}; wl=${4:-dense}; sfx=""; [ $dtype = float64 ] && sfx=_f64
[ $wl = dense ] || sfx=${sfx}_$wl
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_${tag}_n$bins$sfx; export TMPDIR=/tmp
mkdir -p $O
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --converge-steps 0 --reps 0 --bins $bins --dtype $dtype --workload $wl"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq -- $B > $O.sq.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O.fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O.write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- $B > $O.sq2.log 2>&1 || exit 1
python3 - > $R/gpurun_out/pmc_${tag}_n$bins$sfx.txt <<PY
import csv, glob, collections, json
avg = {}
for sub in ("sq","fetch","write","sq2"):
    for f in glob.glob("$O/%s/*/*counter_collection.csv" % sub):
        acc=collections.defaultdict(lambda: [0.0,0])
        kname=None
        for row in csv.DictReader(open(f)):
            if "stress_grad" in row["Kernel_Name"]:
                kname=row["Kernel_Name"]
                a=acc[row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
        for k,(v,n) in sorted(acc.items()):
            avg[k]=(v/max(n,1), n)
            print(sub, k, "avg_per_dispatch", v/max(n,1), "dispatches", n)
for f in glob.glob("$O/trace/*/*kernel_stats.csv"):
    print(open(f).read())
n=$bins; es = 4 if "$dtype"=="float32" else 8
alg = n*(n-1)//2*es
if "$wl" == "genome10kb":
    import sys; sys.path.insert(0, "$R")
    from blueberry_amd.solver import tiles_from_blocks
    from blueberry_amd.utils import genome_boundaries
    alg = tiles_from_blocks(n, genome_boundaries(n), 1000, "$dtype")[1] * es   # stored pairs
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    hbm = avg["FETCH_SIZE"][0]*1024*2 + avg["WRITE_SIZE"][0]*1024
    entry = {"bins": n, "dtype": "$dtype", "gpus": 1, "workload": "$wl", "kernel": kname,
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python3 bench.py --steps 10 --warmup 2 --bins %d --dtype $dtype --workload $wl, averaged over the kernel's %d dispatches; profiles/r04_pmc_${tag}_n%d$sfx.txt" % (n, avg["FETCH_SIZE"][1], n),
             "FETCH_SIZE_KB_avg": avg["FETCH_SIZE"][0], "WRITE_SIZE_KB_avg": avg["WRITE_SIZE"][0],
             "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read stream -> x2; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
             "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
             "traffic_over_algorithmic": hbm/alg}
    json.dump(entry, open("$R/gpurun_out/pmc_entry_${tag}_n$bins$sfx.json","w"), indent=1)
    print("entry", json.dumps(entry))
PY
