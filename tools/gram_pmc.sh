#!/bin/bash
# Counter pass for gram_kernel (correlation at d = 24,927): MFMA busy cycles and LDS bank
# conflicts, each group in a rocprofv3 pass of its own (--kernel-trace only).
#   -> gpurun_out/gram_pmc.txt
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; O=$R/gpurun_out/grampmc; rm -rf $O; mkdir -p $O
cat > $O/run.py <<PY
import sys, numpy
sys.path.insert(0, "$R")
import blueberry_amd as bb
rng = numpy.random.default_rng(0)
n_bins = 24926
nnz = n_bins * 100
bi = rng.integers(0, n_bins, nnz); bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.002, nnz))
tr = numpy.stack([bi * 10000.0, bj * 10000.0, rng.integers(1, 500, nnz).astype(float)], 1)
cm = bb.ContactMap.from_triples(tr, 10000, n_bins)
cm.correlation()
PY
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o -E "SQ_[A-Z_0-9]*(MFMA|LDS)[A-Z_0-9]*" $O/avail.txt | sort -u > $O/names.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 $O/run.py > $O/$tag.log 2>&1 || echo "pass failed: $grp"
done
python3 - > $R/gpurun_out/gram_pmc.txt <<PY
import csv, glob
print("available MFMA / LDS counters:", " ".join(open("$O/names.txt").read().split()))
for f in sorted(glob.glob("$O/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "gram_kernel" in r["Kernel_Name"]:
            print(r["Counter_Name"], r["Counter_Value"], "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if r.get("End_Timestamp") else "")
PY
cat $R/gpurun_out/gram_pmc.txt
