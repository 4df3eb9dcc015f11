#!/bin/bash
# Kernel-level times of the ContactMap stage (rocprofv3 --kernel-trace --stats of
# tools/bench_contactmap.py): the product library, then every libabl_<NAME> given.
#   usage: tools/cm_kernel_stats.sh [NAME ...]     -> gpurun_out/cm_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
out=$R/gpurun_out/cm_kernel_stats.txt; : > $out
for v in product "$@"; do
  O=$R/gpurun_out/cmstats_$v; rm -rf $O; mkdir -p $O
  lib=$R/blueberry_amd/libblueberry_hip.so; [ $v = product ] || lib=$R/tools/variants/libabl_$v.so
  BB_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_contactmap.py > $O.log 2>&1 || { tail -3 $O.log; continue; }
  echo "== $v" >> $out
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    import re
    hit = re.search(r"(normalize\w*|symv\w*|column_sums\w*|gram_\w+|center_rows\w*|corr_finalize\w*|scatter_\w+|"
                    r"gather_kernel|pack_units\w*|basis_\w+|lanczos_\w+|scale_\w*kernel|keep_scan\w*)", n)
    if hit:
        print("  %-32s calls %5s  avg %10.2f us  min %10.2f  max %10.2f" % (
            hit.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
cat $out
