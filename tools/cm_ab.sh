#!/bin/bash
# ContactMap stage with the product library and with the given libabl_<NAME> builds.
#   usage: [CM_SIZES="24926 24927"] tools/cm_ab.sh SUM64 ...      -> gpurun_out/cm_ab.txt
out=gpurun_out/cm_ab.txt; : > $out
for v in product "$@"; do
  echo "== $v" >> $out
  if [ $v = product ]; then timeout -k 10 400 python tools/bench_contactmap.py ${CM_SIZES:-} >> $out 2>&1
  else BB_LIB=$PWD/tools/variants/libabl_$v.so timeout -k 10 300 python tools/bench_contactmap.py >> $out 2>&1; fi
done
grep -E "^==|normalize|marginals|filter" $out
