#!/bin/bash
# Interleaved A/B of builds by the dispatch time stamps of a kernel trace (no HIP events):
# product .so + every tools/variants/libabl_*.so (not UTRACE/TRACE), sizes in AB_BINS,
# AB_REPS repetitions.  Prints per build and size: median sweep / reduce duration and the
# span per iteration.   usage: AB_BINS="12000 17700" AB_REPS=3 tools/ab_timeline.sh [dtype]
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
out=$R/gpurun_out/ab_timeline.txt; : > $out
libs="$R/blueberry_amd/libblueberry_hip.so"
for f in $R/tools/variants/libabl_*.so; do case $f in *TRACE*) ;; *) [ -e $f ] && libs="$libs $f";; esac; done
for rep in $(seq 1 ${AB_REPS:-3}); do
  for lib in $libs; do
    tag=$(basename $lib .so); O=$R/gpurun_out/tl_ab_$tag; rm -rf $O; mkdir -p $O
    BB_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/timeline.py $1 ${AB_BINS:-17700} > $O.log 2>&1 || { tail -3 $O.log; continue; }
    python3 $R/tools/timeline_parse.py $O "$tag" | awk -v t=$tag '/segment of/{seg++; it=$(NF-4)} /dur /{ if (seg%2==0) printf "%-22s %s n=%s med %s | per iteration %s us\n", t, $2, $4, $6, it }' | sed 's/n=n=/n=/' >> $out
  done
done
sort -k2,2 -k1,1 $out
