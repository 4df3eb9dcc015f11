#!/bin/bash
# Kernel-trace timeline of the iteration loop (no HIP events in the stream): per-kernel
# durations and the gaps between dispatches.   usage: timeline.sh <tag> [dtype] n_bins...
tag=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tl_$tag; export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/timeline.py "$@" > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 $R/tools/timeline_parse.py $O "$tag $*" > $R/gpurun_out/timeline_$tag.txt
cat $R/gpurun_out/timeline_$tag.txt
