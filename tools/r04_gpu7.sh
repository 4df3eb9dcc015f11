#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fit_many or correlation or several" > $O/r04_tests7.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_tests7.log
rm -rf $O/cmpmc
timeout -k 10 900 bash tools/cm_pmc.sh; echo "cm_pmc rc=$?"; cat $O/cm_pmc.txt
timeout -k 10 400 python3 tools/bench_contactmap.py > $O/r04_contactmap.txt 2>&1; echo "cm rc=$?"; grep -E "correlation" $O/r04_contactmap.txt
