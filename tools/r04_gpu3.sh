#!/bin/bash
# round 4, third GPU call: whole GPU suite, fp64 MFMA ceiling probe, ContactMap stage timings
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/r04_tests3.log 2>&1; echo "tests rc=$?"
tail -5 $O/r04_tests3.log
timeout -k 10 120 tools/probes/mfma_f64_probe > $O/r04_mfma_f64_probe.txt 2>&1; echo "probe rc=$?"
cat $O/r04_mfma_f64_probe.txt
timeout -k 10 400 python3 tools/bench_contactmap.py > $O/r04_contactmap.txt 2>&1; echo "cm rc=$?"
cat $O/r04_contactmap.txt
