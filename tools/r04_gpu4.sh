#!/bin/bash
# round 4, fourth GPU call: gram kernel A/B, fp64 MFMA ceiling
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 120 tools/probes/mfma_f64_probe > $O/r04_mfma_f64_probe.txt 2>&1; echo "probe rc=$?"
cat $O/r04_mfma_f64_probe.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "correlation or eigen or contactmap" > $O/r04_tests4.log 2>&1; echo "tests rc=$?"
tail -3 $O/r04_tests4.log
for rep in 1 2; do
timeout -k 10 400 python3 tools/bench_contactmap.py 2>&1 | grep -E "correlation" > $O/r04_gram_new_$rep.txt; cat $O/r04_gram_new_$rep.txt
BB_LIB=$R/tools/variants/libabl_GRAMOLD.so timeout -k 10 400 python3 tools/bench_contactmap.py 2>&1 | grep -E "correlation" > $O/r04_gram_old_$rep.txt; cat $O/r04_gram_old_$rep.txt
done
