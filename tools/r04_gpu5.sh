#!/bin/bash
# round 4, fifth GPU call: fit_many tests + timing, gram interleave A/B, fit_triples test
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fit_many or several_maps or triples or state_machine or blocked or genome" > $O/r04_tests5.log 2>&1; echo "tests rc=$?"
tail -15 $O/r04_tests5.log
timeout -k 10 300 python3 tools/batch_timing.py > $O/r04_batch.txt 2>&1; echo "batch rc=$?"
cat $O/r04_batch.txt
timeout -k 10 400 python3 tools/bench_contactmap.py 2>&1 | grep -E "correlation" > $O/r04_gram_new_3.txt; cat $O/r04_gram_new_3.txt
BB_LIB=$R/tools/variants/libabl_GRAMIL.so timeout -k 10 400 python3 tools/bench_contactmap.py 2>&1 | grep -E "correlation" > $O/r04_gram_il.txt; cat $O/r04_gram_il.txt
