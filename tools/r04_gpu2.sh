#!/bin/bash
# round 4, second GPU call: spectral start (multi-rank on the device), timing, cfg5 PMC
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_parity.py -q -x -m gpu -k "spectral" > $O/r04_tests2.log 2>&1; echo "tests rc=$?"
tail -5 $O/r04_tests2.log
timeout -k 10 300 python3 tools/spectral_timing.py > $O/r04_spectral_timing.txt 2>&1; echo "timing rc=$?"
cat $O/r04_spectral_timing.txt
tools/tools_pmc.sh r04 309568 float32 genome10kb; echo "pmc rc=$?"
tail -3 $O/pmc_r04_n309568_genome10kb.txt
