#!/bin/bash
out=gpurun_out/abl.txt; : > $out
run() { echo "== $1" >> $out; env BB_LIB=$2 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', round(d['roofline']['kernel_ms'],4), 'read_ms', round(d['roofline']['stream_read_ms'],4))" >> $out 2>&1; }
run base $PWD/blueberry_amd/libblueberry_hip.so
for v in NODPP NORSQ NOMASK NOCOL NOSTORE ALL; do run $v $PWD/tools/variants/libabl_$v.so; done
run base2 $PWD/blueberry_amd/libblueberry_hip.so
cat $out
