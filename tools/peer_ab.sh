#!/bin/bash
# Fixed cost of the ways an iteration is closed (tools/exchange_timing.py), product
# library against another build, interleaved.   usage: tools/peer_ab.sh /path/to/other.so
out=gpurun_out/peer_ab.txt; : > $out
for rep in 1 2 3; do
  echo "== product" >> $out; timeout -k 10 200 python tools/exchange_timing.py 17700 2>/dev/null | grep "n=" >> $out
  echo "== other" >> $out; BB_LIB=$1 timeout -k 10 200 python tools/exchange_timing.py 17700 2>/dev/null | grep "n=" >> $out
done
cat $out
