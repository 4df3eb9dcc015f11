for rep in 1 2 3; do for w in 4 8; do
  BB_WAVES_PER_CU=$w tools/timeline.sh wpc${w}_$rep 10000 12000 14500 17700 > /dev/null
  echo "wpc=$w rep=$rep: $(grep -A3 'segment of 600' gpurun_out/timeline_wpc${w}_$rep.txt | grep -o 'segment of 600 dispatches, [0-9.]* us\|dur [a-z_]*\[[0-9]*\] *n= 300  med *[0-9.]*' | sed 's/segment of 600 dispatches,/ |/; s/dur //; s/n= 300  med//' | tr '\n' ' ')"
done; done
