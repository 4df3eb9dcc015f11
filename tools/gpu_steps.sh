#!/bin/bash
# Run the given shell commands one after another on the GPU box, each under its own
# `timeout -k 10`, logging to gpurun_out/<name>.log.  An ordinary failure is recorded and
# the next step still runs; a step that was KILLED (timeout / signal) ends the call --
# nothing else is started on a GPU that may be wedged.
#   usage: gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 $secs bash -c "$cmd" > gpurun_out/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 3 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "== $name was killed: stopping"; exit $rc; fi
done
exit 0
