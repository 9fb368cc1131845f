#!/bin/bash
# Runs the given commands (one per argument) in order on the GPU box, each under its own `timeout -k 10`, logging to
# gpurun_out/<name>.log.  A step that fails with an ordinary error does not stop the next one; a step that TIMES OUT or is
# KILLED (rc 124 / 137) ends the sequence — no further GPU step is started after a hang.
# usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out or was killed: stopping here"; exit $rc; fi
done
exit 0
