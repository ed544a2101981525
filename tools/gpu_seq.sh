#!/bin/bash
# Run a sequence of GPU steps on the box (gpurun): each under its own timeout; a step that is KILLED (rc >= 124) ends the
# sequence (no further GPU work after a hang), a step that merely fails (rc 1: a parity mismatch) does not.
# Usage: tools/gpu_seq.sh <tag> '<secs>|<command>' ...   ->  gpurun_out/<tag>_<i>.log
tag=$1; shift
i=0
for step in "$@"; do
  secs=${step%%|*}; cmd=${step#*|}
  i=$((i+1))
  echo "=== step $i: $cmd" | tee gpurun_out/${tag}_$i.log
  timeout -k 10 $secs bash -c "$cmd" >> gpurun_out/${tag}_$i.log 2>&1
  rc=$?
  echo "=== step $i rc=$rc" | tee -a gpurun_out/${tag}_$i.log
  if [ $rc -ge 124 ]; then echo "step $i killed: stopping"; exit $rc; fi
done
exit 0
