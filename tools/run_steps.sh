#!/bin/bash
# Runs the commands of a step file one after the other on the GPU box, each under its own timeout, logs under gpurun_out/<tag>/.
# An ordinary failure (assertion, non-zero exit) does not stop the following steps; a step that had to be KILLED does (a hung
# kernel must not be followed by more GPU work).   usage: tools/run_steps.sh <tag> <stepfile>   (lines: "<seconds> <name> <command...>")
tag=$1; steps=$2
out=gpurun_out/$tag
mkdir -p $out
while read -r secs name cmd; do
  [ -z "$secs" ] && continue
  case "$secs" in \#*) continue;; esac
  echo "=== $name (limit ${secs}s): $cmd" | tee -a $out/steps.log
  t0=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $out/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $out/steps.log
  tail -n 12 $out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping" | tee -a $out/steps.log; exit 1; fi
done < $steps
exit 0
