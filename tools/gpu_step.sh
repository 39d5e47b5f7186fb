#!/bin/bash
# Usage (GPU box): bash tools/gpu_step.sh <label> <seconds> <command...>
# Runs one GPU step under its own timeout, logs to gpurun_out/<label>.log, and exits non-zero ONLY if the step was
# killed (timeout / signal): a failing test or a failed pose check does not stop the steps that follow it, a hang does.
LABEL=$1; SECS=$2; shift 2
mkdir -p gpurun_out
timeout -k 10 "$SECS" "$@" > gpurun_out/$LABEL.log 2> gpurun_out/$LABEL.err
rc=$?
echo "[$LABEL] rc=$rc" | tee -a gpurun_out/steps.txt
if [ $rc -ge 124 ]; then exit $rc; fi
exit 0
