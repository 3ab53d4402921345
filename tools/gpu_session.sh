#!/bin/bash
# usage (on the GPU box, through gpurun): tools/gpu_session.sh TAG 'cmd1' 'cmd2' ...
# Runs each command with its own timeout, logs everything into gpurun_out/TAG/ (replaces the per-session scripts of round 2).
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
n=0
for cmd in "$@"; do
  n=$((n+1))
  echo "=== [$n] $cmd" | tee -a $OUT/session.log
  ( timeout ${DFX_CMD_TIMEOUT:-900} bash -c "$cmd" ) > $OUT/cmd$n.out 2> $OUT/cmd$n.err
  echo "rc=$? " | tee -a $OUT/session.log
  tail -c 1500 $OUT/cmd$n.out | tee -a $OUT/session.log
  tail -c 600 $OUT/cmd$n.err | tee -a $OUT/session.log
done
