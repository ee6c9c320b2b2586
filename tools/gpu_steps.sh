#!/bin/bash
# Run GPU steps one after another; stop at the first step that ends with anything but 0 or 1 (1 = failed assertions /
# a survey that found bad scenes; everything else — timeout 124/137, abort 134, segfault 139 — may be a GPU fault, and
# nothing more is started on the GPU after one).  Usage: tools/gpu_steps.sh 'cmd1' 'cmd2' ...
for step in "$@"; do
  echo "[gpu_steps] $step"
  bash -o pipefail -c "$step"
  rc=$?
  echo "[gpu_steps] rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "[gpu_steps] stopping"; exit $rc; fi
done
exit 0
