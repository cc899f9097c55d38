#!/bin/bash
# A/B of two builds of libpandrs_hip.so on ONE box: experiments/ab/base (git archive of a commit, built there) against the tree's.
# usage: experiments/ab.sh SCRIPT [args]   -> alternates base / new / base / new, separate processes
set -e
B=experiments/ab/base/pandrs_amd/libpandrs_hip.so
for round in 1 2; do
  echo "== base"; PANDRS_HIP_LIB=$B python "$@" 2>&1 | grep -v amdgpu.ids
  echo "== new";  python "$@" 2>&1 | grep -v amdgpu.ids
done
