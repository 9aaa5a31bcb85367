#!/bin/bash
# A/B of two builds of the library on one GPU box, alternating:  bash scripts/debug/lib_ab.sh <other.so> <reps> <python script + args ...>
# (the other build is copied over tps_pp_amd/libtpspp_hip.so in the box's scratch copy of the repo, the original restored at the end;
# the last line of the script's output is printed per run)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; REPS=$2; shift 2
cp tps_pp_amd/libtpspp_hip.so /tmp/tpspp_this.so
for rep in $(seq 1 "$REPS"); do
  cp /tmp/tpspp_this.so tps_pp_amd/libtpspp_hip.so; echo "this tree : $(python3 "$@" 2>/dev/null | tail -1)"
  cp "$VAR" tps_pp_amd/libtpspp_hip.so;             echo "other .so : $(python3 "$@" 2>/dev/null | tail -1)"
done
cp /tmp/tpspp_this.so tps_pp_amd/libtpspp_hip.so
