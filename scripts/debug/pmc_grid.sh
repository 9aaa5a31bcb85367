#!/bin/bash
# ON THE GPU BOX: hardware counters per (kernel substring, grid size) of one python script -- mean per dispatch.
#   scripts/debug/pmc_grid.sh <kernel-substring> "<COUNTERS ...>" script.py [args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
K=$1; C=$2; shift 2
rm -rf /tmp/pk; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pk -o p -- python3 "$@" > /tmp/pk.log 2>&1
python3 scripts/debug/pmc_by_kernel.py $(find /tmp/pk -name "*counter_collection.csv" | head -1) "$K"
