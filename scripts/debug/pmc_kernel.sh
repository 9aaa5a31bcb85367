#!/bin/bash
# ON THE GPU BOX: a few hardware counters for one kernel name (substring) of one python script.
#   scripts/debug/pmc_kernel.sh <kernel-substring> "<COUNTERS ...>" script.py [args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
K=$1; C=$2; shift 2
rm -rf /tmp/pk; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pk -o p -- python3 "$@" > /tmp/pk.log 2>&1
python3 - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
f = glob.glob('/tmp/pk/**/*counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/pk.log').read()[-2000:]); sys.exit(1)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if k in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in acc.items():
    print(f"{c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
