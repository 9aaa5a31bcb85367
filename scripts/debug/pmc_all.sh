#!/bin/bash
# ON THE GPU BOX: instruction mix per kernel (vector / matrix / scalar instructions per wavefront) for one script.
#   scripts/debug/pmc_all.sh script.py [args]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/pa; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/pa -o p -- python3 "$@" > /tmp/pa.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pa/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:90]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
rows = []
for k, c in acc.items():
    w = max(c['SQ_WAVES'], 1.0)
    rows.append((c['SQ_INSTS_VALU'], k, n[k], c['SQ_INSTS_VALU'] / w, c['SQ_INSTS_MFMA'] / w, c['SQ_INSTS_SALU'] / w))
for tot, k, calls, v, m, s in sorted(rows, reverse=True)[:22]:
    print(f"{k:90s} calls {calls:5d} per wave: VALU {v:7.0f} MFMA {m:6.0f} SALU {s:6.0f}  VALU/MFMA {v / m if m else float('nan'):6.1f}")
PY
