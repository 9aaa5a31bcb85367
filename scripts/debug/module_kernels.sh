#!/bin/bash
# ON THE GPU BOX: per-kernel times of TPS_PP.forward (batch 512) in the three arithmetic modes.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for mode in ${MODES:-fp32only x3only bf16only}; do
  rm -rf /tmp/mk_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk_$mode -o m -- python3 scripts/bench_module.py 512 $mode 2>&1 | grep "TPS_PP"
  python3 scripts/kstats.py $(find /tmp/mk_$mode -name "*kernel_stats.csv" | head -1) ${MINPCT:-1.0}
done
