#!/bin/bash
# LDS counters of the bench kernel (own rocprofv3 --pmc pass; no trace domains)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_lds; mkdir -p gpurun_out/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_lds -o bench -- \
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/pmc_lds/run.log 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/pmc_lds/bench_counter_collection.csv")):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "warp" in k:
        print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
