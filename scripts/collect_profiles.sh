#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of bench.py + separate PMC passes
# (HBM read / write bytes), plus the same PMC passes over the plain-copy micro-benchmark, whose byte
# count is known, to calibrate the counters (MI355X_MICROARCH.md, HBM section).
# Outputs land in gpurun_out/prof/ ; scripts/summarize_profiles.py turns them into profiles/*.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
STEPS=${STEPS:-400}
# 1. kernel trace + stats of the bench command (steps round-robin on 3 streams: launches overlap), and of the same steps on
#    ONE stream (the kernel's own start-to-end duration)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python3 bench.py --steps $STEPS --warmup 50 --no-cpu-baseline --no-extras > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o bench -- \
    python3 bench.py --steps $STEPS --warmup 50 --no-cpu-baseline --no-extras --streams 1 > $OUT/bench_trace1.log 2>&1
# 1b. the driver's own command, untouched: `python bench.py --gpus 1 --steps 20 --warmup 5` (two streams at this step count,
#     5 regions per protocol, extras and CPU baseline included); the summary lists period and duration of every region
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o bench -- \
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_trace_driver.log 2>&1
# 2. PMC passes (own runs, nothing but --pmc)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- \
      python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_pmc_$c.log 2>&1
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -o copy -- \
      ./scripts/ubench/copy_bench > $OUT/copy_pmc_$c.log 2>&1
done
[ "${ONLY_BENCH:-0}" = "1" ] && { ls -R $OUT | head -30; tail -2 $OUT/bench_trace.log; exit 0; }
# 3. kernel traces of the wider rows: whole TPS++ module, whole recogniser (F1), warp backward (F2)
for w in module head backward; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o $w -- \
      python3 scripts/bench_$w.py > $OUT/$w.log 2>&1
done
# 3b. the bf16 configuration: TPS_PP on bf16 tensors (BASELINE.json configs[2]) and the bf16 conv kernel beside MIOpen
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_module_bf16 -o module_bf16 -- \
    python3 scripts/bench_module.py 512 bf16only > $OUT/module_bf16.log 2>&1
timeout 300 python3 scripts/bench_conv.py > $OUT/conv_bf16.log 2>&1
# 4. MFMA-busy counters (own pass) for the regressor kernels + the calibration kernel (pure MFMA)
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_pmc -o module -- \
    python3 scripts/bench_module.py > $OUT/mfma_module.log 2>&1
for mode in bf16only x3only; do
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_pmc_$mode -o module -- \
      python3 scripts/bench_module.py 512 $mode > $OUT/mfma_module_$mode.log 2>&1
done
# 4b. (round 6) the bf16 backbone + TPS++ (the wide-tile 3x3 kernel, the stem kernel) and the wide layers alone, hot
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_pmc_backbone -o module -- \
    python3 scripts/debug/bench_backbone.py bf16 > $OUT/mfma_backbone.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_pmc_wide -o module -- \
    python3 scripts/debug/bench_wide.py both > $OUT/mfma_wide.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_backbone_bf16 -o backbone_bf16 -- \
    python3 scripts/debug/bench_backbone.py bf16 > $OUT/backbone_bf16.log 2>&1
timeout 300 python3 scripts/debug/backbone_layers.py > $OUT/backbone_layers.log 2>&1
timeout 300 python3 scripts/debug/bench_wide.py both > $OUT/bench_wide.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_module_x3 -o module_x3 -- \
    python3 scripts/bench_module.py 512 x3only > $OUT/module_x3.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma_cal -o cal -- \
    ./scripts/ubench/mfma_bench > $OUT/mfma_cal.log 2>&1
# 5. summaries are made HERE (the raw kernel traces of the recogniser / the driver's command are tens of MB each and
#    gpurun copies back at most 64 MiB): gpurun_out/prof_summary/* is what gets committed under profiles/
TAG=${TAG:-r06}
TPSPP_PROFILE_DST=gpurun_out/prof_summary python3 scripts/summarize_profiles.py $TAG > $OUT/summarize.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +3M -delete
find $OUT -name "*counter_collection.csv" -size +3M -delete
du -sh $OUT gpurun_out/prof_summary
tail -3 $OUT/summarize.log
tail -2 $OUT/bench_trace.log
