# lab: the token GEMM's block -> tile maps / diagnostics (TPSPP_TOKGEMM_MAP, TPSPP_TOKGEMM_DIAG) under a kernel trace
mkdir -p gpurun_out/r4
for cfg in ${CFGS:-0_0 1_0 2_0 0_2 0_6}; do
  set -- ${cfg/_/ }
  export TPSPP_TOKGEMM_MAP=$1 TPSPP_TOKGEMM_DIAG=$2
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o th -- python3 $GRAFT_REPO_ROOT/scripts/debug/bench_encoder.py 512 5 > /tmp/enc.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && timeout 60 python3 scripts/debug/trace_by_grid.py $f tok_gemm > gpurun_out/r4/enc_m$1_d$2.txt
  grep encoder /tmp/enc.log >> gpurun_out/r4/enc_m$1_d$2.txt
done
