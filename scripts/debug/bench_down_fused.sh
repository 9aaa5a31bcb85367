# lab: TPS_PP bf16 with the fused down0+down0_1 / down1+down1_1 kernel against the two-kernel route
# (TPSPP_NO_DOWN_FUSED=1), same box, interleaved; then a kernel trace of the fused route
mkdir -p gpurun_out/r4
: > gpurun_out/r4/down_fused.txt
for rep in 1 2; do
  for nb in 512 1024; do
    for sw in 0 1; do
      echo "batch $nb TPSPP_NO_DOWN_FUSED=$sw" >> gpurun_out/r4/down_fused.txt
      TPSPP_NO_DOWN_FUSED=$sw timeout 300 python3 scripts/bench_module.py $nb bf16only 2>&1 | grep -i "img/s" >> gpurun_out/r4/down_fused.txt
    done
  done
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o m -- python3 $GRAFT_REPO_ROOT/scripts/bench_module.py 512 bf16only > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && timeout 60 python3 scripts/kstats.py $f 0.8 > gpurun_out/r4/down_fused_stats.txt
