cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_lab
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_lab -o lab -- scripts/ubench/warp_lab scripts/ubench/warp_lab_consts.bin tps_pp_amd/libtpspp_hip.so 300 "m8 nload=3 store=1" > gpurun_out/pmc_lab/run.txt 2>&1
ls -R gpurun_out/pmc_lab | head -30
