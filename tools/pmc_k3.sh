# SQ counter passes for two rollout workgroup sizes (historical diagnostic: needs the TOLG_ROLL_WG switch, removed after the experiment;
# results: profiles/r01_k3_cu_sharing_pmc.json)
export TMPDIR=/tmp
for wg in 64 256; do
  export TOLG_ROLL_WG=$wg
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH -d gpurun_out/pmc_k3b_$wg -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_k3b_$wg.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH_LEVEL -d gpurun_out/pmc_k3c_$wg -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_k3c_$wg.log 2>&1 || exit 1
done
echo done
