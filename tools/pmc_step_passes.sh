cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GE_B=1048576 GE_REPS=4
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc1m/p$i --output-format csv -- python3 $R/tools/step_loop.py > $R/gpurun_out/pmc1m_p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc1m --kernel step_path64 > $R/gpurun_out/pmc1m_summary.txt 2>&1
cat $R/gpurun_out/pmc1m_summary.txt
