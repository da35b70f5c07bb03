# rocprofv3 evidence behind profiles/r03_* (the summaries are copied there): kernel-trace stats of the default
# bench command, HBM and SQ counters of the EDT kernels, the counter passes over the 1 M x 1 M ICP kernels (with
# GRBM_GUI_ACTIVE for the issue fraction), the per-pass kernel times of that pair and the counters of a full Go-ICP run.
set -e
bash $GRAFT_REPO_ROOT/tools/prof_round.sh
bash $GRAFT_REPO_ROOT/tools/pmc_icp1m.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"
bash $GRAFT_REPO_ROOT/tools/prof_icp1m.sh > $GRAFT_REPO_ROOT/gpurun_out/icp_1m_per_iteration.txt 2>&1
GOICP_SETTING=1,255 bash $GRAFT_REPO_ROOT/tools/pmc_goicp.sh "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"
