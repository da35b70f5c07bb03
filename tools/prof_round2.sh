# rocprofv3 evidence of the round: tools/prof_round.sh (kernel-trace stats of the default bench command, HBM and SQ
# counters of the EDT kernels) followed by the counter passes over the 1 M x 1 M ICP kernels.
set -e
bash $GRAFT_REPO_ROOT/tools/prof_round.sh
bash $GRAFT_REPO_ROOT/tools/pmc_icp1m.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
bash $GRAFT_REPO_ROOT/tools/prof_icp1m.sh > $GRAFT_REPO_ROOT/gpurun_out/icp_1m_per_iteration.txt 2>&1
bash $GRAFT_REPO_ROOT/tools/prof_icp.sh > $GRAFT_REPO_ROOT/gpurun_out/icp_bunny_per_iteration.txt 2>&1
