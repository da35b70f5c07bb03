# tools/prof_icp1m.sh for every library build under build_ab/ (ablation / A-B variants)
for d in $GRAFT_REPO_ROOT/build_ab/*/; do echo "== $d"; ICPX_LIB_DIR=$d bash $GRAFT_REPO_ROOT/tools/prof_icp1m.sh 2>&1 | grep "us per pass" | cut -c1-400; done
