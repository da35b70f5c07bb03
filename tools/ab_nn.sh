# A/B of the library builds under build_ab/: per-kernel times of the 1 M x 1 M ICP passes, wall time of the 30 passes,
# and the bunny pair through the host class.
for d in $GRAFT_REPO_ROOT/build_ab/*/; do
  echo "== $d"
  ICPX_LIB_DIR=$d bash $GRAFT_REPO_ROOT/tools/prof_icp1m.sh 2>&1 | grep "us per pass" | cut -c1-400 || exit 1
  ICPX_LIB_DIR=$d timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py | grep total || exit 1
  ICPX_LIB_DIR=$d timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/icp_bunny_time.py | tail -1 || exit 1
done
