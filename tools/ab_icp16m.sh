for rep in 1 2; do
for d in $GRAFT_REPO_ROOT/build_ab/*/; do
  echo "== $d"
  ICPX_LIB_DIR=$d timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py 2>&1 | grep -E "total" | cut -c1-330 || exit 1
  ICPX_LIB_DIR=$d timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py 16000000 10 2>&1 | grep -E "ms per pass" | cut -c1-330 || exit 1
done
done
