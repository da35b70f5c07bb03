for rep in 1 2; do
for d in $GRAFT_REPO_ROOT/build_ab/*/; do
for c in 0 3000000 300000; do
  echo "== $d cand $c"
  ICPX_CAND=$c ICPX_LIB_DIR=$d timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py 2>&1 | grep -E "ms per pass|total|set_target" | cut -c1-130 || exit 1
done
done
done
