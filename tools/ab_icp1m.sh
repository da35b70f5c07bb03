# A/B on one box: wall time of the passes of the 1 M x 1 M ICP for every library build under build_ab/, twice, interleaved.
for rep in 1 2; do
for d in $GRAFT_REPO_ROOT/build_ab/*/; do
  echo "== $d"
  ICPX_LIB_DIR=$d timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/icp1m_time.py 2>&1 | grep -E "ms per pass|total" | cut -c1-330 || exit 1
done
done
