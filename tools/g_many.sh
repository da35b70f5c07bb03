for rep in 1 2; do
for m in 8 4 3 2 1; do
  echo "== many $m"
  ICPX_BALL_MANY=$m timeout -k 10 120 python3 tools/icp1m_time.py 2>&1 | grep -E "ms per pass|total" | cut -c1-200 || exit 1
done
done
