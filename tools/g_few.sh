for rep in 1 2; do
for m in 0 512 1024 2048 4096 8192; do
  echo "== few $m"
  ICPX_BALL_FEW=$m timeout -k 10 120 python3 tools/icp1m_time.py 2>&1 | grep -E "total" | cut -c1-200 || exit 1
done
done
