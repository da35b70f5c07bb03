for c in 3000000 8000000 60000000; do
  echo "== cand $c"
  ICPX_CAND=$c timeout -k 10 200 python3 tools/icp1m_time.py 16000000 10 2>&1 | grep -E "ms per pass|set_target" | cut -c1-330 || exit 1
done
