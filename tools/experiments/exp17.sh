python -m pytest tests/test_tiers_gpu.py tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -2
for cfg in "S2M_SPLIT=0" "S2M_SPLIT=2 S2M_LEAN_EPW=1" "S2M_SPLIT=2 S2M_LEAN_EPW=1 S2M_SEARCH_GRID=8" "S2M_SPLIT=2 S2M_LEAN_EPW=1 S2M_SEARCH_GRID=64" "S2M_SPLIT=2 S2M_LEAN_EPW=2" "S2M_SPLIT=2 S2M_LEAN_EPW=1 S2M_SPLIT_FROM=6"; do
  echo "== $cfg"
  env $cfg python tools/bench_batch.py kitti64 20 4,8 0 2>/dev/null | grep '"B"' | cut -c1-150 | head -2
done
