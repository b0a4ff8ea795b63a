for t in "220,1000000,0" "220,1000000,1" "180,1000000,1" "260,1000000,1"; do
  echo "== TUNE=$t"
  S2M_TUNE=$t python tools/prof_kernel.py kitti64 150 2>/dev/null | head -1 | cut -c1-140
done
