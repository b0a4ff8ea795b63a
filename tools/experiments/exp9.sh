for w in kitti64 ouster128 dense1m; do
for t in "1000000,1000000" "220,1000000" "180,1000000" "150,1000000" "180,180" "150,120" "120,120"; do
  echo "== $w TUNE=$t"
  S2M_TUNE=$t python tools/prof_kernel.py $w 150 2>/dev/null | head -1 | cut -c1-130
done; done
