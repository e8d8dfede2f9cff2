for sc in bvh_stress_200 bvh_stress_500 bvh_stress_1000 bvh_stress_2000 bvh_stress_5000 bvh_stress_20000; do
  for r in 0 100; do
    echo -n "ratio=$r "; RTX_TREE_STATS=1 RTX_WIDE_MIN_RATIO=$r timeout -k 10 120 python tools/gpu_time.py $sc 1920 1080 256 0 50 2 3 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
  done
done
