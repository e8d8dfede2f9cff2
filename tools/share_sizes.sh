for cfg in "random_field 1920 1080 1024" "random_field 1920 1080 128" "bvh_stress 1920 1080 1024" "bvh_stress 1920 1080 128" "cornell_extra 3840 2160 256" "cornell_extra 3840 2160 32" "cornell 1920 1080 512" "cornell 1920 1080 64"; do
  set -- $cfg
  acc=2; case $1 in cornell*) acc=1;; esac
  timeout -k 10 200 python tools/gpu_time.py $1 $2 $3 $4 0 50 $acc 3 2>&1 | grep Grays || exit 1
done
