#!/bin/bash
# tools/mode_cells.py over the named scenes, linear scan (accel 1) and BVH (accel 2), on the GPU box.
for s in small cornell_classic cornell_extra textured sky_spheres sky_sphere_lit cornell_light_tree cornell_side_lights cornell_many_lights random_field bvh_stress_2000; do
  timeout -k 10 200 python tools/mode_cells.py $s 960 540 512 1 2>/dev/null | tail -n 1 || exit 1
done
for s in random_field bvh_stress_2000 cornell_extra textured cornell_many_lights_10; do
  timeout -k 10 200 python tools/mode_cells.py $s 960 540 512 2 2>/dev/null | tail -n 1 || exit 1
done
