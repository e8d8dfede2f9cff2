#!/bin/bash
# One bench.py line per workload at one GPU's share of the eight-GPU configuration and at the full launch
# (GPU box): bash tools/share_bench.sh <out.jsonl>
OUT=${1:-gpurun_out/share_lines.jsonl}
: > $OUT
for wl in cornell_1080p random_field_1080p bvh_stress_1080p cornell_extra_4k; do
  timeout -k 10 300 python bench.py --workload $wl --share-of 8 --steps 8 --warmup 2 --no-cpu-baseline >> $OUT || exit 1
  timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline >> $OUT || exit 1
done
python - "$OUT" <<'PY'
import json, sys
lines = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
for share, full in zip(lines[0::2], lines[1::2]):
    print(f"{share['config']['workload']:20s} share {share['config']['spp_per_step_per_gpu']:5d} spp: {share['value'] / 1e3:7.2f} Grays/s   "
          f"full {full['config']['spp_per_step_per_gpu']:5d} spp: {full['value'] / 1e3:7.2f} Grays/s   efficiency_vs_full_launch {share['value'] / full['value']:.3f}")
PY
