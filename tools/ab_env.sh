#!/bin/bash
# A/B timing of one library under two environments on the GPU box, alternating runs:
#   bash tools/ab_env.sh "<VAR=value>" <gpu_time args...>     (prints base, with, base, with)
SETTING=$1; shift
for rep in 1 2; do
  echo -n "base: "; timeout -k 10 150 python tools/gpu_time.py "$@" 2>/dev/null | tail -n 1 || exit 1
  echo -n "$SETTING: "; env "$SETTING" timeout -k 10 150 python tools/gpu_time.py "$@" 2>/dev/null | tail -n 1 || exit 1
done
