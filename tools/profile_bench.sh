#!/bin/bash
# The rocprofv3 evidence for the default bench.py run (GPU box): bash tools/profile_bench.sh <tag>
# Writes gpurun_out/prof_<tag>/... ; tools/summarize_profile.py turns it into the text kept under profiles/.
exec bash "$(dirname "$0")/profile_workload.sh" "${1:-run}" cornell_1080p 12 3
