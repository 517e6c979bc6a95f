#!/bin/bash
# Build-container side wrapper of a gpurun call: stamps the commit the snapshot is cut from into .build_commit (the GPU box's copy
# has no .git; tools/pmc_traffic.py writes it into the PMC summaries) and forwards everything to gpurun.
#   tools/gpu.sh --timeout 900 -- 'bash tools/hw_xyz.sh'
cd "$(dirname "$0")/.."
c=$(git rev-parse HEAD 2>/dev/null)
if [ -n "$(git status --porcelain 2>/dev/null | grep -v '^??' | head -1)" ]; then c="$c+dirty"; fi
echo "$c" > .build_commit
# the GPU box runs the prebuilt in-tree libraries: make sure they are the current sources'
python -c "import importlib, fsv2v_amd; b = importlib.import_module('few-shot-vid2vid_amd.build'); b.build_hip(); b.build_emu()" || exit 1
exec /usr/local/graft/bin/gpurun "$@"
