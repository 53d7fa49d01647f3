#!/bin/bash
# Build-container side of a GPU call: stamp the tree (commit, dirty flag, time) into .dk_build_stamp -- the snapshot gpurun pushes has no .git --
# then hand the command line to gpurun.  scripts/gpu_round.sh copies the stamp into the header of every profile it writes.
#   scripts/gpu_call.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
DIRTY=$(git status --porcelain -- diffusionkit_amd include bench.py scripts tests | grep -v '^??' | wc -l)
echo "commit $(git rev-parse --short HEAD)$([ "$DIRTY" -gt 0 ] && echo "+${DIRTY} uncommitted files") $(date -u +%Y-%m-%dT%H:%MZ)" > .dk_build_stamp
exec /usr/local/graft/bin/gpurun "$@"
