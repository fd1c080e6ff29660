#!/bin/bash
# builds the library first and refuses to spend GPU time on a stale one; usage: scripts/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -m isaac_aligner_amd.build > /tmp/isaac_build.log 2>&1 || { grep -E "error" -A5 /tmp/isaac_build.log | head -40; echo "BUILD FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
