#!/bin/bash
# Build here (the prebuilt .so travels with the snapshot and must match the sources: cdml_build_id), then run on the GPU box.
# usage: tools/gpu.sh TIMEOUT_S 'command'
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
