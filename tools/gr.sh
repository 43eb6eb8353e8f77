#!/bin/bash
# build the library (make decides what is stale), then hand the command line to gpurun:  tools/gr.sh [--timeout N] -- 'command'
set -e
make -C "$(dirname "$0")/../hybrid-ctunet_amd/csrc" -j8 >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
