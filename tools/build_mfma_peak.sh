#!/bin/bash
# builds tools/_probe/mfma_peak (matrix-pipe ceiling probe; run it on the GPU box)
set -e
cd "$(dirname "$0")"
mkdir -p _probe
/opt/rocm/bin/hipcc -w -O3 --offload-arch=gfx950 mfma_peak.hip -o _probe/mfma_peak
