#!/bin/bash
# Diagnostic build of the library with per-phase stamps in the LDS finisher (GPU box only).
set -e
cd "$(dirname "$0")/.."
cp mergenet_amd/libmergenet_hip.so /tmp/libmergenet_hip.product.so
for T in ${FIN_THREADS:-256}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMN_FIN_STAMPS -DMN_FIN2_THREADS=$T -shared \
    mergenet_amd/csrc/mergenet_hip.hip -o mergenet_amd/libmergenet_hip.so
echo "threads $T"
python tools/gpu_prof.py 1024 2048 16 8192 | grep -E "fin stamps|ms_merge" | sed 's/.*ms_merge/ms_merge/' | cut -c1-260 | tail -2
done
cp /tmp/libmergenet_hip.product.so mergenet_amd/libmergenet_hip.so
