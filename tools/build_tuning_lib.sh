#!/bin/bash
# Measurement build of the library: the same sources with -DGCN_TUNING, which turns on the GCN_TUNE_* environment
# overrides (grid sizes, row-chunk length, product slices, traffic probes).  Written next to the product library as
# lib/libgcn_uelb_hip_tuning.so; select it with GCN_UELB_LIB=<path> (read by the Python binding only).
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGCN_TUNING "$@" \
    -o graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so graph-convnet-uelb_amd/csrc/api.hip
# source stamp (first word: hash of csrc/, checked by the binding when it loads the library; then the build flags)
python3 -c "import sys; sys.path.insert(0, 'graph-convnet-uelb_amd'); from src.gcn.hip import lib as L; print(L.source_hash(), '-DGCN_TUNING', *sys.argv[1:])" "$@" \
    > graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so.srchash
echo "built graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so"
