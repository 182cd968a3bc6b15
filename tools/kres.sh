#!/bin/bash
# Register / LDS usage per kernel of one csrc file: device-only compile + the code object's notes.
#   tools/kres.sh conv_mfma [filter-regex]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; PAT=${2:-.}
EXTRA=""
case $F in voxel_pool|lidar_depth|bri_shell) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -I$R/include -I$R/stereoscene_amd/csrc --cuda-device-only \
    -c $R/stereoscene_amd/csrc/$F.hip -o /tmp/kres_$F.bundle 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
    --input=/tmp/kres_$F.bundle --output=/tmp/kres_$F.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kres_$F.co | python3 $R/tools/kres_parse.py | grep -E "$PAT"
