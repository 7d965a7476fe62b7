#!/bin/bash
# Builds colbert_amd/libmaxsim.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -I"$ROOT/include" -shared -fPIC \
    -I"$HERE" "$HERE/maxsim.hip" -o "$ROOT/colbert_amd/libmaxsim.so" "$@"
