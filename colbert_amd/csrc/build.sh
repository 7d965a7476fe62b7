#!/bin/bash
# Builds colbert_amd/libmaxsim.so for gfx950 (cross-compiles without a GPU).  The translation units are compiled in
# parallel and linked into one shared library.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="${MAXSIM_OUT:-$ROOT/colbert_amd/libmaxsim.so}"
OBJ="$ROOT/build/obj$(echo "$OUT" | md5sum | cut -c1-8)"
mkdir -p "$OBJ"
FLAGS=(-O3 --offload-arch=gfx950 -std=c++17 -I"$ROOT/include" -I"$HERE" -fPIC "$@")
pids=()
for tu in maxsim tu_stream tu_stream_small tu_allpairs tu_bigh_rerank tu_bigh_rerank_list tu_bigh_rerank_small tu_bigh_dense tu_bigh_dense_am; do
  "$HIPCC" "${FLAGS[@]}" -c "$HERE/$tu.hip" -o "$OBJ/$tu.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/maxsim.o "$OBJ"/tu_stream.o "$OBJ"/tu_stream_small.o "$OBJ"/tu_allpairs.o "$OBJ"/tu_bigh_rerank.o "$OBJ"/tu_bigh_rerank_list.o "$OBJ"/tu_bigh_rerank_small.o \
    "$OBJ"/tu_bigh_dense.o "$OBJ"/tu_bigh_dense_am.o -o "$OUT"
# the CPython glue of the online call (host side, above the C ABI): only with the product library
if [ -z "${MAXSIM_OUT:-}" ]; then
  PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
  gcc -O2 -shared -fPIC -Wall -I"$PYINC" "$HERE/fastrank.c" -o "$ROOT/colbert_amd/_fastrank.so"
fi
