#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: the small-launch threshold of pick_docs_per_wave (MAXSIM_MIN_WGS) across workloads and batch sizes.
for wl in "c2" "ragged" "c4" "c5" "c2 --index-dtype fp16"; do for nq in 2 4 8 16 32; do for m in 2048 448 224; do
  nd=200000; [ "${wl:0:2}" = "c4" ] && nd=2000000; [ "$wl" = "c5" ] && nd=40000
  echo -n "[$wl] nq=$nq min_wgs=$m: "; MAXSIM_MIN_WGS=$m python bench.py --workload $wl --nq $nq --ndocs $nd --steps 60 --no-cpu-baseline 2>/dev/null | python tools/ms.py
done; done; done
