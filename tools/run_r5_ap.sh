#!/bin/bash
# On the GPU box: interleaved A/B of all-pairs (training-form forward) builds: tools/run_r5_ap.sh lib1.so lib2.so ...
for rep in 1 2 3; do
  for l in "$@"; do echo -n "$l: "; MAXSIM_LIB=$PWD/$l python tools/bench_allpairs_fwd.py 2>&1 | grep all-pairs; done
done
