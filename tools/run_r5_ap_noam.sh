#!/bin/bash
for rep in 1 2; do
  for l in "$@"; do echo -n "$l: "; ARGMAX=0 MAXSIM_LIB=$PWD/$l python tools/bench_allpairs_fwd.py 2>&1 | grep all-pairs; done
done
