#!/bin/bash
# On the GPU box: A/B of alternative builds (tools/ab/*.so, selected through MAXSIM_LIB) against the in-tree library,
# interleaved so that box-to-box and thermal drift cancel.  usage: tools/ab.sh "<bench args>" lib1.so lib2.so ...
ARGS="$1"; shift
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for rep in 1 2 3; do
  echo -n "base: "; python bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
  for l in "$@"; do echo -n "$l: "; MAXSIM_LIB=$PWD/$l python bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
done
