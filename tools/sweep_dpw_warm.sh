export MAXSIM_LIB=$PWD/tools/ab/diag.so
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py --workload $1 --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2; do
  for d in 8 10 12 16 24; do echo -n "c2 dpw=$d: "; MAXSIM_DPW=$d run c2; done
done
