set -e
O=gpurun_out/r3_ilv.txt; : > $O
for rep in 1 2; do for lib in base ilv; do
  for cfg in "dep768 fp16" "c5 bf16"; do set -- $cfg
    WL=$1 DT=$2 NQS=1,16,256 MAXSIM_LIB=$PWD/tools/ab/$lib.so timeout -k 10 200 python tools/probe_nq_sweep.py 2>&1 | tail -1 | tee -a $O
  done
done; done
