set -e
O=gpurun_out/r3_nqsweep.txt; : > $O
for cfg in ${CFGS:-"dep768 fp16" "c5 bf16" "c2 fp16" "c2 fp32" "ragged fp32"}; do set -- $cfg
  for lib in ${LIBS:-base new base new}; do
    WL=$1 DT=$2 MAXSIM_LIB=$PWD/tools/ab/$lib.so timeout -k 10 200 python tools/probe_nq_sweep.py 2>&1 | tail -1 | tee -a $O
  done
done
