set -e
O=gpurun_out/r3_kchain2.txt; : > $O
for lib in ${LIBS:-diag kc1 kc2 kc3}; do
  echo "== $lib" | tee -a $O
  MAXSIM_LIB=$PWD/tools/ab/$lib.so MAXSIM_KCHAIN=1 OUT=/tmp/x.pt timeout -k 10 300 python tools/probe_kchain.py 2>&1 | grep KCHAIN | tee -a $O
done
