set -e
O=gpurun_out/r3_probe8.txt; : > $O
for rep in 1 2; do for lib in colbert_amd/libmaxsim.so tools/ab/touch.so; do for dt in fp32 fp16; do
  echo "== $lib $dt" | tee -a $O
  MAXSIM_LIB=$PWD/$lib NDOCS=1000000 NQS=1,2 DTYPE=$dt python tools/bench_small.py 2>&1 | grep -v topk | tee -a $O
done; done; done
