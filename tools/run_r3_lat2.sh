O=gpurun_out/r3_lat_cpp.txt; : > $O
for rep in 1 2; do for dt in fp32 fp16; do for f in 1 0; do
  DT=$dt MAXSIM_FUSED=$f tools/micro/rank_forward_lat 2>&1 | tee -a $O
done; done; done
