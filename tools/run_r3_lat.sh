export NDOCS=1000000
O=gpurun_out/r3_lat_fused.txt; : > $O
for rep in 1 2; do
for f in 1 0; do
  echo "== fp32 MAXSIM_FUSED=$f" | tee -a $O
  MAXSIM_LIB=tools/ab/diag.so MAXSIM_FUSED=$f python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "rank_forward total\|GPU span\|sync=" | tee -a $O
  echo "== fp16 MAXSIM_FUSED=$f" | tee -a $O
  MAXSIM_LIB=tools/ab/diag.so MAXSIM_FUSED=$f DTYPE=fp16 python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | grep "rank_forward total\|GPU span\|sync=" | tee -a $O
done
done
