set -e
python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r5b_tests.log 2>&1 || { tail -30 gpurun_out/r5b_tests.log; exit 1; }
tail -3 gpurun_out/r5b_tests.log
for shape in 1 2 3; do
  echo "== shape $shape" 
  MAXSIM_LIB=tools/ab/diag.so MAXSIM_UNI16_SHAPE=$shape WL=mv128,mv128x16 DT=fp16 QDT=fp32 python tools/probe_multiview.py 2>&1 | grep frac
  MAXSIM_LIB=tools/ab/diag.so MAXSIM_UNI16_SHAPE=$shape WL=mv128 DT=fp16 QDT=fp32 NQ=2048 python tools/probe_multiview.py 2>&1 | grep frac
done
echo "== bf16 default"
WL=mv128 DT=bf16 python tools/probe_multiview.py 2>&1 | grep frac
