set -e
export MAXSIM_LIB=tools/ab/diag.so NOLIST=1
O=gpurun_out/r3_probe7.txt; : > $O
for v in 0 1 2; do
  MAXSIM_VARIANT=$v WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
  MAXSIM_VARIANT=$v WL=ragged UNIFORM=1 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
  MAXSIM_VARIANT=$v WL=ragged RAGGED=120,5,8,180 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
  MAXSIM_VARIANT=$v WL=ragged RAGGED=120,40,96,180 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
done
