export MAXSIM_LIB=tools/ab/diag.so NOLIST=1
O=gpurun_out/r3_probe5.txt; : > $O
for v in 0 1 2 4; do MAXSIM_VARIANT=$v WL=ragged DT=fp16 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for v in 0 1 2; do MAXSIM_VARIANT=$v WL=c2 DT=fp16 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
