set -e
export MAXSIM_LIB=tools/ab/diag.so NOLIST=1
O=gpurun_out/r3_probe3.txt; : > $O
for d in 4 7 14 28; do MAXSIM_BIGH_SHAPE=81 MAXSIM_DPW=$d WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for s in 0 81 121 141 62; do MAXSIM_BIGH_SHAPE=$s WL=c5 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for s in 0 81 121 141; do MAXSIM_BIGH_SHAPE=$s QDT=fp16 WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for s in 0 81 42; do MAXSIM_BIGH_SHAPE=$s DT=fp32 WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
