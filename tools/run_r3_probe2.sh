set -e
export MAXSIM_LIB=tools/ab/diag.so
O=gpurun_out/r3_probe2.txt; : > $O
for s in 0 81 42 41; do MAXSIM_BIGH_SHAPE=$s WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for d in 2 4 14; do MAXSIM_DPW=$d WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
UNIFORM=1 WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
QDT=fp16 WL=dep768 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
WL=c2 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
WL=c2 DT=fp16 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
WL=ragged DT=fp16 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
