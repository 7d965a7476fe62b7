set -e
export MAXSIM_LIB=tools/ab/diag.so NOLIST=1
O=gpurun_out/r3_probe4.txt; : > $O
for d in 0 6 8 16 24 48; do MAXSIM_DPW=$d WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
for v in 6 3 4; do MAXSIM_VARIANT=$v WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
MAXSIM_VARIANT=1 WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
MAXSIM_VARIANT=2 WL=ragged python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
MAXSIM_VARIANT=1 WL=c2 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
MAXSIM_VARIANT=2 WL=c2 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
WL=c2 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
