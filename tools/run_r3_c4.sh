export NOLIST=1
O=gpurun_out/r3_c4_uni2.txt; : > $O
export MAXSIM_LIB=tools/ab/diag.so
for rep in 1 2; do
WL=c4 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
for d in 64 50 40 32 25; do MAXSIM_UNI_WAVES=4 MAXSIM_DPW=$d WL=c4 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O; done
done
