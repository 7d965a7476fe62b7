export NOLIST=1
O=gpurun_out/r3_c4_uni3.txt; : > $O
python -m pytest tests -m gpu -x -q -k "uniform or c4 or multiview or short or half or q_mask or sweep or dispatch" 2>&1 | tail -3 | tee -a $O
for rep in 1 2 3; do
MAXSIM_LIB=tools/ab/diag.so WL=c4 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
MAXSIM_LIB=tools/ab/base_uni.so WL=c4 python tools/probe_workload.py 2>&1 | tail -1 | tee -a $O
done
