set -e
export MAXSIM_LIB=$PWD/tools/ab/diag.so
O=gpurun_out/r3_balance.txt; : > $O
python tools/probe_single_768_balance.py 2>&1 | tail -1 | tee -a $O
MAXSIM_SPLIT=2 python tools/probe_single_768_balance.py 2>&1 | tail -1 | tee -a $O
MAXSIM_SPLIT=8 MAXSIM_DPW=4 python tools/probe_single_768_balance.py 2>&1 | tail -1 | tee -a $O
MAXSIM_SPLIT=4 MAXSIM_DPW=2 python tools/probe_single_768_balance.py 2>&1 | tail -1 | tee -a $O
MAXSIM_SPLIT=8 MAXSIM_DPW=8 PER_WG=8 python tools/probe_single_768_balance.py 2>&1 | tail -1 | tee -a $O
