set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -15 > gpurun_out/r3_t1.log || { cat gpurun_out/r3_t1.log; exit 1; }
cat gpurun_out/r3_t1.log
python tools/probe_worklist.py 2>&1 | tail -4 | tee gpurun_out/r3_wl_default.log
for w in 512 1024 2048 8192 16384; do MAXSIM_LIB=tools/ab/diag.so MAXSIM_LIST_WGS=$w python tools/probe_worklist.py 2>&1 | tail -3 | tee -a gpurun_out/r3_wl_sweep.log; done
