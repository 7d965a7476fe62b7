O=gpurun_out/r3_small_f32.txt; : > $O
for rep in 1 2; do for f in 0 1; do DT=fp32 MAXSIM_SMALL_F32=$f tools/micro/rank_forward_lat 2>&1 | sed "s/^/SMALL_F32=$f /" | tee -a $O; done; done
MAXSIM_LIB=tools/ab/stamp.so MAXSIM_SMALL_F32=1 python tools/probe_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O
MAXSIM_LIB=tools/ab/stamp.so MAXSIM_SMALL_F32=0 python tools/probe_timeline.py 2>&1 | grep -v amdgpu.ids | tee -a $O
