O=gpurun_out/r3_probe6.txt; : > $O
for d in 0 4 8 16 32; do echo -n "DPW=$d " | tee -a $O; MAXSIM_DPW=$d ROUNDS=1 python tools/probe_share_ab.py tools/ab/diag.so 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O; done
