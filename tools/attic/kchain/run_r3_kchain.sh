set -e
export MAXSIM_LIB=$PWD/tools/ab/diag.so
O=gpurun_out/r3_kchain.txt; : > $O
for q in fp32 fp16; do
  MAXSIM_KCHAIN=0 QDT=$q OUT=/tmp/k0_$q.pt timeout -k 10 300 python tools/probe_kchain.py 2>&1 | grep KCHAIN | tee -a $O
  MAXSIM_KCHAIN=1 QDT=$q OUT=/tmp/k1_$q.pt timeout -k 10 300 python tools/probe_kchain.py 2>&1 | grep KCHAIN | tee -a $O
  python - <<PY 2>&1 | tee -a $O
import torch
a, b = torch.load("/tmp/k0_$q.pt"), torch.load("/tmp/k1_$q.pt")
for k in a:
    same = torch.equal(a[k], b[k])
    d = (a[k] - b[k]).abs().nan_to_num(0, 0, 0).max().item()
    print("q=$q", k, "bit-identical" if same else f"DIFFERENT max |d| = {d:.3g}", tuple(a[k].shape))
PY
done
