#!/usr/bin/env python3
"""What the box lets an ordinary user read about board power and clocks (sysfs hwmon of the amdgpu device): the inputs of
bench.py's PowerSampler.  Prints the files found and a few samples while a rerank loop runs."""
import glob
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rd(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return f"<{type(e).__name__}>"


def main():
    print("pci_bus_id attr:", getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None),
          getattr(torch.cuda.get_device_properties(0), "pci_device_id", None), getattr(torch.cuda.get_device_properties(0), "pci_domain_id", None))
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        print(k, "=", os.environ.get(k))
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    print("cards:", len(cards))
    for c in cards[:16]:
        hw = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*")))
        print(c, "->", os.path.realpath(c), "vendor", rd(os.path.join(c, "vendor")), "hwmon", hw)
        for h in hw:
            for f in ("name", "power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input", "freq2_input", "temp1_input"):
                print("   ", f, rd(os.path.join(h, f)))
        print("    pp_dpm_sclk", rd(os.path.join(c, "pp_dpm_sclk")).replace("\n", " | ")[:200])
        print("    gpu_busy_percent", rd(os.path.join(c, "gpu_busy_percent")))
    # under load
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
        for c in cards[:16]:
            for h in glob.glob(os.path.join(c, "hwmon", "hwmon*")):
                print("load", os.path.basename(c), rd(os.path.join(h, "power1_average")), rd(os.path.join(h, "power1_input")), rd(os.path.join(h, "freq1_input")),
                      rd(os.path.join(c, "gpu_busy_percent")))
        time.sleep(0.3)


if __name__ == "__main__":
    main()
