// smi_sample -- measurement helper of bench.py (not part of the product): the firmware's own counters of one GPU, read
// through the SMI library's gpu_metrics table (a sysfs read; no HIP, nothing written).  bench.py's PowerSampler takes one
// sample before and one after its timed steps: the differences say how much energy the package took, and for what share
// of the firmware's accumulation cycles the package-power (PPT) and thermal limiters were active -- "this box ran the
// kernel at a lower clock because it sat on its power cap" in the bench line itself.  Built by __graft_entry__.build():
//   gcc -O2 -shared -fPIC tools/smi_sample.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/libsmi_sample.so
#include <rocm_smi/rocm_smi.h>
#include <stdint.h>
#include <string.h>

static int g_open = 0;

// -> device index of the GPU at PCI domain:bus (the SMI library enumerates every amdgpu card it can see), -1: none / no library
int smi_open(uint32_t domain, uint32_t bus) {
  if (!g_open) {
    if (rsmi_init(0) != RSMI_STATUS_SUCCESS) return -1;
    g_open = 1;
  }
  uint32_t n = 0;
  if (rsmi_num_monitor_devices(&n) != RSMI_STATUS_SUCCESS) return -1;
  for (uint32_t d = 0; d < n; ++d) {
    uint64_t bdf = 0;
    if (rsmi_dev_pci_id_get(d, &bdf) != RSMI_STATUS_SUCCESS) continue;
    if ((uint32_t)(bdf >> 32) == domain && (uint32_t)((bdf >> 8) & 0xff) == bus) return (int)d;
  }
  return -1;
}

// out[0] accumulation_counter  [1] ppt_residency_acc  [2] socket_thm_residency_acc  [3] prochot_residency_acc
// [4] vr_thm_residency_acc  [5] hbm_thm_residency_acc  [6] energy_accumulator (counts)  [7] current socket power, W
// [8] mean of the XCDs' current (target) shader clocks, MHz  [9] hotspot C  [10] memory C  [11] energy counter resolution, uJ
// [12] the energy counter's timestamp, ns
// fields the table does not carry are -1.  returns 0, or -1 when the table cannot be read.
int smi_read(int dev, double* out) {
  static rsmi_gpu_metrics_t m;
  if (!g_open || dev < 0) return -1;
  if (rsmi_dev_gpu_metrics_info_get((uint32_t)dev, &m) != RSMI_STATUS_SUCCESS) return -1;
#define U64(v) ((v) == UINT64_MAX ? -1.0 : (double)(v))
#define U16(v) ((v) == 0xffff ? -1.0 : (double)(v))
  out[0] = U64(m.accumulation_counter);
  out[1] = U64(m.ppt_residency_acc);
  out[2] = U64(m.socket_thm_residency_acc);
  out[3] = U64(m.prochot_residency_acc);
  out[4] = U64(m.vr_thm_residency_acc);
  out[5] = U64(m.hbm_thm_residency_acc);
  out[6] = -1.0;
  out[11] = out[12] = -1.0;
  {
    uint64_t e = 0, ts = 0;
    float res = 0.f;
    if (rsmi_dev_energy_count_get((uint32_t)dev, &e, &res, &ts) == RSMI_STATUS_SUCCESS) {
      out[6] = (double)e;
      out[11] = (double)res;
      out[12] = (double)ts;
    }
  }
  out[7] = U16(m.current_socket_power);
  if (out[7] < 0) out[7] = U16(m.average_socket_power);
  double c = 0;
  int k = 0;
  for (int x = 0; x < RSMI_MAX_NUM_GFX_CLKS; ++x)
    if (m.current_gfxclks[x] != 0xffff && m.current_gfxclks[x]) { c += m.current_gfxclks[x]; ++k; }
  out[8] = k ? c / k : -1.0;
  out[9] = U16(m.temperature_hotspot);
  out[10] = U16(m.temperature_mem);
  return 0;
}

void smi_close(void) {
  if (g_open) rsmi_shut_down();
  g_open = 0;
}
