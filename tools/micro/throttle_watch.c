// throttle_watch -- which limiter holds the shader clock down while a kernel runs?  (GPU box; reads only.)
// Samples the amdgpu gpu_metrics table of every GPU the SMI library enumerates (rsmi_dev_gpu_metrics_info_get: a sysfs read, no
// HIP, nothing written) every PERIOD ms for SECONDS s and prints one line per device and second: average of the XCDs' current
// shader clocks, socket power, hotspot / memory temperature, and the firmware's accumulated throttler residencies as a share of
// its accumulation cycles in that second --
//   PVIOL = d ppt_residency_acc / d accumulation_counter   (package power tracking: the POWER limiter was active)
//   TVIOL = d socket_thm_residency_acc / d accumulation_counter (socket thermal limiter), likewise prochot / vr_thm / hbm_thm
// and, where the table carries them (v1.8), the per-XCD "clock below the host limit because of ppt / thermal" accumulators.
// Run beside a long bench loop (tools/run_throttle_watch.sh): the busy device is the one whose power rises.
//   gcc -O2 tools/micro/throttle_watch.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/micro/throttle_watch
#include <rocm_smi/rocm_smi.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

typedef struct {
  uint64_t acc, ppt, thm, prochot, vr, hbm, below_ppt, below_thm, below_total, low_util;
} Acc;

static int valid16(uint16_t v) { return v != 0xffff; }
static int valid64(uint64_t v) { return v != UINT64_MAX; }

static void take(const rsmi_gpu_metrics_t* m, Acc* a) {
  a->acc = m->accumulation_counter;
  a->ppt = m->ppt_residency_acc;
  a->thm = m->socket_thm_residency_acc;
  a->prochot = m->prochot_residency_acc;
  a->vr = m->vr_thm_residency_acc;
  a->hbm = m->hbm_thm_residency_acc;
  a->below_ppt = a->below_thm = a->below_total = a->low_util = 0;
  for (int x = 0; x < RSMI_MAX_NUM_XCC; ++x) {   // partition 0 holds all XCDs in SPX mode
    const struct amdgpu_xcp_metrics_t* s = &m->xcp_stats[0];
    if (valid64(s->gfx_below_host_limit_ppt_acc[x])) a->below_ppt += s->gfx_below_host_limit_ppt_acc[x];
    if (valid64(s->gfx_below_host_limit_thm_acc[x])) a->below_thm += s->gfx_below_host_limit_thm_acc[x];
    if (valid64(s->gfx_below_host_limit_total_acc[x])) a->below_total += s->gfx_below_host_limit_total_acc[x];
    if (valid64(s->gfx_low_utilization_acc[x])) a->low_util += s->gfx_low_utilization_acc[x];
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
  const int period_ms = argc > 2 ? atoi(argv[2]) : 10;
  const char* stop_file = argc > 3 ? argv[3] : NULL;   // ends early once this file exists
  if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("{\"error\": \"rsmi_init failed\"}\n"); return 1; }
  uint32_t n = 0;
  rsmi_num_monitor_devices(&n);
  if (n > 16) n = 16;
  static rsmi_gpu_metrics_t m;
  Acc first[16], prev[16];
  int ok[16];
  for (uint32_t d = 0; d < n; ++d) {
    ok[d] = rsmi_dev_gpu_metrics_info_get(d, &m) == RSMI_STATUS_SUCCESS;
    uint64_t bdf = 0;
    rsmi_dev_pci_id_get(d, &bdf);
    if (ok[d]) take(&m, &first[d]);
    prev[d] = first[d];
    printf("device %u bdfid 0x%llx metrics %s table v%u.%u\n", d, (unsigned long long)bdf, ok[d] ? "ok" : "unreadable",
           ok[d] ? m.common_header.format_revision : 0, ok[d] ? m.common_header.content_revision : 0);
  }
  printf("# dev  t_s  sclk_MHz(avg of XCDs, min..max over the second)  power_W(avg,max)  hotspot_C mem_C  PVIOL%% TVIOL%% prochot%% vr%% hbm%%  "
         "below_limit: ppt thm total low_util (sum over XCDs, counts)  throttle_status|indep\n");
  const double t0 = now_s();
  double next_line = 1.0;
  double clk_sum[16] = {0}, pw_sum[16] = {0};
  double clk_min[16], clk_max[16], pw_max[16] = {0};
  int cnt[16] = {0};
  uint64_t thr[16] = {0}, ind[16] = {0};
  uint16_t hot[16] = {0}, memt[16] = {0};
  for (uint32_t d = 0; d < 16; ++d) { clk_min[d] = 1e9; clk_max[d] = 0; }
  struct timespec nap = {0, (long)period_ms * 1000000L};
  for (;;) {
    const double t = now_s() - t0;
    for (uint32_t d = 0; d < n; ++d) {
      if (!ok[d] || rsmi_dev_gpu_metrics_info_get(d, &m) != RSMI_STATUS_SUCCESS) continue;
      double c = 0;
      int k = 0;
      for (int x = 0; x < RSMI_MAX_NUM_GFX_CLKS; ++x)
        if (valid16(m.current_gfxclks[x]) && m.current_gfxclks[x]) { c += m.current_gfxclks[x]; ++k; }
      if (k) {
        c /= k;
        clk_sum[d] += c;
        if (c < clk_min[d]) clk_min[d] = c;
        if (c > clk_max[d]) clk_max[d] = c;
      }
      const double p = valid16(m.current_socket_power) ? m.current_socket_power : (valid16(m.average_socket_power) ? m.average_socket_power : 0);
      pw_sum[d] += p;
      if (p > pw_max[d]) pw_max[d] = p;
      ++cnt[d];
      if (m.throttle_status != UINT32_MAX) thr[d] |= m.throttle_status;
      if (valid64(m.indep_throttle_status)) ind[d] |= m.indep_throttle_status;
      hot[d] = m.temperature_hotspot;
      memt[d] = m.temperature_mem;
      if (t >= next_line) {
        Acc a;
        take(&m, &a);
        const double da = (double)(a.acc - prev[d].acc);
        const double f = da > 0 ? 100.0 / da : 0.0;
        printf("%u %5.1f  %6.0f %6.0f..%-6.0f  %6.0f %6.0f  %3u %3u  %6.1f %6.1f %6.1f %6.1f %6.1f  %llu %llu %llu %llu  0x%llx|0x%llx\n", d, t,
               cnt[d] ? clk_sum[d] / cnt[d] : 0.0, clk_min[d], clk_max[d], cnt[d] ? pw_sum[d] / cnt[d] : 0.0, pw_max[d], hot[d], memt[d],
               (a.ppt - prev[d].ppt) * f, (a.thm - prev[d].thm) * f, (a.prochot - prev[d].prochot) * f, (a.vr - prev[d].vr) * f,
               (a.hbm - prev[d].hbm) * f, (unsigned long long)(a.below_ppt - prev[d].below_ppt),
               (unsigned long long)(a.below_thm - prev[d].below_thm), (unsigned long long)(a.below_total - prev[d].below_total),
               (unsigned long long)(a.low_util - prev[d].low_util), (unsigned long long)thr[d], (unsigned long long)ind[d]);
        prev[d] = a;
        clk_sum[d] = pw_sum[d] = pw_max[d] = 0;
        clk_min[d] = 1e9;
        clk_max[d] = 0;
        cnt[d] = 0;
        thr[d] = ind[d] = 0;
      }
    }
    if (t >= next_line) {
      next_line += 1.0;
      fflush(stdout);
      if (stop_file) {
        FILE* f = fopen(stop_file, "r");
        if (f) { fclose(f); break; }
      }
    }
    if (t >= seconds) break;
    nanosleep(&nap, NULL);
  }
  rsmi_shut_down();
  return 0;
}
