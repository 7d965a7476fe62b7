// maxsim_backward.h -- backward of the all-pairs MaxSim operator (training form: the reference differentiates
// BaseModel.score through torch autograd, colbert/modeling/colbert_model.py:87-96, materialising the
// [q, d, m, n] similarity tensor).  Here the forward records arg-max indices (ReducerArg) and the backward routes
// gradients through them:
//     dQ[q,m,:] = q_mask[q,m] * sum_d g[q,d] * d_mask[d,i] * D[d,i,:]          i = argmax[q,d,m]
//     dD[d,i,:] = d_mask[d,i] * sum_{(q,m): argmax[q,d,m] == i} g[q,d] * q_mask[q,m] * Q[q,m,:]
// Both are gather-reduce problems over rows that sit in L2 / Infinity Cache.  Indices and coefficients are loaded 64
// at a time into lanes and walked with v_readlane, so the row loads of consecutive items are independent and stay
// in flight together (a scalar index load per item would serialise on memory latency).  fp32 accumulation.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// 8 consecutive elements starting at element index i (i % 8 == 0, 16-byte aligned rows) as floats
template <int DT>
__device__ __forceinline__ void load8(const void* p, int64_t i, float (&x)[8]) {
  if constexpr (DT == MAXSIM_F32) {
    const f32x4 a = *(const f32x4*)((const float*)p + i), b = *(const f32x4*)((const float*)p + i + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = a[j]; x[4 + j] = b[j]; }
  } else {
    const u32x4 w = *(const u32x4*)((const uint16_t*)p + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint16_t lo = (uint16_t)(w[j] & 0xffffu), hi = (uint16_t)(w[j] >> 16);
      x[2 * j] = DT == MAXSIM_F16 ? f16_to_f32(lo) : bf16_to_f32(lo);
      x[2 * j + 1] = DT == MAXSIM_F16 ? f16_to_f32(hi) : bf16_to_f32(hi);
    }
  }
}

__device__ __forceinline__ float lane_f32(float v, int l) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

// dQ: one workgroup per (q, m); thread t owns dims t, t + 256, ... (h <= 1024).
template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dq(const void* __restrict__ D, const void* __restrict__ q_mask,
                                                       const void* __restrict__ d_mask, int mask_dtype,
                                                       const int32_t* __restrict__ argmax,
                                                       const float* __restrict__ grad, float* __restrict__ dQ, int nd,
                                                       int Lq, int Ld, int h) {
  const int qm = blockIdx.x;  // q * Lq + m
  const int q = qm / Lq;
  const int m = qm - q * Lq;
  const int lane = threadIdx.x & 63;
  const float qs = mask_dtype != MAXSIM_MASK_NONE ? load_mask(q_mask, mask_dtype, qm) : 1.0f;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int d0 = 0; d0 < nd; d0 += 64) {
    // lane j describes doc d0 + j: row of its arg-max token and the coefficient g * d_mask
    const int d = d0 + lane;
    int64_t row = 0;
    float g = 0.0f;
    if (d < nd) {
      const int i = argmax[((int64_t)q * nd + d) * Lq + m];
      row = (int64_t)d * Ld + i;
      g = grad[(int64_t)q * nd + d];
      if (mask_dtype != MAXSIM_MASK_NONE) g *= load_mask(d_mask, mask_dtype, row);
    }
    // 4 docs at a time: their row loads are independent and in flight together; docs past the end have g = 0, row 0
    for (int j0 = 0; j0 < 64; j0 += 4) {
      if (d0 + j0 >= nd) break;
      float gj[4];
      int64_t base[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gj[t] = lane_f32(g, j0 + t);
        base[t] = (((int64_t)__builtin_amdgcn_readlane((int)(row >> 32), j0 + t) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)row, j0 + t)) * h;
      }
      float x[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = threadIdx.x + 256 * u;
          x[t][u] = k < h ? load_elem<DT>(D, base[t] + k) : 0.0f;
        }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(gj[t], x[t][u], acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = threadIdx.x + 256 * u;
    if (k < h) dQ[(int64_t)qm * h + k] = acc[u] * qs;
  }
}

// dQ, vector form (h % 8 == 0, h <= 1024): 128 threads, thread t owns dims 8t .. 8t+7 and reads them with one 16-byte
// load per row (16-bit inputs) -- the scalar form above moves 2 bytes per lane per load.
template <int DT>
__global__ void __launch_bounds__(128) k_maxsim_bwd_dq_v8(const void* __restrict__ D, const void* __restrict__ q_mask,
                                                          const void* __restrict__ d_mask, int mask_dtype,
                                                          const int32_t* __restrict__ argmax,
                                                          const float* __restrict__ grad, float* __restrict__ dQ,
                                                          int nd, int Lq, int Ld, int h) {
  const int qm = blockIdx.x;
  const int q = qm / Lq;
  const int m = qm - q * Lq;
  const int lane = threadIdx.x & 63;
  const int k0 = threadIdx.x * 8;
  const bool act = k0 < h;
  const float qs = mask_dtype != MAXSIM_MASK_NONE ? load_mask(q_mask, mask_dtype, qm) : 1.0f;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int d0 = 0; d0 < nd; d0 += 64) {
    const int d = d0 + lane;
    int64_t row = 0;
    float g = 0.0f;
    if (d < nd) {
      const int i = argmax[((int64_t)q * nd + d) * Lq + m];
      row = (int64_t)d * Ld + i;
      g = grad[(int64_t)q * nd + d];
      if (mask_dtype != MAXSIM_MASK_NONE) g *= load_mask(d_mask, mask_dtype, row);
    }
    for (int j0 = 0; j0 < 64; j0 += 4) {  // 4 docs' rows in flight
      if (d0 + j0 >= nd) break;
      float gj[4], x[4][8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gj[t] = lane_f32(g, j0 + t);
        const int64_t base = (((int64_t)__builtin_amdgcn_readlane((int)(row >> 32), j0 + t) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)row, j0 + t)) * h;
        if (act) load8<DT>(D, base + k0, x[t]);
      }
      if (act) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[u] = fmaf(gj[t], x[t][u], acc[u]);
      }
    }
  }
  if (act) {
    float* o = dQ + (int64_t)qm * h + k0;
    *(f32x4*)o = f32x4{acc[0] * qs, acc[1] * qs, acc[2] * qs, acc[3] * qs};
    *(f32x4*)(o + 4) = f32x4{acc[4] * qs, acc[5] * qs, acc[6] * qs, acc[7] * qs};
  }
}

// dD: one workgroup (16 waves) per (doc d, 64-dim chunk); the doc's [Ld][64] fp32 gradient slab lives in LDS and
// is written out once -- no global atomics, no read-modify-write of dD.  LDS float atomics are very slow on gfx950
// (~170 cycles per wave-instruction measured), so the slab rows are OWNED: wave w accumulates only the items whose
// arg-max token n has n % 16 == w, with plain ds_read / ds_write.  Every wave scans all (q, m) items 64 at a time
// (indices in lanes), picks its own with a ballot and walks them 4 at a time so the 4 query-row loads are in flight
// together.  The sum order is fixed, so dD is bitwise reproducible.  LDS = Ld * 256 B (Ld <= 600).
template <int DT>
__global__ void __launch_bounds__(1024) k_maxsim_bwd_dd_lds(const void* __restrict__ Q, const void* __restrict__ q_mask,
                                                            const void* __restrict__ d_mask, int mask_dtype,
                                                            const int32_t* __restrict__ argmax,
                                                            const float* __restrict__ grad, float* __restrict__ dD,
                                                            int nq, int nd, int Lq, int Ld, int h, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* slab = (float*)lds;  // [Ld][64]
  const int d = blockIdx.x / nchunk;
  const int k0 = (blockIdx.x - d * nchunk) * 64;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < Ld * 64; i += 1024) slab[i] = 0.0f;
  __syncthreads();
  const int k = min(k0 + lane, h - 1);  // lanes past h (last chunk of a non-multiple-of-64 h) add into unused columns
  const int nitem = nq * Lq;            // items (q, m) in row-major order
  for (int p0 = 0; p0 < nitem; p0 += 64) {
    const int pi = p0 + lane;
    int idx = 0;
    float coef = 0.0f;
    if (pi < nitem) {
      const int q = pi / Lq, m = pi - q * Lq;
      idx = argmax[((int64_t)q * nd + d) * Lq + m];
      coef = grad[(int64_t)q * nd + d];
      if (mask_dtype != MAXSIM_MASK_NONE)
        coef *= load_mask(q_mask, mask_dtype, pi) * load_mask(d_mask, mask_dtype, (int64_t)d * Ld + idx);
    }
    uint64_t own = __ballot(((idx & 15) == wave) && coef != 0.0f);
    while (own) {
      int jj[4];
      float x[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        jj[t] = own ? (int)__builtin_ctzll(own) : -1;
        own = own ? (own & (own - 1)) : own;
        x[t] = jj[t] >= 0 ? load_elem<DT>(Q, (int64_t)(p0 + jj[t]) * h + k) : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (jj[t] >= 0) {
          const float cj = lane_f32(coef, jj[t]);
          float* cell = slab + __builtin_amdgcn_readlane(idx, jj[t]) * 64 + lane;
          *cell = fmaf(cj, x[t], *cell);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Ld * 64; i += 1024) {
    const int n = i >> 6, kk = k0 + (i & 63);
    if (kk < h) dD[((int64_t)d * Ld + n) * h + kk] = slab[i];
  }
}

// ---- dD through a per-doc inverse index (needs a caller-provided workspace) ------------------------------------
// Step 1, one workgroup per doc: the (q, m) items that contribute to the doc (coefficient != 0) are counting-sorted by
// their arg-max token, STABLY -- inside a token's bucket the items stand in ascending item order, so the sums of step 2
// have a fixed order (bitwise reproducible).  ws_start[d][0..Ld] = bucket offsets, ws_items[d][*] = item ids.
// The sort is stable by construction and its cost does not depend on how the arg-maxes spread over the tokens: each of
// the 8 waves owns a contiguous range of items and counts its keys into its own histogram row; a scan over (token, wave)
// gives every wave its first position in every bucket; the wave then places its items 64 at a time in item order.  A
// lane's rank among the batch's lanes with the same key comes from rounds of LDS `min`: every unplaced lane offers its
// lane id to the key's scratch word, the smallest wins the round and takes rank = round number -- as many rounds as the
// batch's most frequent key has lanes (1-3 when the keys spread, 64 when they all agree), each three LDS operations.
// (Round 3's form placed items with LDS atomics and insertion-sorted every bucket with one thread: 0.26 ms of the 2.0 ms
// backward at the reference's step, and quadratic in the longest bucket -- a doc with one live token took ~80 ms.)
// The keys are computed once (a dependent chain of global loads: arg-max, gradient, two masks) and parked in LDS as
// 16-bit values for the placement pass.
// LDS: wcnt[8][Ld + 1] u32, slot[8][Ld + 1] u32, beg[Ld + 1] u32, keys[nitem] u16 (when nitem <= 32768: else recomputed).
constexpr int BWD_INDEX_WAVES = 8;
constexpr int BWD_INDEX_MAX_PARKED = 32768;
__global__ void __launch_bounds__(64 * BWD_INDEX_WAVES) k_maxsim_bwd_index(const void* __restrict__ q_mask,
                                                           const void* __restrict__ d_mask, int mask_dtype,
                                                           const int32_t* __restrict__ argmax,
                                                           const float* __restrict__ grad, int32_t* __restrict__ ws_start,
                                                           int32_t* __restrict__ ws_items, int nq, int nd, int Lq,
                                                           int Ld) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = BWD_INDEX_WAVES, NT = 64 * NW;
  const int L1 = Ld + 1;
  uint32_t* wcnt = (uint32_t*)lds;       // [NW][L1]: per-wave histogram, then per-wave running write positions
  uint32_t* slot = wcnt + NW * L1;       // [NW][L1]: per-wave scratch of the ranking rounds (0xFFFFFFFF = free)
  uint32_t* beg = slot + NW * L1;        // [L1]: bucket starts
  uint16_t* parked = (uint16_t*)(beg + L1);
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nitem = nq * Lq;
  const bool park = nitem <= BWD_INDEX_MAX_PARKED;
  const int per = (nitem + NW - 1) / NW;                       // items per wave (contiguous ranges, in item order)
  const int it0 = min(nitem, wave * per), it1 = min(nitem, it0 + per);
  for (int i = tid; i < NW * L1; i += NT) { wcnt[i] = 0; slot[i] = 0xFFFFFFFFu; }
  __syncthreads();
  auto key_of = [&](int it) -> int {  // arg-max token of item `it`, or -1 if it contributes nothing
    const int q = it / Lq, m = it - q * Lq;
    const int n = argmax[((int64_t)q * nd + d) * Lq + m];
    float c = grad[(int64_t)q * nd + d];
    if (mask_dtype != MAXSIM_MASK_NONE)
      c *= load_mask(q_mask, mask_dtype, it) * load_mask(d_mask, mask_dtype, (int64_t)d * Ld + n);
    return (c != 0.0f && n >= 0 && n < Ld) ? n : -1;
  };
  uint32_t* const mycnt = wcnt + wave * L1;
  uint32_t* const myslot = slot + wave * L1;
#pragma unroll 4
  for (int it = it0 + lane; it < it1; it += 64) {
    const int k = key_of(it);
    if (park) parked[it] = (uint16_t)k;                        // (-1 -> 0xFFFF; Ld <= 1024)
    if (k >= 0) atomicAdd(&mycnt[k], 1u);
  }
  __syncthreads();
  // bucket sizes -> bucket starts (exclusive scan over the tokens, two per thread), then every wave's first position in
  // every bucket
  const int t0 = 2 * tid, t1 = 2 * tid + 1;                    // (Ld <= 2 * NT = 1024 for this kernel: checked on the host)
  uint32_t c0 = 0, c1 = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    c0 += t0 < Ld ? wcnt[w * L1 + t0] : 0u;
    c1 += t1 < Ld ? wcnt[w * L1 + t1] : 0u;
  }
  uint32_t incl = c0 + c1;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t u = __shfl_up(incl, s);
    if (lane >= s) incl += u;
  }
  __shared__ uint32_t wave_tot[NW];
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint32_t u = wave_tot[w];
    base += w < wave ? u : 0u;
    total += u;
  }
  uint32_t run0 = base + incl - c0 - c1, run1 = run0 + c0;     // starts of buckets t0, t1
  if (t0 < Ld) beg[t0] = run0;
  if (t1 < Ld) beg[t1] = run1;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    if (t0 < Ld) { const uint32_t c = wcnt[w * L1 + t0]; wcnt[w * L1 + t0] = run0; run0 += c; }
    if (t1 < Ld) { const uint32_t c = wcnt[w * L1 + t1]; wcnt[w * L1 + t1] = run1; run1 += c; }
  }
  if (tid == 0) beg[Ld] = total;
  __syncthreads();
  for (int i = tid; i <= Ld; i += NT) ws_start[(int64_t)d * L1 + i] = (int32_t)beg[i];
  // placement, in item order, batch by batch: position = the wave's running position in the bucket + the lane's rank among
  // the batch's lanes with the same key.  (Only this wave touches its rows; LDS operations of a wave execute in order.)
  int32_t* const out = ws_items + (int64_t)d * nitem;
  for (int b0 = it0; b0 < it1; b0 += 64) {
    const int it = b0 + lane;
    int k = -1;
    if (it < it1) {
      if (park) { const uint16_t v = parked[it]; k = v == 0xFFFFu ? -1 : (int)v; }
      else k = key_of(it);
    }
    const uint32_t p0 = k >= 0 ? mycnt[k] : 0u;
    bool open = k >= 0;
    int rank = 0;
    for (int round = 0; __any(open); ++round) {
      if (open) atomicMin(&myslot[k], (uint32_t)lane);
      if (open && myslot[k] == (uint32_t)lane) {
        rank = round;
        open = false;
        myslot[k] = 0xFFFFFFFFu;
      }
    }
    if (k >= 0) {
      out[p0 + rank] = it;
      atomicAdd(&mycnt[k], 1u);
    }
  }
}

// Step 2, one wave per doc token (row of dD): dD[d, n, :] = d_mask * sum over the row's items of g * q_mask * Q[item, :].
// Lanes cover the hidden dimension (lane + 64 j); 64 items' coefficients are prepared in lanes and walked with readlane.
template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dd_rows(const void* __restrict__ Q, const void* __restrict__ q_mask,
                                                            const void* __restrict__ d_mask, int mask_dtype,
                                                            const float* __restrict__ grad,
                                                            const int32_t* __restrict__ ws_start,
                                                            const int32_t* __restrict__ ws_items,
                                                            float* __restrict__ dD, int nq, int nd, int Lq, int Ld,
                                                            int h) {
  // lanes cover the hidden dimension 8 dims at a time: lane l owns dims 512 j + 8 l .. + 7 (j = 0, 1; h <= 1024, h % 8 == 0)
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // d * Ld + n
  if (row >= (int64_t)nd * Ld) return;
  const int d = (int)(row / Ld), n = (int)(row - (int64_t)d * Ld);
  const int nitem = nq * Lq;
  const int s = ws_start[(int64_t)d * (Ld + 1) + n], e = ws_start[(int64_t)d * (Ld + 1) + n + 1];
  const int32_t* items = ws_items + (int64_t)d * nitem;
  const float ds = mask_dtype != MAXSIM_MASK_NONE ? load_mask(d_mask, mask_dtype, row) : 1.0f;
  const int ka = 8 * lane, kb = 512 + 8 * lane;
  const bool acta = ka < h, actb = kb < h;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.0f;
  for (int p0 = s; p0 < e; p0 += 64) {
    int it = 0;
    float c = 0.0f;
    if (p0 + lane < e) {
      it = items[p0 + lane];
      const int q = it / Lq;
      c = grad[(int64_t)q * nd + d] * ds;
      if (mask_dtype != MAXSIM_MASK_NONE) c *= load_mask(q_mask, mask_dtype, it);
    }
    const int cnt = min(64, e - p0);
    for (int j0 = 0; j0 < cnt; j0 += 2) {  // two items' rows in flight
      const int i0 = __builtin_amdgcn_readlane(it, j0), i1 = __builtin_amdgcn_readlane(it, min(j0 + 1, cnt - 1));
      const float c0 = lane_f32(c, j0), c1 = (j0 + 1 < cnt) ? lane_f32(c, j0 + 1) : 0.0f;
      float x0a[8], x1a[8], x0b[8], x1b[8];
      if (acta) { load8<DT>(Q, (int64_t)i0 * h + ka, x0a); load8<DT>(Q, (int64_t)i1 * h + ka, x1a); }
      if (actb) { load8<DT>(Q, (int64_t)i0 * h + kb, x0b); load8<DT>(Q, (int64_t)i1 * h + kb, x1b); }
      if (acta) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] = fmaf(c1, x1a[j], fmaf(c0, x0a[j], acc[0][j]));
      }
      if (actb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[1][j] = fmaf(c1, x1b[j], fmaf(c0, x0b[j], acc[1][j]));
      }
    }
  }
  float* o = dD + row * h;
  if (acta) {
    *(f32x4*)(o + ka) = f32x4{acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
    *(f32x4*)(o + ka + 4) = f32x4{acc[0][4], acc[0][5], acc[0][6], acc[0][7]};
  }
  if (actb) {
    *(f32x4*)(o + kb) = f32x4{acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
    *(f32x4*)(o + kb + 4) = f32x4{acc[1][4], acc[1][5], acc[1][6], acc[1][7]};
  }
}

// dD fallback for very long documents (slab does not fit in LDS): global float atomics into a zeroed dD.
template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dd_atomic(const void* __restrict__ Q,
                                                              const void* __restrict__ q_mask,
                                                              const void* __restrict__ d_mask, int mask_dtype,
                                                              const int32_t* __restrict__ argmax,
                                                              const float* __restrict__ grad, float* __restrict__ dD,
                                                              int nd, int Lq, int Ld, int h) {
  const int64_t qd = blockIdx.x;  // q * nd + d
  const int q = (int)(qd / nd);
  const int d = (int)(qd - (int64_t)q * nd);
  const float g = grad[qd];
  if (g == 0.0f) return;
  for (int m = 0; m < Lq; ++m) {
    const int i = argmax[qd * Lq + m];
    float coef = g;
    if (mask_dtype != MAXSIM_MASK_NONE)
      coef *= load_mask(q_mask, mask_dtype, (int64_t)q * Lq + m) * load_mask(d_mask, mask_dtype, (int64_t)d * Ld + i);
    if (coef == 0.0f) continue;
    const int64_t src = ((int64_t)q * Lq + m) * h;
    float* dst = dD + ((int64_t)d * Ld + i) * h;
    for (int k = threadIdx.x; k < h; k += 256) atomicAdd(dst + k, coef * load_elem<DT>(Q, src + k));
  }
}

}  // namespace maxsim
