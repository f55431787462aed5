// EEGNet temporal convolution (1 x 64 taps, 8 filters) on the matrix cores -- bf16 storage / compute only.
// Replaces, for dtype bf16, the VALU kernels k_eeg_conv1 / the weight-gradient half of k_eeg_conv1_bwd (eeg.hip), which
// are bound by fp32 FMA issue (2.5 GFLOP per step each).  Reference: root/src/models/models.py:250-253,272 (conv1 of EEGNet).
//
// Per (sample, electrode) row the convolution is a GEMM with a Toeplitz operand:
//   forward   D[f][t] = sum_k W[f][k] * x[t + k - 31]          A = W (8 filters zero-padded to 16 rows), B = x windows
//   wgrad     D[f][k] = sum_t dC1[f][t] * x[t + k - 31]        A = dC1, B = x windows, reduction over time
// A B-fragment lane needs 8 consecutive samples starting at an arbitrary (per lane) position.  The padded row is kept in
// LDS as bf16 in TWO copies, the second shifted by one sample, so any window starts on a 4-byte boundary of copy
// (start & 1) and is fetched with four ds_read_b32 (copy pitch = 64 bytes mod 128: the two copies sit on disjoint bank
// halves).  8.5 KB of LDS per row keeps every row's workgroup resident at once (eight shifted copies -- one ds_read_b128
// per window -- cost 34 KB, four workgroups per CU, and 1216 rows then ran as a full round plus a 19 % tail round).
#include "bx_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define EM_K 64            // taps
#define EM_PADL 31         // 'same' padding of an even kernel: left (K-1)/2
#define EM_OFF 32          // xpad[j] = x[j - EM_OFF]

struct EmGeom { int B, Ch, T, NP; };      // NP = samples per shifted copy (multiple of 8)

__host__ __device__ __forceinline__ int em_np(int T) {      // samples per copy: multiple of 8, pitch in bytes = 64 mod 128
  int np = ((T + 112 + 7) / 8) * 8;
  while ((np * 2) % 128 != 64) np += 8;
  return np;
}

// builds the two bf16 copies of one row in LDS: copy0[i] = xpad[i], copy1[i] = xpad[i + 1]  (copy c at xs + c*NP)
__device__ __forceinline__ void em_build_copies(const float* __restrict__ xrow, bf16_t* xs, int T, int NP) {
  const __amdgpu_buffer_rsrc_t xres = bx_rsrc(xrow, (uint32_t)T * 4u);
  for (int m0 = threadIdx.x; m0 < NP / 8; m0 += 256) {
    float v[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int t = m0 * 8 + j - EM_OFF;
      v[j] = bx_ldf_or0(xres, t, t >= 0 && t < T);          // zero padding without a branch per load
    }
    uint32_t w0[4], w1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      w0[k] = pack2bf(v[2 * k], v[2 * k + 1]);
      w1[k] = pack2bf(v[2 * k + 1], v[2 * k + 2]);
    }
    *reinterpret_cast<uint4*>(xs + m0 * 8) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
    *reinterpret_cast<uint4*>(xs + NP + m0 * 8) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
  }
  __syncthreads();
}
// the same in two halves, so that a kernel can request the NEXT row while it still multiplies the current one: a thread's (up to)
// two items m0 = tid, tid + 256 (rows with NP / 8 <= 512, i.e. T <= 3984)
struct EmRow { float v[2][9]; };
__device__ __forceinline__ void em_row_request(const float* __restrict__ xrow, int T, int NP, EmRow& r) {
  const __amdgpu_buffer_rsrc_t xres = bx_rsrc(xrow, (uint32_t)T * 4u);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m0 = threadIdx.x + 256 * q;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int t = m0 * 8 + j - EM_OFF;
      r.v[q][j] = bx_ldf_or0(xres, t, m0 < NP / 8 && t >= 0 && t < T);
    }
  }
}
__device__ __forceinline__ void em_row_store(bf16_t* xs, int NP, const EmRow& r) {      // the caller's barrier follows
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m0 = threadIdx.x + 256 * q;
    if (m0 < NP / 8) {
      uint32_t w0[4], w1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        w0[k] = pack2bf(r.v[q][2 * k], r.v[q][2 * k + 1]);
        w1[k] = pack2bf(r.v[q][2 * k + 1], r.v[q][2 * k + 2]);
      }
      *reinterpret_cast<uint4*>(xs + m0 * 8) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
      *reinterpret_cast<uint4*>(xs + NP + m0 * 8) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    }
  }
}
// 8 consecutive samples xpad[e .. e+8): four 4-byte reads from the copy whose parity matches e
__device__ __forceinline__ bf16x8 em_window(const bf16_t* xs, int NP, int e) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(xs + (size_t)(e & 1) * NP + (e & ~1));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 d = {p[0], p[1], p[2], p[3]};
  return __builtin_bit_cast(bf16x8, d);
}

// ---- forward: c1[b][f][ch][t] (bf16) + BatchNorm partial sums [row][2][8]; one workgroup (4 waves) per row.
// MFMA roles: A = x windows (16 time steps), B = W^T (8 filters zero-padded to 16 columns): a lane of D then holds FOUR
// CONSECUTIVE time steps of one filter -> one packed conversion pair and one 8-byte store per lane and tile (with the
// roles swapped a lane held four filters of one time step: four 2-byte stores and 4x the conversion work made the kernel
// VALU-bound).
__global__ __launch_bounds__(256) void k_eeg_conv1_mfma(const float* __restrict__ x, const float* __restrict__ w1, bf16_t* __restrict__ c1,
                                                        float* __restrict__ partials, EmGeom g, int want_stats) {
  extern __shared__ __attribute__((aligned(16))) bf16_t xs[];         // [2][NP]
  __shared__ float red[4][16];
  const int row = blockIdx.x, b = row / g.Ch, ch = row % g.Ch;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gq = lane >> 4, li = lane & 15;
  // B fragments: W[f = li][32 s + 8 gq + j] (k = 8 gq + j, column f), columns 8..15 are zero
  const __amdgpu_buffer_rsrc_t wres = bx_rsrc(w1, 8u * EM_K * 4u);
  bf16x8 wfr[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) wfr[s][j] = (short)f2bf(bx_ldf_or0(wres, li * EM_K + 32 * s + 8 * gq + j, li < 8));
  em_build_copies(x + (size_t)row * g.T, xs, g.T, g.NP);
  float sa = 0.f, qa = 0.f;
  const int ntile = (g.T + 15) / 16;
  const bool vec = (g.T & 3) == 0;
  bf16_t* orow = c1 + (((size_t)b * 8 + (li & 7)) * g.Ch + ch) * g.T;
  for (int tt = wave; tt < ntile; tt += 4) {
    const int t0 = tt * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // A[t][k]: lane (row t0 + li, k = 32 s + 8 gq + j) -> x[t + k - 31] = xpad[t + k + 1]
      const bf16x8 xfr = em_window(xs, g.NP, t0 + li + 32 * s + 8 * gq + (EM_OFF - EM_PADL));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xfr, wfr[s], acc, 0, 0, 0);
    }
    // D: lane (col f = li, rows t0 + 4 gq + r)
    const int t = t0 + 4 * gq;
    if (li < 8 && t < g.T) {
      const uint32_t p0 = pack2bf(acc[0], acc[1]), p1 = pack2bf(acc[2], acc[3]);
      if (vec) {
        *reinterpret_cast<uint2*>(orow + t) = make_uint2(p0, p1);
      } else {
        const bf16_t h[4] = {(bf16_t)(p0 & 0xffffu), (bf16_t)(p0 >> 16), (bf16_t)(p1 & 0xffffu), (bf16_t)(p1 >> 16)};
#pragma unroll
        for (int r = 0; r < 4; ++r) if (t + r < g.T) orow[t + r] = h[r];
      }
      const float v[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t + r < g.T) { sa += v[r]; qa += v[r] * v[r]; }          // statistics of the stored values
    }
  }
  if (!want_stats) return;
  sa += __shfl_xor(sa, 16, 64); qa += __shfl_xor(qa, 16, 64);        // the four row groups of filter li
  sa += __shfl_xor(sa, 32, 64); qa += __shfl_xor(qa, 32, 64);
  if (lane < 8) { red[wave][lane] = sa; red[wave][8 + lane] = qa; }
  __syncthreads();
  if (threadIdx.x < 16)      // layout [row][2][8], as k_eeg_conv1 writes it
    partials[(size_t)row * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

size_t bx_eeg_conv1_mfma_lds(int T) { return (size_t)2 * (size_t)em_np(T) * sizeof(bf16_t); }

int bx_eeg_conv1_mfma_launch(const float* x, const float* w1, void* c1, float* partials, int B, int Ch, int T, int want_stats, hipStream_t s) {
  EmGeom g;
  g.B = B; g.Ch = Ch; g.T = T; g.NP = em_np(T);
  const size_t lds = bx_eeg_conv1_mfma_lds(T);
  if (lds > 150 * 1024) return -1;
  if (hipFuncSetAttribute((const void*)k_eeg_conv1_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_eeg_conv1_mfma, dim3(B * Ch), dim3(256), lds, s, x, w1, (bf16_t*)c1, partials, g, want_stats);
  return 0;
}

// ---- backward, weight gradient of the temporal convolution: one workgroup per (sample, electrode) row.
// dC1[f][t] (gradient at the conv1 output, after the BatchNorm1 / depthwise backward algebra of k_eeg_conv1_bwd) is rebuilt
// into LDS as bf16, four filters per pass ([4][TXB]: 16 KB, so that with the two x copies six workgroups fit a CU and all
// rows are resident together); D[f][k] = sum_t dC1[f][t] x[t + k - 31] runs as 16x16x32 MFMAs with time as the reduction
// dimension (A = dC1 rows 0..3, B = shifted x windows, 4 tap tiles of 16); the four waves split the time steps and are
// summed in wave order through LDS.  Partials [row][8][64] are reduced by the caller (k_sum_partials).
__global__ __launch_bounds__(256) void k_eeg_conv1_wgrad_mfma(const bf16_t* __restrict__ c1, const float* __restrict__ dd, const float* __restrict__ x,
    const float* __restrict__ dw, const float* __restrict__ mean1, const float* __restrict__ inv1, const float* __restrict__ coef,
    float* __restrict__ w1part, EmGeom g, int FD, int coef_stride) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_em[];
  const int TXB = ((g.T + 31) / 32) * 32 + 8;
  bf16_t* xs = lds_em;                          // [2][NP]
  bf16_t* sdc = lds_em + (size_t)2 * g.NP;      // [4][TXB]
  // Workgroups are dealt to the 8 XCDs round-robin by id; the 19 electrode rows of a sample all read that sample's gradient
  // rows dd[b] (16 x T fp32), so consecutive rows must land on ONE XCD's L2 instead of eight (PMC: 116 MB fetched per launch
  // for 57 MB of operands).  XCD k = id % 8 takes the contiguous row range [start_k, start_k + count_k).
  const int nrows = (int)gridDim.x, xq = nrows >> 3, xr = nrows & 7;
  const int xk = (int)blockIdx.x & 7, xj = (int)blockIdx.x >> 3;
  const int row = xk * xq + (xk < xr ? xk : xr) + xj, b = row / g.Ch, ch = row % g.Ch;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gq = lane >> 4, li = lane & 15;
  const int Tn = g.T, T4 = Tn & ~3;
  for (int i = threadIdx.x; i < 4 * (TXB - T4); i += 256)          // the last K-step reads up to TXB - 8: keep the tails zero
    sdc[(i / (TXB - T4)) * TXB + T4 + i % (TXB - T4)] = 0;
  em_build_copies(x + (size_t)row * Tn, xs, Tn, g.NP);             // ends with a barrier
  f32x4 acc[2][4];
#pragma unroll
  for (int fg = 0; fg < 2; ++fg)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[fg][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = (Tn + 31) / 32;
#pragma unroll
  for (int fg = 0; fg < 2; ++fg) {
    // rebuild dC1 of filters 4 fg .. 4 fg + 3:  dbn = w0 * dd[2f] + w1 * dd[2f+1];  dC1 = a * (dbn - k1 - xhat * k2)
    float w0[4], w1d[4], mu[4], is[4], ca[4], k1[4], k2[4];
#pragma unroll
    for (int f4 = 0; f4 < 4; ++f4) {
      const int f = fg * 4 + f4;
      w0[f4] = dw[(f * 2) * g.Ch + ch]; w1d[f4] = dw[(f * 2 + 1) * g.Ch + ch];
      mu[f4] = mean1[f]; is[f4] = inv1[f]; ca[f4] = coef[f]; k1[f4] = coef[coef_stride + f]; k2[f4] = coef[2 * coef_stride + f];
    }
    if ((Tn & 3) == 0) {
      const int nq = Tn / 4;
      for (int q0 = threadIdx.x; q0 < nq; q0 += 512) {
        float4 a0[2][4], a1[2][4];
        float cv[2][4][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = q0 + u * 256;
          if (q < nq) {
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
              const int f = fg * 4 + f4;
              const float* d0 = dd + ((size_t)b * FD + f * 2) * Tn;
              a0[u][f4] = reinterpret_cast<const float4*>(d0)[q];
              a1[u][f4] = reinterpret_cast<const float4*>(d0 + Tn)[q];
              ld4(c1, (((size_t)b * 8 + f) * g.Ch + ch) * Tn + 4 * (size_t)q, cv[u][f4]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int q = q0 + u * 256;
          if (q < nq) {
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
              const float e0[4] = {a0[u][f4].x, a0[u][f4].y, a0[u][f4].z, a0[u][f4].w};
              const float e1[4] = {a1[u][f4].x, a1[u][f4].y, a1[u][f4].z, a1[u][f4].w};
              float o[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float dbn = fmaf(w1d[f4], e1[j], w0[f4] * e0[j]);
                const float xh = (cv[u][f4][j] - mu[f4]) * is[f4];
                o[j] = ca[f4] * (dbn - k1[f4] - xh * k2[f4]);
              }
              *reinterpret_cast<uint2*>(sdc + f4 * TXB + 4 * q) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4) {
        const int f = fg * 4 + f4;
        const float* d0 = dd + ((size_t)b * FD + f * 2) * Tn;
        const size_t cb = (((size_t)b * 8 + f) * g.Ch + ch) * Tn;
        for (int t = threadIdx.x; t < Tn; t += 256) {
          const float dbn = fmaf(w1d[f4], d0[Tn + t], w0[f4] * d0[t]);
          const float xh = (bf2f(c1[cb + t]) - mu[f4]) * is[f4];
          sdc[f4 * TXB + t] = f2bf(ca[f4] * (dbn - k1[f4] - xh * k2[f4]));
        }
      }
    }
    __syncthreads();
    for (int ks = wave; ks < nk; ks += 4) {
      const int t0 = 32 * ks;
      bf16x8 afr = {0, 0, 0, 0, 0, 0, 0, 0};
      if (li < 4) afr = *reinterpret_cast<const bf16x8*>(sdc + li * TXB + t0 + 8 * gq);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        // B[t][k]: lane (col k = 16 n + li, time t0 + 8 gq + j) -> x[t + k - 31] = xpad[t + k + 1]
        const bf16x8 bfr = em_window(xs, g.NP, t0 + 8 * gq + 16 * n + li + (EM_OFF - EM_PADL));
        acc[fg][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[fg][n], 0, 0, 0);
      }
    }
    __syncthreads();                            // sdc is rewritten by the next pass / reused below
  }
  // D: lane (col k = 16 n + li, rows 4 gq + r): the real rows (filters 4 fg + r) sit in lanes gq == 0
  float* red = reinterpret_cast<float*>(sdc);           // [4 waves][2 passes][4 tiles][16 lanes][4]  = 8 KB <= 4 * TXB * 2
  if (gq == 0) {
#pragma unroll
    for (int fg = 0; fg < 2; ++fg)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        *reinterpret_cast<float4*>(red + ((((wave * 2 + fg) * 4 + n) * 16) + li) * 4) =
            make_float4(acc[fg][n][0], acc[fg][n][1], acc[fg][n][2], acc[fg][n][3]);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 512; o += 256) {        // 8 filters x 64 taps
    const int f = o >> 6, k = o & 63, fg = f >> 2, r = f & 3, n = k >> 4, l2 = k & 15;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) s += red[((((wv * 2 + fg) * 4 + n) * 16) + l2) * 4 + r];
    w1part[(size_t)row * 512 + o] = s;
  }
}

size_t bx_eeg_conv1_wgrad_mfma_lds(int T) {
  const size_t txb = ((T + 31) / 32) * 32 + 8;
  size_t b = (2 * (size_t)em_np(T) + 4 * txb) * sizeof(bf16_t);
  const size_t need_red = 2 * (size_t)em_np(T) * sizeof(bf16_t) + 8192;
  return b < need_red ? need_red : b;
}
int bx_eeg_conv1_wgrad_mfma_launch(const void* c1, const float* dd, const float* x, const float* dw, const float* mean1, const float* inv1,
                                   const float* coef, float* w1part, int B, int Ch, int T, int FD, int coef_stride, hipStream_t s) {
  EmGeom g;
  g.B = B; g.Ch = Ch; g.T = T; g.NP = em_np(T);
  const size_t lds = bx_eeg_conv1_wgrad_mfma_lds(T);
  if (lds > 150 * 1024) return -1;
  if (hipFuncSetAttribute((const void*)k_eeg_conv1_wgrad_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_eeg_conv1_wgrad_mfma, dim3(B * Ch), dim3(256), lds, s, (const bf16_t*)c1, dd, x, dw, mean1, inv1, coef, w1part, g, FD, coef_stride);
  return 0;
}


// ---- collapsed front end (eeg_collapse.hip), backward: C[fd][ch][k] = sum_{b,t} g[b][fd][t] x~[b][ch][t + k - 31] ---------------------
// One workgroup per (electrode, split): it walks the samples b = split, split + nsplit, ... of its electrode.  Per sample the row's
// two shifted bf16 copies are rebuilt in LDS and D[fd][k] += A[fd][t] B[t][k] runs as 16x16x32 MFMAs with time as the reduction
// dimension: A = g rows (ALL 16 rows are real here; the per-filter kernel above fills 4 of 16), B = shifted x windows, 4 tap tiles.
// g arrives as bf16 [B][16][T] (written by the BatchNorm2 backward apply): a lane's A operand of a K-step is ONE 16-byte load, so a
// wave's share of a WHOLE sample (NJ K-steps) fits 4 NJ registers and the NEXT sample's share -- and the next row's samples -- are
// requested before the current sample is multiplied (with fp32 g each K-step group was an exposed round trip: 39 -> see DESIGN).
// The four waves split the K-steps and keep their accumulators across samples; one fixed-order sum through LDS at the end.
// Partials [ch][split][16][64]; the electrode-0 workgroups also leave sum_t g.
template <int NJ>
__global__ __launch_bounds__(256) void k_eegc_corr(const bf16_t* __restrict__ g16, const float* __restrict__ x, float* __restrict__ cpart,
                                                   float* __restrict__ gpart, EmGeom g, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_ec[];
  bf16_t* xs = lds_ec;                                  // [2][NP]
  // Workgroups are dealt to the 8 XCDs round-robin by id: the electrode workgroups of a split read the same samples' gradient
  // rows and should share ONE XCD's L2.  With nsplit % 8 == 0: XCD k = id % 8 takes the splits k, k + 8, ...
  int ch, split;
  if (nsplit % 8 == 0) {
    const int xk = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3, per = nsplit >> 3;
    split = xk + 8 * (idx % per); ch = idx / per;
  } else {
    ch = (int)blockIdx.x / nsplit; split = (int)blockIdx.x % nsplit;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gq = lane >> 4, li = lane & 15;
  const int Tn = g.T, nk = (Tn + 31) / 32;
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gs = 0.f;
  const int nj = (nk - wave + 3) / 4;                   // K-steps of this wave: ks = wave + 4 j  (nj <= NJ, checked by the launcher)
  const int e0 = 8 * gq + li + (EM_OFF - EM_PADL);
  // B[t][k]: lane (col k = 16 n + li, time 32 ks + 8 gq + j) -> xpad[e], e = 32 ks + 16 n + e0: the copy (parity of e) and the aligned
  // start are per-lane constants, so a window is wbase + a uniform offset + an immediate
  const uint32_t* wbase = reinterpret_cast<const uint32_t*>(xs + (size_t)(e0 & 1) * g.NP + (e0 & ~1));
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  auto request = [&](int b, u32x4_ (&dst)[NJ]) {
    const __amdgpu_buffer_rsrc_t gres = bx_rsrc(g16 + (size_t)b * 16 * Tn, (uint32_t)(16 * Tn) * 2u);        // wave-uniform: the sample's 16 rows
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int t0 = 32 * (wave + 4 * j) + 8 * gq;
      // T % 8 == 0: a group of 8 is inside the row or past it; K-steps past this wave's share read zeros as well
      const uint32_t off = (j < nj && t0 < Tn) ? (uint32_t)(li * Tn + t0) * 2u : 0x80000000u;
      dst[j] = __builtin_amdgcn_raw_buffer_load_b128(gres, off, 0, 0);
    }
  };
  auto multiply = [&](const u32x4_ (&src)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) break;                               // wave-uniform
      const u32x4_ a = src[j];
      if (ch == 0) {                                    // workgroup-uniform: sum_t g
#pragma unroll
        for (int q = 0; q < 4; ++q) gs += __uint_as_float(a[q] << 16) + __uint_as_float(a[q] & 0xffff0000u);
      }
      const uint32_t* pw = wbase + 16 * (wave + 4 * j);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const u32x4_ d = {pw[8 * n], pw[8 * n + 1], pw[8 * n + 2], pw[8 * n + 3]};
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, d), acc[n], 0, 0, 0);
      }
    }
  };
  EmRow row;
  u32x4_ A0[NJ], A1[NJ];
  if (split < g.B) {
    em_row_request(x + ((size_t)split * g.Ch + ch) * Tn, Tn, g.NP, row);
    request(split, A0);
  }
  auto sample = [&](int b, const u32x4_ (&cur)[NJ], u32x4_ (&nxt)[NJ]) {
    __syncthreads();                                    // the previous sample's window reads are done
    em_row_store(xs, g.NP, row);                        // requested while the previous sample was multiplied
    __syncthreads();
    if (b + nsplit < g.B) {
      em_row_request(x + ((size_t)(b + nsplit) * g.Ch + ch) * Tn, Tn, g.NP, row);
      request(b + nsplit, nxt);
    }
    multiply(cur);
  };
  for (int b = split; b < g.B; b += 2 * nsplit) {
    sample(b, A0, A1);
    if (b + nsplit < g.B) sample(b + nsplit, A1, A0);
  }
  // D: lane (col k = 16 n + li, rows fd = 4 gq + r).  Sum the four waves in wave order.
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds_ec);        // [4 waves][16 fd][64 k] = 16 KB
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * gq + r) * 64 + 16 * n + li] = acc[n][r];
  __syncthreads();
  for (int o = threadIdx.x; o < 1024; o += 256)
    cpart[((size_t)ch * nsplit + split) * 1024 + o] = (red[o] + red[1024 + o]) + (red[2048 + o] + red[3072 + o]);
  if (ch == 0) {                                        // sum_t g of this split's samples: lanes of row li (4 k-groups) x 4 waves
    gs += __shfl_xor(gs, 16, 64); gs += __shfl_xor(gs, 32, 64);
    __syncthreads();
    if (lane < 16) red[wave * 16 + lane] = gs;
    __syncthreads();
    if (threadIdx.x < 16) gpart[split * 16 + threadIdx.x] = (red[threadIdx.x] + red[16 + threadIdx.x]) + (red[32 + threadIdx.x] + red[48 + threadIdx.x]);
  }
}
int bx_eegc_corr_max_T() { return 3072; }               // 24 K-steps per wave: two sample buffers of 96 registers
int bx_eegc_corr_launch(const void* g16, const float* x, float* cpart, float* gpart, int B, int Ch, int T, int nsplit, hipStream_t s) {
  EmGeom g;
  g.B = B; g.Ch = Ch; g.T = T; g.NP = em_np(T);
  if (g.NP / 8 > 512 || T > bx_eegc_corr_max_T() || T % 8) return -3;
  size_t lds = (size_t)2 * g.NP * sizeof(bf16_t);
  if (lds < 16384) lds = 16384;
  if (lds > 150 * 1024) return -1;
  const int njmax = ((T + 31) / 32 + 3) / 4;
  if (njmax <= 16) {
    if (hipFuncSetAttribute((const void*)k_eegc_corr<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_eegc_corr<16>, dim3(Ch * nsplit), dim3(256), lds, s, (const bf16_t*)g16, x, cpart, gpart, g, nsplit);
  } else {
    if (hipFuncSetAttribute((const void*)k_eegc_corr<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_eegc_corr<24>, dim3(Ch * nsplit), dim3(256), lds, s, (const bf16_t*)g16, x, cpart, gpart, g, nsplit);
  }
  return 0;
}
