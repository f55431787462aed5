// Classifier heads, fusion head and KL-divergence loss, forward and backward.  All fp32 except the
// channels-last feature map fed to / produced by the global-average-pool.
//   GAP -> Linear(256,6) -> LogSoftmax   reference root/src/models/models.py:92-94,103-106
//   Linear(F2*T/32, 6) -> LogSoftmax     reference root/src/models/models.py:263-269,286-288
//   cat -> Linear(12,128) -> ReLU -> Linear(128,6) -> LogSoftmax   XAI_Multimodality.py:1095-1105
//   nn.KLDivLoss()                        XAI_Multimodality.py:1989,1599
// Parameter gradients are summed over the batch in a fixed order (deterministic).
#include "bx_common.h"

#define HEAD_MAX_N 32

template <typename T>
__global__ __launch_bounds__(256) void k_gap(const T* __restrict__ feat, float* __restrict__ gap, int HW, int C) {
  const int b = blockIdx.x;
  const float inv = 1.f / (float)HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= HW; p += 8) {              // eight pixel loads in flight, added in pixel order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldf(feat, ((size_t)b * HW + p + u) * C + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < HW; ++p) s += ldf(feat, ((size_t)b * HW + p) * C + c);
    gap[(size_t)b * C + c] = s * inv;
  }
}

// one workgroup (256 threads = 4 waves) per sample
__global__ __launch_bounds__(256) void k_linear_lsm_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ logp, int K, int N) {
  __shared__ float logit[HEAD_MAX_N];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int n = wave; n < N; n += 4) {
    float s = 0.f;
    const float* wr = w + (size_t)n * K;
    const float* xr = x + (size_t)b * K;
    int k = lane;
    for (; k + 192 < K; k += 256) {          // eight loads in flight per trip; the adds keep k order
      const float w0 = wr[k], w1 = wr[k + 64], w2 = wr[k + 128], w3 = wr[k + 192];
      const float x0 = xr[k], x1 = xr[k + 64], x2 = xr[k + 128], x3 = xr[k + 192];
      s = fmaf(w0, x0, s); s = fmaf(w1, x1, s); s = fmaf(w2, x2, s); s = fmaf(w3, x3, s);
    }
    for (; k < K; k += 64) s = fmaf(wr[k], xr[k], s);
    s = wave_sum(s);
    if (lane == 0) logit[n] = s + bias[n];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, logit[n]);
    float se = 0.f;
    for (int n = 0; n < N; ++n) se += expf(logit[n] - m);
    const float lse = m + logf(se);
    for (int n = 0; n < N; ++n) logp[(size_t)b * N + n] = logit[n] - lse;
  }
}

// dlogit = dlogp - softmax * sum(dlogp)
// (loops are fully unrolled over HEAD_MAX_N with guards: a runtime-indexed per-thread array would live in scratch memory)
// evaluated without the (1 - p_n) cancellation (bx_common.h: bx_lsm_bwd), here with prefix / suffix sums so that every index is static:
//   dz_n = dy_n * (P_<n + P_>n) - p_n * (D_<n + D_>n)
__device__ __forceinline__ void lsm_bwd_row(const float* dlogp, const float* logp, int N, float* dlogit) {
  float p[HEAD_MAX_N], pl[HEAD_MAX_N], dl[HEAD_MAX_N];
  float pr = 0.f, dr = 0.f;
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n) {
    p[n] = n < N ? expf(logp[n]) : 0.f;
    pl[n] = pr; dl[n] = dr;                       // sums over j < n
    pr += p[n]; dr += n < N ? dlogp[n] : 0.f;
  }
  pr = 0.f; dr = 0.f;                             // now sums over j > n
#pragma unroll
  for (int n = HEAD_MAX_N - 1; n >= 0; --n) {
    dlogit[n] = n < N ? dlogp[n] * (pl[n] + pr) - p[n] * (dl[n] + dr) : 0.f;
    pr += p[n]; dr += n < N ? dlogp[n] : 0.f;
  }
}

// input gradient: dx[b][k] = sum_n dlogit[b][n] W[n][k]; GAP variant broadcasts dx/HW over the pixels
template <typename T, bool GAP>
__global__ __launch_bounds__(256) void k_linear_lsm_bwd_dx(const float* __restrict__ dlogp, const float* __restrict__ logp,
    const float* __restrict__ w, float* __restrict__ dx, T* __restrict__ dfeat, int K, int N, int HW) {
  const int b = blockIdx.x;
  float dl[HEAD_MAX_N];
  lsm_bwd_row(dlogp + (size_t)b * N, logp + (size_t)b * N, N, dl);
  for (int k = blockIdx.y * 256 + threadIdx.x; k < K; k += 256 * gridDim.y) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) s = fmaf(dl[n], w[(size_t)n * K + k], s);
    if (GAP) {
      s /= (float)HW;
      for (int p = 0; p < HW; ++p) stf(dfeat, ((size_t)b * HW + p) * K + k, s);
    } else {
      dx[(size_t)b * K + k] = s;
    }
  }
}
// parameter gradients: the workgroup first builds dlogit[b][n] in LDS (B*N values); a thread owns one input column k
// of a 16-column slab and every 16th sample (all of its x loads are in flight together -- the batch loop used to be one
// thread's serial chain of HBM round trips), then the 16 sample groups are summed through LDS in group order.
// Thread 0.. of block 0 sums the bias.  LDS: B*N + 16*N*16 floats.
__global__ __launch_bounds__(256) void k_linear_lsm_bwd_w(const float* __restrict__ dlogp, const float* __restrict__ logp,
    const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int B, int K, int N) {
  extern __shared__ float sdl[];   // [B][N] | [16 groups][N][16 columns]
  float* sred = sdl + (size_t)B * N;
  for (int b = threadIdx.x; b < B; b += 256) {
    float dl[HEAD_MAX_N];
    lsm_bwd_row(dlogp + (size_t)b * N, logp + (size_t)b * N, N, dl);
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) sdl[b * N + n] = dl[n];
  }
  __syncthreads();
  const int kk = threadIdx.x & 15, bg = threadIdx.x >> 4;
  const int k = blockIdx.x * 16 + kk;
  if (dw) {
    float acc[HEAD_MAX_N];
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) acc[n] = 0.f;
    if (k < K) {
#pragma unroll 4
      for (int b = bg; b < B; b += 16) {
        const float xv = x[(size_t)b * K + k];
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) acc[n] = fmaf(sdl[b * N + n], xv, acc[n]);
      }
    }
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) sred[(bg * N + n) * 16 + kk] = acc[n];
    __syncthreads();
    for (int i = threadIdx.x; i < N * 16; i += 256) {
      const int n = i >> 4, kc = blockIdx.x * 16 + (i & 15);
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += sred[(g * N + n) * 16 + (i & 15)];
      if (kc < K) dw[(size_t)n * K + kc] = s;
    }
  }
  if (db && blockIdx.x == 0 && (int)threadIdx.x < N) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sdl[b * N + threadIdx.x];
    db[threadIdx.x] = s;
  }
}

extern "C" int bx_linear_lsm_fwd(const float* x, const float* w, const float* b, float* logp, int B, int K, int N, bxStream stream) {
  BX_REQUIRE(x && w && b && logp && B > 0 && K > 0 && N > 0 && N <= HEAD_MAX_N, "bx_linear_lsm_fwd: bad arguments (N<=%d)", HEAD_MAX_N);
  hipLaunchKernelGGL(k_linear_lsm_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, x, w, b, logp, K, N);
  BX_CHECK_LAUNCH("bx_linear_lsm_fwd");
  return BX_OK;
}
extern "C" int bx_linear_lsm_bwd(const float* dlogp, const float* logp, const float* x, const float* w, float* dx,
                                 float* dw, float* db, int B, int K, int N, bxStream stream) {
  BX_REQUIRE(dlogp && logp && x && w && B > 0 && K > 0 && N > 0 && N <= HEAD_MAX_N, "bx_linear_lsm_bwd: bad arguments");
  BX_REQUIRE(((size_t)B * N + 256 * N) * sizeof(float) <= 60 * 1024, "bx_linear_lsm_bwd: batch %d too large for the LDS gradient tile ((B+256)*N <= 15360)", B);
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL((k_linear_lsm_bwd_dx<float, false>), dim3(B, bx_ceil_div(K, 256)), dim3(256), 0, s, dlogp, logp, w, dx, (float*)nullptr, K, N, 1);
    BX_CHECK_LAUNCH("bx_linear_lsm_bwd(dx)");
  }
  if (dw || db) {
    hipLaunchKernelGGL(k_linear_lsm_bwd_w, dim3(bx_ceil_div(K, 16)), dim3(256), ((size_t)B * N + 256 * N) * sizeof(float), s, dlogp, logp, x, dw, db, B, K, N);
    BX_CHECK_LAUNCH("bx_linear_lsm_bwd(w)");
  }
  return BX_OK;
}

extern "C" int bx_gap_fc_lsm_fwd(const void* feat, const float* w, const float* b, float* gap_out, float* logp,
                                 int B, int HW, int C, int N, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(feat && w && b && gap_out && logp && B > 0 && HW > 0 && C > 0 && N > 0 && N <= HEAD_MAX_N, "bx_gap_fc_lsm_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  BX_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((k_gap<T>), dim3(B), dim3(256), 0, s, (const T*)feat, gap_out, HW, C));
  BX_CHECK_LAUNCH("bx_gap_fc_lsm_fwd(gap)");
  hipLaunchKernelGGL(k_linear_lsm_fwd, dim3(B), dim3(256), 0, s, gap_out, w, b, logp, C, N);
  BX_CHECK_LAUNCH("bx_gap_fc_lsm_fwd(fc)");
  return BX_OK;
}
extern "C" int bx_gap_fc_lsm_bwd(const float* dlogp, const float* logp, const float* gap_out, const float* w,
                                 void* dfeat, float* dw, float* db, int B, int HW, int C, int N, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(dlogp && logp && gap_out && w && B > 0 && HW > 0 && C > 0 && N > 0 && N <= HEAD_MAX_N, "bx_gap_fc_lsm_bwd: bad arguments");
  BX_REQUIRE(((size_t)B * N + 256 * N) * sizeof(float) <= 60 * 1024, "bx_gap_fc_lsm_bwd: batch %d too large for the LDS gradient tile ((B+256)*N <= 15360)", B);
  hipStream_t s = (hipStream_t)stream;
  if (dfeat) {
    BX_DISPATCH_DTYPE(dtype, T,
      hipLaunchKernelGGL((k_linear_lsm_bwd_dx<T, true>), dim3(B), dim3(256), 0, s, dlogp, logp, w, (float*)nullptr, (T*)dfeat, C, N, HW));
    BX_CHECK_LAUNCH("bx_gap_fc_lsm_bwd(dfeat)");
  }
  if (dw || db) {
    hipLaunchKernelGGL(k_linear_lsm_bwd_w, dim3(bx_ceil_div(C, 16)), dim3(256), ((size_t)B * N + 256 * N) * sizeof(float), s, dlogp, logp, gap_out, dw, db, B, C, N);
    BX_CHECK_LAUNCH("bx_gap_fc_lsm_bwd(w)");
  }
  return BX_OK;
}

// ------------------------------------------------------------------------------------------------
// fusion head: one workgroup per sample, one thread per hidden unit
__global__ void k_fusion_fwd(const float* __restrict__ e, const float* __restrict__ sp, const float* __restrict__ w1,
                             const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                             float* __restrict__ hidden, float* __restrict__ logp, int N, int Hd) {
  extern __shared__ float sm[];  // z[2N] | h[Hd] | logit[N]
  float* z = sm; float* h = sm + 2 * N; float* logit = h + Hd;
  const int b = blockIdx.x, j = threadIdx.x;
  if (j < N) { z[j] = e[(size_t)b * N + j]; z[N + j] = sp[(size_t)b * N + j]; }
  __syncthreads();
  if (j < Hd) {
    float s = b1[j];
#pragma unroll 4
    for (int i = 0; i < 2 * N; ++i) s = fmaf(w1[(size_t)j * 2 * N + i], z[i], s);
    s = fmaxf(s, 0.f);
    h[j] = s;
    hidden[(size_t)b * Hd + j] = s;
  }
  __syncthreads();
  if (j < N) {
    float s = b2[j];
#pragma unroll 8
    for (int k = 0; k < Hd; ++k) s = fmaf(w2[(size_t)j * Hd + k], h[k], s);
    logit[j] = s;
  }
  __syncthreads();
  if (j == 0) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, logit[n]);
    float se = 0.f;
    for (int n = 0; n < N; ++n) se += expf(logit[n] - m);
    const float lse = m + logf(se);
    for (int n = 0; n < N; ++n) logp[(size_t)b * N + n] = logit[n] - lse;
  }
}
// gradient w.r.t. the two branch outputs
__global__ void k_fusion_bwd_in(const float* __restrict__ dlogp, const float* __restrict__ logp, const float* __restrict__ hidden,
                                const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ de,
                                float* __restrict__ dsp, int N, int Hd) {
  extern __shared__ float sm[];  // dh[Hd]
  const int b = blockIdx.x, j = threadIdx.x;
  float dl[HEAD_MAX_N];
  lsm_bwd_row(dlogp + (size_t)b * N, logp + (size_t)b * N, N, dl);
  if (j < Hd) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) s = fmaf(dl[n], w2[(size_t)n * Hd + j], s);
    sm[j] = hidden[(size_t)b * Hd + j] > 0.f ? s : 0.f;
  }
  __syncthreads();
  if (j < 2 * N) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < Hd; ++k) s = fmaf(sm[k], w1[(size_t)k * 2 * N + j], s);
    if (j < N) { if (de) de[(size_t)b * N + j] = s; }
    else if (dsp) dsp[(size_t)b * N + (j - N)] = s;
  }
}
// parameter gradients: one wave per hidden unit j, lanes over the batch, fixed shuffle-tree sums
__global__ __launch_bounds__(64) void k_fusion_bwd_w(const float* __restrict__ dlogp, const float* __restrict__ logp, const float* __restrict__ hidden,
                               const float* __restrict__ e, const float* __restrict__ sp, const float* __restrict__ w2,
                               float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                               int B, int N, int Hd) {
  const int j = blockIdx.x, lane = threadIdx.x;
  float a1e[HEAD_MAX_N], a1s[HEAD_MAX_N], a2[HEAD_MAX_N], ab2[HEAD_MAX_N], ab1 = 0.f;
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n) a1e[n] = a1s[n] = a2[n] = ab2[n] = 0.f;
  for (int b = lane; b < B; b += 64) {
    float dl[HEAD_MAX_N];
    lsm_bwd_row(dlogp + (size_t)b * N, logp + (size_t)b * N, N, dl);
    const float hj = hidden[(size_t)b * Hd + j];
    float dh = 0.f;
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n)
      if (n < N) { dh = fmaf(dl[n], w2[(size_t)n * Hd + j], dh); a2[n] = fmaf(dl[n], hj, a2[n]); ab2[n] += dl[n]; }
    dh = hj > 0.f ? dh : 0.f;
    ab1 += dh;
#pragma unroll
    for (int i = 0; i < HEAD_MAX_N; ++i)
      if (i < N) { a1e[i] = fmaf(dh, e[(size_t)b * N + i], a1e[i]); a1s[i] = fmaf(dh, sp[(size_t)b * N + i], a1s[i]); }
  }
#pragma unroll
  for (int i = 0; i < HEAD_MAX_N; ++i)
    if (i < N) {
      const float ve = wave_sum(a1e[i]), vs = wave_sum(a1s[i]);
      if (lane == 0 && dw1) { dw1[(size_t)j * 2 * N + i] = ve; dw1[(size_t)j * 2 * N + N + i] = vs; }
    }
  { const float v = wave_sum(ab1); if (lane == 0 && db1) db1[j] = v; }
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n)
    if (n < N) {
      const float v = wave_sum(a2[n]), vb = wave_sum(ab2[n]);
      if (lane == 0 && dw2) dw2[(size_t)n * Hd + j] = v;
      if (lane == 0 && db2 && j == 0) db2[n] = vb;
    }
}

extern "C" int bx_fusion_head_fwd(const float* eeg_logp, const float* spec_logp, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* hidden, float* logp, int B, int N, int Hd, bxStream stream) {
  BX_REQUIRE(eeg_logp && spec_logp && w1 && b1 && w2 && b2 && hidden && logp, "bx_fusion_head_fwd: null pointer");
  BX_REQUIRE(B > 0 && N > 0 && N <= HEAD_MAX_N && Hd >= 2 * N && Hd <= 1024, "bx_fusion_head_fwd: need N<=%d, 2N<=Hd<=1024", HEAD_MAX_N);
  const int threads = (Hd + 63) / 64 * 64;
  hipLaunchKernelGGL(k_fusion_fwd, dim3(B), dim3(threads), (3 * N + Hd) * sizeof(float), (hipStream_t)stream,
                     eeg_logp, spec_logp, w1, b1, w2, b2, hidden, logp, N, Hd);
  BX_CHECK_LAUNCH("bx_fusion_head_fwd");
  return BX_OK;
}
extern "C" int bx_fusion_head_bwd(const float* dlogp, const float* logp, const float* hidden, const float* eeg_logp,
                                  const float* spec_logp, const float* w1, const float* w2, float* d_eeg_logp,
                                  float* d_spec_logp, float* dw1, float* db1, float* dw2, float* db2, int B, int N, int Hd,
                                  bxStream stream) {
  BX_REQUIRE(dlogp && logp && hidden && eeg_logp && spec_logp && w1 && w2, "bx_fusion_head_bwd: null pointer");
  BX_REQUIRE(B > 0 && N > 0 && N <= HEAD_MAX_N && Hd >= 2 * N && Hd <= 1024, "bx_fusion_head_bwd: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const int threads = (Hd + 63) / 64 * 64;
  if (d_eeg_logp || d_spec_logp) {
    hipLaunchKernelGGL(k_fusion_bwd_in, dim3(B), dim3(threads), Hd * sizeof(float), s, dlogp, logp, hidden, w1, w2, d_eeg_logp, d_spec_logp, N, Hd);
    BX_CHECK_LAUNCH("bx_fusion_head_bwd(in)");
  }
  if (dw1 || db1 || dw2 || db2) {
    hipLaunchKernelGGL(k_fusion_bwd_w, dim3(Hd), dim3(64), 0, s, dlogp, logp, hidden, eeg_logp, spec_logp, w2,
                       dw1, db1, dw2, db2, B, N, Hd);
    BX_CHECK_LAUNCH("bx_fusion_head_bwd(w)");
  }
  return BX_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kldiv(const float* __restrict__ logp, const float* __restrict__ target, float* __restrict__ loss,
                                                float* __restrict__ dlogp, int n, float inv_denom, float gscale) {
  __shared__ float part[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float t = target[i];
    acc += t > 0.f ? t * (logf(t) - logp[i]) : 0.f;
    if (dlogp) dlogp[i] = -t * inv_denom * gscale;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && loss) loss[0] = (part[0] + part[1] + part[2] + part[3]) * inv_denom;
}
extern "C" int bx_kldiv_fwd_bwd(const float* logp, const float* target, float* loss, float* dlogp, int B, int N,
                                int reduction, float grad_scale, bxStream stream) {
  BX_REQUIRE(logp && target && B > 0 && N > 0, "bx_kldiv_fwd_bwd: bad arguments");
  BX_REQUIRE(reduction >= 0 && reduction <= 2, "bx_kldiv_fwd_bwd: reduction must be 0 (mean), 1 (batchmean) or 2 (sum)");
  const float denom = reduction == 0 ? (float)B * N : reduction == 1 ? (float)B : 1.f;
  hipLaunchKernelGGL(k_kldiv, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, target, loss, dlogp, B * N, 1.f / denom, grad_scale);
  BX_CHECK_LAUNCH("bx_kldiv_fwd_bwd");
  return BX_OK;
}

// ================================================================================================
// Fused multimodal head: everything between the two feature extractors and the loss in ONE launch forward and TWO
// backward (the separate ops above cost 4 + 6 launches of ~5-12 us for a few kFLOP per sample):
//   gap = mean_hw feat5;  s = LogSoftmax(fc gap + b)           (models.py:103-106)
//   e = LogSoftmax(dense eegfeat + b)                           (models.py:286-288)
//   out = LogSoftmax(fc2 relu(fc1 cat(e, s) + b1) + b2)         (XAI_Multimodality.py:1099-1103)
// One workgroup per sample; parameter gradients are batch sums computed by a second kernel, one thread per entry.
#define MMH_MAXK 4096
#define MMH_MAXC 1024
#define MMH_MAXHD 256

__device__ __forceinline__ void mmh_lsm(const float* logit, int N, float* out) {      // serial, N <= HEAD_MAX_N
  float m = -INFINITY;
  for (int n = 0; n < N; ++n) m = fmaxf(m, logit[n]);
  float se = 0.f;
  for (int n = 0; n < N; ++n) se += expf(logit[n] - m);
  const float lse = m + logf(se);
  for (int n = 0; n < N; ++n) out[n] = logit[n] - lse;
}
// dot of a global row with an LDS vector by one wave (all loads of a lane issued before the adds)
__device__ __forceinline__ float mmh_wave_dot(const float* __restrict__ wrow, const float* vec, int n, int lane) {
  float acc = 0.f;
  for (int i0 = lane; i0 < n; i0 += 64 * 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; w[u] = i < n ? wrow[i] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; if (i < n) acc = fmaf(w[u], vec[i], acc); }
  }
  return wave_sum(acc);
}

// FAST (2N <= 12, N <= 8, K <= 1024, C <= 1024): every weight a thread will need is loaded into registers at kernel start,
// together with the feature loads, so the chain gap -> logits -> hidden -> logits costs one memory round trip instead of six.
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void k_mm_head_fwd(const T* __restrict__ feat, const float* __restrict__ ef, const float* __restrict__ fcw,
    const float* __restrict__ fcb, const float* __restrict__ dw_, const float* __restrict__ db_, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gap_out,
    float* __restrict__ s_logp, float* __restrict__ e_logp, float* __restrict__ hidden, float* __restrict__ logp, int HW, int C, int K, int N, int Hd) {
  __shared__ float gs[MMH_MAXC], es[MMH_MAXK], hs[MMH_MAXHD], lg[2 * HEAD_MAX_N], z[2 * HEAD_MAX_N], lo[HEAD_MAX_N];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float inv_hw = 1.f / (float)HW;
  // FAST: register copies of this thread's weights (rows q = wave + 4r of [dense ; fc], row tid of fc1, rows wave, wave+4 of fc2)
  float wq[FAST ? 3 : 1][FAST ? 16 : 1], bq[FAST ? 3 : 1], w1r[FAST ? 12 : 1], b1r = 0.f, w2r[FAST ? 2 : 1][FAST ? 4 : 1], b2r[FAST ? 2 : 1];
  if (FAST) {
#pragma unroll
    // (clamped indices + selects, never `cond ? p[i] : 0`: each conditional load became an exec-mask branch, ~75 of them in a row)
    for (int r = 0; r < 3; ++r) {
      const int q = wave + 4 * r;
      const bool live = q < 2 * N, eeg = q < N || !live;
      const int n = !live ? 0 : eeg ? q : q - N, len = eeg ? K : C;
      const float* row = eeg ? dw_ + (size_t)n * K : fcw + (size_t)n * C;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = lane + 64 * u;
        const float v = row[i < len ? i : 0];
        wq[r][u] = i < len ? v : 0.f;                            // rows q >= 2N load row 0 and are never used (a `live &&` here
      }                                                           // turned every load into a branch with its own s_waitcnt)
      bq[r] = eeg ? db_[n] : fcb[n];
    }
    const int jt = tid < Hd ? tid : 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) w1r[i] = w1[(size_t)jt * 2 * N + (i < 2 * N ? i : 0)];     // entries i >= 2N are never used
    b1r = b1[jt];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int n = wave + 4 * r, nc = n < N ? n : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = lane + 64 * u; const float v = w2[(size_t)nc * Hd + (j < Hd ? j : 0)]; w2r[r][u] = j < Hd ? v : 0.f; }
      b2r[r] = b2[nc];
    }
  }
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    const T* fp = feat + (size_t)b * HW * C + c;
    for (int p0 = 0; p0 < HW; p0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ldf(fp, (size_t)(p0 + u < HW ? p0 + u : 0) * C);
#pragma unroll
      for (int u = 0; u < 16; ++u) s += p0 + u < HW ? v[u] : 0.f;
    }
    s *= inv_hw;
    gs[c] = s;
    gap_out[(size_t)b * C + c] = s;
  }
  lds_fill<4>(es, K, [&](int i) { return ef[(size_t)b * K + i]; });
  __syncthreads();
  if (FAST) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int q = wave + 4 * r;
      const float* vec = q < N ? es : gs;
      const int len = q < N ? K : C;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int i = lane + 64 * u; if (i < len) acc = fmaf(wq[r][u], vec[i], acc); }
      acc = wave_sum(acc);
      if (lane == 0 && q < 2 * N) lg[q] = acc + bq[r];
    }
  } else {
    for (int q = wave; q < 2 * N; q += 4) {                   // 2N dot products over the 4 waves: eeg rows first, then spectrogram rows
      const bool eeg = q < N;
      const int n = eeg ? q : q - N;
      const float v = eeg ? mmh_wave_dot(dw_ + (size_t)n * K, es, K, lane) : mmh_wave_dot(fcw + (size_t)n * C, gs, C, lane);
      if (lane == 0) lg[q] = v + (eeg ? db_[n] : fcb[n]);
    }
  }
  __syncthreads();
  if (tid < 2) mmh_lsm(lg + tid * N, N, z + tid * N);
  __syncthreads();
  if (tid < 2 * N) (tid < N ? e_logp : s_logp)[(size_t)b * N + (tid < N ? tid : tid - N)] = z[tid];
  if (FAST) {
    if (tid < Hd) {
      float s = b1r;
#pragma unroll
      for (int i = 0; i < 12; ++i) if (i < 2 * N) s = fmaf(w1r[i], z[i], s);
      s = fmaxf(s, 0.f);
      hs[tid] = s;
      hidden[(size_t)b * Hd + tid] = s;
    }
  } else {
    for (int j = tid; j < Hd; j += 256) {
      float wv[2 * HEAD_MAX_N];
#pragma unroll
      for (int i = 0; i < 2 * HEAD_MAX_N; ++i) wv[i] = i < 2 * N ? w1[(size_t)j * 2 * N + i] : 0.f;
      float s = b1[j];
#pragma unroll
      for (int i = 0; i < 2 * HEAD_MAX_N; ++i) if (i < 2 * N) s = fmaf(wv[i], z[i], s);
      s = fmaxf(s, 0.f);
      hs[j] = s;
      hidden[(size_t)b * Hd + j] = s;
    }
  }
  __syncthreads();
  if (FAST) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int n = wave + 4 * r;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = lane + 64 * u; if (j < Hd) acc = fmaf(w2r[r][u], hs[j], acc); }
      acc = wave_sum(acc);
      if (lane == 0 && n < N) lo[n] = acc + b2r[r];
    }
  } else {
    for (int n = wave; n < N; n += 4) {
      const float v = mmh_wave_dot(w2 + (size_t)n * Hd, hs, Hd, lane);
      if (lane == 0) lo[n] = v + b2[n];
    }
  }
  __syncthreads();
  if (tid == 0) mmh_lsm(lo, N, lg);
  __syncthreads();
  if (tid < N) logp[(size_t)b * N + tid] = lg[tid];
}

// backward 1: per sample, gradients w.r.t. the two feature inputs; the per-sample logit / pre-activation gradients go to
// the workspace ([B][N] dl2 | [B][Hd] dpre | [B][N] dle | [B][N] dls) for the parameter-gradient kernel
template <typename T>
__global__ __launch_bounds__(256) void k_mm_head_bwd_in(const float* __restrict__ dlogp, const float* __restrict__ logp, const float* __restrict__ hidden,
    const float* __restrict__ s_logp, const float* __restrict__ e_logp, const float* __restrict__ fcw, const float* __restrict__ dw_,
    const float* __restrict__ w1, const float* __restrict__ w2, T* __restrict__ dfeat, float* __restrict__ def, float* __restrict__ ws,
    int B, int HW, int C, int K, int N, int Hd) {
  __shared__ float w1s[MMH_MAXHD * 2 * HEAD_MAX_N / 4];      // Hd * 2N <= 4096 floats (checked by the launcher)
  __shared__ float dl2[HEAD_MAX_N], dpre[MMH_MAXHD], dz[2 * HEAD_MAX_N], dbr[2 * HEAD_MAX_N];
  const int b = blockIdx.x, tid = threadIdx.x;
  // everything this thread will read from global memory is requested up front (one round trip for the whole kernel)
  constexpr int PN = 8, PK = 4;                               // register prefetch covers N <= 8, K <= 1024, C <= 256, Hd <= 256
  const bool pre = N <= PN && K <= 256 * PK && C <= 256 && Hd <= 256;
  float w2c[PN], hid = 0.f, fcc[PN], dwc[PK][PN];
  if (pre) {
#pragma unroll
    for (int n = 0; n < PN; ++n) {                              // clamped indices + selects: no branch per load
      const int nc = n < N ? n : 0;
      w2c[n] = w2[(size_t)nc * Hd + (tid < Hd ? tid : 0)];     // entries n >= N / threads beyond the row are never used
      fcc[n] = fcw[(size_t)nc * C + (tid < C ? tid : 0)];
#pragma unroll
      for (int u = 0; u < PK; ++u) { const int k = tid + 256 * u; dwc[u][n] = dw_[(size_t)nc * K + (k < K ? k : 0)]; }
    }
    hid = hidden[(size_t)b * Hd + (tid < Hd ? tid : 0)];
  }
  lds_fill<8>(w1s, Hd * 2 * N, [&](int i) { return w1[i]; });
  if (tid == 0) bx_lsm_bwd(dlogp + (size_t)b * N, logp + (size_t)b * N, N, dl2);
  __syncthreads();
  float* ws_dl2 = ws; float* ws_dpre = ws + (size_t)B * N; float* ws_dle = ws_dpre + (size_t)B * Hd; float* ws_dls = ws_dle + (size_t)B * N;
  if (tid < N) ws_dl2[(size_t)b * N + tid] = dl2[tid];
  if (pre) {
    if (tid < Hd) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < PN; ++n) if (n < N) s = fmaf(dl2[n], w2c[n], s);
      s = hid > 0.f ? s : 0.f;
      dpre[tid] = s;
      ws_dpre[(size_t)b * Hd + tid] = s;
    }
  } else {
    for (int j = tid; j < Hd; j += 256) {
      float wv[HEAD_MAX_N];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) wv[n] = n < N ? w2[(size_t)n * Hd + j] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) s = fmaf(dl2[n], wv[n], s);
      s = hidden[(size_t)b * Hd + j] > 0.f ? s : 0.f;
      dpre[j] = s;
      ws_dpre[(size_t)b * Hd + j] = s;
    }
  }
  __syncthreads();
  if (tid < 2 * N) {
    float s = 0.f;
    for (int j = 0; j < Hd; ++j) s = fmaf(dpre[j], w1s[j * 2 * N + tid], s);
    dz[tid] = s;
  }
  __syncthreads();
  if (tid < 2) {                                               // log-softmax backward of the two branch outputs
    const float* lp = (tid == 0 ? e_logp : s_logp) + (size_t)b * N;
    bx_lsm_bwd(dz + tid * N, lp, N, dbr + tid * N);
  }
  __syncthreads();
  if (tid < 2 * N) (tid < N ? ws_dle : ws_dls)[(size_t)b * N + (tid < N ? tid : tid - N)] = dbr[tid];
  const float inv_hw = 1.f / (float)HW;
  if (pre) {
    if (def) {
#pragma unroll
      for (int u = 0; u < PK; ++u) {
        const int k = tid + 256 * u;
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < PN; ++n) if (n < N) s = fmaf(dbr[n], dwc[u][n], s);
        if (k < K) def[(size_t)b * K + k] = s;
      }
    }
    if (dfeat && tid < C) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < PN; ++n) if (n < N) s = fmaf(dbr[N + n], fcc[n], s);
      s *= inv_hw;
      for (int p = 0; p < HW; ++p) stf(dfeat, ((size_t)b * HW + p) * C + tid, s);
    }
    return;
  }
  if (def)
    for (int k = tid; k < K; k += 256) {
      float wv[HEAD_MAX_N];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) wv[n] = n < N ? dw_[(size_t)n * K + k] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) s = fmaf(dbr[n], wv[n], s);
      def[(size_t)b * K + k] = s;
    }
  if (dfeat) {
    for (int c = tid; c < C; c += 256) {
      float wv[HEAD_MAX_N];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) wv[n] = n < N ? fcw[(size_t)n * C + c] : 0.f;
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) if (n < N) s = fmaf(dbr[N + n], wv[n], s);
      s *= inv_hw;
      for (int p = 0; p < HW; ++p) stf(dfeat, ((size_t)b * HW + p) * C + c, s);
    }
  }
}

// backward 2: every parameter-gradient entry is a batch sum  sum_b A[b][m] * V[b][n]  (V = 1 for biases); one thread per entry
struct MmHeadSeg { const float* A; const float* V; const float* V2; float* out; int M, Nn, sa, sv, split; };   // V2: second half of a concatenated V
struct MmHeadJobs { MmHeadSeg seg[8]; int start[9]; };
__global__ __launch_bounds__(256) void k_mm_head_bwd_w(MmHeadJobs jobs, int B) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= jobs.start[8]) return;
  int si = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q) if (idx >= jobs.start[q]) si = q;
  const MmHeadSeg sg = jobs.seg[si];
  if (!sg.out) return;
  const int e = idx - jobs.start[si];
  const bool bias = sg.V == nullptr;
  const int m = bias ? e : e / sg.Nn, n = bias ? 0 : e - m * sg.Nn;
  const float* vp = bias ? nullptr : (sg.V2 && n >= sg.split ? sg.V2 + (n - sg.split) : sg.V + n);
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float av[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { av[u] = sg.A[(size_t)(b + u) * sg.sa + m]; vv[u] = bias ? 1.f : vp[(size_t)(b + u) * sg.sv]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) s = fmaf(av[u], vv[u], s);
  }
  for (; b < B; ++b) s = fmaf(sg.A[(size_t)b * sg.sa + m], bias ? 1.f : vp[(size_t)b * sg.sv], s);
  sg.out[e] = s;
}

extern "C" size_t bx_mm_head_workspace(int B, int N, int Hd) { return (size_t)B * (3 * (size_t)N + Hd) * sizeof(float); }
static int mm_head_check(const char* who, int B, int HW, int C, int K, int N, int Hd) {
  BX_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= MMH_MAXC && K > 0 && K <= MMH_MAXK, "%s: need C <= %d and K <= %d", who, MMH_MAXC, MMH_MAXK);
  BX_REQUIRE(N > 0 && N <= HEAD_MAX_N && Hd > 0 && Hd <= MMH_MAXHD && Hd * 2 * N <= MMH_MAXHD * 2 * HEAD_MAX_N / 4, "%s: need N <= %d, Hd <= %d, Hd*2N <= %d",
             who, HEAD_MAX_N, MMH_MAXHD, MMH_MAXHD * 2 * HEAD_MAX_N / 4);
  return BX_OK;
}
extern "C" int bx_mm_head_fwd(const void* feat, const float* eeg_feat, const float* fc_w, const float* fc_b, const float* dense_w,
                              const float* dense_b, const float* w1, const float* b1, const float* w2, const float* b2, float* gap_out,
                              float* spec_logp, float* eeg_logp, float* hidden, float* logp, int B, int HW, int C, int K, int N, int Hd,
                              int dtype, bxStream stream) {
  BX_REQUIRE(feat && eeg_feat && fc_w && fc_b && dense_w && dense_b && w1 && b1 && w2 && b2 && gap_out && spec_logp && eeg_logp && hidden && logp,
             "bx_mm_head_fwd: null pointer");
  BX_DTYPE_OK(dtype);
  const int rc = mm_head_check("bx_mm_head_fwd", B, HW, C, K, N, Hd);
  if (rc) return rc;
  const bool fast = 2 * N <= 12 && N <= 8 && K <= 1024 && C <= 1024;
  BX_DISPATCH_DTYPE(dtype, T,
    if (fast) hipLaunchKernelGGL((k_mm_head_fwd<T, true>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const T*)feat, eeg_feat, fc_w, fc_b, dense_w,
                                 dense_b, w1, b1, w2, b2, gap_out, spec_logp, eeg_logp, hidden, logp, HW, C, K, N, Hd);
    else hipLaunchKernelGGL((k_mm_head_fwd<T, false>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const T*)feat, eeg_feat, fc_w, fc_b, dense_w,
                            dense_b, w1, b1, w2, b2, gap_out, spec_logp, eeg_logp, hidden, logp, HW, C, K, N, Hd));
  BX_CHECK_LAUNCH("bx_mm_head_fwd");
  return BX_OK;
}
extern "C" int bx_mm_head_bwd(const float* dlogp, const float* logp, const float* hidden, const float* spec_logp, const float* eeg_logp,
                              const float* gap, const float* eeg_feat, const float* fc_w, const float* dense_w, const float* w1, const float* w2,
                              void* dfeat, float* d_eeg_feat, float* d_fc_w, float* d_fc_b, float* d_dense_w, float* d_dense_b, float* dw1,
                              float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, int B, int HW, int C, int K, int N,
                              int Hd, int dtype, bxStream stream) {
  BX_REQUIRE(dlogp && logp && hidden && spec_logp && eeg_logp && gap && eeg_feat && fc_w && dense_w && w1 && w2, "bx_mm_head_bwd: null pointer");
  BX_DTYPE_OK(dtype);
  const int rc = mm_head_check("bx_mm_head_bwd", B, HW, C, K, N, Hd);
  if (rc) return rc;
  if (!workspace || workspace_bytes < bx_mm_head_workspace(B, N, Hd)) BX_FAIL(BX_EWORKSPACE, "bx_mm_head_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  BX_DISPATCH_DTYPE(dtype, T,
    hipLaunchKernelGGL((k_mm_head_bwd_in<T>), dim3(B), dim3(256), 0, s, dlogp, logp, hidden, spec_logp, eeg_logp, fc_w, dense_w, w1, w2, (T*)dfeat,
                       d_eeg_feat, ws, B, HW, C, K, N, Hd));
  BX_CHECK_LAUNCH("bx_mm_head_bwd(inputs)");
  if (d_fc_w || d_fc_b || d_dense_w || d_dense_b || dw1 || db1 || dw2 || db2) {
    const float* dl2 = ws; const float* dpre = ws + (size_t)B * N; const float* dle = dpre + (size_t)B * Hd; const float* dls = dle + (size_t)B * N;
    MmHeadJobs j;
    const MmHeadSeg segs[8] = {
        {dl2, hidden, nullptr, dw2, N, Hd, N, Hd, 0},            {dl2, nullptr, nullptr, db2, N, 1, N, 0, 0},
        {dpre, eeg_logp, spec_logp, dw1, Hd, 2 * N, Hd, N, N},   {dpre, nullptr, nullptr, db1, Hd, 1, Hd, 0, 0},
        {dle, eeg_feat, nullptr, d_dense_w, N, K, N, K, 0},      {dle, nullptr, nullptr, d_dense_b, N, 1, N, 0, 0},
        {dls, gap, nullptr, d_fc_w, N, C, N, C, 0},              {dls, nullptr, nullptr, d_fc_b, N, 1, N, 0, 0}};
    int acc = 0;
    for (int q = 0; q < 8; ++q) { j.seg[q] = segs[q]; j.start[q] = acc; acc += segs[q].M * segs[q].Nn; }
    j.start[8] = acc;
    hipLaunchKernelGGL(k_mm_head_bwd_w, dim3(bx_ceil_div(acc, 256)), dim3(256), 0, s, j, B);
    BX_CHECK_LAUNCH("bx_mm_head_bwd(weights)");
  }
  return BX_OK;
}
