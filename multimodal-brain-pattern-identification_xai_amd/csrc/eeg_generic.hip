// EEGNet feature extractor for ARBITRARY geometry (round 3): the plain, correct kernel set behind the register-tiled one of
// eeg.hip, which covers the reference's defaults only (F1 = 8, D = 2, F2 = 16, kernLength <= 64, Chans <= 64).  The reference class
// is fully parametric (root/src/models/models.py:239-262: nb_classes, Chans, Samples, kernLength, F1, D, F2), so any other
// configuration used to raise; it now runs here -- fp32 storage, one thread per output element, deterministic fixed-order sums,
// no tiling tricks -- at a fraction of the tuned path's speed (the branch is 2 % of the model's FLOPs).
//   conv1 (1 x K1, 'same', no bias) -> BN1 -> depthwise (Chans x 1, D per filter) -> BN2 -> ELU -> avgpool (1 x P1) -> dropout
//   -> separable conv (1 x K2, 'same', F1*D -> F2, no bias) -> BN3 -> ELU -> avgpool (1 x P2) -> dropout -> flatten
// 'same' padding of an even kernel K: left (K-1)/2, right K/2 (models.py:246,257 via nn.Conv2d(padding='same')).
// Forward and backward (reference: autograd through models.py:271-285), training and evaluation mode, Dropout / Dropout2d masks
// from the same counter hash as eeg.hip (bx_dropout_scale), optionally a second rate for the second dropout
// (EEGNetAttentionDeep.dropout2, models.py:152-164).
#include "bx_common.h"

struct GGeom {
  int B, Ch, T, F1, D, FD, F2, K1, K2, P1, P2, T1, T2, pl1, pl2;
  size_t off_c1, off_d, off_p1, off_s, off_stats, saved_total;
  size_t off_part, off_du3, off_dp1, off_du2, off_dc1, off_coef, off_w1p, ws_total;
  int rows1, nchunk1, maxc;
};
static int g_geom(const bxEegDesc* d, GGeom* g) {
  g->B = d->B; g->Ch = d->Chans; g->T = d->T; g->F1 = d->F1; g->D = d->D; g->FD = d->F1 * d->D; g->F2 = d->F2;
  g->K1 = d->K1; g->K2 = d->K2; g->P1 = d->P1; g->P2 = d->P2;
  if (d->B <= 0 || d->Chans <= 0 || d->T <= 0) return -1;
  if (d->F1 < 1 || d->D < 1 || d->F2 < 1 || g->FD > 1024 || d->F2 > 1024 || d->F1 > 1024) return -2;
  if (d->K1 < 1 || d->K2 < 1 || d->K1 > 4096 || d->K2 > 4096) return -3;
  if (d->P1 < 1 || d->P2 < 1 || d->Chans > 65535) return -4;
  g->T1 = d->T / d->P1; g->T2 = g->T1 / d->P2;
  if (g->T2 < 1) return -5;
  if ((long long)d->B * d->F1 * d->Chans * d->T >= (1ll << 31) || (long long)d->B * g->FD * d->T >= (1ll << 31)) return -6;    // 32-bit element indices
  g->pl1 = (d->K1 - 1) / 2; g->pl2 = (d->K2 - 1) / 2;
  g->maxc = g->F1 > g->FD ? g->F1 : g->FD;
  if (g->F2 > g->maxc) g->maxc = g->F2;
  size_t o = 0;
  g->off_c1 = o; o += bx_align_up((size_t)g->B * g->F1 * g->Ch * g->T * 4, 256);
  g->off_d = o;  o += bx_align_up((size_t)g->B * g->FD * g->T * 4, 256);
  g->off_p1 = o; o += bx_align_up((size_t)g->B * g->FD * g->T1 * 4, 256);
  g->off_s = o;  o += bx_align_up((size_t)g->B * g->F2 * g->T1 * 4, 256);
  g->off_stats = o; o += bx_align_up((size_t)4 * (g->F1 + g->FD + g->F2) * 4, 256);
  g->saved_total = o;
  // workspace
  g->rows1 = g->B * g->Ch * ((g->T + 255) / 256);                    // partial rows of the conv1 statistics (the largest producer)
  size_t rows = (size_t)g->rows1;
  if (rows < (size_t)g->B * g->maxc) rows = (size_t)g->B * g->maxc;
  g->nchunk1 = (g->B * g->Ch + 3) / 4;                               // conv1 weight-gradient partials: four rows per workgroup
  o = 0;
  g->off_part = o; o += bx_align_up(rows * 2 * g->maxc * 4, 256);
  g->off_du3 = o; o += bx_align_up((size_t)g->B * g->F2 * g->T1 * 4, 256);
  g->off_dp1 = o; o += bx_align_up((size_t)g->B * g->FD * g->T1 * 4, 256);
  g->off_du2 = o; o += bx_align_up((size_t)g->B * g->FD * g->T * 4, 256);
  g->off_dc1 = o; o += bx_align_up((size_t)g->B * g->F1 * g->Ch * g->T * 4, 256);
  g->off_coef = o; o += bx_align_up((size_t)3 * 3 * g->maxc * 4, 256);
  g->off_w1p = o; o += bx_align_up((size_t)g->nchunk1 * g->F1 * g->K1 * 4, 256);
  g->ws_total = o;
  return 0;
}
size_t bx_eegg_saved_bytes(const bxEegDesc* d) { GGeom g; return (d && g_geom(d, &g) == 0) ? g.saved_total : 0; }
size_t bx_eegg_workspace(const bxEegDesc* d) { GGeom g; return (d && g_geom(d, &g) == 0) ? g.ws_total : 0; }

struct GStats { float *mean1, *inv1, *sc1, *sh1, *mean2, *inv2, *sc2, *sh2, *mean3, *inv3, *sc3, *sh3; };
static GStats g_stats(const GGeom& g, void* saved) {
  float* p = (float*)((char*)saved + g.off_stats);
  GStats s;
  s.mean1 = p; s.inv1 = p + g.F1; s.sc1 = p + 2 * g.F1; s.sh1 = p + 3 * g.F1; p += 4 * g.F1;
  s.mean2 = p; s.inv2 = p + g.FD; s.sc2 = p + 2 * g.FD; s.sh2 = p + 3 * g.FD; p += 4 * g.FD;
  s.mean3 = p; s.inv3 = p + g.F2; s.sc3 = p + 2 * g.F2; s.sh3 = p + 3 * g.F2;
  return s;
}
__device__ __forceinline__ float g_elu(float z) { return z > 0.f ? z : expm1f(z); }

// workgroup sum of (s, q) in a fixed order: 16-lane rows by DPP inside wave_sum, then the four waves through LDS.  Threads 0 gets it.
__device__ __forceinline__ void g_block_sum2(float& s, float& q, float (*red)[2]) {
  s = wave_sum(s); q = wave_sum(q);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[wave][0] = s; red[wave][1] = q; }
  __syncthreads();
  s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  q = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

// ---- forward --------------------------------------------------------------------------------------------------------------
// G1: c1[b,f,ch,t] = sum_k w1[f,k] x[b,ch,t+k-pl1];  grid (ceil(T/256), B*Ch); BatchNorm1 partial row per workgroup: [row][2][F1]
__global__ __launch_bounds__(256) void gk_conv1(const float* __restrict__ x, const float* __restrict__ w1, float* __restrict__ c1,
                                                float* __restrict__ partials, GGeom g, int want_stats) {
  __shared__ float red[4][2];
  const int row = blockIdx.y, b = row / g.Ch, ch = row - b * g.Ch;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < g.T;
  const float* xr = x + (size_t)row * g.T;
  const int prow = blockIdx.y * gridDim.x + blockIdx.x;
  for (int f = 0; f < g.F1; ++f) {
    float acc = 0.f;
    if (live) {
      const float* w = w1 + (size_t)f * g.K1;
      for (int k = 0; k < g.K1; ++k) {
        const int tt = t + k - g.pl1;
        if (tt >= 0 && tt < g.T) acc = fmaf(w[k], xr[tt], acc);
      }
      c1[(((size_t)b * g.F1 + f) * g.Ch + ch) * g.T + t] = acc;
    }
    if (want_stats) {
      float s = live ? acc : 0.f, q = live ? acc * acc : 0.f;
      g_block_sum2(s, q, red);
      if (threadIdx.x == 0) { partials[((size_t)prow * 2 + 0) * g.F1 + f] = s; partials[((size_t)prow * 2 + 1) * g.F1 + f] = q; }
    }
  }
}
// G3: d[b,fd,t] = sum_ch wd[fd,ch] * (sc1[f] c1[b,f,ch,t] + sh1[f]),  f = fd / D;  grid (ceil(T/256), B); partial row [row][2][FD]
__global__ __launch_bounds__(256) void gk_dw(const float* __restrict__ c1, const float* __restrict__ wd, const float* __restrict__ sc1,
                                             const float* __restrict__ sh1, float* __restrict__ dmap, float* __restrict__ partials, GGeom g, int want_stats) {
  __shared__ float red[4][2];
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < g.T;
  const int prow = blockIdx.y * gridDim.x + blockIdx.x;
  for (int fd = 0; fd < g.FD; ++fd) {
    const int f = fd / g.D;
    float v = 0.f;
    if (live) {
      float acc = 0.f, wsum = 0.f;
      const float* cb = c1 + (((size_t)b * g.F1 + f) * g.Ch) * g.T + t;
      for (int ch = 0; ch < g.Ch; ++ch) { const float w = wd[(size_t)fd * g.Ch + ch]; acc = fmaf(w, cb[(size_t)ch * g.T], acc); wsum += w; }
      v = sc1[f] * acc + sh1[f] * wsum;
      dmap[((size_t)b * g.FD + fd) * g.T + t] = v;
    }
    if (want_stats) {
      float s = live ? v : 0.f, q = live ? v * v : 0.f;
      g_block_sum2(s, q, red);
      if (threadIdx.x == 0) { partials[((size_t)prow * 2 + 0) * g.FD + fd] = s; partials[((size_t)prow * 2 + 1) * g.FD + fd] = q; }
    }
  }
}
// G5 / G8: BN apply + ELU + average pool (1 x P, floor) + dropout.  in [B,C,Tin] -> out [B,C,Tout]   (eeg.hip's k_eeg_bn_elu_pool)
__global__ void gk_bn_elu_pool(const float* __restrict__ in, const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ out,
                               int B, int C, int Tin, int Tout, int P, const uint64_t* __restrict__ seed, float dropout_p, uint32_t salt) {
  const long long n = (long long)B * C * Tout;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long bc = i / Tout;
    const int to = (int)(i - bc * Tout), c = (int)(bc % C);
    const float a = sc[c], cc = sh[c];
    const float* src = in + bc * Tin + (size_t)to * P;
    float s = 0.f;
    for (int j = 0; j < P; ++j) s += g_elu(src[j] * a + cc);
    s /= (float)P;
    if (dropout_p > 0.f) s *= bx_dropout_scale(sd, salt, (salt >> 31) ? (uint64_t)bc : (uint64_t)i, dropout_p, inv_keep);
    out[i] = s;
  }
}
// G6: s[b,f2,t] = sum_fd sum_k ws[f2,fd,k] p1[b,fd,t+k-pl2];  grid (ceil(T1/256), B); partial row [row][2][F2]
__global__ __launch_bounds__(256) void gk_sep(const float* __restrict__ p1, const float* __restrict__ ws, float* __restrict__ smap,
                                              float* __restrict__ partials, GGeom g, int want_stats) {
  __shared__ float red[4][2];
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < g.T1;
  const int prow = blockIdx.y * gridDim.x + blockIdx.x;
  for (int f2 = 0; f2 < g.F2; ++f2) {
    float acc = 0.f;
    if (live) {
      for (int fd = 0; fd < g.FD; ++fd) {
        const float* w = ws + ((size_t)f2 * g.FD + fd) * g.K2;
        const float* pr = p1 + ((size_t)b * g.FD + fd) * g.T1;
        for (int k = 0; k < g.K2; ++k) {
          const int tt = t + k - g.pl2;
          if (tt >= 0 && tt < g.T1) acc = fmaf(w[k], pr[tt], acc);
        }
      }
      smap[((size_t)b * g.F2 + f2) * g.T1 + t] = acc;
    }
    if (want_stats) {
      float s = live ? acc : 0.f, q = live ? acc * acc : 0.f;
      g_block_sum2(s, q, red);
      if (threadIdx.x == 0) { partials[((size_t)prow * 2 + 0) * g.F2 + f2] = s; partials[((size_t)prow * 2 + 1) * g.F2 + f2] = q; }
    }
  }
}

int bx_eegg_forward(const bxEegDesc* d, const bxEegParams* p, const float* x, const uint64_t* seed, float* feat, void* saved, void* workspace,
                    size_t workspace_bytes, hipStream_t s) {
  GGeom g;
  const int ge = g_geom(d, &g);
  BX_REQUIRE(ge == 0, "bx_eeg_features_fwd: geometry outside even the general kernels' range (code %d)", ge);
  if (!workspace || workspace_bytes < g.ws_total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_features_fwd: workspace %zu < %zu", workspace_bytes, g.ws_total);
  const GStats st = g_stats(g, saved);
  float* part = (float*)((char*)workspace + g.off_part);
  float* c1 = (float*)((char*)saved + g.off_c1);
  float* dmap = (float*)((char*)saved + g.off_d);
  float* p1 = (float*)((char*)saved + g.off_p1);
  float* smap = (float*)((char*)saved + g.off_s);
  const int tr = d->training;
  const float pd1 = tr ? d->dropout_p : 0.f, pd2 = tr ? (d->dropout_p2 >= 0.f ? d->dropout_p2 : d->dropout_p) : 0.f;
  BX_REQUIRE((pd1 == 0.f && pd2 == 0.f) || seed, "bx_eeg_features_fwd: dropout needs a device seed");
  // k_bn_finalize (bx_common.h): training -> batch statistics from the partial rows + running-stat update; evaluation -> running statistics
  dim3 g1((unsigned)((g.T + 255) / 256), (unsigned)(g.B * g.Ch));
  hipLaunchKernelGGL(gk_conv1, g1, dim3(256), 0, s, x, p->conv1_w, c1, part, g, tr);
  hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.F1)), dim3(1024), 0, s, part, (int)(g1.x * g1.y), (double)g.B * g.Ch * g.T, g.F1, tr, p->bn1_w,
                     p->bn1_b, p->bn1_rm, p->bn1_rv, p->bn1_nbt, d->momentum, d->eps, st.sc1, st.sh1, st.mean1, st.inv1);
  dim3 g3((unsigned)((g.T + 255) / 256), (unsigned)g.B);
  hipLaunchKernelGGL(gk_dw, g3, dim3(256), 0, s, c1, p->dw_w, st.sc1, st.sh1, dmap, part, g, tr);
  hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.FD)), dim3(1024), 0, s, part, (int)(g3.x * g3.y), (double)g.B * g.T, g.FD, tr, p->bn2_w,
                     p->bn2_b, p->bn2_rm, p->bn2_rv, p->bn2_nbt, d->momentum, d->eps, st.sc2, st.sh2, st.mean2, st.inv2);
  {
    const long long n = (long long)g.B * g.FD * g.T1;
    hipLaunchKernelGGL(gk_bn_elu_pool, dim3(bx_ceil_div(n, 256)), dim3(256), 0, s, dmap, st.sc2, st.sh2, p1, g.B, g.FD, g.T, g.T1, g.P1, seed, pd1, d->salt);
  }
  dim3 g6((unsigned)((g.T1 + 255) / 256), (unsigned)g.B);
  hipLaunchKernelGGL(gk_sep, g6, dim3(256), 0, s, p1, p->sep_w, smap, part, g, tr);
  hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.F2)), dim3(1024), 0, s, part, (int)(g6.x * g6.y), (double)g.B * g.T1, g.F2, tr, p->bn3_w,
                     p->bn3_b, p->bn3_rm, p->bn3_rv, p->bn3_nbt, d->momentum, d->eps, st.sc3, st.sh3, st.mean3, st.inv3);
  {
    const long long n = (long long)g.B * g.F2 * g.T2;
    hipLaunchKernelGGL(gk_bn_elu_pool, dim3(bx_ceil_div(n, 256)), dim3(256), 0, s, smap, st.sc3, st.sh3, feat, g.B, g.F2, g.T1, g.T2, g.P2, seed, pd2, d->salt + 1);
  }
  BX_CHECK_LAUNCH("bx_eeg_features_fwd(general geometry)");
  return BX_OK;
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// gradient through dropout / average pool / ELU up to the BatchNorm output: du[b,c,t] and the BatchNorm-backward partial sums
// s1 = sum du, s2 = sum du * xhat; grid (B, C): one workgroup per (sample, map); partial layout [b][2][C]
__global__ __launch_bounds__(256) void gk_act_bwd(const float* __restrict__ dpool, const float* __restrict__ pre, const float* __restrict__ mean,
    const float* __restrict__ inv, const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ du, float* __restrict__ partials,
    int C, int Tin, int Tout, int P, const uint64_t* __restrict__ seed, float dropout_p, uint32_t salt) {
  __shared__ float red[4][2];
  const int b = blockIdx.x, c = blockIdx.y;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  const size_t bc = (size_t)b * C + c;
  const float a = sc[c], cc = sh[c], mu = mean[c], is = inv[c];
  float s1 = 0.f, s2 = 0.f;
  for (int t = threadIdx.x; t < Tin; t += 256) {
    const int to = t / P;
    float gz = 0.f;
    const float x = pre[bc * Tin + t];
    if (to < Tout) {
      float gp = dpool[bc * Tout + to] / (float)P;
      if (dropout_p > 0.f) gp *= bx_dropout_scale(sd, salt, (salt >> 31) ? (uint64_t)bc : (uint64_t)(bc * Tout + to), dropout_p, inv_keep);
      const float z = x * a + cc;
      gz = gp * (z > 0.f ? 1.f : expf(z));
    }
    du[bc * Tin + t] = gz;
    s1 += gz; s2 += gz * ((x - mu) * is);
  }
  g_block_sum2(s1, s2, red);
  if (threadIdx.x == 0) { partials[((size_t)b * 2 + 0) * C + c] = s1; partials[((size_t)b * 2 + 1) * C + c] = s2; }
}
// sums over the rows in row order -> d gamma, d beta and the coefficients of dx = coef0 * (du - coef1 - xhat * coef2)
__global__ void gk_bn_bwd_finalize(const float* __restrict__ partials, int nrows, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ inv, int training, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < nrows; ++r) { s1 += (double)partials[((size_t)r * 2 + 0) * C + c]; s2 += (double)partials[((size_t)r * 2 + 1) * C + c]; }
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  coef[c] = gamma[c] * inv[c];
  coef[C + c] = training ? (float)(s1 / count) : 0.f;
  coef[2 * C + c] = training ? (float)(s2 / count) : 0.f;
}
// in place: du <- d(pre-BN tensor) = coef0 * (du - coef1 - xhat * coef2);  tensor [B,C,R] (R = inner extent per channel)
__global__ void gk_bn_bwd_apply(float* __restrict__ du, const float* __restrict__ pre, const float* __restrict__ mean, const float* __restrict__ inv,
                                const float* __restrict__ coef, int C, long long R, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / R) % C);
    const float xhat = (pre[i] - mean[c]) * inv[c];
    du[i] = coef[c] * (du[i] - coef[C + c] - xhat * coef[2 * C + c]);
  }
}
// dp1[b,fd,t] = sum_f2 sum_k ws[f2,fd,k] ds[b,f2,t-k+pl2]
__global__ __launch_bounds__(256) void gk_sep_bwd_x(const float* __restrict__ ds, const float* __restrict__ ws, float* __restrict__ dp1, GGeom g) {
  const int b = blockIdx.z, fd = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.T1) return;
  float acc = 0.f;
  for (int f2 = 0; f2 < g.F2; ++f2) {
    const float* w = ws + ((size_t)f2 * g.FD + fd) * g.K2;
    const float* dr = ds + ((size_t)b * g.F2 + f2) * g.T1;
    for (int k = 0; k < g.K2; ++k) {
      const int tt = t - k + g.pl2;
      if (tt >= 0 && tt < g.T1) acc = fmaf(w[k], dr[tt], acc);
    }
  }
  dp1[((size_t)b * g.FD + fd) * g.T1 + t] = acc;
}
// dws[f2,fd,k] = sum_{b,t} ds[b,f2,t] p1[b,fd,t+k-pl2]: one workgroup per (f2, fd), thread = tap k (strided), sequential over (b, t)
__global__ __launch_bounds__(256) void gk_sep_bwd_w(const float* __restrict__ ds, const float* __restrict__ p1, float* __restrict__ dws, GGeom g) {
  const int f2 = blockIdx.x, fd = blockIdx.y;
  for (int k = threadIdx.x; k < g.K2; k += 256) {
    float acc = 0.f;
    for (int b = 0; b < g.B; ++b) {
      const float* dr = ds + ((size_t)b * g.F2 + f2) * g.T1;
      const float* pr = p1 + ((size_t)b * g.FD + fd) * g.T1;
      const int lo = g.pl2 - k > 0 ? g.pl2 - k : 0, hi = g.T1 + g.pl2 - k < g.T1 ? g.T1 + g.pl2 - k : g.T1;
      for (int t = lo; t < hi; ++t) acc = fmaf(dr[t], pr[t + k - g.pl2], acc);
    }
    dws[((size_t)f2 * g.FD + fd) * g.K2 + k] = acc;
  }
}
// dwd[fd,ch] = sum_{b,t} dd[b,fd,t] * (sc1[f] c1[b,f,ch,t] + sh1[f]): one workgroup per (fd, ch)
__global__ __launch_bounds__(256) void gk_dw_bwd_w(const float* __restrict__ dd, const float* __restrict__ c1, const float* __restrict__ sc1,
                                                   const float* __restrict__ sh1, float* __restrict__ dwd, GGeom g) {
  __shared__ float red[4][2];
  const int fd = blockIdx.x, ch = blockIdx.y, f = fd / g.D;
  const float a = sc1[f], c = sh1[f];
  float acc = 0.f, dummy = 0.f;
  for (int b = 0; b < g.B; ++b) {
    const float* dr = dd + ((size_t)b * g.FD + fd) * g.T;
    const float* cr = c1 + (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T;
    for (int t = threadIdx.x; t < g.T; t += 256) acc = fmaf(dr[t], a * cr[t] + c, acc);
  }
  g_block_sum2(acc, dummy, red);
  if (threadIdx.x == 0) dwd[(size_t)fd * g.Ch + ch] = acc;
}
// g1[b,f,ch,t] = sum_{j<D} wd[f*D+j, ch] dd[b,f*D+j,t]  (gradient at BatchNorm1's output) -> dc1 buffer, + BatchNorm1 partial sums per row
__global__ __launch_bounds__(256) void gk_dw_bwd_x(const float* __restrict__ dd, const float* __restrict__ wd, const float* __restrict__ c1,
    const float* __restrict__ mean1, const float* __restrict__ inv1, float* __restrict__ g1, float* __restrict__ partials, GGeom g, int want_stats) {
  __shared__ float red[4][2];
  const int row = blockIdx.y, b = row / g.Ch, ch = row - b * g.Ch;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool live = t < g.T;
  const int prow = blockIdx.y * gridDim.x + blockIdx.x;
  for (int f = 0; f < g.F1; ++f) {
    float v = 0.f, xh = 0.f;
    if (live) {
      for (int j = 0; j < g.D; ++j) v = fmaf(wd[(size_t)(f * g.D + j) * g.Ch + ch], dd[((size_t)b * g.FD + f * g.D + j) * g.T + t], v);
      const size_t o = (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T + t;
      g1[o] = v;
      xh = (c1[o] - mean1[f]) * inv1[f];
    }
    if (want_stats) {
      float s1 = live ? v : 0.f, s2 = live ? v * xh : 0.f;
      g_block_sum2(s1, s2, red);
      if (threadIdx.x == 0) { partials[((size_t)prow * 2 + 0) * g.F1 + f] = s1; partials[((size_t)prow * 2 + 1) * g.F1 + f] = s2; }
    }
  }
}
// dw1 partials: workgroup (f, chunk of four (b, ch) rows), thread = tap k (strided): part[chunk][f][k] = sum_{rows, t} dc1[b,f,ch,t] x[b,ch,t+k-pl1]
__global__ __launch_bounds__(256) void gk_conv1_bwd_w(const float* __restrict__ dc1, const float* __restrict__ x, float* __restrict__ part, GGeom g) {
  const int f = blockIdx.x, chunk = blockIdx.y;
  const int nrows = g.B * g.Ch;
  for (int k = threadIdx.x; k < g.K1; k += 256) {
    float acc = 0.f;
    for (int r = chunk * 4; r < chunk * 4 + 4 && r < nrows; ++r) {
      const int b = r / g.Ch, ch = r - b * g.Ch;
      const float* dr = dc1 + (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T;
      const float* xr = x + (size_t)r * g.T;
      const int lo = g.pl1 - k > 0 ? g.pl1 - k : 0, hi = g.T + g.pl1 - k < g.T ? g.T + g.pl1 - k : g.T;
      for (int t = lo; t < hi; ++t) acc = fmaf(dr[t], xr[t + k - g.pl1], acc);
    }
    part[((size_t)chunk * g.F1 + f) * g.K1 + k] = acc;
  }
}
// dx[b,ch,t] = sum_f sum_k w1[f,k] dc1[b,f,ch,t-k+pl1]
__global__ __launch_bounds__(256) void gk_conv1_bwd_x(const float* __restrict__ dc1, const float* __restrict__ w1, float* __restrict__ dx, GGeom g) {
  const int row = blockIdx.y, b = row / g.Ch, ch = row - b * g.Ch;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.T) return;
  float acc = 0.f;
  for (int f = 0; f < g.F1; ++f) {
    const float* dr = dc1 + (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T;
    const float* w = w1 + (size_t)f * g.K1;
    for (int k = 0; k < g.K1; ++k) {
      const int tt = t - k + g.pl1;
      if (tt >= 0 && tt < g.T) acc = fmaf(w[k], dr[tt], acc);
    }
  }
  dx[(size_t)row * g.T + t] = acc;
}

int bx_eegg_backward(const bxEegDesc* d, const bxEegParams* p, const float* x, const float* dfeat, const uint64_t* seed, const void* saved,
                     const bxEegGrads* gr, float* dx, void* workspace, size_t workspace_bytes, hipStream_t s) {
  GGeom g;
  const int ge = g_geom(d, &g);
  BX_REQUIRE(ge == 0, "bx_eeg_features_bwd: geometry outside even the general kernels' range (code %d)", ge);
  if (!workspace || workspace_bytes < g.ws_total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_features_bwd: workspace %zu < %zu", workspace_bytes, g.ws_total);
  const GStats st = g_stats(g, const_cast<void*>(saved));
  const float* c1 = (const float*)((const char*)saved + g.off_c1);
  const float* dmap = (const float*)((const char*)saved + g.off_d);
  const float* p1 = (const float*)((const char*)saved + g.off_p1);
  const float* smap = (const float*)((const char*)saved + g.off_s);
  float* part = (float*)((char*)workspace + g.off_part);
  float* du3 = (float*)((char*)workspace + g.off_du3);
  float* dp1 = (float*)((char*)workspace + g.off_dp1);
  float* du2 = (float*)((char*)workspace + g.off_du2);
  float* dc1 = (float*)((char*)workspace + g.off_dc1);
  float* coef = (float*)((char*)workspace + g.off_coef);
  float* w1p = (float*)((char*)workspace + g.off_w1p);
  float* coef3 = coef, *coef2 = coef + 3 * g.maxc, *coef1 = coef + 6 * g.maxc;
  const int tr = d->training;
  const float pd1 = tr ? d->dropout_p : 0.f, pd2 = tr ? (d->dropout_p2 >= 0.f ? d->dropout_p2 : d->dropout_p) : 0.f;
  const bool want_w = gr && gr->conv1_w;
  // stage 3: pool2 / dropout2 / ELU / BatchNorm3
  hipLaunchKernelGGL(gk_act_bwd, dim3(g.B, g.F2), dim3(256), 0, s, dfeat, smap, st.mean3, st.inv3, st.sc3, st.sh3, du3, part, g.F2, g.T1, g.T2, g.P2, seed, pd2, d->salt + 1);
  hipLaunchKernelGGL(gk_bn_bwd_finalize, dim3(bx_ceil_div(g.F2, 64)), dim3(64), 0, s, part, g.B, g.F2, (double)g.B * g.T1, p->bn3_w, st.inv3, tr,
                     want_w ? gr->bn3_w : (float*)nullptr, want_w ? gr->bn3_b : (float*)nullptr, coef3);
  {
    const long long n = (long long)g.B * g.F2 * g.T1;
    hipLaunchKernelGGL(gk_bn_bwd_apply, dim3(bx_ceil_div(n, 256)), dim3(256), 0, s, du3, smap, st.mean3, st.inv3, coef3, g.F2, (long long)g.T1, n);
  }
  if (want_w) hipLaunchKernelGGL(gk_sep_bwd_w, dim3(g.F2, g.FD), dim3(256), 0, s, du3, p1, gr->sep_w, g);
  hipLaunchKernelGGL(gk_sep_bwd_x, dim3((g.T1 + 255) / 256, g.FD, g.B), dim3(256), 0, s, du3, p->sep_w, dp1, g);
  // stage 2: pool1 / dropout1 / ELU / BatchNorm2
  hipLaunchKernelGGL(gk_act_bwd, dim3(g.B, g.FD), dim3(256), 0, s, dp1, dmap, st.mean2, st.inv2, st.sc2, st.sh2, du2, part, g.FD, g.T, g.T1, g.P1, seed, pd1, d->salt);
  hipLaunchKernelGGL(gk_bn_bwd_finalize, dim3(bx_ceil_div(g.FD, 64)), dim3(64), 0, s, part, g.B, g.FD, (double)g.B * g.T, p->bn2_w, st.inv2, tr,
                     want_w ? gr->bn2_w : (float*)nullptr, want_w ? gr->bn2_b : (float*)nullptr, coef2);
  {
    const long long n = (long long)g.B * g.FD * g.T;
    hipLaunchKernelGGL(gk_bn_bwd_apply, dim3(bx_ceil_div(n, 256)), dim3(256), 0, s, du2, dmap, st.mean2, st.inv2, coef2, g.FD, (long long)g.T, n);
  }
  if (want_w) hipLaunchKernelGGL(gk_dw_bwd_w, dim3(g.FD, g.Ch), dim3(256), 0, s, du2, c1, st.sc1, st.sh1, gr->dw_w, g);
  if (!want_w && !dx) { BX_CHECK_LAUNCH("bx_eeg_features_bwd(general geometry)"); return BX_OK; }
  // stage 1: depthwise mix -> BatchNorm1 -> temporal convolution
  dim3 g1((unsigned)((g.T + 255) / 256), (unsigned)(g.B * g.Ch));
  hipLaunchKernelGGL(gk_dw_bwd_x, g1, dim3(256), 0, s, du2, p->dw_w, c1, st.mean1, st.inv1, dc1, part, g, 1);
  hipLaunchKernelGGL(gk_bn_bwd_finalize, dim3(bx_ceil_div(g.F1, 64)), dim3(64), 0, s, part, (int)(g1.x * g1.y), g.F1, (double)g.B * g.Ch * g.T, p->bn1_w, st.inv1, tr,
                     want_w ? gr->bn1_w : (float*)nullptr, want_w ? gr->bn1_b : (float*)nullptr, coef1);
  {
    const long long n = (long long)g.B * g.F1 * g.Ch * g.T;
    hipLaunchKernelGGL(gk_bn_bwd_apply, dim3(bx_ceil_div(n, 256)), dim3(256), 0, s, dc1, c1, st.mean1, st.inv1, coef1, g.F1, (long long)g.Ch * g.T, n);
  }
  if (want_w) {
    hipLaunchKernelGGL(gk_conv1_bwd_w, dim3(g.F1, g.nchunk1), dim3(256), 0, s, dc1, x, w1p, g);
    BX_SUM_PARTIALS(w1p, gr->conv1_w, g.nchunk1, g.F1 * g.K1, s);
  }
  if (dx) hipLaunchKernelGGL(gk_conv1_bwd_x, g1, dim3(256), 0, s, dc1, p->conv1_w, dx, g);
  BX_CHECK_LAUNCH("bx_eeg_features_bwd(general geometry)");
  return BX_OK;
}
