// EEGNetAttentionDeep: everything after EEGNet's second block, forward and backward (SURVEY 8(a) row C').
// Replaces reference root/src/models/models.py:215-233 (EEGNetAttentionDeep.forward from conv2 on), the Attention
// module it calls (models.py:109-134) and their autograd backward:
//   feat [B,16,T2] -> conv(1x16 'same', 16 -> 32, no bias) -> BN4 -> ELU -> avgpool(1x8) -> dropout -> tokens [B,L,32]
//   -> Q,K,V = Linear(32,32) -> softmax(Q K^T / sqrt(32)) V -> back to [B,32,L] -> flatten -> Linear(32L,Hd)
//   -> Linear(Hd,N) (no activation in between) -> LogSoftmax.
// Everything here is tiny (B x 32 x T2 floats with T2 = Samples/32) and latency-bound, so the work is organised as a
// few one-workgroup-per-sample kernels that keep a sample's tensors in LDS; batch-wide sums (BatchNorm statistics,
// parameter gradients) go through per-workgroup partials and a fixed-order reduce (deterministic).  All fp32.
#include "bx_common.h"

#define DP_F2 16
#define DP_F3 32
#define DP_K 16
#define DP_P 8
#define DP_TC 128            // time steps per convolution workgroup
#define DP_XP (DP_TC + 16)   // padded input row pitch (float4 aligned)
#define DP_ZP (DP_TC + 17)   // gradient row pitch (odd: 32 rows on 32 banks)
#define DP_MAXL 32
#define DP_LP 33
#define DP_MAXHD 256
#define DP_MAXN 16
#define DP_NQKV (3 * DP_F3 * DP_F3 + 3 * DP_F3)

struct DeepGeom {
  int B, T2, L, K, Hd, N, nchunk;
  size_t off_z, off_x3, off_qkv, off_flat, off_h1, off_logp, off_stats, total;                       // saved arena (bytes)
  size_t w_part, w_dv, w_dlog, w_dh1, w_qkvp, w_wpart, w_coef, w_total;                              // workspace (bytes)
};
static int deep_geom(const bxEegDeepDesc* d, DeepGeom* g) {
  if (!d || d->B <= 0 || d->T2 < DP_P) return -1;
  if (d->F2 != DP_F2 || d->F3 != DP_F3 || d->K3 != DP_K || d->P3 != DP_P) return -2;
  g->B = d->B; g->T2 = d->T2; g->L = d->T2 / DP_P; g->K = DP_F3 * g->L; g->Hd = d->Hd; g->N = d->N;
  if (g->L > DP_MAXL) return -3;
  if (d->Hd < 32 || d->Hd > DP_MAXHD || (d->Hd & (d->Hd - 1))) return -4;
  if (d->N < 1 || d->N > DP_MAXN) return -5;
  g->nchunk = (d->T2 + DP_TC - 1) / DP_TC;
  size_t o = 0;
  g->off_z = o;     o += bx_align_up((size_t)g->B * DP_F3 * g->T2 * 4, 256);
  g->off_x3 = o;    o += bx_align_up((size_t)g->B * g->L * DP_F3 * 4, 256);
  g->off_qkv = o;   o += bx_align_up((size_t)g->B * 3 * g->L * DP_F3 * 4, 256);
  g->off_flat = o;  o += bx_align_up((size_t)g->B * g->K * 4, 256);
  g->off_h1 = o;    o += bx_align_up((size_t)g->B * g->Hd * 4, 256);
  g->off_logp = o;  o += bx_align_up((size_t)g->B * g->N * 4, 256);
  g->off_stats = o; o += bx_align_up((size_t)4 * DP_F3 * 4, 256);
  g->total = o;
  const size_t nblk = (size_t)g->B * g->nchunk;
  o = 0;
  g->w_part = o;  o += bx_align_up(nblk * 2 * DP_F3 * 4, 256);
  g->w_dv = o;    o += bx_align_up((size_t)g->B * DP_F3 * g->T2 * 4, 256);
  g->w_dlog = o;  o += bx_align_up((size_t)g->B * g->N * 4, 256);
  g->w_dh1 = o;   o += bx_align_up((size_t)g->B * g->Hd * 4, 256);
  g->w_qkvp = o;  o += bx_align_up((size_t)g->B * DP_NQKV * 4, 256);
  g->w_wpart = o; o += bx_align_up(nblk * DP_F3 * DP_F2 * DP_K * 4, 256);
  g->w_coef = o;  o += bx_align_up((size_t)3 * DP_F3 * 4, 256);
  g->w_total = o;
  return 0;
}
extern "C" size_t bx_eeg_deep_saved_bytes(const bxEegDeepDesc* d) { DeepGeom g; return deep_geom(d, &g) == 0 ? g.total : 0; }
extern "C" size_t bx_eeg_deep_workspace(const bxEegDeepDesc* d) { DeepGeom g; return deep_geom(d, &g) == 0 ? g.w_total : 0; }

// Q/K/V weights -> LDS as [which][d][c] with pitch 33 (all 12 loads of a thread in flight before the first LDS store)
__device__ __forceinline__ void deep_fill_qkv_w(float (*wl)[DP_F3][DP_LP], const float* __restrict__ wq, const float* __restrict__ wk,
                                                const float* __restrict__ wv) {
  float tmp[12];
#pragma unroll
  for (int u = 0; u < 12; ++u) {
    const int i = (int)threadIdx.x + u * 256, which = i >> 10;
    tmp[u] = (which == 0 ? wq : which == 1 ? wk : wv)[i & 1023];
  }
#pragma unroll
  for (int u = 0; u < 12; ++u) {
    const int i = (int)threadIdx.x + u * 256;
    wl[i >> 10][(i >> 5) & 31][i & 31] = tmp[u];
  }
}

// ------------------------------------------------------------------------------------------------
// D1: third temporal convolution.  grid (B, nchunk): a workgroup owns 128 time steps of one sample; the padded input
// rows and the weights (transposed to [i][k][o], pitch 33) sit in LDS; thread = (output map o, 16 consecutive steps).
// 'same' padding for K = 16: 7 left, 8 right.  Also the BatchNorm partial sums of the chunk.
__global__ __launch_bounds__(256) void k_deep_conv(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ z,
                                                   float* __restrict__ partials, int T2, int want_stats) {
  __shared__ __attribute__((aligned(16))) float xp[DP_F2][DP_XP];
  __shared__ float wt[DP_F2 * DP_K][DP_LP];
  __shared__ float red[8][2][DP_F3];
  const int b = blockIdx.x, t0 = blockIdx.y * DP_TC, tid = threadIdx.x;
  {
    float tmp[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) tmp[u] = w[tid + u * 256];                       // w[o][i][k], flat index o*256 + (i*16+k)
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int i = tid + u * 256; wt[i & 255][i >> 8] = tmp[u]; }
  }
  const float* xb = x + (size_t)b * DP_F2 * T2;
  lds_fill<5>(&xp[0][0], DP_F2 * DP_XP, [&](int i) {
    const int r = i / DP_XP, j = i - r * DP_XP, t = t0 + j - 7;
    return (t >= 0 && t < T2) ? xb[(size_t)r * T2 + t] : 0.f;
  });
  __syncthreads();
  const int o = tid & 31, tl = (tid >> 5) * 16;
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
#pragma unroll 1
  for (int i = 0; i < DP_F2; ++i) {
    float win[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(&xp[i][tl + 4 * q]);
      win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < DP_K; ++k) {
      const float wv = wt[i * DP_K + k][o];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] += win[u + k] * wv;
    }
  }
  float s1 = 0.f, s2 = 0.f;
  float* zr = z + ((size_t)b * DP_F3 + o) * T2;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int t = t0 + tl + u;
    if (t < T2) { zr[t] = acc[u]; s1 += acc[u]; s2 += acc[u] * acc[u]; }
  }
  if (!want_stats) return;
  red[tid >> 5][0][o] = s1; red[tid >> 5][1][o] = s2;
  __syncthreads();
  if (tid < 64) {
    const int v = tid >> 5, c = tid & 31;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][v][c];
    partials[(((size_t)b * gridDim.y + blockIdx.y) * 2 + v) * DP_F3 + c] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// D2: one workgroup per sample: BN4 apply -> ELU -> avgpool8 -> dropout -> Q,K,V -> softmax attention -> flatten ->
// dense1 -> dense2 -> log-softmax.  Saves the tokens, Q/K/V, the attention weights, the flattened output and dense1's
// output for the backward pass.
struct DeepHeadArgs {
  const float *z, *stats, *wq, *bq, *wk, *bk, *wv, *bv, *w1, *b1, *w2, *b2;
  float *x3, *qkv, *attn, *flat, *h1, *logp_saved, *logp;
  const uint64_t* seed;
  int T2, L, Hd, N;
  float dropout_p, scale;
  uint32_t salt;
};
__global__ __launch_bounds__(256) void k_deep_head(DeepHeadArgs a) {
  __shared__ float wl[3][DP_F3][DP_LP];
  __shared__ float x3s[DP_MAXL][DP_LP];
  __shared__ float qkvs[3][DP_MAXL][DP_LP];
  __shared__ float As[DP_MAXL][DP_LP];
  __shared__ __attribute__((aligned(16))) float flat_s[DP_F3 * DP_MAXL];
  __shared__ float h1s[DP_MAXHD];
  __shared__ float lg[DP_MAXN];
  const int b = blockIdx.x, tid = threadIdx.x, L = a.L, T2 = a.T2, K = DP_F3 * L;
  deep_fill_qkv_w(wl, a.wq, a.wk, a.wv);
  const float* sc = a.stats + 2 * DP_F3;
  const float* sh = a.stats + 3 * DP_F3;
  const uint64_t sd = (a.dropout_p > 0.f && a.seed) ? a.seed[0] : 0;
  const float inv_keep = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
  for (int idx = tid; idx < K; idx += 256) {                                       // (c, l): pooled, dropped token feature
    const int c = idx / L, l = idx - c * L;
    const float* zr = a.z + ((size_t)b * DP_F3 + c) * T2 + l * DP_P;
    float v[DP_P];
#pragma unroll
    for (int u = 0; u < DP_P; ++u) v[u] = zr[u];
    const float s_ = sc[c], h_ = sh[c];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < DP_P; ++u) { const float y = v[u] * s_ + h_; s += y > 0.f ? y : expm1f(y); }
    s *= 1.f / DP_P;
    if (a.dropout_p > 0.f) s *= bx_dropout_scale(sd, a.salt, (a.salt >> 31) ? (uint64_t)((size_t)b * DP_F3 + c) : (uint64_t)((size_t)b * K + idx), a.dropout_p, inv_keep);   // bit 31: nn.Dropout2d
    x3s[l][c] = s;
    a.x3[((size_t)b * L + l) * DP_F3 + c] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < 3 * K; idx += 256) {                                   // Q, K, V
    const int which = idx / K, r = idx - which * K, l = r >> 5, d = r & 31;
    float acc = (which == 0 ? a.bq : which == 1 ? a.bk : a.bv)[d];
#pragma unroll
    for (int c = 0; c < DP_F3; ++c) acc += x3s[l][c] * wl[which][d][c];
    qkvs[which][l][d] = acc;
    a.qkv[(((size_t)b * 3 + which) * L + l) * DP_F3 + d] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) {                                   // scores
    const int l = idx / L, m = idx - l * L;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DP_F3; ++d) s += qkvs[0][l][d] * qkvs[1][m][d];
    As[l][m] = s * a.scale;
  }
  __syncthreads();
  if (tid < L) {                                                                   // row softmax
    float mx = -INFINITY;
    for (int m = 0; m < L; ++m) mx = fmaxf(mx, As[tid][m]);
    float sum = 0.f;
    for (int m = 0; m < L; ++m) { const float e = expf(As[tid][m] - mx); As[tid][m] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int m = 0; m < L; ++m) As[tid][m] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) { const int l = idx / L, m = idx - l * L; a.attn[(size_t)b * L * L + idx] = As[l][m]; }
  for (int idx = tid; idx < K; idx += 256) {                                       // attended tokens, stored channel-major
    const int l = idx >> 5, d = idx & 31;
    float o = 0.f;
    for (int m = 0; m < L; ++m) o += As[l][m] * qkvs[2][m][d];
    flat_s[d * L + l] = o;
    a.flat[(size_t)b * K + d * L + l] = o;
  }
  __syncthreads();
  {                                                                                // dense1: thread = (row j, segment of K)
    const int nparts = 256 / a.Hd, seg = K / nparts;                               // seg is a multiple of 4 (K = 32 L, nparts <= 8)
    const int j = tid / nparts, part = tid - j * nparts;
    const float* wr = a.w1 + (size_t)j * K + part * seg;
    const float* fs = flat_s + part * seg;
    float acc = 0.f;
    for (int i0 = 0; i0 < seg; i0 += 16) {
      float4 wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = (i0 + 4 * u < seg) ? *reinterpret_cast<const float4*>(wr + i0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + 4 * u < seg) {
          const float4 f = *reinterpret_cast<const float4*>(fs + i0 + 4 * u);
          acc += wv[u].x * f.x; acc += wv[u].y * f.y; acc += wv[u].z * f.z; acc += wv[u].w * f.w;
        }
    }
    for (int off = 1; off < nparts; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (part == 0) { const float h = acc + a.b1[j]; h1s[j] = h; a.h1[(size_t)b * a.Hd + j] = h; }
  }
  __syncthreads();
  {                                                                                // dense2: 16 lanes per class
    const int n = tid >> 4, p = tid & 15;
    float acc = 0.f;
    if (n < a.N)
      for (int j = p; j < a.Hd; j += 16) acc += a.w2[(size_t)n * a.Hd + j] * h1s[j];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (p == 0 && n < a.N) lg[n] = acc + a.b2[n];
  }
  __syncthreads();
  if (tid < a.N) {
    float mx = -INFINITY;
    for (int n = 0; n < a.N; ++n) mx = fmaxf(mx, lg[n]);
    float sum = 0.f;
    for (int n = 0; n < a.N; ++n) sum += expf(lg[n] - mx);
    const float lp = lg[tid] - mx - logf(sum);
    a.logp[(size_t)b * a.N + tid] = lp;
    a.logp_saved[(size_t)b * a.N + tid] = lp;
  }
}

extern "C" int bx_eeg_deep_fwd(const bxEegDeepDesc* d, const bxEegDeepParams* p, const float* feat, const uint64_t* seed, float* logp,
                               float* attn, void* saved, void* workspace, size_t workspace_bytes, bxStream stream) {
  DeepGeom g;
  const int rc = deep_geom(d, &g);
  if (rc) BX_FAIL(BX_EINVAL, "bx_eeg_deep_fwd: unsupported geometry (code %d): needs F2=16, F3=32, K3=16, P3=8, T2/8 <= %d, Hd a power of two in [32,%d], N <= %d",
                  rc, DP_MAXL, DP_MAXHD, DP_MAXN);
  BX_REQUIRE(p && feat && logp && attn && saved, "bx_eeg_deep_fwd: null pointer");
  BX_REQUIRE(p->conv2_w && p->bn4_w && p->bn4_b && p->bn4_rm && p->bn4_rv && p->wq && p->bq && p->wk && p->bk && p->wv && p->bv && p->w1 && p->b1 &&
             p->w2 && p->b2, "bx_eeg_deep_fwd: null parameter");
  BX_REQUIRE(((uintptr_t)p->w1 & 15) == 0, "bx_eeg_deep_fwd: dense1.weight must be 16-byte aligned");
  BX_REQUIRE(!(d->training && d->dropout_p > 0.f) || seed, "bx_eeg_deep_fwd: dropout needs a seed");
  if (!workspace || workspace_bytes < g.w_total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_deep_fwd: workspace %zu < %zu", workspace_bytes, g.w_total);
  hipStream_t s = (hipStream_t)stream;
  char* sv = (char*)saved; char* ws = (char*)workspace;
  float* z = (float*)(sv + g.off_z);
  float* stats = (float*)(sv + g.off_stats);
  float* part = (float*)(ws + g.w_part);
  hipLaunchKernelGGL(k_deep_conv, dim3(g.B, g.nchunk), dim3(256), 0, s, feat, p->conv2_w, z, part, g.T2, d->training);
  hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(DP_F3)), dim3(256), 0, s, part, g.B * g.nchunk, (double)g.B * g.T2, DP_F3, d->training,
                     p->bn4_w, p->bn4_b, p->bn4_rm, p->bn4_rv, p->bn4_nbt, d->momentum, d->eps, stats + 2 * DP_F3, stats + 3 * DP_F3, stats,
                     stats + DP_F3);
  DeepHeadArgs a;
  a.z = z; a.stats = stats; a.wq = p->wq; a.bq = p->bq; a.wk = p->wk; a.bk = p->bk; a.wv = p->wv; a.bv = p->bv;
  a.w1 = p->w1; a.b1 = p->b1; a.w2 = p->w2; a.b2 = p->b2;
  a.x3 = (float*)(sv + g.off_x3); a.qkv = (float*)(sv + g.off_qkv); a.attn = attn; a.flat = (float*)(sv + g.off_flat);
  a.h1 = (float*)(sv + g.off_h1); a.logp_saved = (float*)(sv + g.off_logp); a.logp = logp;
  a.seed = seed; a.T2 = g.T2; a.L = g.L; a.Hd = g.Hd; a.N = g.N;
  a.dropout_p = d->training ? d->dropout_p : 0.f; a.scale = 1.0f / sqrtf((float)DP_F3); a.salt = d->salt;
  hipLaunchKernelGGL(k_deep_head, dim3(g.B), dim3(256), 0, s, a);
  BX_CHECK_LAUNCH("bx_eeg_deep_fwd");
  return BX_OK;
}

// ================================================================================================
// backward
// DB1: one workgroup per sample: log-softmax / dense2 / dense1 backward, attention backward (dV, dA, softmax, dQ, dK),
// token gradient through the three Linear maps, per-sample partials of the Q/K/V parameter gradients, then
// dropout / avgpool / ELU backward up to BN4's output with the BatchNorm-backward partial sums.
struct DeepHeadBwdArgs {
  const float *dlogp, *logp, *h1, *flat, *attn, *qkv, *x3, *z, *stats, *wq, *wk, *wv, *w1, *w2;
  float *dlog, *dh1, *qkvp, *dv, *partials;
  const uint64_t* seed;
  int T2, L, Hd, N, want_w;
  float dropout_p, scale;
  uint32_t salt;
};
__global__ __launch_bounds__(256) void k_deep_head_bwd(DeepHeadBwdArgs a) {
  __shared__ float wl[3][DP_F3][DP_LP];
  __shared__ float x3s[DP_MAXL][DP_LP];
  __shared__ float qkvs[3][DP_MAXL][DP_LP];
  __shared__ float dG[3][DP_MAXL][DP_LP];
  __shared__ float As[DP_MAXL][DP_LP];
  __shared__ float dSs[DP_MAXL][DP_LP];
  __shared__ float dOs[DP_MAXL][DP_LP];           // dO[l][d]; later the token gradient dx3[l][c]
  __shared__ float tmp[1024];
  __shared__ float dh1s[DP_MAXHD];
  __shared__ float dlg[DP_MAXN];
  const int b = blockIdx.x, tid = threadIdx.x, L = a.L, T2 = a.T2, K = DP_F3 * L, Hd = a.Hd, N = a.N;
  deep_fill_qkv_w(wl, a.wq, a.wk, a.wv);
  {                                                                                // saved tensors of the sample -> LDS
    const float* src = a.qkv + (size_t)b * 3 * K;
    for (int i0 = tid; i0 < 3 * K; i0 += 256 * 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; v[u] = i < 3 * K ? src[i] : 0.f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; if (i < 3 * K) { const int wh = i / K, r = i - wh * K; qkvs[wh][r >> 5][r & 31] = v[u]; } }
    }
    for (int i0 = tid; i0 < K; i0 += 256 * 4) {
      float v[4], w_[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        v[u] = i < K ? a.x3[(size_t)b * K + i] : 0.f;
        w_[u] = i < L * L ? a.attn[(size_t)b * L * L + i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < K) x3s[i >> 5][i & 31] = v[u];
        if (i < L * L) { const int l = i / L; As[l][i - l * L] = w_[u]; }
      }
    }
  }
  if (tid < DP_MAXN) {
    float v = 0.f;
    if (tid < N) {
      float ps = 0.f, gs = 0.f;                          // cancellation-free LogSoftmax backward (bx_common.h: bx_lsm_bwd)
      for (int n = 0; n < N; ++n)
        if (n != tid) { ps += expf(a.logp[(size_t)b * N + n]); gs += a.dlogp[(size_t)b * N + n]; }
      v = a.dlogp[(size_t)b * N + tid] * ps - expf(a.logp[(size_t)b * N + tid]) * gs;
      a.dlog[(size_t)b * N + tid] = v;
    }
    dlg[tid] = v;
  }
  __syncthreads();
  if (tid < Hd) {
    float wv[DP_MAXN];
#pragma unroll
    for (int n = 0; n < DP_MAXN; ++n) wv[n] = n < N ? a.w2[(size_t)n * Hd + tid] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < DP_MAXN; ++n) s += wv[n] * dlg[n];
    dh1s[tid] = s;
    a.dh1[(size_t)b * Hd + tid] = s;
  }
  __syncthreads();
  {                                                                                // dflat = W1^T dh1: thread = (4 columns, slice of rows)
    const int ngrp = K >> 2;
    int nparts = 1;
    while (nparts * 2 * ngrp <= 256 && nparts * 2 <= 8 && (Hd / (nparts * 2)) >= 8) nparts *= 2;
    const int grp = tid % ngrp, part = tid / ngrp;
    if (part < nparts) {
      const int rows = Hd / nparts, j0 = part * rows;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int jj = 0; jj < rows; jj += 8) {
        float4 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float4*>(a.w1 + (size_t)(j0 + jj + u) * K + 4 * grp);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float g_ = dh1s[j0 + jj + u];
          acc.x += wv[u].x * g_; acc.y += wv[u].y * g_; acc.z += wv[u].z * g_; acc.w += wv[u].w * g_;
        }
      }
      float* t = tmp + part * K + 4 * grp;
      t[0] = acc.x; t[1] = acc.y; t[2] = acc.z; t[3] = acc.w;
    }
    __syncthreads();
    for (int i = tid; i < K; i += 256) {
      float s = tmp[i];
      for (int q = 1; q < nparts; ++q) s += tmp[q * K + i];
      const int dd = i / L, l = i - dd * L;                                        // flat index = d * L + l
      dOs[l][dd] = s;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < K; idx += 256) {                                       // dV[m][d] = sum_l A[l][m] dO[l][d]
    const int m = idx >> 5, dd = idx & 31;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += As[l][m] * dOs[l][dd];
    dG[2][m][dd] = s;
  }
  for (int idx = tid; idx < L * L; idx += 256) {                                   // dA[l][m] = sum_d dO[l][d] V[m][d]
    const int l = idx / L, m = idx - l * L;
    float s = 0.f;
#pragma unroll
    for (int dd = 0; dd < DP_F3; ++dd) s += dOs[l][dd] * qkvs[2][m][dd];
    dSs[l][m] = s;
  }
  __syncthreads();
  if (tid < L) {                                                                   // softmax backward, then the 1/sqrt(d) scale
    float r = 0.f;
    for (int m = 0; m < L; ++m) r += dSs[tid][m] * As[tid][m];
    for (int m = 0; m < L; ++m) dSs[tid][m] = As[tid][m] * (dSs[tid][m] - r) * a.scale;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * K; idx += 256) {                                   // dQ[l][d] = sum_m dS[l][m] K[m][d];  dK[m][d] = sum_l dS[l][m] Q[l][d]
    const int which = idx / K, r = idx - which * K, row = r >> 5, dd = r & 31;
    float s = 0.f;
    if (which == 0) for (int m = 0; m < L; ++m) s += dSs[row][m] * qkvs[1][m][dd];
    else            for (int l = 0; l < L; ++l) s += dSs[l][row] * qkvs[0][l][dd];
    dG[which][row][dd] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < K; idx += 256) {                                       // token gradient (overwrites dO: last read two barriers ago)
    const int l = idx >> 5, c = idx & 31;
    float s = 0.f;
#pragma unroll
    for (int wh = 0; wh < 3; ++wh)
#pragma unroll
      for (int dd = 0; dd < DP_F3; ++dd) s += dG[wh][l][dd] * wl[wh][dd][c];
    dOs[l][c] = s;
  }
  if (a.want_w) {                                                                  // per-sample partials of dWq/dWk/dWv and their biases
    float* out = a.qkvp + (size_t)b * DP_NQKV;
    for (int idx = tid; idx < 3 * DP_F3 * DP_F3; idx += 256) {
      const int wh = idx >> 10, dd = (idx >> 5) & 31, c = idx & 31;
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += dG[wh][l][dd] * x3s[l][c];
      out[idx] = s;
    }
    if (tid < 3 * DP_F3) {
      const int wh = tid >> 5, dd = tid & 31;
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += dG[wh][l][dd];
      out[3 * DP_F3 * DP_F3 + tid] = s;
    }
  }
  __syncthreads();
  {                                                                                // dropout / avgpool / ELU backward + BN-backward partials
    const int c = tid >> 3, sub = tid & 7;
    const float mu = a.stats[c], is = a.stats[DP_F3 + c], s_ = a.stats[2 * DP_F3 + c], h_ = a.stats[3 * DP_F3 + c];
    const uint64_t sd = (a.dropout_p > 0.f && a.seed) ? a.seed[0] : 0;
    const float inv_keep = a.dropout_p > 0.f ? 1.f / (1.f - a.dropout_p) : 1.f;
    const float* zr = a.z + ((size_t)b * DP_F3 + c) * T2;
    float* dvr = a.dv + ((size_t)b * DP_F3 + c) * T2;
    float s1 = 0.f, s2 = 0.f;
    for (int t0 = sub; t0 < T2; t0 += 32) {
      float zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int t = t0 + 8 * u; zv[u] = t < T2 ? zr[t] : 0.f; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 8 * u;
        if (t >= T2) continue;
        const int l = t >> 3;
        float g_ = 0.f;
        if (l < L) {
          g_ = dOs[l][c] * (1.f / DP_P);
          if (a.dropout_p > 0.f) g_ *= bx_dropout_scale(sd, a.salt, (a.salt >> 31) ? (uint64_t)((size_t)b * DP_F3 + c) : (uint64_t)((size_t)b * K + c * L + l), a.dropout_p, inv_keep);
          const float y = zv[u] * s_ + h_;
          if (y <= 0.f) g_ *= expf(y);
        }
        dvr[t] = g_;
        s1 += g_;
        s2 += g_ * (zv[u] - mu) * is;
      }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (sub == 0) { a.partials[((size_t)b * 2) * DP_F3 + c] = s1; a.partials[((size_t)b * 2 + 1) * DP_F3 + c] = s2; }
  }
}

// dgamma, dbeta and the per-channel coefficients of BN backward: a = gamma * invstd, k1 = s1 / n, k2 = s2 / n (0 in eval mode)
__global__ __launch_bounds__(256) void k_deep_bn_bwd_finalize(const float* __restrict__ partials, int nblk, double count, int training,
                                                              const float* __restrict__ gamma, const float* __restrict__ inv,
                                                              float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  double s[2] = {0.0, 0.0};
  sum_partials_256<2>(partials, nblk, DP_F3, 0, DP_F3, s);
  const int f = threadIdx.x;
  if (f >= DP_F3) return;
  if (dbeta) dbeta[f] = (float)s[0];
  if (dgamma) dgamma[f] = (float)s[1];
  coef[f] = gamma[f] * inv[f];
  coef[DP_F3 + f] = training ? (float)(s[0] / count) : 0.f;
  coef[2 * DP_F3 + f] = training ? (float)(s[1] / count) : 0.f;
}

// DB2: grid (B, nchunk): BN backward applied on the fly while staging the chunk's gradient rows (with halo) in LDS, then
//  (a) dfeat[i][t] = sum_{o,k} dz[o][t+7-k] w[o][i][k]      thread = (input map i, 8 consecutive steps)
//  (b) partial dw[o][i][k] = sum_t dz[o][t] x[i][t+k-7]      thread = (output map o, 2 input maps, 16 taps)
__global__ __launch_bounds__(256) void k_deep_conv_bwd(const float* __restrict__ dv, const float* __restrict__ z, const float* __restrict__ stats,
                                                       const float* __restrict__ coef, const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ dfeat, float* __restrict__ wpart, int T2, int want_w) {
  __shared__ float dzs[DP_F3][DP_ZP];                       // column j <-> time t0 + j - 8
  __shared__ __attribute__((aligned(16))) float xp[DP_F2][DP_XP];   // column j <-> time t0 + j - 7
  __shared__ float wt[DP_F3 * DP_K][DP_F2 + 1];             // [o][k][i], pitch 17
  const int b = blockIdx.x, t0 = blockIdx.y * DP_TC, tid = threadIdx.x;
  // blockIdx.z selects the role when both are wanted (two half-length workgroups instead of one): 0 = (a), 1 = (b)
  if (gridDim.z == 2) { if (blockIdx.z == 0) want_w = 0; else dfeat = nullptr; }
  if (dfeat) {
    float tmp[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) tmp[u] = w[tid + u * 256];
#pragma unroll
    for (int u = 0; u < 32; ++u) { const int i = tid + u * 256; wt[(i >> 8) * DP_K + (i & 15)][(i >> 4) & 15] = tmp[u]; }
  }
  {
    const float* dvb = dv + (size_t)b * DP_F3 * T2;
    const float* zb = z + (size_t)b * DP_F3 * T2;
    const int n = DP_F3 * (DP_TC + 16);
    for (int i0 = tid; i0 < n; i0 += 256 * 6) {
      float g_[6], zv[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + u * 256, o = i / (DP_TC + 16), t = t0 + (i - o * (DP_TC + 16)) - 8;
        const bool ok = i < n && t >= 0 && t < T2;
        g_[u] = ok ? dvb[(size_t)o * T2 + t] : 0.f;
        zv[u] = ok ? zb[(size_t)o * T2 + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + u * 256;
        if (i < n) {
          const int o = i / (DP_TC + 16), j = i - o * (DP_TC + 16), t = t0 + j - 8;
          const bool ok = t >= 0 && t < T2;
          dzs[o][j] = ok ? coef[o] * (g_[u] - coef[DP_F3 + o] - (zv[u] - stats[o]) * stats[DP_F3 + o] * coef[2 * DP_F3 + o]) : 0.f;
        }
      }
    }
  }
  const float* xb = x + (size_t)b * DP_F2 * T2;
  if (want_w)
    lds_fill<5>(&xp[0][0], DP_F2 * DP_XP, [&](int i) {
      const int r = i / DP_XP, j = i - r * DP_XP, t = t0 + j - 7;
      return (t >= 0 && t < T2) ? xb[(size_t)r * T2 + t] : 0.f;
    });
  __syncthreads();
  if (dfeat) {
    const int i = tid & 15, tl = (tid >> 4) * 8;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
#pragma unroll 2
    for (int o = 0; o < DP_F3; ++o) {
      float win[23];                                                               // dz at time t+7-k  <->  column tl + u + 15 - k
#pragma unroll
      for (int q = 0; q < 23; ++q) win[q] = dzs[o][tl + q];
#pragma unroll
      for (int k = 0; k < DP_K; ++k) {
        const float wv = wt[o * DP_K + k][i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += win[u + 15 - k] * wv;
      }
    }
    float* dr = dfeat + ((size_t)b * DP_F2 + i) * T2;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int t = t0 + tl + u; if (t < T2) dr[t] = acc[u]; }
  }
  if (want_w) {
    const int o = tid & 31, i0 = (tid >> 5) * 2;
    float acc[2][DP_K];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int k = 0; k < DP_K; ++k) acc[q][k] = 0.f;
#pragma unroll 1
    for (int tt = 0; tt < DP_TC; tt += 16) {
      float g_[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) g_[u] = dzs[o][tt + u + 8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float win[32];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float4 v = *reinterpret_cast<const float4*>(&xp[i0 + q][tt + 4 * r]);
          win[4 * r] = v.x; win[4 * r + 1] = v.y; win[4 * r + 2] = v.z; win[4 * r + 3] = v.w;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int k = 0; k < DP_K; ++k) acc[q][k] += g_[u] * win[u + k];
      }
    }
    float* out = wpart + ((size_t)b * gridDim.y + blockIdx.y) * (DP_F3 * DP_F2 * DP_K) + ((size_t)o * DP_F2 + i0) * DP_K;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float4*>(out + q * DP_K + 4 * r) = make_float4(acc[q][4 * r], acc[q][4 * r + 1], acc[q][4 * r + 2], acc[q][4 * r + 3]);
  }
}

// Q/K/V parameter gradients: fixed-order sum of the per-sample partials, scattered to the six tensors
struct DeepQkvOut { float* p[6]; };
__global__ void k_deep_qkv_reduce(const float* __restrict__ part, int B, DeepQkvOut out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= DP_NQKV) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * DP_NQKV + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < B; ++b) s += part[(size_t)b * DP_NQKV + i];
  if (i < 3 * DP_F3 * DP_F3) { float* p = out.p[i >> 10]; if (p) p[i & 1023] = s; }
  else { const int r = i - 3 * DP_F3 * DP_F3; float* p = out.p[3 + (r >> 5)]; if (p) p[r & 31] = s; }
}

// dW[m][n] = sum_r A[r][m] Bm[r][n],  db[m] = sum_r A[r][m]   (dense layers: r = sample)
__global__ __launch_bounds__(256) void k_deep_outer(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ dW,
                                                    float* __restrict__ db, int R, int M, int Nn) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * Nn + M) return;
  const bool bias = idx >= M * Nn;
  const int m = bias ? idx - M * Nn : idx / Nn, n = bias ? 0 : idx - m * Nn;
  float s = 0.f;
  int r = 0;
  for (; r + 8 <= R; r += 8) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { av[u] = A[(size_t)(r + u) * M + m]; bv[u] = bias ? 1.f : Bm[(size_t)(r + u) * Nn + n]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += av[u] * bv[u];
  }
  for (; r < R; ++r) s += A[(size_t)r * M + m] * (bias ? 1.f : Bm[(size_t)r * Nn + n]);
  if (bias) { if (db) db[m] = s; } else if (dW) dW[idx] = s;
}

extern "C" int bx_eeg_deep_bwd(const bxEegDeepDesc* d, const bxEegDeepParams* p, const float* feat, const float* dlogp, const float* attn,
                               const uint64_t* seed, const void* saved, const bxEegDeepGrads* gr, float* dfeat, void* workspace,
                               size_t workspace_bytes, bxStream stream) {
  DeepGeom g;
  const int rc = deep_geom(d, &g);
  if (rc) BX_FAIL(BX_EINVAL, "bx_eeg_deep_bwd: unsupported geometry (code %d)", rc);
  BX_REQUIRE(p && feat && dlogp && attn && saved, "bx_eeg_deep_bwd: null pointer");
  BX_REQUIRE(((uintptr_t)p->w1 & 15) == 0, "bx_eeg_deep_bwd: dense1.weight must be 16-byte aligned");
  BX_REQUIRE(!(d->training && d->dropout_p > 0.f) || seed, "bx_eeg_deep_bwd: dropout needs a seed");
  if (!workspace || workspace_bytes < g.w_total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_deep_bwd: workspace %zu < %zu", workspace_bytes, g.w_total);
  const int want_w = gr != nullptr;
  hipStream_t s = (hipStream_t)stream;
  const char* sv = (const char*)saved; char* ws = (char*)workspace;
  const float* stats = (const float*)(sv + g.off_stats);
  float* part = (float*)(ws + g.w_part);
  float* coef = (float*)(ws + g.w_coef);
  DeepHeadBwdArgs a;
  a.dlogp = dlogp; a.logp = (const float*)(sv + g.off_logp); a.h1 = (const float*)(sv + g.off_h1); a.flat = (const float*)(sv + g.off_flat);
  a.attn = attn; a.qkv = (const float*)(sv + g.off_qkv); a.x3 = (const float*)(sv + g.off_x3); a.z = (const float*)(sv + g.off_z);
  a.stats = stats; a.wq = p->wq; a.wk = p->wk; a.wv = p->wv; a.w1 = p->w1; a.w2 = p->w2;
  a.dlog = (float*)(ws + g.w_dlog); a.dh1 = (float*)(ws + g.w_dh1); a.qkvp = (float*)(ws + g.w_qkvp); a.dv = (float*)(ws + g.w_dv);
  a.partials = part; a.seed = seed; a.T2 = g.T2; a.L = g.L; a.Hd = g.Hd; a.N = g.N; a.want_w = want_w;
  a.dropout_p = d->training ? d->dropout_p : 0.f; a.scale = 1.0f / sqrtf((float)DP_F3); a.salt = d->salt;
  hipLaunchKernelGGL(k_deep_head_bwd, dim3(g.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_deep_bn_bwd_finalize, dim3(1), dim3(256), 0, s, part, g.B, (double)g.B * g.T2, d->training, p->bn4_w, stats + DP_F3, coef,
                     want_w ? gr->bn4_w : nullptr, want_w ? gr->bn4_b : nullptr);
  float* wpart = (float*)(ws + g.w_wpart);
  if (dfeat || want_w)
    hipLaunchKernelGGL(k_deep_conv_bwd, dim3(g.B, g.nchunk, (dfeat && want_w) ? 2 : 1), dim3(256), 0, s, a.dv, a.z, stats, coef, feat, p->conv2_w, dfeat, wpart, g.T2, want_w);
  if (want_w) {
    if (gr->conv2_w) BX_SUM_PARTIALS(wpart, gr->conv2_w, g.B * g.nchunk, DP_F3 * DP_F2 * DP_K, s);
    DeepQkvOut qo;
    qo.p[0] = gr->wq; qo.p[1] = gr->wk; qo.p[2] = gr->wv; qo.p[3] = gr->bq; qo.p[4] = gr->bk; qo.p[5] = gr->bv;
    hipLaunchKernelGGL(k_deep_qkv_reduce, dim3(bx_ceil_div(DP_NQKV, 256)), dim3(256), 0, s, a.qkvp, g.B, qo);
    hipLaunchKernelGGL(k_deep_outer, dim3(bx_ceil_div((long long)g.Hd * g.K + g.Hd, 256)), dim3(256), 0, s, a.dh1, a.flat, gr->w1, gr->b1, g.B, g.Hd, g.K);
    hipLaunchKernelGGL(k_deep_outer, dim3(bx_ceil_div((long long)g.N * g.Hd + g.N, 256)), dim3(256), 0, s, a.dlog, a.h1, gr->w2, gr->b2, g.B, g.N, g.Hd);
  }
  BX_CHECK_LAUNCH("bx_eeg_deep_bwd");
  return BX_OK;
}

// ================================================================================================
// Stand-alone Attention module (reference models.py:109-134): tokens x [B,L,32] -> (output [B,L,32], weights [B,L,L]).
// Same arithmetic as the attention stage of the fused head above, one workgroup per sample; input_dim = attention_dim = 32,
// L <= 32.  The backward also accepts a gradient w.r.t. the returned weights (the module returns them).
struct AttnFwdArgs { const float *x, *wq, *bq, *wk, *bk, *wv, *bv; float *out, *attn, *qkv; int L; float scale; };
__global__ __launch_bounds__(256) void k_attn_fwd(AttnFwdArgs a) {
  __shared__ float wl[3][DP_F3][DP_LP];
  __shared__ float xs[DP_MAXL][DP_LP];
  __shared__ float qkvs[3][DP_MAXL][DP_LP];
  __shared__ float As[DP_MAXL][DP_LP];
  const int b = blockIdx.x, tid = threadIdx.x, L = a.L, K = DP_F3 * L;
  deep_fill_qkv_w(wl, a.wq, a.wk, a.wv);
  for (int i0 = tid; i0 < K; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; v[u] = a.x[(size_t)b * K + (i < K ? i : 0)]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; if (i < K) xs[i >> 5][i & 31] = v[u]; }
  }
  __syncthreads();
  for (int idx = tid; idx < 3 * K; idx += 256) {
    const int which = idx / K, r = idx - which * K, l = r >> 5, d = r & 31;
    float acc = (which == 0 ? a.bq : which == 1 ? a.bk : a.bv)[d];
#pragma unroll
    for (int c = 0; c < DP_F3; ++c) acc += xs[l][c] * wl[which][d][c];
    qkvs[which][l][d] = acc;
    a.qkv[(((size_t)b * 3 + which) * L + l) * DP_F3 + d] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) {
    const int l = idx / L, m = idx - l * L;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DP_F3; ++d) s += qkvs[0][l][d] * qkvs[1][m][d];
    As[l][m] = s * a.scale;
  }
  __syncthreads();
  if (tid < L) {
    float mx = -INFINITY;
    for (int m = 0; m < L; ++m) mx = fmaxf(mx, As[tid][m]);
    float sum = 0.f;
    for (int m = 0; m < L; ++m) { const float e = expf(As[tid][m] - mx); As[tid][m] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int m = 0; m < L; ++m) As[tid][m] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < L * L; idx += 256) { const int l = idx / L, m = idx - l * L; a.attn[(size_t)b * L * L + idx] = As[l][m]; }
  for (int idx = tid; idx < K; idx += 256) {
    const int l = idx >> 5, d = idx & 31;
    float o = 0.f;
    for (int m = 0; m < L; ++m) o += As[l][m] * qkvs[2][m][d];
    a.out[(size_t)b * K + idx] = o;
  }
}

struct AttnBwdArgs { const float *dout, *dattn, *x, *attn, *qkv, *wq, *wk, *wv; float *dx, *qkvp; int L, want_w; float scale; };
__global__ __launch_bounds__(256) void k_attn_bwd(AttnBwdArgs a) {
  __shared__ float wl[3][DP_F3][DP_LP];
  __shared__ float xs[DP_MAXL][DP_LP];
  __shared__ float qkvs[3][DP_MAXL][DP_LP];
  __shared__ float dG[3][DP_MAXL][DP_LP];
  __shared__ float As[DP_MAXL][DP_LP];
  __shared__ float dSs[DP_MAXL][DP_LP];
  __shared__ float dOs[DP_MAXL][DP_LP];
  const int b = blockIdx.x, tid = threadIdx.x, L = a.L, K = DP_F3 * L;
  deep_fill_qkv_w(wl, a.wq, a.wk, a.wv);
  for (int i0 = tid; i0 < 3 * K; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; v[u] = a.qkv[(size_t)b * 3 * K + (i < 3 * K ? i : 0)]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; if (i < 3 * K) { const int wh = i / K, r = i - wh * K; qkvs[wh][r >> 5][r & 31] = v[u]; } }
  }
  for (int i0 = tid; i0 < K; i0 += 256 * 4) {
    float v[4], g_[4], w_[4], da[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256, ic = i < K ? i : 0, il = i < L * L ? i : 0;
      v[u] = a.x[(size_t)b * K + ic];
      g_[u] = a.dout[(size_t)b * K + ic];
      w_[u] = a.attn[(size_t)b * L * L + il];
      da[u] = a.dattn ? a.dattn[(size_t)b * L * L + il] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      if (i < K) { xs[i >> 5][i & 31] = v[u]; dOs[i >> 5][i & 31] = g_[u]; }
      if (i < L * L) { const int l = i / L; As[l][i - l * L] = w_[u]; dSs[l][i - l * L] = da[u]; }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < K; idx += 256) {                                       // dV[m][d] = sum_l A[l][m] dO[l][d]
    const int m = idx >> 5, dd = idx & 31;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += As[l][m] * dOs[l][dd];
    dG[2][m][dd] = s;
  }
  for (int idx = tid; idx < L * L; idx += 256) {                                   // dA[l][m] = sum_d dO[l][d] V[m][d]  (+ the weights' own gradient)
    const int l = idx / L, m = idx - l * L;
    float s = dSs[l][m];
#pragma unroll
    for (int dd = 0; dd < DP_F3; ++dd) s += dOs[l][dd] * qkvs[2][m][dd];
    dSs[l][m] = s;
  }
  __syncthreads();
  if (tid < L) {
    float r = 0.f;
    for (int m = 0; m < L; ++m) r += dSs[tid][m] * As[tid][m];
    for (int m = 0; m < L; ++m) dSs[tid][m] = As[tid][m] * (dSs[tid][m] - r) * a.scale;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * K; idx += 256) {
    const int which = idx / K, r = idx - which * K, row = r >> 5, dd = r & 31;
    float s = 0.f;
    if (which == 0) for (int m = 0; m < L; ++m) s += dSs[row][m] * qkvs[1][m][dd];
    else            for (int l = 0; l < L; ++l) s += dSs[l][row] * qkvs[0][l][dd];
    dG[which][row][dd] = s;
  }
  __syncthreads();
  if (a.dx)
    for (int idx = tid; idx < K; idx += 256) {
      const int l = idx >> 5, c = idx & 31;
      float s = 0.f;
#pragma unroll
      for (int wh = 0; wh < 3; ++wh)
#pragma unroll
        for (int dd = 0; dd < DP_F3; ++dd) s += dG[wh][l][dd] * wl[wh][dd][c];
      a.dx[(size_t)b * K + idx] = s;
    }
  if (a.want_w) {
    float* out = a.qkvp + (size_t)b * DP_NQKV;
    for (int idx = tid; idx < 3 * DP_F3 * DP_F3; idx += 256) {
      const int wh = idx >> 10, dd = (idx >> 5) & 31, c = idx & 31;
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += dG[wh][l][dd] * xs[l][c];
      out[idx] = s;
    }
    if (tid < 3 * DP_F3) {
      const int wh = tid >> 5, dd = tid & 31;
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += dG[wh][l][dd];
      out[3 * DP_F3 * DP_F3 + tid] = s;
    }
  }
}

extern "C" size_t bx_attention_workspace(int B) { return B > 0 ? (size_t)B * DP_NQKV * sizeof(float) : 0; }
extern "C" int bx_attention_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                                const float* bv, float* out, float* attn, float* qkv_saved, int B, int L, int D, bxStream stream) {
  BX_REQUIRE(x && wq && bq && wk && bk && wv && bv && out && attn && qkv_saved, "bx_attention_fwd: null pointer");
  BX_REQUIRE(B > 0 && L > 0 && L <= DP_MAXL && D == DP_F3, "bx_attention_fwd: needs input_dim = attention_dim = %d and L <= %d", DP_F3, DP_MAXL);
  AttnFwdArgs a;
  a.x = x; a.wq = wq; a.bq = bq; a.wk = wk; a.bk = bk; a.wv = wv; a.bv = bv; a.out = out; a.attn = attn; a.qkv = qkv_saved; a.L = L;
  a.scale = 1.0f / sqrtf((float)D);
  hipLaunchKernelGGL(k_attn_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  BX_CHECK_LAUNCH("bx_attention_fwd");
  return BX_OK;
}
extern "C" int bx_attention_bwd(const float* dout, const float* dattn, const float* x, const float* attn, const float* qkv_saved,
                                const float* wq, const float* wk, const float* wv, float* dx, float* dwq, float* dbq, float* dwk, float* dbk,
                                float* dwv, float* dbv, void* workspace, size_t workspace_bytes, int B, int L, int D, bxStream stream) {
  BX_REQUIRE(dout && x && attn && qkv_saved && wq && wk && wv, "bx_attention_bwd: null pointer");
  BX_REQUIRE(B > 0 && L > 0 && L <= DP_MAXL && D == DP_F3, "bx_attention_bwd: needs input_dim = attention_dim = %d and L <= %d", DP_F3, DP_MAXL);
  const int want_w = dwq || dbq || dwk || dbk || dwv || dbv;
  if (want_w && (!workspace || workspace_bytes < bx_attention_workspace(B))) BX_FAIL(BX_EWORKSPACE, "bx_attention_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  AttnBwdArgs a;
  a.dout = dout; a.dattn = dattn; a.x = x; a.attn = attn; a.qkv = qkv_saved; a.wq = wq; a.wk = wk; a.wv = wv; a.dx = dx;
  a.qkvp = (float*)workspace; a.L = L; a.want_w = want_w; a.scale = 1.0f / sqrtf((float)D);
  hipLaunchKernelGGL(k_attn_bwd, dim3(B), dim3(256), 0, s, a);
  if (want_w) {
    DeepQkvOut qo;
    qo.p[0] = dwq; qo.p[1] = dwk; qo.p[2] = dwv; qo.p[3] = dbq; qo.p[4] = dbk; qo.p[5] = dbv;
    hipLaunchKernelGGL(k_deep_qkv_reduce, dim3(bx_ceil_div(DP_NQKV, 256)), dim3(256), 0, s, (const float*)workspace, B, qo);
  }
  BX_CHECK_LAUNCH("bx_attention_bwd");
  return BX_OK;
}
