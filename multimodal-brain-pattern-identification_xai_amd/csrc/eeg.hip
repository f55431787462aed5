// EEGNet feature extractor over raw EEG [B,1,Chans,T], forward and backward.
// Replaces reference root/src/models/models.py:271-285 (EEGNet.forward up to Flatten) and its autograd
// backward:  conv(1xK1,'same') -> BN1 -> depthwise(Chans x 1, D per filter) -> BN2 -> ELU -> avgpool(1xP1)
//            -> dropout -> conv(1xK2,'same', F1*D -> F2) -> BN3 -> ELU -> avgpool(1xP2) -> dropout -> flatten
// 'same' padding for an even kernel K: left (K-1)/2, right K/2 (31/32 for K=64, 7/8 for K=16).
//
// The only large tensor is the temporal-conv output c1 [B,F1,Chans,T] (stored in `dtype`); everything
// after the depthwise electrode mix is [B,16,T] or smaller and stays fp32.  All batch reductions use
// per-workgroup partials + a fixed-order finalize (deterministic).
#include "bx_common.h"

#define EEG_TT 256           // time steps per workgroup in the temporal conv
#define EEG_MAXK 64
#define EEG_MAXF 16          // F1*D and F2 upper bound
#define EEG_MAXCH 64

// eeg_mfma.hip
size_t bx_eeg_conv1_mfma_lds(int T);
int bx_eeg_conv1_mfma_launch(const float* x, const float* w1, void* c1, float* partials, int B, int Ch, int T, int want_stats, hipStream_t s);
size_t bx_eeg_conv1_wgrad_mfma_lds(int T);
int bx_eeg_conv1_wgrad_mfma_launch(const void* c1, const float* dd, const float* x, const float* dw, const float* mean1, const float* inv1,
                                   const float* coef, float* w1part, int B, int Ch, int T, int FD, int coef_stride, hipStream_t s);

// general-geometry kernel set (eeg_generic.hip): whatever the register-tiled kernels below do not cover
size_t bx_eegg_saved_bytes(const bxEegDesc* d);
size_t bx_eegg_workspace(const bxEegDesc* d);
int bx_eegg_forward(const bxEegDesc* d, const bxEegParams* p, const float* x, const uint64_t* seed, float* feat, void* saved, void* workspace,
                    size_t workspace_bytes, hipStream_t s);
int bx_eegg_backward(const bxEegDesc* d, const bxEegParams* p, const float* x, const float* dfeat, const uint64_t* seed, const void* saved,
                     const bxEegGrads* gr, float* dx, void* workspace, size_t workspace_bytes, hipStream_t s);

// collapsed front end (eeg_collapse.hip / eeg_mfma.hip)
size_t bx_eegc_stat_floats();
int bx_eegc_forward(const float* x, const float* w1, const float* wd, const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt,
                    float momentum, float eps, float* mean1, float* inv1, float* sc1, float* sh1, double* RS, float* stat_ws, float* u,
                    float* bn2_partials, int* bn2_rows, int B, int Ch, int T, hipStream_t s);
int bx_eegc_corr_launch(const void* g16, const float* x, float* cpart, float* gpart, int B, int Ch, int T, int nsplit, hipStream_t s);
int bx_eegc_corr_max_T();
int bx_eegc_forward_eval(const float* x, const float* w1, const float* wd, const float* sc1, const float* sh1, float* u, int B, int Ch, int T,
                         hipStream_t s, const float* const* x_slot = nullptr);
int bx_eegc_dx_launch(const float* g, const float* w1, const float* wd, const float* sc1, float* dx, int B, int Ch, int T, hipStream_t s);
int bx_eegc_grads(const float* cpart, const float* gpart, int nsplit, const float* w1, const float* wd, const float* mean1, const float* inv1,
                  const float* sc1, const float* sh1, const double* RS, float* ep, float* d_wd, float* d_gamma, float* d_beta, float* d_w1,
                  int B, int Ch, int T, hipStream_t s);
#define EEGC_RS_BYTES ((size_t)(64 * 64 + 64) * sizeof(double))
#define EEGC_MAX_SPLIT 32
// The collapsed front end applies to: training mode, bf16 storage (its backward rounds the MFMA operands to bf16, which the fp32
// parity path must not), the reference's 64-tap kernel, rows long enough for the edge terms, T % 8 == 0, no input gradient.
static bool eeg_collapsed(const bxEegDesc* d) {
  return d->collapse && d->training && d->dtype == BX_BF16 && d->K1 == 64 && d->T >= 96 && d->T % 8 == 0 && d->T <= bx_eegc_corr_max_T()
         && d->F2 == d->F1 * d->D && d->P1 == 4;       // (the bf16 gradient map reuses the du3 | dp1 regions: 2 x B*16*(T/4)*4 bytes)
}

// Evaluation mode (inference, Grad-CAM sweeps, attribution passes) with no parameter gradient wanted: BatchNorm1 is a fixed affine
// map, so the forward is k_eegc_fwd alone (any storage type: it computes in fp32 and never forms the conv1 tensor) and the input
// gradient, if asked for, k_eegc_dx.  The caller (ops.py) sets `collapse` only when no parameter needs a gradient.
static bool eeg_collapsed_eval(const bxEegDesc* d) { return d->collapse && !d->training && d->K1 == 64; }

struct EegGeom {
  int B, Ch, T, F1, D, FD, F2, K1, K2, P1, P2, T1, T2, padl1, padl2;
  size_t off_c1, off_d, off_p1, off_s, off_stats, total;   // saved arena (bytes)
};
static int eeg_geom(const bxEegDesc* d, EegGeom* g) {
  g->B = d->B; g->Ch = d->Chans; g->T = d->T; g->F1 = d->F1; g->D = d->D; g->FD = d->F1 * d->D; g->F2 = d->F2;
  g->K1 = d->K1; g->K2 = d->K2; g->P1 = d->P1; g->P2 = d->P2;
  if (d->B <= 0 || d->Chans <= 0 || d->T <= 0) return -1;
  if (d->F1 != 8 || g->FD != 16 || d->F2 != 16) return -2;          // register-tiled for the EEGNet-8,2 family
  if (d->K1 < 1 || d->K1 > EEG_MAXK || d->K2 != 16) return -3;
  if (d->Chans > EEG_MAXCH || d->P1 < 1 || d->P2 < 1) return -4;
  g->T1 = d->T / d->P1; g->T2 = g->T1 / d->P2;
  if (g->T2 < 1) return -5;
  g->padl1 = (d->K1 - 1) / 2; g->padl2 = (d->K2 - 1) / 2;
  size_t o = 0;
  {   // conv1 output, or (collapsed front end) the input's sufficient statistics R[64][64], S[64] in fp64
    size_t c1b = (size_t)g->B * g->F1 * g->Ch * g->T * bx_esize(d->dtype);
    if (c1b < EEGC_RS_BYTES) c1b = EEGC_RS_BYTES;
    g->off_c1 = o; o += bx_align_up(c1b, 256);
  }
  g->off_d = o;  o += bx_align_up((size_t)g->B * g->FD * g->T * 4, 256);
  g->off_p1 = o; o += bx_align_up((size_t)g->B * g->FD * g->T1 * 4, 256);
  g->off_s = o;  o += bx_align_up((size_t)g->B * g->F2 * g->T1 * 4, 256);
  g->off_stats = o; o += bx_align_up((size_t)4 * (g->F1 + g->FD + g->F2) * 4, 256);
  g->total = o;
  return 0;
}
// stats block: [mean1 F1][invstd1 F1][scale1 F1][shift1 F1][mean2 FD]...[mean3 F2]...
struct EegStats { float *mean1, *inv1, *sc1, *sh1, *mean2, *inv2, *sc2, *sh2, *mean3, *inv3, *sc3, *sh3; };
static EegStats eeg_stats(const EegGeom& g, void* saved) {
  float* p = (float*)((char*)saved + g.off_stats);
  EegStats s;
  s.mean1 = p; s.inv1 = p + g.F1; s.sc1 = p + 2 * g.F1; s.sh1 = p + 3 * g.F1; p += 4 * g.F1;
  s.mean2 = p; s.inv2 = p + g.FD; s.sc2 = p + 2 * g.FD; s.sh2 = p + 3 * g.FD; p += 4 * g.FD;
  s.mean3 = p; s.inv3 = p + g.F2; s.sc3 = p + 2 * g.F2; s.sh3 = p + 3 * g.F2;
  return s;
}
// the tuned path needs the reference's default family AND rows that fit its one-row LDS tile; everything else: general kernels
static bool eeg_tuned(const bxEegDesc* d, EegGeom* g) { return eeg_geom(d, g) == 0 && d->T <= 15000; }
extern "C" size_t bx_eeg_saved_bytes(const bxEegDesc* d) {
  EegGeom g;
  if (!d) return 0;
  return eeg_tuned(d, &g) ? g.total : bx_eegg_saved_bytes(d);
}

// workspace: forward needs BN partials only; backward needs gradient maps + partial buffers
struct EegWs { size_t off_part, off_du3, off_dp1, off_du2, off_r, off_w1p, off_sepp, off_coef, total; int nblk_rows; };
static EegWs eeg_ws(const EegGeom& g) {
  EegWs w; size_t o = 0;
  const int rows = g.B * g.Ch;
  w.nblk_rows = rows;
  size_t npart = (size_t)w.nblk_rows;
  const size_t n_dw = (size_t)g.B * ((g.T + 127) / 128), n_sep = (size_t)g.B * ((g.T1 + 255) / 256);
  if (npart < n_dw) npart = n_dw;
  if (npart < n_sep) npart = n_sep;
  if (npart < (size_t)g.B) npart = (size_t)g.B;
  w.off_part = o; o += bx_align_up(npart * 2 * EEG_MAXF * 4, 256);
  w.off_du3 = o; o += bx_align_up((size_t)g.B * g.F2 * g.T1 * 4, 256);
  w.off_dp1 = o; o += bx_align_up((size_t)g.B * g.FD * g.T1 * 4, 256);
  w.off_du2 = o; o += bx_align_up((size_t)g.B * g.FD * g.T * 4, 256);
  {   // also the collapsed backward's per-electrode contributions [Ch][520] and sum-of-g partials [16][16]
    size_t rb = (size_t)g.B * (g.FD * g.Ch + g.FD) * 4;
    const size_t cb = ((size_t)g.Ch * 520 + EEGC_MAX_SPLIT * 16) * 4;
    if (rb < cb) rb = cb;
    w.off_r = o;   o += bx_align_up(rb, 256);
  }
  {   // also the collapsed front end's statistic partials (forward) and correlation partials [Ch][split][16][64] (backward)
    size_t wb = (size_t)rows * g.F1 * g.K1 * 4;
    const size_t sb = bx_eegc_stat_floats() * 4, cb = (size_t)g.Ch * EEGC_MAX_SPLIT * 1024 * 4;
    if (wb < sb) wb = sb;
    if (wb < cb) wb = cb;
    w.off_w1p = o; o += bx_align_up(wb, 256);
  }
  w.off_sepp = o; o += bx_align_up((size_t)g.B * 4 * g.F2 * g.FD * g.K2 * 4, 256);
  w.off_coef = o; o += bx_align_up((size_t)3 * 3 * EEG_MAXF * 4 + 256, 256);
  w.total = o;
  return w;
}
extern "C" size_t bx_eeg_workspace(const bxEegDesc* d) {
  EegGeom g;
  if (!d) return 0;
  if (!eeg_tuned(d, &g)) return bx_eegg_workspace(d);
  return eeg_ws(g).total;
}

// ------------------------------------------------------------------------------------------------
// E1: temporal convolution, one workgroup per row (sample, electrode): the whole padded row and the transposed
// filter bank ([k][8 filters]) sit in LDS; a thread produces 4 consecutive time steps x 8 filters (32 accumulators)
// with a sliding input window: per tap 1 input read + 2 broadcast ds_read_b128 of weights feed 32 FMAs.  BatchNorm
// partial sums stay in registers: ONE partial per row.
template <typename T>
__global__ __launch_bounds__(EEG_TT) void k_eeg_conv1(const float* __restrict__ x, const float* __restrict__ w1, T* __restrict__ c1,
                                                      float* __restrict__ partials, EegGeom g, int want_stats) {
  extern __shared__ __attribute__((aligned(16))) float sm1[];
  float* swt = sm1;                    // [K1][8]
  float* sx = sm1 + 8 * EEG_MAXK;      // [T + K1 + 3]
  __shared__ float red[4][16];
  const int row = blockIdx.x;
  const int b = row / g.Ch, ch = row % g.Ch;
  const float* xr = x + (size_t)row * g.T;
  {
    const int Tn = g.T, pl = g.padl1, K1 = g.K1;
    lds_fill<8>(sx, Tn + K1 + 3, [&](int i) { const int t = i - pl; return (t >= 0 && t < Tn) ? xr[t] : 0.f; });
    lds_fill<2>(swt, 8 * K1, [&](int j) { return w1[(j & 7) * K1 + (j >> 3)]; });          // [k][8 filters]
  }
  __syncthreads();
  float s[8], q[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) s[f] = q[f] = 0.f;
  const int nq = (g.T + 3) / 4;
  for (int i = threadIdx.x; i < nq; i += EEG_TT) {
    const int t0 = 4 * i;
    float acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int f = 0; f < 8; ++f) acc[j][f] = 0.f;
    float x0 = sx[t0], x1 = sx[t0 + 1], x2 = sx[t0 + 2];
#pragma unroll 4
    for (int k = 0; k < g.K1; ++k) {
      const float x3 = sx[t0 + k + 3];
      const float4 wa = *reinterpret_cast<const float4*>(swt + k * 8);
      const float4 wb = *reinterpret_cast<const float4*>(swt + k * 8 + 4);
      const float wv[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
      const float xw[4] = {x0, x1, x2, x3};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[j][f] = fmaf(wv[f], xw[j], acc[j][f]);
      x0 = x1; x1 = x2; x2 = x3;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const size_t o = (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T + t0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t0 + j < g.T) {
          stf(c1, o + j, acc[j][f]);
          const float v = round_as(c1, acc[j][f]);                                // statistics of the stored value
          s[f] += v; q[f] += v * v;
        }
    }
  }
  if (!want_stats) return;
#pragma unroll
  for (int f = 0; f < 8; ++f) { s[f] = wave_sum(s[f]); q[f] = wave_sum(q[f]); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
#pragma unroll
    for (int f = 0; f < 8; ++f) { red[wave][f] = s[f]; red[wave][8 + f] = q[f]; }
  __syncthreads();
  if (threadIdx.x < 16)      // layout [row][2][F1]
    partials[(size_t)blockIdx.x * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// E2: BN1 apply + depthwise electrode mix.  A thread owns DWV consecutive time steps (8-byte c1 loads for bf16 when
// T % 4 == 0, else one step) and produces all FD maps for them.  grid (ceil(T / (128*DWV)), B), 128 threads.
template <typename T, int DWV>
__global__ __launch_bounds__(128) void k_eeg_dw(const T* __restrict__ c1, const float* __restrict__ dw, const float* __restrict__ sc1,
                                                 const float* __restrict__ sh1, float* __restrict__ dmap, float* __restrict__ partials,
                                                 EegGeom g, int want_stats) {
  __shared__ float sdw[EEG_MAXF * EEG_MAXCH];
  __shared__ float red[2][32];
  for (int i = threadIdx.x; i < g.FD * g.Ch; i += 128) sdw[i] = dw[i];
  __syncthreads();
  const int b = blockIdx.y, t0 = (blockIdx.x * 128 + threadIdx.x) * DWV;
  float ssum[EEG_MAXF], ssq[EEG_MAXF];
#pragma unroll
  for (int i = 0; i < EEG_MAXF; ++i) ssum[i] = ssq[i] = 0.f;
  if (t0 < g.T) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const float a = sc1[f], c = sh1[f];
      float o0[DWV], o1[DWV];
#pragma unroll
      for (int j = 0; j < DWV; ++j) o0[j] = o1[j] = 0.f;
      const size_t rb = (((size_t)b * g.F1 + f) * g.Ch) * g.T + t0;
      // electrodes in batches of 10 with every load of a batch issued before the first use (clamped index, no per-load
      // branch): the plain `for ch` loop was one memory round trip per electrode, 8 * Ch of them per thread
      for (int ch0 = 0; ch0 < g.Ch; ch0 += 10) {
        float v[10][DWV];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
          const int ch = ch0 + q < g.Ch ? ch0 + q : g.Ch - 1;
          if (DWV == 4) ld4(c1, rb + (size_t)ch * g.T, v[q]);
          else v[q][0] = ldf(c1, rb + (size_t)ch * g.T);
        }
#pragma unroll
        for (int q = 0; q < 10; ++q) {
          const int ch = ch0 + q;
          const bool live = ch < g.Ch;
          const int chc = live ? ch : g.Ch - 1;
          const float w0 = live ? sdw[(2 * f) * g.Ch + chc] : 0.f, w1 = live ? sdw[(2 * f + 1) * g.Ch + chc] : 0.f;
#pragma unroll
          for (int j = 0; j < DWV; ++j) {
            const float u = v[q][j] * a + c;
            o0[j] = fmaf(w0, u, o0[j]);
            o1[j] = fmaf(w1, u, o1[j]);
          }
        }
      }
      float* d0 = dmap + ((size_t)b * g.FD + 2 * f) * g.T + t0;
      if (DWV == 4) {
        *reinterpret_cast<float4*>(d0) = make_float4(o0[0], o0[1], o0[2], o0[3]);
        *reinterpret_cast<float4*>(d0 + g.T) = make_float4(o1[0], o1[1], o1[2], o1[3]);
      } else {
        d0[0] = o0[0]; d0[g.T] = o1[0];
      }
#pragma unroll
      for (int j = 0; j < DWV; ++j) {
        ssum[2 * f] += o0[j]; ssq[2 * f] += o0[j] * o0[j];
        ssum[2 * f + 1] += o1[j]; ssq[2 * f + 1] += o1[j] * o1[j];
      }
    }
  }
  if (!want_stats) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < EEG_MAXF; ++i) {
    const float s = wave_sum(ssum[i]), q = wave_sum(ssq[i]);
    if (lane == 0) { red[wave][i] = s; red[wave][16 + i] = q; }
  }
  __syncthreads();
  if (threadIdx.x < 32)
    partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];   // [blk][2][16]
}

__device__ __forceinline__ float elu1(float u) { return u > 0.f ? u : expm1f(u); }

// E3/E5: BN apply + ELU + average pool (1xP, floor) + dropout.  in [B,F,Tin] -> out [B,F,Tout]
// Folded finalize (round 3, fold.rows != NULL; F == 16, 256 threads): no k_bn_finalize launch in front of this kernel -- every
// workgroup sums the producer's partial rows ([row][2][16]) itself (bx_rows_total) and derives (scale, shift); workgroup 0 writes the
// running statistics and what the backward reads.
struct EegBnFold { const float* rows; int nrows; BxBnFinalize fin; };
__global__ __launch_bounds__(256) void k_eeg_bn_elu_pool(const float* __restrict__ in, const float* __restrict__ sc, const float* __restrict__ sh,
                                  float* __restrict__ out, int B, int F, int Tin, int Tout, int P, const uint64_t* __restrict__ seed,
                                  float dropout_p, uint32_t salt, EegBnFold fold) {
  __shared__ float ssc[2][16];
  if (fold.rows) {
    __shared__ __attribute__((aligned(8))) double sp[BX_ROWS_TOTAL_LDS(2) / 8];
    bx_rows_total<2>(fold.rows, fold.nrows, 16, sp);
    if (threadIdx.x < 16) {
      const int c = threadIdx.x;
      const BxBnFinalize& f = fold.fin;
      float mean, invstd; double unbiased;
      bx_bn_batch_stats(sp[c], sp[16 + c], f.count, f.eps, mean, invstd, unbiased);
      const float s_ = f.gamma[c] * invstd, h_ = f.beta[c] - mean * s_;
      ssc[0][c] = s_; ssc[1][c] = h_;
      if (blockIdx.x == 0) {
        f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
        f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unbiased;
        if (c == 0 && f.nbt) f.nbt[0] += 1;
        f.scale[c] = s_; f.shift[c] = h_; f.save_mean[c] = mean; f.save_invstd[c] = invstd;
      }
    }
    __syncthreads();
  }
  const long long n = (long long)B * F * Tout;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned i32 = (unsigned)i, bfu = i32 / (unsigned)Tout;      // n < 2^31: 32-bit divisions
    const int to = (int)(i32 - bfu * (unsigned)Tout);
    const long long bf = (long long)bfu;
    const int f = (int)(bfu % (unsigned)F);
    const float a = fold.rows ? ssc[0][f] : sc[f], c = fold.rows ? ssc[1][f] : sh[f];
    const float* src = in + bf * Tin + (size_t)to * P;
    float s = 0.f;
    for (int j = 0; j < P; ++j) s += elu1(src[j] * a + c);
    s /= (float)P;
    if (dropout_p > 0.f) s *= bx_dropout_scale(sd, salt, (salt >> 31) ? (uint64_t)bf : (uint64_t)i, dropout_p, inv_keep);   // salt bit 31: one mask entry per (sample, channel) = nn.Dropout2d
    out[i] = s;
  }
}

// E4: dense temporal conv FD -> F2 over K2 = 16 taps.  grid (ceil(T1/256), B), 256 threads: wave w owns outputs
// 4w..4w+3, a lane owns 4 consecutive time steps -> 16 accumulators.  The input tile [16][256+16] and the weights
// (transposed to [fd][k][o]) sit in LDS; per input map a lane reads its window as 5 aligned float4s and per tap one
// wave-uniform float4 of weights (broadcast): 21 LDS reads feed 256 FMAs (the previous thread-per-time-step layout
// issued 17 reads per 16 FMAs and was LDS bound).  BatchNorm partial sums: one [2][16] row per workgroup.
#define SEP_TT 256
// NO = output maps per wave (round 3): grid.z = 4 / NO workgroups share a (time chunk, sample) and split the 16 outputs -- at T1 = 500
// the (2, B) grid was 128 workgroups of 4 096 FMAs per lane on half of the CUs (14.4 us); the weights a workgroup does not use are
// not staged.  Same sums in the same order per output, so results do not depend on NO.
template <int NO>
__global__ __launch_bounds__(256) void k_eeg_sep(const float* __restrict__ p1, const float* __restrict__ ws, float* __restrict__ s,
                                                  float* __restrict__ partials, EegGeom g, int want_stats) {
  constexpr int TP = SEP_TT + 16, OG = 4 * NO;                      // OG outputs per workgroup
  __shared__ __attribute__((aligned(16))) float sp[16 * TP];        // index j <-> t = t0 + j - padl2 (zero outside [0, T1))
  __shared__ __attribute__((aligned(16))) float sw[16 * 16 * OG];   // [fd][k][o - o0]
  const int b = blockIdx.y, t0 = blockIdx.x * SEP_TT, o0 = blockIdx.z * OG;
  {
    const int T1 = g.T1, pl = g.padl2;
    const float* pb = p1 + (size_t)b * g.FD * T1;
    lds_fill<OG>(sw, 256 * OG, [&](int j) { return ws[((o0 + j % OG) * 16 + j / (16 * OG)) * 16 + (j / OG) % 16]; });   // [fd][k][o] <- [o][fd][k]
    lds_fill<17>(sp, 16 * TP, [&](int i) { const int fd = i / TP, tt = t0 + i % TP - pl; return (tt >= 0 && tt < T1) ? pb[(size_t)fd * T1 + tt] : 0.f; });
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tl = 4 * lane;
  float acc[4][NO];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[j][o] = 0.f;
  for (int fd = 0; fd < 16; ++fd) {
    const float* row = sp + fd * TP + tl;
    float4 cur = *reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const float4 nxt = *reinterpret_cast<const float4*>(row + 4 * (kq + 1));
      const float win[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float wv[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) wv[o] = sw[(fd * 16 + 4 * kq + kk) * OG + NO * wave + o];      // wave-uniform (broadcast) reads
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int o = 0; o < NO; ++o) acc[j][o] = fmaf(wv[o], win[kk + j], acc[j][o]);
      }
      cur = nxt;
    }
  }
  float su[NO], sq[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    su[o] = sq[o] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tt = t0 + tl + j;
      if (tt < g.T1) {
        s[((size_t)b * g.F2 + o0 + NO * wave + o) * g.T1 + tt] = acc[j][o];
        su[o] += acc[j][o]; sq[o] += acc[j][o] * acc[j][o];
      }
    }
  }
  if (!want_stats) return;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float a = wave_sum(su[o]), q = wave_sum(sq[o]);
    if (lane == 0) {
      partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + o0 + NO * wave + o] = a;
      partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + 16 + o0 + NO * wave + o] = q;
    }
  }
}

// eval mode: scale / shift / mean / invstd of all three BatchNorms straight from the running statistics, one launch
// (the training path needs a finalize after each producer; in eval the three sets are known up front)
__global__ void k_eeg_eval_stats(const float* g1, const float* b1, const float* m1, const float* v1, int n1, float* sc1, float* sh1, float* mean1, float* inv1,
                                 const float* g2, const float* b2, const float* m2, const float* v2, int n2, float* sc2, float* sh2, float* mean2, float* inv2,
                                 const float* g3, const float* b3, const float* m3, const float* v3, int n3, float* sc3, float* sh3, float* mean3, float* inv3,
                                 float eps) {
  const int i = threadIdx.x;
  auto one = [&](const float* g, const float* b, const float* m, const float* v, int n, float* sc, float* sh, float* mean, float* inv) {
    if (i < n) {
      const float is = 1.0f / sqrtf(v[i] + eps), s_ = g[i] * is;
      sc[i] = s_; sh[i] = b[i] - m[i] * s_; mean[i] = m[i]; inv[i] = is;
    }
  };
  one(g1, b1, m1, v1, n1, sc1, sh1, mean1, inv1);
  one(g2, b2, m2, v2, n2, sc2, sh2, mean2, inv2);
  one(g3, b3, m3, v3, n3, sc3, sh3, mean3, inv3);
}

extern "C" int bx_eeg_features_fwd(const bxEegDesc* d, const bxEegParams* p, const float* x, const uint64_t* seed,
                                   float* feat, void* saved, void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(d && p && x && feat && saved, "bx_eeg_features_fwd: null pointer");
  BX_DTYPE_OK(d->dtype);
  EegGeom g;
  // x_slot: only the collapsed evaluation-mode front end reads the input in exactly one kernel (nothing is saved for a backward)
  BX_REQUIRE(!d->x_slot || (eeg_tuned(d, &g) && eeg_collapsed_eval(d)), "bx_eeg_features_fwd: x_slot needs the collapsed evaluation-mode path (collapse=1, training=0, default geometry)");
  if (!eeg_tuned(d, &g))                                  // any other F1 / D / F2 / kernLength / Chans (models.py:239-262 is parametric)
    return bx_eegg_forward(d, p, x, seed, feat, saved, workspace, workspace_bytes, (hipStream_t)stream);
  const EegWs w = eeg_ws(g);
  if (!workspace || workspace_bytes < w.total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_features_fwd: workspace %zu < %zu", workspace_bytes, w.total);
  BX_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "bx_eeg_features_fwd: dropout_p must be in [0,1)");
  hipStream_t s = (hipStream_t)stream;
  const EegStats st = eeg_stats(g, saved);
  float* part = (float*)((char*)workspace + w.off_part);
  void* c1 = (char*)saved + g.off_c1;
  float* dmap = (float*)((char*)saved + g.off_d);
  float* p1 = (float*)((char*)saved + g.off_p1);
  float* smap = (float*)((char*)saved + g.off_s);
  const int tr = d->training;
  const float pdrop = tr ? d->dropout_p : 0.f;
  const float pdrop2 = tr ? (d->dropout_p2 >= 0.f ? d->dropout_p2 : d->dropout_p) : 0.f;      // the second dropout's own rate (EEGNetAttentionDeep.dropout2)
  BX_REQUIRE((pdrop == 0.f && pdrop2 == 0.f) || seed, "bx_eeg_features_fwd: dropout needs a device seed");

  if (!tr) {
    hipLaunchKernelGGL(k_eeg_eval_stats, dim3(1), dim3(64), 0, s, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, g.F1, st.sc1, st.sh1, st.mean1, st.inv1,
                       p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, g.FD, st.sc2, st.sh2, st.mean2, st.inv2,
                       p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, g.F2, st.sc3, st.sh3, st.mean3, st.inv3, d->eps);
    BX_CHECK_LAUNCH("eeg eval stats");
  }
  EegBnFold fold2 = {}, fold3 = {};                       // BatchNorm2 / BatchNorm3 finalizes folded into the pooling kernels (few partial rows)
  if (eeg_collapsed_eval(d)) {
    const int rc = bx_eegc_forward_eval(x, p->conv1_w, p->dw_w, st.sc1, st.sh1, dmap, g.B, g.Ch, g.T, s, d->x_slot);
    BX_REQUIRE(rc == 0, "bx_eeg_features_fwd: collapsed evaluation-mode front end failed (code %d)", rc);
  } else if (eeg_collapsed(d)) {
    // conv1 -> BatchNorm1 -> electrode mix without the [B,8,Chans,T] tensor (eeg_collapse.hip): statistics, finalize, forward
    int rows2 = 0;
    const int rc = bx_eegc_forward(x, p->conv1_w, p->dw_w, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, p->bn1_nbt, d->momentum, d->eps, st.mean1, st.inv1,
                                   st.sc1, st.sh1, (double*)c1, (float*)((char*)workspace + w.off_w1p), dmap, part, &rows2, g.B, g.Ch, g.T, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_fwd: collapsed front end failed (code %d)", rc);
    if (tr && (bx_tail_fold_mask() & 2) && g.FD == 16 && (long long)rows2 * 16 <= 8192) {
      fold2 = EegBnFold{part, rows2, BxBnFinalize{p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, p->bn2_nbt, d->momentum, d->eps, st.sc2, st.sh2, st.mean2,
                                                  st.inv2, (double)g.B * g.T}};
    } else {
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.FD)), dim3(1024), 0, s, part, rows2, (double)g.B * g.T, g.FD, tr, p->bn2_w, p->bn2_b,
                       p->bn2_rm, p->bn2_rv, p->bn2_nbt, d->momentum, d->eps, st.sc2, st.sh2, st.mean2, st.inv2);
    BX_CHECK_LAUNCH("eeg bn2 (collapsed front end)");
    }
  } else {
  // bf16 storage with the reference's 64-tap kernel: the temporal convolution runs on the matrix cores (eeg_mfma.hip)
  const bool no_mfma = getenv("BX_EEG_NO_MFMA") != nullptr;          // read per call: tests flip it to compare both paths
  if (d->dtype == BX_BF16 && g.K1 == 64 && !no_mfma && bx_eeg_conv1_mfma_lds(g.T) <= 150 * 1024) {
    const int rc = bx_eeg_conv1_mfma_launch(x, p->conv1_w, c1, part, g.B, g.Ch, g.T, tr, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_fwd: MFMA temporal convolution launch failed (code %d)", rc);
  } else {
    BX_DISPATCH_DTYPE(d->dtype, T,
      hipLaunchKernelGGL((k_eeg_conv1<T>), dim3(w.nblk_rows), dim3(EEG_TT), (size_t)(8 * EEG_MAXK + g.T + g.K1 + 8) * sizeof(float), s, x, p->conv1_w, (T*)c1, part, g, tr));
  }
  BX_CHECK_LAUNCH("eeg conv1");
  if (tr)
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.F1)), dim3(1024), 0, s, part, w.nblk_rows, (double)g.B * g.Ch * g.T, g.F1, tr, p->bn1_w, p->bn1_b,
                     p->bn1_rm, p->bn1_rv, p->bn1_nbt, d->momentum, d->eps, st.sc1, st.sh1, st.mean1, st.inv1);
  BX_CHECK_LAUNCH("eeg bn1");
  const int dwv = 1;      // 4 steps per thread (8-byte loads) measured SLOWER (35 vs 27 us): fewer, longer serial load chains
  dim3 gdw(bx_ceil_div(g.T, 128 * dwv), g.B);
  BX_DISPATCH_DTYPE(d->dtype, T,
    if (dwv == 4) hipLaunchKernelGGL((k_eeg_dw<T, 4>), gdw, dim3(128), 0, s, (const T*)c1, p->dw_w, st.sc1, st.sh1, dmap, part, g, tr);
    else hipLaunchKernelGGL((k_eeg_dw<T, 1>), gdw, dim3(128), 0, s, (const T*)c1, p->dw_w, st.sc1, st.sh1, dmap, part, g, tr));
  BX_CHECK_LAUNCH("eeg depthwise");
  if (tr)
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.FD)), dim3(1024), 0, s, part, (int)(gdw.x * gdw.y), (double)g.B * g.T, g.FD, tr, p->bn2_w, p->bn2_b,
                     p->bn2_rm, p->bn2_rv, p->bn2_nbt, d->momentum, d->eps, st.sc2, st.sh2, st.mean2, st.inv2);
  BX_CHECK_LAUNCH("eeg bn2");
  }
  {
    const long long n = (long long)g.B * g.FD * g.T1;
    const long long nb = bx_ceil_div(n, 256);
    hipLaunchKernelGGL(k_eeg_bn_elu_pool, dim3((unsigned)(fold2.rows && nb > 512 ? 512 : nb)), dim3(256), 0, s, dmap, st.sc2, st.sh2, p1, g.B, g.FD, g.T,
                       g.T1, g.P1, seed, pdrop, d->salt, fold2);      // (folded: every workgroup re-reads the rows -- at most 512 of them)
    BX_CHECK_LAUNCH("eeg pool1");
  }
  dim3 gsep(bx_ceil_div(g.T1, SEP_TT), g.B);
  {
    // outputs per wave: split the 16 outputs over 2 or 4 workgroups while that is what it takes to give every CU one (BX_EEG_SEP_NO overrides)
    static const int no_env = getenv("BX_EEG_SEP_NO") ? atoi(getenv("BX_EEG_SEP_NO")) : 0;
    const long long wgs = (long long)gsep.x * gsep.y;
    // measured at B = 64, T1 = 500 (training step, same box): NO = 4 1.483 ms, 2 1.479, 1 1.486 (one 4-byte weight read per 4 FMAs)
    const int no = no_env == 1 || no_env == 2 || no_env == 4 ? no_env : wgs >= 512 ? 4 : 2;
    if (no == 4) hipLaunchKernelGGL(k_eeg_sep<4>, gsep, dim3(256), 0, s, p1, p->sep_w, smap, part, g, tr);
    else if (no == 2) hipLaunchKernelGGL(k_eeg_sep<2>, dim3(gsep.x, gsep.y, 2), dim3(256), 0, s, p1, p->sep_w, smap, part, g, tr);
    else hipLaunchKernelGGL(k_eeg_sep<1>, dim3(gsep.x, gsep.y, 4), dim3(256), 0, s, p1, p->sep_w, smap, part, g, tr);
  }
  BX_CHECK_LAUNCH("eeg sepconv");
  if (tr && (bx_tail_fold_mask() & 2) && g.F2 == 16 && (long long)gsep.x * gsep.y * 16 <= 8192)
    fold3 = EegBnFold{part, (int)(gsep.x * gsep.y), BxBnFinalize{p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, p->bn3_nbt, d->momentum, d->eps, st.sc3, st.sh3,
                                                                  st.mean3, st.inv3, (double)g.B * g.T1}};
  else if (tr)
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.F2)), dim3(1024), 0, s, part, (int)(gsep.x * gsep.y), (double)g.B * g.T1, g.F2, tr, p->bn3_w, p->bn3_b,
                     p->bn3_rm, p->bn3_rv, p->bn3_nbt, d->momentum, d->eps, st.sc3, st.sh3, st.mean3, st.inv3);
  BX_CHECK_LAUNCH("eeg bn3");
  {
    const long long n = (long long)g.B * g.F2 * g.T2;
    const long long nb = bx_ceil_div(n, 256);
    hipLaunchKernelGGL(k_eeg_bn_elu_pool, dim3((unsigned)(fold3.rows && nb > 512 ? 512 : nb)), dim3(256), 0, s, smap, st.sc3, st.sh3, feat, g.B, g.F2, g.T1,
                       g.T2, g.P2, seed, pdrop2, d->salt + 1, fold3);
    BX_CHECK_LAUNCH("eeg pool2");
  }
  return BX_OK;
}

// ================================================================================================
// backward
// B-act: gradient through dropout/avg-pool/ELU up to the BN output, du = dL/d(bn_out); writes du and the
// BN-backward partial sums  s1 = sum du,  s2 = sum du * xhat.  One workgroup per (b), 256 threads.
__global__ __launch_bounds__(256) void k_eeg_act_bwd(const float* __restrict__ dpool, const float* __restrict__ pre, const float* __restrict__ mean,
    const float* __restrict__ inv, const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ du,
    float* __restrict__ partials, int F, int Tin, int Tout, int P, const uint64_t* __restrict__ seed, float dropout_p, uint32_t salt) {
  // grid (B, F): one workgroup per (sample, feature map); partial layout [b][2][16]
  __shared__ float red[4][2];
  const int b = blockIdx.x, f = blockIdx.y;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s1 = 0.f, s2 = 0.f;
  const float a = sc[f], c = sh[f], mu = mean[f], is = inv[f];
  for (int t = threadIdx.x; t < Tin; t += 256) {
    const size_t i = ((size_t)b * F + f) * Tin + t;
    float g_ = 0.f;
    const int to = t / P;
    if (to < Tout) {
      const size_t io = ((size_t)b * F + f) * Tout + to;
      float go = dpool[io] / (float)P;
      if (dropout_p > 0.f) go *= bx_dropout_scale(sd, salt, (salt >> 31) ? (uint64_t)((size_t)b * F + f) : (uint64_t)io, dropout_p, inv_keep);
      const float v = pre[i];
      const float u = v * a + c;
      g_ = u > 0.f ? go : go * expf(u);
      s2 += g_ * (v - mu) * is;
    }
    du[i] = g_;
    s1 += g_;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
  __syncthreads();
  if (threadIdx.x < 2)
    partials[(size_t)b * 32 + threadIdx.x * 16 + f] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// finalize: dgamma, dbeta, coefficients a = gamma*invstd, k1 = s1/N, k2 = s2/N (zero in eval mode)
// Workgroups > 0 (optional second role): fixed-order sum of an unrelated, already finished set of weight-gradient partials
// (sum_src [nchunk][n] -> sum_dst[n], as k_sum_partials with S slices) that would otherwise be a launch of its own.
__global__ __launch_bounds__(1024) void k_eeg_bn_bwd_finalize(const float* __restrict__ partials, int nblk, double count, int F, int training,
                                      const float* __restrict__ gamma, const float* __restrict__ inv, float* __restrict__ coef,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                      const float* __restrict__ sum_src, float* __restrict__ sum_dst, int nchunk, int n, int S) {
  if (blockIdx.x > 0) {
    __shared__ float sm[1024];
    const int NO = 256 / S, grp = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int o = t % NO, sl = t / NO;
    const int i = (((int)blockIdx.x - 1) * 4 + grp) * NO + o;
    float a = 0.f;
    if (i < n) {
      int k = sl;
      for (; k + 3 * S < nchunk; k += 4 * S) {
        const float v0 = sum_src[(size_t)k * n + i], v1 = sum_src[(size_t)(k + S) * n + i];
        const float v2 = sum_src[(size_t)(k + 2 * S) * n + i], v3 = sum_src[(size_t)(k + 3 * S) * n + i];
        a += v0; a += v1; a += v2; a += v3;
      }
      for (; k < nchunk; k += S) a += sum_src[(size_t)k * n + i];
    }
    sm[threadIdx.x] = a;
    __syncthreads();
    if (sl == 0 && i < n) {
      float r = sm[grp * 256 + o];
      for (int k = 1; k < S; ++k) r += sm[grp * 256 + k * NO + o];
      sum_dst[i] = r;
    }
    return;
  }
  // partial layout [blk][2][16] regardless of F
  double s[2] = {0.0, 0.0};
  sum_partials_256<2>(partials, nblk, 16, 0, 16, s);
  const int f = threadIdx.x;
  if (f >= F) return;
  if (dbeta) dbeta[f] = (float)s[0];
  if (dgamma) dgamma[f] = (float)s[1];
  coef[f] = gamma[f] * inv[f];
  coef[EEG_MAXF + f] = training ? (float)(s[0] / count) : 0.f;
  coef[2 * EEG_MAXF + f] = training ? (float)(s[1] / count) : 0.f;
}
// in place: du -> dpre = a * (du - k1 - xhat * k2)
__global__ void k_eeg_bn_bwd_apply(float* __restrict__ du, const float* __restrict__ pre, const float* __restrict__ mean,
                                   const float* __restrict__ inv, const float* __restrict__ coef, int F, int Tin, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(((unsigned)i / (unsigned)Tin) % (unsigned)F);
    du[i] = coef[f] * (du[i] - coef[EEG_MAXF + f] - (pre[i] - mean[f]) * inv[f] * coef[2 * EEG_MAXF + f]);
  }
}

// the same for the collapsed front end: the result goes out as bf16 (the MFMA operand of k_eegc_corr), 8 elements per thread
__global__ __launch_bounds__(256) void k_eeg_bn_bwd_apply16(const float* __restrict__ du, const float* __restrict__ pre, const float* __restrict__ mean,
                                                             const float* __restrict__ inv, const float* __restrict__ coef, bf16_t* __restrict__ out,
                                                             int F, int Tin, long long n8) {
  for (long long i8 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i8 < n8; i8 += (long long)gridDim.x * blockDim.x) {
    const long long i = i8 * 8;
    const int f = (int)(((unsigned)i / (unsigned)Tin) % (unsigned)F);          // Tin % 8 == 0: the 8 elements share a row
    const float a = coef[f], k1 = coef[EEG_MAXF + f], k2 = coef[2 * EEG_MAXF + f] * inv[f], mu = mean[f];
    const float4 d0 = *reinterpret_cast<const float4*>(du + i), d1 = *reinterpret_cast<const float4*>(du + i + 4);
    const float4 p0 = *reinterpret_cast<const float4*>(pre + i), p1 = *reinterpret_cast<const float4*>(pre + i + 4);
    const float o[8] = {a * (d0.x - k1 - (p0.x - mu) * k2), a * (d0.y - k1 - (p0.y - mu) * k2), a * (d0.z - k1 - (p0.z - mu) * k2),
                        a * (d0.w - k1 - (p0.w - mu) * k2), a * (d1.x - k1 - (p1.x - mu) * k2), a * (d1.y - k1 - (p1.y - mu) * k2),
                        a * (d1.z - k1 - (p1.z - mu) * k2), a * (d1.w - k1 - (p1.w - mu) * k2)};
    *reinterpret_cast<uint4*>(out + i) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
  }
}

// sepconv backward: grid (B, 4); LDS holds the sample's ds[16][.] and p1[16][.] rows (zero halo) and the weights.
// (a) dp1[fd][t] = sum_{o,k} ws[o][fd][k] ds[o][t-k+padl]: wave w of quarter q owns map fd = 4q+w, a lane 8 consecutive
//     time steps; per output map o it reads the 23-value window as 6 aligned float4s and 16 weights as 4 broadcast
//     float4s for 128 FMAs.
// (b) partial dws[o][fd][k] = sum_t ds[o][t] p1[fd][t+k-padl] over the quarter's time range: thread = (o, fd), 16 tap
//     accumulators, sliding p1 window: 2 float4 reads per 64 FMAs.
// Row pitch TP: multiple of 4 with TP/4 odd (the 16 fd rows a wave reads then fall on different bank groups).
__host__ __device__ inline int eeg_sepb_pitch(int T1) { int tp = ((T1 + 7) & ~7) + 24; if (((tp >> 2) & 1) == 0) tp += 4; return tp; }
// The BatchNorm3 backward apply (k_eeg_bn_bwd_apply: ds = a (du - k1 - xhat k2), elementwise) rides in the staging of ds when `pre`
// is given: same expression, evaluated on the way into LDS -- one launch and one pass over du3 fewer.
__global__ __launch_bounds__(256) void k_eeg_sep_bwd(const float* __restrict__ ds, const float* __restrict__ p1, const float* __restrict__ ws,
                                                      float* __restrict__ dp1, float* __restrict__ wpart, EegGeom g, const float* __restrict__ pre,
                                                      const float* __restrict__ mean, const float* __restrict__ inv, const float* __restrict__ coef) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T1 = g.T1, TP = eeg_sepb_pitch(T1);
  // round 3: with grid (B, 8) the two halves run in DIFFERENT workgroups -- blockIdx.y < 4: (a) for quarter q, >= 4: (b) -- each staging
  // only what it reads (LDS: ds + weights, or ds + p1: two workgroups per CU), so that B = 64 launches 512 workgroups of half the work
  // instead of 256 at one per CU (27 us).  grid (B, 4): both halves in one workgroup as before (role 2).  Same sums either way.
  const int role = gridDim.y == 8 ? (int)(blockIdx.y >> 2) : 2;
  float* sds = sm;                 // [16][TP]  index t + 8
  float* sp1 = sm + 16 * TP;       // [16][TP]  index t + 7          (roles 1, 2)
  float* sw = role == 0 ? sm + 16 * TP : sp1 + 16 * TP;       // [o][fd][k]   (roles 0, 2)
  const int b = blockIdx.x, q = blockIdx.y & 3;
  {
    const float* dsb = ds + (size_t)b * 16 * T1;
    const float* p1b = p1 + (size_t)b * 16 * T1;
    if ((T1 & 3) == 0) {
      // zero the halos, then copy the rows as float4s (8 + 8 loads in flight per thread)
      float4* z = reinterpret_cast<float4*>(sm);
      for (int i = threadIdx.x; i < (role == 0 ? 1 : 2) * 16 * TP / 4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
      const int n4 = T1 / 4, tot = 16 * n4;
      for (int i0 = threadIdx.x; i0 < tot; i0 += 256 * 8) {
        float4 a[8], p[8], q8[8];
        const float* preb = pre ? pre + (size_t)b * 16 * T1 : dsb;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 256, ic = i < tot ? i : 0;      // (clamped, unconditional loads: a conditionally filled array went to scratch memory)
          a[u] = reinterpret_cast<const float4*>(dsb)[ic]; p[u] = reinterpret_cast<const float4*>(role == 0 ? dsb : p1b)[ic];
          q8[u] = reinterpret_cast<const float4*>(preb)[ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * 256;
          if (i < tot) {
            const int f = i / n4, c = i % n4;
            if (pre) {
              const float ca = coef[f], k1 = coef[EEG_MAXF + f], k2 = coef[2 * EEG_MAXF + f], mu = mean[f], is = inv[f];
              a[u].x = ca * (a[u].x - k1 - (q8[u].x - mu) * is * k2); a[u].y = ca * (a[u].y - k1 - (q8[u].y - mu) * is * k2);
              a[u].z = ca * (a[u].z - k1 - (q8[u].z - mu) * is * k2); a[u].w = ca * (a[u].w - k1 - (q8[u].w - mu) * is * k2);
            }
            *reinterpret_cast<float4*>(sds + f * TP + 8 + 4 * c) = a[u];
            if (role != 0) {
              float* q = sp1 + f * TP + 7 + 4 * c;
              q[0] = p[u].x; q[1] = p[u].y; q[2] = p[u].z; q[3] = p[u].w;
            }
          }
        }
      }
    } else {
      lds_fill<16>(sds, 16 * TP, [&](int i) {
        const int f = i / TP, ta = i % TP - 8;
        if (!(ta >= 0 && ta < T1)) return 0.f;
        const float d = dsb[(size_t)f * T1 + ta];
        if (!pre) return d;
        return coef[f] * (d - coef[EEG_MAXF + f] - (pre[((size_t)b * 16 + f) * T1 + ta] - mean[f]) * inv[f] * coef[2 * EEG_MAXF + f]);
      });
      if (role != 0) lds_fill<16>(sp1, 16 * TP, [&](int i) { const int f = i / TP, tb = i % TP - 7; return (tb >= 0 && tb < T1) ? p1b[(size_t)f * T1 + tb] : 0.f; });
    }
    if (role != 1) lds_fill<16>(sw, 4096, [&](int i) { return ws[i]; });
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (a)
  if (role != 1) {
    const int fd = 4 * q + wave;
    for (int t8 = lane; t8 * 8 < T1; t8 += 64) {
      const int tb = t8 * 8;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int o = 0; o < 16; ++o) {
        float V[24], Wt[16];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(sds + o * TP + tb + 4 * c);   // ds[o][tb - 8 + 4c ..]
          V[4 * c] = v.x; V[4 * c + 1] = v.y; V[4 * c + 2] = v.z; V[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 v = *reinterpret_cast<const float4*>(sw + (o * 16 + fd) * 16 + 4 * c);
          Wt[4 * c] = v.x; Wt[4 * c + 1] = v.y; Wt[4 * c + 2] = v.z; Wt[4 * c + 3] = v.w;
        }
        // ds[o][t - k + padl] with t = tb + j, padl = 7  ->  V[(t - k + 7) - (tb - 8)] = V[15 - k + j]
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(Wt[k], V[15 - k + j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (tb + j < T1) dp1[((size_t)b * 16 + fd) * T1 + tb + j] = acc[j];
    }
  }
  // (b)
  if (role != 0) {
    const int o = threadIdx.x >> 4, fd = threadIdx.x & 15;
    const int n4 = (T1 + 3) / 4;
    const int c0 = n4 * q / 4, c1 = n4 * (q + 1) / 4;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    const float* prow = sp1 + fd * TP;        // prow[t + k] = p1[fd][t + k - 7]
    const float* drow = sds + o * TP + 8;
    float P[20];
    if (c0 < c1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(prow + 4 * c0 + 4 * c);
        P[4 * c] = v.x; P[4 * c + 1] = v.y; P[4 * c + 2] = v.z; P[4 * c + 3] = v.w;
      }
    }
    for (int c = c0; c < c1; ++c) {
      const int tq = 4 * c;
      const float4 d = *reinterpret_cast<const float4*>(drow + tq);
      const float4 v = *reinterpret_cast<const float4*>(prow + tq + 16);
      P[16] = v.x; P[17] = v.y; P[18] = v.z; P[19] = v.w;
      const float D[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = fmaf(D[i], P[i + k], acc[k]);
#pragma unroll
      for (int k = 0; k < 16; ++k) P[k] = P[k + 4];
    }
    float4* dst = reinterpret_cast<float4*>(wpart + ((size_t)b * 4 + q) * 4096 + (o * 16 + fd) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = make_float4(acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]);
  }
}
// depthwise backward pass A: per sample  R[fd][ch] = sum_t dd[fd][t] * c1[f][ch][t],  Sd[fd] = sum_t dd[fd][t].
// One wave per (filter f, electrode ch) row of c1: the row is read ONCE for both depth maps fd = 2f, 2f+1 (D == 2),
// 8 time steps per lane per trip (16-byte c1 loads, float4 gradient loads) when T % 8 == 0; the last FD wave tasks
// produce Sd.  Shuffle reduce, no workgroup barriers.  grid (B, ceil((F1*Ch + FD)/4)).
template <typename T>
__global__ __launch_bounds__(256) void k_eeg_dw_bwd_a(const T* __restrict__ c1, const float* __restrict__ dd, float* __restrict__ rpart, EegGeom g) {
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nrow = g.F1 * g.Ch, nout = g.FD * g.Ch + g.FD;
  const int task = blockIdx.y * 4 + wave;
  if (task < nrow) {
    const int f = task / g.Ch, ch = task % g.Ch;
    const float* d0 = dd + ((size_t)b * g.FD + 2 * f) * g.T;
    const float* d1 = d0 + g.T;
    const size_t cb = (((size_t)b * g.F1 + f) * g.Ch + ch) * g.T;
    float a0 = 0.f, a1 = 0.f;
    if ((g.T & 7) == 0) {
      for (int i = lane; i < g.T / 8; i += 64) {
        float c[8];
        ld8(c1, cb + 8 * (size_t)i, c);
        const float4 u0 = *reinterpret_cast<const float4*>(d0 + 8 * i), u1 = *reinterpret_cast<const float4*>(d0 + 8 * i + 4);
        const float4 v0 = *reinterpret_cast<const float4*>(d1 + 8 * i), v1 = *reinterpret_cast<const float4*>(d1 + 8 * i + 4);
        a0 += u0.x * c[0] + u0.y * c[1] + u0.z * c[2] + u0.w * c[3] + u1.x * c[4] + u1.y * c[5] + u1.z * c[6] + u1.w * c[7];
        a1 += v0.x * c[0] + v0.y * c[1] + v0.z * c[2] + v0.w * c[3] + v1.x * c[4] + v1.y * c[5] + v1.z * c[6] + v1.w * c[7];
      }
    } else {
      for (int tt = lane; tt < g.T; tt += 64) {
        const float c = ldf(c1, cb + tt);
        a0 = fmaf(d0[tt], c, a0); a1 = fmaf(d1[tt], c, a1);
      }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) {
      rpart[(size_t)b * nout + (2 * f) * g.Ch + ch] = a0;
      rpart[(size_t)b * nout + (2 * f + 1) * g.Ch + ch] = a1;
    }
  } else if (task < nrow + g.FD) {
    const int fd = task - nrow;
    const float* dr = dd + ((size_t)b * g.FD + fd) * g.T;
    float acc = 0.f;
    for (int tt = lane; tt < g.T; tt += 64) acc += dr[tt];
    acc = wave_sum(acc);
    if (lane == 0) rpart[(size_t)b * nout + g.FD * g.Ch + fd] = acc;
  }
}
// finalize of pass A: ddw, dgamma1, dbeta1 and the BN1-backward coefficients
__global__ void k_eeg_dw_bwd_finalize(const float* __restrict__ rpart, int B, EegGeom g, int training, const float* __restrict__ dw,
                                      const float* __restrict__ gamma1, const float* __restrict__ mean1, const float* __restrict__ inv1,
                                      const float* __restrict__ sc1, const float* __restrict__ sh1, float* __restrict__ ddw,
                                      float* __restrict__ dgamma1, float* __restrict__ dbeta1, float* __restrict__ coef) {
  __shared__ double R[EEG_MAXF * EEG_MAXCH + EEG_MAXF];
  const int nout = g.FD * g.Ch + g.FD;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    double s = 0;
    int bb = 0;
    for (; bb + 8 <= B; bb += 8) {             // eight independent loads in flight, added in sample order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rpart[(size_t)(bb + j) * nout + o];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; bb < B; ++bb) s += rpart[(size_t)bb * nout + o];
    R[o] = s;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < g.FD * g.Ch; o += blockDim.x) {
    const int fd = o / g.Ch, f = fd / g.D;
    if (ddw) ddw[o] = (float)(sc1[f] * R[o] + sh1[f] * R[g.FD * g.Ch + fd]);
  }
  if ((int)threadIdx.x < g.F1) {
    const int f = threadIdx.x;
    double db = 0, dg = 0;
    for (int dd = 0; dd < g.D; ++dd) {
      const int fd = f * g.D + dd;
      for (int ch = 0; ch < g.Ch; ++ch) {
        const double w = dw[fd * g.Ch + ch];
        db += w * R[g.FD * g.Ch + fd];
        dg += w * (R[fd * g.Ch + ch] - (double)mean1[f] * R[g.FD * g.Ch + fd]);
      }
    }
    dg *= inv1[f];
    if (dbeta1) dbeta1[f] = (float)db;
    if (dgamma1) dgamma1[f] = (float)dg;
    const double N = (double)g.B * g.Ch * g.T;
    coef[f] = gamma1[f] * inv1[f];
    coef[EEG_MAXF + f] = training ? (float)(db / N) : 0.f;
    coef[2 * EEG_MAXF + f] = training ? (float)(dg / N) : 0.f;
  }
}
// pass B: per (sample, electrode) row rebuild dc1[f][t] in LDS -- four filters at a time, so that three workgroups
// fit a CU -- then the temporal-conv weight gradient partial dW1[f][k] = sum_t dc1[f][t] x[t+k-padl] (thread = filter x
// 4 consecutive taps x one time quarter, sliding x window: 1 dc1 + 1 x LDS read per 4 FMAs) and, optionally,
// dx[t] = sum_{f,k} w1[f][k] dc1[f][t-k+padl].
#define EEG_DX_MAX 16          // dx values a thread can own: T <= 256*16
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__global__ __launch_bounds__(256) void k_eeg_conv1_bwd(const T* __restrict__ c1, const float* __restrict__ dd, const float* __restrict__ x,
    const float* __restrict__ dw, const float* __restrict__ w1, const float* __restrict__ mean1, const float* __restrict__ inv1,
    const float* __restrict__ coef, float* __restrict__ w1part, float* __restrict__ dx, EegGeom g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Tn = g.T, TX = (Tn + 2 * EEG_MAXK + 3) & ~3;       // rows stay 16-byte aligned for the float4 reads
  float* sdc = sm;                 // [4][TX]  dc1 of the current filter group, zero halo, index t + 64
  float* sxr = sm + 4 * TX;        // [TX]     x row with zero halo, index t + 64
  float* comb = sxr + TX;          // [256][4]
  const int row = blockIdx.x, b = row / g.Ch, ch = row % g.Ch;
  for (int i = threadIdx.x; i < 4 * TX; i += 256) sdc[i] = 0.f;
  {
    const float* xrow = x + (size_t)row * Tn;
    lds_fill<8>(sxr, TX, [&](int i) { const int t = i - EEG_MAXK; return (t >= 0 && t < Tn) ? xrow[t] : 0.f; });
  }
  float dxa[EEG_DX_MAX];
#pragma unroll
  for (int k = 0; k < EEG_DX_MAX; ++k) dxa[k] = 0.f;
  for (int fg = 0; fg < 2; ++fg) {
    __syncthreads();
    if ((Tn & 3) == 0) {
      // all four filters of the group per trip, four consecutive time steps per thread: 24 vector loads in flight, so the
      // group's rebuild is one memory round trip (a per-filter scalar loop exposes one per trip)
      float w0[4], w1d[4], mu[4], is[4], ca[4], k1[4], k2[4];
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4) {
        const int f = fg * 4 + f4;
        w0[f4] = dw[(f * 2) * g.Ch + ch]; w1d[f4] = dw[(f * 2 + 1) * g.Ch + ch];      // D == 2 (eeg_geom enforces it)
        mu[f4] = mean1[f]; is[f4] = inv1[f]; ca[f4] = coef[f]; k1[f4] = coef[EEG_MAXF + f]; k2[f4] = coef[2 * EEG_MAXF + f];
      }
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
              const float* d0 = dd + ((size_t)b * g.FD + f * 2) * Tn;
              a0[u][f4] = reinterpret_cast<const float4*>(d0)[q];
              a1[u][f4] = reinterpret_cast<const float4*>(d0 + Tn)[q];
              ld4(c1, (((size_t)b * g.F1 + f) * g.Ch + ch) * Tn + 4 * (size_t)q, cv[u][f4]);
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
              *reinterpret_cast<float4*>(sdc + f4 * TX + EEG_MAXK + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4) {
        const int f = fg * 4 + f4;
        const float w0 = dw[(f * 2) * g.Ch + ch], w1d = dw[(f * 2 + 1) * g.Ch + ch];
        const float mu = mean1[f], is = inv1[f], ca = coef[f], k1 = coef[EEG_MAXF + f], k2 = coef[2 * EEG_MAXF + f];
        const float* d0 = dd + ((size_t)b * g.FD + f * 2) * Tn;
        const float* d1 = d0 + Tn;
        const size_t cb = (((size_t)b * g.F1 + f) * g.Ch + ch) * Tn;
#pragma unroll 4
        for (int t = threadIdx.x; t < Tn; t += 256) {
          const float dbn = fmaf(w1d, d1[t], w0 * d0[t]);
          const float xh = (ldf(c1, cb + t) - mu) * is;
          sdc[f4 * TX + t + EEG_MAXK] = ca * (dbn - k1 - xh * k2);
        }
      }
    }
    __syncthreads();
    if (w1part) {
      // thread = (filter pair fp, tap octet k8, time part tp): 2 filters x 8 taps = 16 accumulators; per 4 time steps it
      // reads 2 dc1 float4 + 1 x float4 from LDS and issues 64 FMAs (the previous version was LDS-throughput bound).
      const int tp = threadIdx.x >> 4, fp = (threadIdx.x >> 3) & 1, k8 = threadIdx.x & 7, kb = k8 * 8;
      float acc[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
      if (kb < g.K1) {
        const int T4 = Tn & ~3;
        int tb = (int)((long long)(T4 / 4) * tp / 16) * 4, te = (int)((long long)(T4 / 4) * (tp + 1) / 16) * 4;
        if (tp == 15) te = Tn;
        const float* dc0 = sdc + (2 * fp) * TX + EEG_MAXK;
        const float* dc1 = dc0 + TX;
        const float* xr = sxr + EEG_MAXK + kb - g.padl1;         // xr[t + j] = x[t + kb + j - padl]
        float win[12];                                          // x window xr[t .. t+10]
        f32x2 acc2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[j] = (f32x2){0.f, 0.f};
        int t = tb;
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = xr[t + j];
        for (; t + 4 <= te; t += 4) {
          const float4 d0 = *reinterpret_cast<const float4*>(dc0 + t);
          const float4 d1 = *reinterpret_cast<const float4*>(dc1 + t);
          win[8] = xr[t + 8]; win[9] = xr[t + 9]; win[10] = xr[t + 10]; win[11] = xr[t + 11];
          // the two filters of the pair ride in one packed fp32 FMA (v_pk_fma_f32: twice the scalar FMA rate)
          const f32x2 dv[4] = {{d0.x, d1.x}, {d0.y, d1.y}, {d0.z, d1.z}, {d0.w, d1.w}};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const f32x2 xw = {win[i + j], win[i + j]};
              acc2[j] = dv[i] * xw + acc2[j];
            }
#pragma unroll
          for (int j = 0; j < 8; ++j) win[j] = win[j + 4];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[0][j] = acc2[j].x; acc[1][j] = acc2[j].y; }
        for (; t < te; ++t) {
          const float v0 = dc0[t], v1 = dc1[t];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float xv = xr[t + j]; acc[0][j] = fmaf(v0, xv, acc[0][j]); acc[1][j] = fmaf(v1, xv, acc[1][j]); }
        }
      }
      // combine the 16 time parts: comb[(tp*16 + fp*8 + k8)][16 values] -> thread o = (filter, tap) sums 16 entries
      __syncthreads();
      float* cb = comb;                                        // needs 256*16 floats (sized by the launcher)
#pragma unroll
      for (int j = 0; j < 8; ++j) { cb[threadIdx.x * 16 + j] = acc[0][j]; cb[threadIdx.x * 16 + 8 + j] = acc[1][j]; }
      __syncthreads();
      {
        const int o = threadIdx.x;                             // 4 filters x 64 taps = 256 outputs
        const int f4 = o >> 6, kk = o & 63;
        if (kk < g.K1) {
          const int fpo = f4 >> 1, fi = f4 & 1, k8o = kk >> 3, j = kk & 7;
          float s = 0.f;
          for (int p = 0; p < 16; ++p) s += cb[(p * 16 + fpo * 8 + k8o) * 16 + fi * 8 + j];
          w1part[(size_t)row * 8 * g.K1 + (fg * 4 + f4) * g.K1 + kk] = s;
        }
      }
    }
    if (dx) {
#pragma unroll
      for (int k = 0; k < EEG_DX_MAX; ++k) {
        const int t = threadIdx.x + k * 256;
        if (t < Tn) {
          float acc = dxa[k];
          for (int f4 = 0; f4 < 4; ++f4)
            for (int kk = 0; kk < g.K1; ++kk)
              acc = fmaf(w1[(fg * 4 + f4) * g.K1 + kk], sdc[f4 * TX + EEG_MAXK + t - kk + g.padl1], acc);
          dxa[k] = acc;
        }
      }
    }
  }
  if (dx) {
#pragma unroll
    for (int k = 0; k < EEG_DX_MAX; ++k) {
      const int t = threadIdx.x + k * 256;
      if (t < Tn) dx[(size_t)row * Tn + t] = dxa[k];
    }
  }
}

extern "C" int bx_eeg_features_bwd(const bxEegDesc* d, const bxEegParams* p, const float* x, const float* dfeat,
                                   const uint64_t* seed, const void* saved, const bxEegGrads* gr, float* dx,
                                   void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(d && p && x && dfeat && saved, "bx_eeg_features_bwd: null pointer");
  BX_DTYPE_OK(d->dtype);
  EegGeom g;
  if (!eeg_tuned(d, &g))
    return bx_eegg_backward(d, p, x, dfeat, seed, saved, gr, dx, workspace, workspace_bytes, (hipStream_t)stream);
  const EegWs w = eeg_ws(g);
  if (!workspace || workspace_bytes < w.total) BX_FAIL(BX_EWORKSPACE, "bx_eeg_features_bwd: workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  const EegStats st = eeg_stats(g, (void*)saved);
  const void* c1 = (const char*)saved + g.off_c1;
  const float* dmap = (const float*)((const char*)saved + g.off_d);
  const float* p1 = (const float*)((const char*)saved + g.off_p1);
  const float* smap = (const float*)((const char*)saved + g.off_s);
  char* W = (char*)workspace;
  float* part = (float*)(W + w.off_part);
  float* du3 = (float*)(W + w.off_du3);
  float* dp1 = (float*)(W + w.off_dp1);
  float* du2 = (float*)(W + w.off_du2);
  float* rpart = (float*)(W + w.off_r);
  float* w1part = (float*)(W + w.off_w1p);
  float* sepp = (float*)(W + w.off_sepp);
  float* coef3 = (float*)(W + w.off_coef);
  float* coef2 = coef3 + 3 * EEG_MAXF;
  float* coef1 = coef2 + 3 * EEG_MAXF;
  const int tr = d->training;
  const float pdrop = tr ? d->dropout_p : 0.f;
  const float pdrop2 = tr ? (d->dropout_p2 >= 0.f ? d->dropout_p2 : d->dropout_p) : 0.f;
  bxEegGrads none = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!gr) gr = &none;

  // pool2/dropout/ELU/BN3
  hipLaunchKernelGGL(k_eeg_act_bwd, dim3(g.B, g.F2), dim3(256), 0, s, dfeat, smap, st.mean3, st.inv3, st.sc3, st.sh3, du3, part, g.F2, g.T1, g.T2, g.P2,
                     seed, pdrop2, d->salt + 1);
  BX_CHECK_LAUNCH("eeg act3 bwd");
  hipLaunchKernelGGL(k_eeg_bn_bwd_finalize, dim3(1), dim3(1024), 0, s, part, g.B, (double)g.B * g.T1, g.F2, tr, p->bn3_w, st.inv3, coef3, gr->bn3_w, gr->bn3_b,
                     (const float*)nullptr, (float*)nullptr, 0, 0, 4);
  BX_CHECK_LAUNCH("eeg bn3 bwd finalize");
  // separable conv (the BatchNorm3 backward apply rides in its staging)
  {
    // the two halves of the kernel in separate workgroups while that is what gives every CU two (BX_EEG_SEPB_SPLIT=0|1 overrides);
    // B = 64: 27.0 -> 24.0 us
    static const int split_env = getenv("BX_EEG_SEPB_SPLIT") ? atoi(getenv("BX_EEG_SEPB_SPLIT")) : -1;
    const bool split = split_env >= 0 ? split_env != 0 : g.B * 4 < 1024;
    const size_t tile = (size_t)16 * eeg_sepb_pitch(g.T1);
    const size_t lds = (split ? (tile + (tile > 4096 ? tile : 4096)) : (2 * tile + 4096)) * sizeof(float);
    BX_REQUIRE(lds <= 160 * 1024, "bx_eeg_features_bwd: T/P1 too long for the LDS tile (%zu bytes)", lds);
    if (hipFuncSetAttribute((const void*)k_eeg_sep_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      BX_FAIL(BX_EHIP, "bx_eeg_features_bwd: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(k_eeg_sep_bwd, dim3(g.B, split ? 8 : 4), dim3(256), lds, s, du3, p1, p->sep_w, dp1, sepp, g, smap, st.mean3, st.inv3, (const float*)coef3);
    BX_CHECK_LAUNCH("eeg sep bwd");
  }
  // pool1/dropout/ELU/BN2
  hipLaunchKernelGGL(k_eeg_act_bwd, dim3(g.B, g.FD), dim3(256), 0, s, dp1, dmap, st.mean2, st.inv2, st.sc2, st.sh2, du2, part, g.FD, g.T, g.T1, g.P1,
                     seed, pdrop, d->salt);
  BX_CHECK_LAUNCH("eeg act2 bwd");
  {
    // BN2 finalize; the separable convolution's weight-gradient partials (finished two launches ago) are summed by extra workgroups
    const int S2 = bx_partial_slices(4096, g.B * 4);
    const int nsum = gr->sep_w ? bx_ceil_div(4096, 4 * (256 / S2)) : 0;
    hipLaunchKernelGGL(k_eeg_bn_bwd_finalize, dim3(1 + nsum), dim3(1024), 0, s, part, g.B, (double)g.B * g.T, g.FD, tr, p->bn2_w, st.inv2, coef2, gr->bn2_w,
                       gr->bn2_b, (const float*)sepp, gr->sep_w, g.B * 4, 4096, S2);
  }
  BX_CHECK_LAUNCH("eeg bn2 bwd finalize");
  void* g16 = W + w.off_du3;                              // collapsed front end: dL/du as bf16 over the (dead by now) du3 | dp1 regions
  if (eeg_collapsed(d)) {
    const long long n8 = (long long)g.B * g.FD * g.T / 8;
    hipLaunchKernelGGL(k_eeg_bn_bwd_apply16, dim3(bx_ceil_div(n8, 256)), dim3(256), 0, s, du2, dmap, st.mean2, st.inv2, coef2, (bf16_t*)g16, g.FD, g.T, n8);
    BX_CHECK_LAUNCH("eeg bn2 bwd apply (bf16)");
  } else {
    const long long n = (long long)g.B * g.FD * g.T;
    hipLaunchKernelGGL(k_eeg_bn_bwd_apply, dim3(bx_ceil_div(n, 256) > 4096 ? 4096 : bx_ceil_div(n, 256)), dim3(256), 0, s, du2, dmap, st.mean2, st.inv2,
                       coef2, g.FD, g.T, n);
    BX_CHECK_LAUNCH("eeg bn2 bwd apply");
  }
  if (eeg_collapsed(d)) {
    // gradients of conv1 / BatchNorm1 / depthwise from C = g (*) x, G = sum g and the saved (R, S): eeg_collapse.hip
    BX_REQUIRE(!dx, "bx_eeg_features_bwd: the collapsed front end (bxEegDesc.collapse) has no input gradient; clear the flag when x needs one");
    static const int want_split = getenv("BX_EEGC_NSPLIT") ? atoi(getenv("BX_EEGC_NSPLIT")) : EEGC_MAX_SPLIT;
    int nsplit = want_split < 1 ? 1 : want_split > EEGC_MAX_SPLIT ? EEGC_MAX_SPLIT : want_split;
    if (nsplit > g.B) nsplit = g.B;
    float* cpart = w1part;
    float* ep = rpart;
    float* gpart = rpart + (size_t)g.Ch * 520;
    int rc = bx_eegc_corr_launch(g16, x, cpart, gpart, g.B, g.Ch, g.T, nsplit, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_bwd: correlation launch failed (code %d)", rc);
    rc = bx_eegc_grads(cpart, gpart, nsplit, p->conv1_w, p->dw_w, st.mean1, st.inv1, st.sc1, st.sh1, (const double*)c1, ep, gr->dw_w, gr->bn1_w, gr->bn1_b,
                       gr->conv1_w, g.B, g.Ch, g.T, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_bwd: collapsed gradient launch failed (code %d)", rc);
    return BX_OK;
  }
  const bool front_params = gr->conv1_w || gr->dw_w || gr->bn1_w || gr->bn1_b;
  if (eeg_collapsed_eval(d)) {
    BX_REQUIRE(!front_params, "bx_eeg_features_bwd: the forward pass ran the collapsed evaluation-mode front end (bxEegDesc.collapse), which keeps no "
                              "conv1 output; clear the flag when conv1 / batchnorm1 / depthwise gradients are wanted");
    if (dx) {
      const int rc = bx_eegc_dx_launch(du2, p->conv1_w, p->dw_w, st.sc1, dx, g.B, g.Ch, g.T, s);
      BX_REQUIRE(rc == 0, "bx_eeg_features_bwd: collapsed input-gradient launch failed (code %d)", rc);
    }
    return BX_OK;
  }
  if (dx && !tr && !front_params && g.K1 == 64 && !getenv("BX_EEG_DX_LAYERED")) {
    // attribution pass (evaluation mode, only the input gradient wanted): the front end's adjoint in collapsed form, eeg_collapse.hip
    const int rc = bx_eegc_dx_launch(du2, p->conv1_w, p->dw_w, st.sc1, dx, g.B, g.Ch, g.T, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_bwd: collapsed input-gradient launch failed (code %d)", rc);
    return BX_OK;
  }
  // depthwise + BN1
  BX_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((k_eeg_dw_bwd_a<T>), dim3(g.B, (g.F1 * g.Ch + g.FD + 3) / 4), dim3(256), 0, s, (const T*)c1, du2, rpart, g));
  BX_CHECK_LAUNCH("eeg dw bwd A");
  hipLaunchKernelGGL(k_eeg_dw_bwd_finalize, dim3(1), dim3(512), 0, s, rpart, g.B, g, tr, p->dw_w, p->bn1_w, st.mean1, st.inv1, st.sc1, st.sh1,
                     gr->dw_w, gr->bn1_w, gr->bn1_b, coef1);
  BX_CHECK_LAUNCH("eeg dw bwd finalize");
  const bool no_mfma = getenv("BX_EEG_NO_MFMA") != nullptr;
  if (gr->conv1_w && !dx && d->dtype == BX_BF16 && g.K1 == 64 && !no_mfma && bx_eeg_conv1_wgrad_mfma_lds(g.T) <= 150 * 1024) {
    // bf16 storage, no input gradient wanted (training): the weight gradient runs on the matrix cores (eeg_mfma.hip)
    const int rc = bx_eeg_conv1_wgrad_mfma_launch(c1, du2, x, p->dw_w, st.mean1, st.inv1, coef1, w1part, g.B, g.Ch, g.T, g.FD, EEG_MAXF, s);
    BX_REQUIRE(rc == 0, "bx_eeg_features_bwd: MFMA weight-gradient launch failed (code %d)", rc);
    BX_CHECK_LAUNCH("eeg conv1 wgrad (mfma)");
    BX_SUM_PARTIALS(w1part, gr->conv1_w, g.B * g.Ch, 8 * g.K1, s);
    BX_CHECK_LAUNCH("eeg conv1 wgrad reduce");
  } else if (gr->conv1_w || dx) {
    const size_t lds = ((size_t)5 * ((g.T + 2 * EEG_MAXK + 3) & ~3) + 4096) * sizeof(float);
    BX_REQUIRE(lds <= 160 * 1024 && g.T <= 256 * EEG_DX_MAX, "bx_eeg_features_bwd: T too long (LDS tile %zu bytes, T <= %d)", lds, 256 * EEG_DX_MAX);
    BX_DISPATCH_DTYPE(d->dtype, T,
      if (hipFuncSetAttribute((const void*)k_eeg_conv1_bwd<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        BX_FAIL(BX_EHIP, "bx_eeg_features_bwd: cannot reserve %zu bytes of LDS", lds);
      hipLaunchKernelGGL((k_eeg_conv1_bwd<T>), dim3(g.B * g.Ch), dim3(256), lds, s, (const T*)c1, du2, x, p->dw_w, p->conv1_w, st.mean1, st.inv1,
                         coef1, gr->conv1_w ? w1part : (float*)nullptr, dx, g));
    BX_CHECK_LAUNCH("eeg conv1 bwd");
    if (gr->conv1_w) {
      BX_SUM_PARTIALS(w1part, gr->conv1_w, g.B * g.Ch, 8 * g.K1, s);
      BX_CHECK_LAUNCH("eeg conv1 wgrad reduce");
    }
  }
  return BX_OK;
}
