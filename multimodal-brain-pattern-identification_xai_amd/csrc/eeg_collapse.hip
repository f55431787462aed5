// EEGNet front end without its big tensor (training, bf16 storage mode): conv1 -> BatchNorm1 -> depthwise electrode mix are all
// linear (reference root/src/models/models.py:250-253,272-274: no bias, no activation between them), so
//   u[b,fd,t] = a_f * sum_k w1[f,k] v~[b,fd,t+k-31] + c_f * Wsum[fd],      v[b,fd,t] = sum_ch wd[fd,ch] x[b,ch,t]   (f = fd / D)
// with a_f = gamma_f * invstd_f, c_f = beta_f - a_f * mu_f: the electrodes are mixed FIRST and the 64-tap convolution runs on
// 16 rows per sample instead of 8 * Chans; the conv1 output [B,8,Chans,T] (39 MB per step at the benchmark shape, written once
// and read three times) is never formed.  BatchNorm1's batch statistics come from the input's sufficient statistics
//   S[k] = sum_{b,ch,t} x~[t+k-31],   R[k,k'] = sum_{b,ch,t} x~[t+k-31] x~[t+k'-31]       (x~ = x with the conv's zero padding)
//   mu_f = w1[f].S / N,   E[z^2]_f = w1[f]^T R w1[f] / N
// and R is assembled from the 64 autocorrelation lags r[d] = sum x[s] x[s+d] minus head / tail partial sums (the windows that
// hang over the row's ends).  The backward pass needs, besides (R, S), only
//   C[fd,ch,k] = sum_{b,t} g[b,fd,t] x~[b,ch,t+k-31],   G[fd] = sum_{b,t} g[b,fd,t]       (g = dL/du)
// from which d depthwise.weight, d bn1.weight, d bn1.bias and d conv1.weight are closed forms (k_eegc_grads_ch / _final).
// The algebra is checked in fp64 against autograd through the reference sequence by tools/eeg_collapse_proto.py (1e-15).
#include "bx_common.h"

#define EC_K 64
#define EEGC_MAXS 32              // EEGC_MAX_SPLIT of eeg.hip
#define EEG_MAXCH_C 64            // EEG_MAXCH of eeg.hip
#define EC_PADL 31
#define EC_EDGE 96                 // samples of a row's head / tail that the edge terms touch (31 + 63 + 1, rounded up)
#define EC_NR 608                  // workgroups of the lag-sum role (two rows each at the benchmark shape: several resident per CU hide the row loads)
#define EC_NE 32                   // workgroups of the edge role
#define EC_RCOLS 65                // r[0..63], sum x
#define EC_ECOLS (66 * EC_EDGE)    // E_head[32][96], E_tail[32][96] (products of edge samples), column sums of head and tail [96] each

// ---- forward statistics --------------------------------------------------------------------------------------------
// role r (blockIdx.x < EC_NR): rows w, w + EC_NR, ...; a thread takes 4 consecutive samples and keeps the 67-sample window in
//   registers: 4 x 64 products per 17 aligned 16-byte LDS reads.  One partial row [65] per workgroup.
// role e: rows e, e + EC_NE, ...; wave rg takes every 4th of them; see below.
template <int ROLE>      // 0: both roles in one launch; 1 / 2: one role only (profiling: BX_EEGC_SPLIT_STATS=1)
__global__ __launch_bounds__(256) void k_eegc_stats(const float* __restrict__ x, float* __restrict__ rpart, float* __restrict__ epart, int nrows, int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int blk = ROLE == 2 ? (int)blockIdx.x + EC_NR : (int)blockIdx.x;
  if (blk < EC_NR) {
    // lag sums straight from global memory: a thread takes 4 consecutive samples and requests its 68-sample window as 17
    // 16-byte buffer loads, all in flight together (past the row: zeros); neighbouring threads' windows overlap in L1.  No LDS
    // staging, no barrier per row.  4 x 64 products per window.
    float acc[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    float xsum = 0.f;
    const int nq = (T + 3) / 4;                         // quads per row
    for (int row = blk; row < nrows; row += EC_NR) {
      const __amdgpu_buffer_rsrc_t xres = bx_rsrc(x + (size_t)row * T, (uint32_t)T * 4u);
      for (int q0 = tid; q0 < nq; q0 += 256) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        float w[68];
#pragma unroll
        for (int q = 0; q < 17; ++q) {
          const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xres, (uint32_t)(4 * q0 + 4 * q) * 4u, 0, 0));
          w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
        }
        xsum += (w[0] + w[1]) + (w[2] + w[3]);
#pragma unroll
        for (int d = 0; d < 64; ++d) acc[d] += w[0] * w[d] + w[1] * w[d + 1] + w[2] * w[d + 2] + w[3] * w[d + 3];
      }
    }
    // workgroup sum per lag (fixed order): 16-lane rows by DPP, then the 16 row totals (4 rows x 4 waves) through LDS
    float* red = sm;                                    // [16][65]
#pragma unroll
    for (int d = 0; d < 64; ++d) {
      float v = acc[d];
      v += bx_dpp<0xB1>(v); v += bx_dpp<0x4E>(v); v += bx_dpp<0x124>(v); v += bx_dpp<0x128>(v);
      if ((lane & 15) == 0) red[(wave * 4 + (lane >> 4)) * EC_RCOLS + d] = v;
    }
    {
      float v = xsum;
      v += bx_dpp<0xB1>(v); v += bx_dpp<0x4E>(v); v += bx_dpp<0x124>(v); v += bx_dpp<0x128>(v);
      if ((lane & 15) == 0) red[(wave * 4 + (lane >> 4)) * EC_RCOLS + 64] = v;
    }
    __syncthreads();
    if (tid < EC_RCOLS) {
      float t_ = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t_ += red[r * EC_RCOLS + tid];
      rpart[(size_t)blk * EC_RCOLS + tid] = t_;
    }
    return;
  }
  // edge role: E_head[s][c] = sum_rows x[s] x[c] (s < 32, c < 96) and the same for the reversed tail (xr[i] = x[T-1-i]).  A lane owns
  // columns c = lane and lane + 64; x[s] of the row comes from lane s through a readlane broadcast: no LDS, no dependent loads
  // (the next row's four values are requested before this row's 128 FMAs).  The finalize turns E into the prefix sums it needs.
  const int e = blk - EC_NR;
  float h0[32], h1[32], t0[32], t1[32];
#pragma unroll
  for (int m = 0; m < 32; ++m) h0[m] = h1[m] = t0[m] = t1[m] = 0.f;
  float sh0 = 0.f, sh1 = 0.f, st0 = 0.f, st1 = 0.f;
  const int c1 = lane + 64 < EC_EDGE ? lane + 64 : EC_EDGE - 1;           // clamped: lanes >= 32 of the second column are unused
  int row = e + EC_NE * wave;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (row < nrows) {
    const float* xr = x + (size_t)row * T;
    a0 = xr[lane]; a1 = xr[c1]; b0 = xr[T - 1 - lane]; b1 = xr[T - 1 - c1];
  }
  for (; row < nrows; row += 4 * EC_NE) {
    const float x0 = a0, x1 = a1, y0 = b0, y1 = b1;
    const int nxt = row + 4 * EC_NE;
    if (nxt < nrows) {
      const float* xr = x + (size_t)nxt * T;
      a0 = xr[lane]; a1 = xr[c1]; b0 = xr[T - 1 - lane]; b1 = xr[T - 1 - c1];
    }
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const float bh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), m));
      const float bt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y0), m));
      h0[m] = fmaf(bh, x0, h0[m]); h1[m] = fmaf(bh, x1, h1[m]);
      t0[m] = fmaf(bt, y0, t0[m]); t1[m] = fmaf(bt, y1, t1[m]);
    }
    sh0 += x0; sh1 += x1; st0 += y0; st1 += y1;
  }
  // the four waves' sums, in wave order: head half, then tail half, through a [4][33 * 96] buffer (the merged launch sizes every
  // workgroup's LDS by this role).  epart columns: [head E 32 x 96][tail E 32 x 96][head column sums 96][tail column sums 96]
  constexpr int HALF = 33 * EC_EDGE;
  float* red = sm;                                      // [4][HALF]
  float* mine = red + (size_t)wave * HALF;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      mine[m * EC_EDGE + lane] = half ? t0[m] : h0[m];
      if (lane < 32) mine[m * EC_EDGE + 64 + lane] = half ? t1[m] : h1[m];
    }
    mine[32 * EC_EDGE + lane] = half ? st0 : sh0;
    if (lane < 32) mine[32 * EC_EDGE + 64 + lane] = half ? st1 : sh1;
    __syncthreads();
    for (int i = tid; i < HALF; i += 256) {
      const float v = (red[i] + red[HALF + i]) + (red[2 * HALF + i] + red[3 * HALF + i]);
      const int col = i < 32 * EC_EDGE ? half * 32 * EC_EDGE + i : (64 + half) * EC_EDGE + (i - 32 * EC_EDGE);
      epart[(size_t)e * EC_ECOLS + col] = v;
    }
  }
}

// ---- BatchNorm1 finalize from the sufficient statistics (one workgroup, 1024 threads) --------------------------------
// RS (saved for backward): R[64][64] doubles, then S[64].
__global__ __launch_bounds__(1024) void k_eegc_finalize1(const float* __restrict__ rpart, const float* __restrict__ epart, const float* __restrict__ w1,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
    float momentum, float eps, double count, float* __restrict__ mean1, float* __restrict__ inv1, float* __restrict__ sc1, float* __restrict__ sh1,
    double* __restrict__ RS) {
  __shared__ double sR[64 * 64];
  __shared__ double sS[64];
  __shared__ double tot_r[EC_RCOLS];
  __shared__ float tot_e[EC_ECOLS];
  double (*slice)[EC_RCOLS] = reinterpret_cast<double (*)[EC_RCOLS]>(sR);      // [15][65]: consumed (-> tot_r) before R is assembled in sR
  __shared__ double qf[8][2][2];
  __shared__ double Hh[64][33], Ht[64][33], hx[33], tx[33];
  __shared__ float sw[8 * EC_K];
  const int tid = threadIdx.x;
  {                                                     // lag sums: 15 slices of the partial rows, combined in slice order
    const int col = tid % EC_RCOLS, sl = tid / EC_RCOLS;
    constexpr int PER = (EC_NR + 14) / 15;
    if (sl < 15) {
      double a = 0.0;
      for (int r0 = 0; r0 < PER; r0 += 16) {            // sixteen independent loads per trip
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int r = sl * PER + r0 + u;
          v[u] = rpart[(size_t)(r < EC_NR ? r : EC_NR - 1) * EC_RCOLS + col];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) a += (r0 + u < PER && sl * PER + r0 + u < EC_NR) ? (double)v[u] : 0.0;
      }
      slice[sl][col] = a;
    }
  }
  for (int i = tid; i < EC_ECOLS; i += 2048) {          // edge sums, two columns (2 x EC_NE loads) per trip; float is enough (~2 % of R)
    float v[2][EC_NE];
    const int i1 = i + 1024 < EC_ECOLS ? i + 1024 : i;
#pragma unroll
    for (int r = 0; r < EC_NE; ++r) { v[0][r] = epart[(size_t)r * EC_ECOLS + i]; v[1][r] = epart[(size_t)r * EC_ECOLS + i1]; }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int r = 0; r < EC_NE; ++r) { a += (double)v[0][r]; b += (double)v[1][r]; }
    tot_e[i] = (float)a;
    if (i + 1024 < EC_ECOLS) tot_e[i + 1024] = (float)b;
  }
  __syncthreads();
  if (tid < EC_RCOLS) {
    double a = 0.0;
    for (int sl = 0; sl < 15; ++sl) a += slice[sl][tid];
    tot_r[tid] = a;
  }
  __syncthreads();
  // prefix sums along the diagonals of E (one thread per lag and end): Hh[d][m] = sum_{s<m} x[s] x[s+d],  Ht[d][j] = the last j
  // products of lag d; hx / tx likewise from the column sums.  (Summed per entry of R instead, a thread walked up to 62 dependent
  // LDS reads for each of its four entries.)
  const float* Eh = tot_e;                              // E_head[s][c] = sum_rows x[s] x[c]
  const float* Et = tot_e + 32 * EC_EDGE;               // E_tail[i][c] = sum_rows x[T-1-i] x[T-1-c]
  if (tid < 128) {
    const int d = tid & 63;
    const float* E = tid < 64 ? Eh : Et;
    double* P = tid < 64 ? Hh[d] : Ht[d];
    double run = 0.0;
    P[0] = 0.0;
#pragma unroll 8
    for (int s_ = 0; s_ < 32; ++s_) { run += (double)E[s_ * EC_EDGE + s_ + d]; P[s_ + 1] = run; }
  } else if (tid < 130) {
    const float* cs = tot_e + (64 + (tid - 128)) * EC_EDGE;
    double* P = tid == 128 ? hx : tx;
    double run = 0.0;
    P[0] = 0.0;
    for (int s_ = 0; s_ < 32; ++s_) { run += (double)cs[s_]; P[s_ + 1] = run; }
  }
  for (int i = tid; i < 8 * EC_K; i += 1024) sw[i] = w1[i];
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 1024) {           // R[k][k'] = r[d] - [m > 0] Hh[d][m] - [m' < 0] Ht[d][-m']   (k <= k', d = k' - k)
    int k = i >> 6, k2 = i & 63;
    if (k > k2) { const int t_ = k; k = k2; k2 = t_; }
    const int m = k - EC_PADL, m2 = k2 - EC_PADL, d = k2 - k;
    double v = tot_r[d];
    if (m > 0) v -= Hh[d][m];
    if (m2 < 0) v -= Ht[d][-m2];
    sR[i] = v;
    RS[i] = v;
  }
  if (tid < 64) {                                       // S[k] = sum x - [m > 0] sum_{s<m} x[s] - [m < 0] sum_{s >= T+m} x[s]
    const int m = tid - EC_PADL;
    double v = tot_r[64];
    if (m > 0) v -= hx[m];
    if (m < 0) v -= tx[-m];
    sS[tid] = v;
    RS[4096 + tid] = v;
  }
  __syncthreads();
  {                                                     // per filter: w.S and w^T R w, 128 threads each
    const int f = tid >> 7, t = tid & 127;
    const float* w = sw + f * EC_K;
    double q = 0.0, l = 0.0;
    for (int i = t; i < 64 * 64; i += 128) q += (double)w[i >> 6] * (double)w[i & 63] * sR[i];
    if (t < 64) l = (double)w[t] * sS[t];
    for (int o = 32; o > 0; o >>= 1) { q += __shfl_xor(q, o, 64); l += __shfl_xor(l, o, 64); }
    if ((tid & 63) == 0) { qf[f][t >> 6][0] = q; qf[f][t >> 6][1] = l; }
  }
  __syncthreads();
  if (tid < 8) {
    const int f = tid;
    const double m = (qf[f][0][1] + qf[f][1][1]) / count;
    double var = (qf[f][0][0] + qf[f][1][0]) / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[f] = (1.f - momentum) * rmean[f] + momentum * mean;
    rvar[f] = (1.f - momentum) * rvar[f] + momentum * (float)unbiased;
    if (f == 0 && nbt) nbt[0] += 1;
    const float sc = gamma[f] * invstd;
    sc1[f] = sc; sh1[f] = beta[f] - mean * sc; mean1[f] = mean; inv1[f] = invstd;
  }
}

// ---- forward: mix the electrodes, convolve the 16 mixed rows, apply BatchNorm1 -------------------------------------------
// grid (ceil(T / 256), B).  Phase 1: v[fd][tau] for the chunk + 63 halo samples into LDS (thread per tau, the 16 x Chans mix
// weights broadcast from LDS).  Phase 2: wave w owns rows fd = 4w .. 4w+3, a lane 4 consecutive time steps: the filter's 64
// weights and the 67-sample window live in registers (17 aligned 16-byte LDS reads per 256 FMAs).  u is written as fp32
// [B][16][T] (the tensor the depthwise kernel used to write) together with this workgroup's BatchNorm2 partial sums.
#define EC_TC 256
#define EC_VP (EC_TC + 64)        // row pitch of v in LDS (multiple of 4)
__global__ __launch_bounds__(256) void k_eegc_fwd(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ wd,
    const float* __restrict__ sc1, const float* __restrict__ sh1, float* __restrict__ u, float* __restrict__ partials, int Ch, int T, int want_stats,
    const float* const* xslot = nullptr) {
  if (xslot) x = *xslot;                                   // bxEegDesc.x_slot: the input's address comes from a device slot
  __shared__ __attribute__((aligned(16))) float sv[16 * EC_VP];
  __shared__ __attribute__((aligned(16))) float swt[EEG_MAXCH_C * 16];   // mix weights transposed: [ch][fd] (four 16-byte broadcast reads per electrode)
  __shared__ __attribute__((aligned(16))) float sw1[8 * EC_K];
  __shared__ float swsum[16];
  __shared__ float red[4][2][4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, t0 = blockIdx.x * EC_TC;
  // (weights through LDS as 16-byte broadcast reads: as uniform global loads they became 777 serialised s_load_dword with SGPR spills)
  for (int i = tid; i < 16 * Ch; i += 256) { const int fd = i / Ch, ch = i - fd * Ch; swt[ch * 16 + fd] = wd[i]; }
  for (int i = tid; i < 8 * EC_K; i += 256) sw1[i] = w1[i];
  const float* xb = x + (size_t)b * Ch * T;
  // v[fd][tau] = sum_ch wd[fd][ch] x[ch][t0 - 31 + tau] for tau = tid and (first wave) 256 + tid: every electrode's sample of both
  // columns is requested before the first FMA (a rolled loop over electrodes is one memory round trip per electrode)
  const int nt = tid < EC_VP - 256 ? 2 : 1;
  float acc[2][16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int fd = 0; fd < 16; ++fd) acc[q][fd] = 0.f;
  for (int ch0 = 0; ch0 < Ch; ch0 += 24) {
    float xv[2][24];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = t0 - EC_PADL + tid + 256 * q;
      const bool ok = q < nt && t >= 0 && t < T;
#pragma unroll
      for (int c = 0; c < 24; ++c) {
        const int chc = ch0 + c < Ch ? ch0 + c : Ch - 1;
        xv[q][c] = (q == 0 || tid < 64) ? xb[(size_t)chc * T + (ok ? t : 0)] : 0.f;
        if (!ok || ch0 + c >= Ch) xv[q][c] = 0.f;
      }
    }
    __syncthreads();                                    // swt is complete (first trip) -- uniform trip count
#pragma unroll
    for (int c = 0; c < 24; ++c) {
      const int chc = ch0 + c < Ch ? ch0 + c : Ch - 1;
      float w[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 v = *reinterpret_cast<const float4*>(swt + chc * 16 + 4 * q4);
        w[4 * q4] = v.x; w[4 * q4 + 1] = v.y; w[4 * q4 + 2] = v.z; w[4 * q4 + 3] = v.w;
      }
#pragma unroll
      for (int fd = 0; fd < 16; ++fd) { acc[0][fd] = fmaf(w[fd], xv[0][c], acc[0][fd]); acc[1][fd] = fmaf(w[fd], xv[1][c], acc[1][fd]); }
    }
  }
#pragma unroll
  for (int fd = 0; fd < 16; ++fd) {
    sv[fd * EC_VP + tid] = acc[0][fd];
    if (nt == 2) sv[fd * EC_VP + 256 + tid] = acc[1][fd];
  }
  if (tid < 16) {                                       // Wsum[fd]
    float a = 0.f;
    for (int ch = 0; ch < Ch; ++ch) a += swt[ch * 16 + tid];
    swsum[tid] = a;
  }
  __syncthreads();
  float s1[4], s2[4];
  const int t = t0 + 4 * lane;
#pragma unroll
  for (int half = 0; half < 2; ++half) {                // the wave's two filters (rows 4w, 4w+1 | 4w+2, 4w+3)
    const int f = 2 * wave + half;
    float w[64];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const float4 v = *reinterpret_cast<const float4*>(sw1 + f * EC_K + 4 * k4);
      w[4 * k4] = v.x; w[4 * k4 + 1] = v.y; w[4 * k4 + 2] = v.z; w[4 * k4 + 3] = v.w;
    }
    const float a = sc1[f], c = sh1[f];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const int fd = 2 * f + dd, r = 2 * half + dd;
      float win[68];
#pragma unroll
      for (int q = 0; q < 17; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(sv + fd * EC_VP + 4 * lane + 4 * q);
        win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        o[0] = fmaf(w[k], win[k], o[0]); o[1] = fmaf(w[k], win[k + 1], o[1]);
        o[2] = fmaf(w[k], win[k + 2], o[2]); o[3] = fmaf(w[k], win[k + 3], o[3]);
      }
      const float cc = c * swsum[fd];
      s1[r] = s2[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = fmaf(a, o[j], cc);
        if (t + j < T) { s1[r] += o[j]; s2[r] += o[j] * o[j]; }
      }
      float* dst = u + ((size_t)b * 16 + fd) * T + t;
      if (t + 3 < T && (T & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (t + j < T) dst[j] = o[j];
      }
    }
  }
  if (!want_stats) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = wave_sum(s1[r]), q = wave_sum(s2[r]);
    if (lane == 0) { red[wave][0][r] = a; red[wave][1][r] = q; }
  }
  __syncthreads();
  if (tid < 32) {                                       // partial layout [row][2][16] (k_bn_finalize), row = this workgroup
    const int v = tid >> 4, fd = tid & 15;
    partials[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + v) * 16 + fd] = red[fd >> 2][v][fd & 3];
  }
}

// ---- backward: per-electrode closed forms ----------------------------------------------------------------------------------
// grid = Chans.  Sums the nsplit partial slabs C[fd][k] of its electrode, then
//   zg[fd] = sum_k w1[f][k] C[fd][k]                      (= sum_{b,t} g z of this electrode)
//   d wd[fd][ch] = a_f zg[fd] + c_f G[fd]
//   per-electrode contributions  Q[f] += wd[fd][ch] zg[fd],   dyx[f][k] += wd[fd][ch] C[fd][k]     -> ep[ch][8 + 8*64]
__global__ __launch_bounds__(256) void k_eegc_grads_ch(const float* __restrict__ cpart, const float* __restrict__ gpart, int nsplit,
    const float* __restrict__ w1, const float* __restrict__ wd, const float* __restrict__ sc1, const float* __restrict__ sh1,
    float* __restrict__ d_wd, float* __restrict__ ep, int Ch) {
  __shared__ float sC[16 * 64];
  __shared__ float sG[16];
  __shared__ float szg[16];
  const int tid = threadIdx.x, ch = blockIdx.x;
  {                                                     // all of a thread's slab loads are issued before the first add (sum stays in split order)
    float v[4][EEGC_MAXS];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int sp = 0; sp < EEGC_MAXS; ++sp) v[q][sp] = cpart[((size_t)ch * nsplit + (sp < nsplit ? sp : nsplit - 1)) * 1024 + tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = 0.f;
#pragma unroll
      for (int sp = 0; sp < EEGC_MAXS; ++sp) a += sp < nsplit ? v[q][sp] : 0.f;
      sC[tid + 256 * q] = a;
    }
  }
  if (tid < 16) {                                       // nsplit <= EEGC_MAXS loads in flight together, summed in split order
    float v[EEGC_MAXS];
#pragma unroll
    for (int sp = 0; sp < EEGC_MAXS; ++sp) v[sp] = gpart[(sp < nsplit ? sp : nsplit - 1) * 16 + tid];
    float a = 0.f;
#pragma unroll
    for (int sp = 0; sp < EEGC_MAXS; ++sp) a += sp < nsplit ? v[sp] : 0.f;
    sG[tid] = a;
  }
  __syncthreads();
  {                                                     // zg: 16 rows x 64 taps, 16 threads per row
    const int fd = tid >> 4, l = tid & 15;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) a += w1[(fd >> 1) * EC_K + l + 16 * q] * sC[fd * 64 + l + 16 * q];
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
    if (l == 0) szg[fd] = a;
  }
  __syncthreads();
  if (tid < 16 && d_wd) d_wd[tid * Ch + ch] = sc1[tid >> 1] * szg[tid] + sh1[tid >> 1] * sG[tid];
  float* o = ep + (size_t)ch * 520;
  if (tid < 8) o[tid] = wd[(2 * tid) * Ch + ch] * szg[2 * tid] + wd[(2 * tid + 1) * Ch + ch] * szg[2 * tid + 1];
  for (int i = tid; i < 512; i += 256) {
    const int f = i >> 6, k = i & 63;
    o[8 + i] = wd[(2 * f) * Ch + ch] * sC[(2 * f) * 64 + k] + wd[(2 * f + 1) * Ch + ch] * sC[(2 * f + 1) * 64 + k];
  }
}
// one workgroup: sums the electrodes' contributions and finishes d bn1.weight, d bn1.bias, d conv1.weight
__global__ __launch_bounds__(512) void k_eegc_grads_final(const float* __restrict__ ep, const float* __restrict__ gpart, int nsplit, int Ch,
    const float* __restrict__ w1, const float* __restrict__ wd, const float* __restrict__ mean1, const float* __restrict__ inv1,
    const float* __restrict__ sc1, const double* __restrict__ RS, double count, float* __restrict__ d_gamma, float* __restrict__ d_beta,
    float* __restrict__ d_w1) {
  __shared__ double sRS[64 * 64 + 64];                  // R and S, staged once with coalesced loads
  __shared__ float sw[8 * 64];
  __shared__ float swd[16 * EEG_MAXCH_C];
  __shared__ double sQ[8], sk1[8], sk2[8];
  __shared__ float sG[16];
  const int tid = threadIdx.x;
  {                                                     // all nine of a thread's R / S loads in flight together (a rolled copy loop is
    double rs[9];                                       // nine serial round trips in a one-workgroup kernel: nothing else hides them)
#pragma unroll
    for (int q = 0; q < 9; ++q) { const int i = tid + 512 * q; rs[q] = RS[i < 64 * 64 + 64 ? i : 64 * 64 + 63]; }
#pragma unroll
    for (int q = 0; q < 9; ++q) { const int i = tid + 512 * q; if (i < 64 * 64 + 64) sRS[i] = rs[q]; }
  }
  sw[tid] = w1[tid];
  // sum over the electrodes: this thread's column of ep (8 + 512 columns, thread tid -> column 8 + tid), loads batched by 32
  double dyx = 0.0;
  for (int c0 = 0; c0 < Ch; c0 += 32) {
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = ep[(size_t)(c0 + c < Ch ? c0 + c : Ch - 1) * 520 + 8 + tid];
#pragma unroll
    for (int c = 0; c < 32; ++c) dyx += c0 + c < Ch ? (double)v[c] : 0.0;
  }
  if (tid < 16) {                                       // nsplit <= EEGC_MAXS loads in flight together, summed in split order
    float v[EEGC_MAXS];
#pragma unroll
    for (int sp = 0; sp < EEGC_MAXS; ++sp) v[sp] = gpart[(sp < nsplit ? sp : nsplit - 1) * 16 + tid];
    float a = 0.f;
#pragma unroll
    for (int sp = 0; sp < EEGC_MAXS; ++sp) a += sp < nsplit ? v[sp] : 0.f;
    sG[tid] = a;
  }
  if (tid >= 64 && tid < 72) {                          // Q[f]: the electrodes' contributions, loads batched by 8
    const int f = tid - 64;
    double a = 0.0;
    for (int c0 = 0; c0 < Ch; c0 += 32) {
      float v[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = ep[(size_t)(c0 + c < Ch ? c0 + c : Ch - 1) * 520 + f];
#pragma unroll
      for (int c = 0; c < 32; ++c) a += c0 + c < Ch ? (double)v[c] : 0.0;
    }
    sQ[f] = a;
  }
  for (int i = tid; i < 16 * Ch; i += 512) swd[i] = wd[i];
  __syncthreads();
  if (tid < 8) {
    const int f = tid;
    double ws0 = 0.0, ws1 = 0.0;
    for (int ch = 0; ch < Ch; ++ch) { ws0 += (double)swd[(2 * f) * Ch + ch]; ws1 += (double)swd[(2 * f + 1) * Ch + ch]; }
    const double db = ws0 * (double)sG[2 * f] + ws1 * (double)sG[2 * f + 1];               // sum dy
    const double dg = (double)inv1[f] * (sQ[f] - (double)mean1[f] * db);                    // sum dy zhat
    sk1[f] = db / count; sk2[f] = dg / count;
    if (d_beta) d_beta[f] = (float)db;
    if (d_gamma) d_gamma[f] = (float)dg;
  }
  __syncthreads();
  if (!d_w1) return;
  {                                                     // d w1[f][k] = a_f (dyx - k1 S[k] - k2 invstd_f (sum_k' w1[f][k'] R[k'][k] - mu_f S[k]))
    const int f = tid >> 6, k = tid & 63;
    double zx = 0.0;
#pragma unroll 8
    for (int k2 = 0; k2 < 64; ++k2) zx += (double)sw[f * EC_K + k2] * sRS[k2 * 64 + k];
    const double S = sRS[4096 + k];
    const double zhx = (double)inv1[f] * (zx - (double)mean1[f] * S);
    d_w1[f * EC_K + k] = (float)((double)sc1[f] * (dyx - sk1[f] * S - sk2[f] * zhx));
  }
}

// ---- input gradient in evaluation mode (attribution passes: saliency, integrated gradients, SHAP-style estimators) -------------
// With BatchNorm1 on its running statistics, conv1 -> BatchNorm1 -> depthwise is the fixed linear map of the header comment, and its
// adjoint is    dx[b,ch,t] = sum_fd wd[fd,ch] e[b,fd,t],      e[b,fd,t] = a_f sum_k w1[f,k] g[b,fd,t-k+31]       (g = dL/du, a_f = sc1[f])
// -- a 64-tap filter over 16 rows per sample and a 16 -> Chans mix, where the layer-by-layer kernel (k_eeg_conv1_bwd) rebuilds
// dL/dconv1 [8][T] per (sample, electrode) row and runs 8 x 64 taps on each: Chans / 2 times the arithmetic and the conv1 tensor re-read.
// grid (ceil(T / 256), B); phases as k_eegc_fwd run backwards: stage g with its halo, filter (wave = four rows, lane = four
// consecutive time steps, taps and window in registers), mix (thread = time step, all electrodes).
__global__ __launch_bounds__(256) void k_eegc_dx(const float* __restrict__ g, const float* __restrict__ w1, const float* __restrict__ wd,
    const float* __restrict__ sc1, float* __restrict__ dx, int Ch, int T) {
  __shared__ __attribute__((aligned(16))) float sg[16 * EC_VP];          // column i <-> time t0 - 32 + i
  __shared__ __attribute__((aligned(16))) float se[16 * EC_TC];
  __shared__ __attribute__((aligned(16))) float swt[EEG_MAXCH_C * 16];   // [ch][fd]
  __shared__ __attribute__((aligned(16))) float sw1[8 * EC_K];           // taps reversed: the adjoint of a correlation is a convolution
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, t0 = blockIdx.x * EC_TC;
  for (int i = tid; i < 16 * Ch; i += 256) { const int fd = i / Ch, ch = i - fd * Ch; swt[ch * 16 + fd] = wd[i]; }
  for (int i = tid; i < 8 * EC_K; i += 256) sw1[i] = w1[(i & ~(EC_K - 1)) + (EC_K - 1 - (i & (EC_K - 1)))];
  {
    constexpr int NQ = 16 * EC_VP / 256;                // 20 loads per thread, all in flight before the first LDS write
    const float* gb = g + (size_t)b * 16 * T;
    float gv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + 256 * q, fd = i / EC_VP, col = i - fd * EC_VP;
      const int t = t0 - (EC_K - 1 - EC_PADL) + col;
      const bool ok = t >= 0 && t < T;
      gv[q] = gb[(size_t)fd * T + (ok ? t : 0)];
      if (!ok) gv[q] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) sg[tid + 256 * q] = gv[q];
  }
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int f = 2 * wave + half;
    float w[64];
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const float4 v = *reinterpret_cast<const float4*>(sw1 + f * EC_K + 4 * k4);
      w[4 * k4] = v.x; w[4 * k4 + 1] = v.y; w[4 * k4 + 2] = v.z; w[4 * k4 + 3] = v.w;
    }
    const float a = sc1[f];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const int fd = 2 * f + dd;
      float win[68];
#pragma unroll
      for (int q = 0; q < 17; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(sg + fd * EC_VP + 4 * lane + 4 * q);
        win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        o[0] = fmaf(w[k], win[k], o[0]); o[1] = fmaf(w[k], win[k + 1], o[1]);
        o[2] = fmaf(w[k], win[k + 2], o[2]); o[3] = fmaf(w[k], win[k + 3], o[3]);
      }
      *reinterpret_cast<float4*>(se + fd * EC_TC + 4 * lane) = make_float4(a * o[0], a * o[1], a * o[2], a * o[3]);
    }
  }
  __syncthreads();
  float ev[16];
#pragma unroll
  for (int fd = 0; fd < 16; ++fd) ev[fd] = se[fd * EC_TC + tid];
  const int t = t0 + tid;
  if (t >= T) return;
  float* dst = dx + (size_t)b * Ch * T + t;
  for (int ch = 0; ch < Ch; ++ch) {
    float acc = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = *reinterpret_cast<const float4*>(swt + ch * 16 + 4 * q4);
      acc = fmaf(v.x, ev[4 * q4], acc); acc = fmaf(v.y, ev[4 * q4 + 1], acc); acc = fmaf(v.z, ev[4 * q4 + 2], acc); acc = fmaf(v.w, ev[4 * q4 + 3], acc);
    }
    dst[(size_t)ch * T] = acc;
  }
}
int bx_eegc_dx_launch(const float* g, const float* w1, const float* wd, const float* sc1, float* dx, int B, int Ch, int T, hipStream_t s) {
  if (Ch > EEG_MAXCH_C) return -2;
  dim3 grid((unsigned)((T + EC_TC - 1) / EC_TC), (unsigned)B);
  hipLaunchKernelGGL(k_eegc_dx, grid, dim3(256), 0, s, g, w1, wd, sc1, dx, Ch, T);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- launchers (called from eeg.hip) ---------------------------------------------------------------------------------------
size_t bx_eegc_stat_floats() { return (size_t)EC_NR * EC_RCOLS + (size_t)EC_NE * EC_ECOLS; }
int bx_eegc_forward(const float* x, const float* w1, const float* wd, const float* gamma, const float* beta, float* rmean, float* rvar, int64_t* nbt,
                    float momentum, float eps, float* mean1, float* inv1, float* sc1, float* sh1, double* RS, float* stat_ws, float* u,
                    float* bn2_partials, int* bn2_rows, int B, int Ch, int T, hipStream_t s) {
  float* rpart = stat_ws;
  float* epart = stat_ws + (size_t)EC_NR * EC_RCOLS;
  const size_t lds_r = (size_t)16 * EC_RCOLS * sizeof(float), lds_e = (size_t)4 * 33 * EC_EDGE * sizeof(float);
  const size_t lds = lds_r > lds_e ? lds_r : lds_e;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k_eegc_stats<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_eegc_stats<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_eegc_stats<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -1;
    attr = true;
  }
  if (lds > 150 * 1024) return -2;
  static const bool split_roles = getenv("BX_EEGC_SPLIT_STATS") != nullptr;
  if (split_roles) {
    hipLaunchKernelGGL(k_eegc_stats<1>, dim3(EC_NR), dim3(256), lds_r, s, x, rpart, epart, B * Ch, T);
    hipLaunchKernelGGL(k_eegc_stats<2>, dim3(EC_NE), dim3(256), lds_e, s, x, rpart, epart, B * Ch, T);
  } else
    hipLaunchKernelGGL(k_eegc_stats<0>, dim3(EC_NR + EC_NE), dim3(256), lds, s, x, rpart, epart, B * Ch, T);
  hipLaunchKernelGGL(k_eegc_finalize1, dim3(1), dim3(1024), 0, s, rpart, epart, w1, gamma, beta, rmean, rvar, nbt, momentum, eps,
                     (double)B * Ch * T, mean1, inv1, sc1, sh1, RS);
  dim3 grid((unsigned)((T + EC_TC - 1) / EC_TC), (unsigned)B);
  hipLaunchKernelGGL(k_eegc_fwd, grid, dim3(256), 0, s, x, w1, wd, sc1, sh1, u, bn2_partials, Ch, T, 1);
  *bn2_rows = (int)(grid.x * grid.y);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int bx_eegc_forward_eval(const float* x, const float* w1, const float* wd, const float* sc1, const float* sh1, float* u, int B, int Ch, int T,
                         hipStream_t s, const float* const* x_slot) {
  if (Ch > EEG_MAXCH_C) return -2;
  dim3 grid((unsigned)((T + EC_TC - 1) / EC_TC), (unsigned)B);
  hipLaunchKernelGGL(k_eegc_fwd, grid, dim3(256), 0, s, x, w1, wd, sc1, sh1, u, (float*)nullptr, Ch, T, 0, x_slot);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int bx_eegc_grads(const float* cpart, const float* gpart, int nsplit, const float* w1, const float* wd, const float* mean1, const float* inv1,
                  const float* sc1, const float* sh1, const double* RS, float* ep, float* d_wd, float* d_gamma, float* d_beta, float* d_w1,
                  int B, int Ch, int T, hipStream_t s) {
  hipLaunchKernelGGL(k_eegc_grads_ch, dim3(Ch), dim3(256), 0, s, cpart, gpart, nsplit, w1, wd, sc1, sh1, d_wd, ep, Ch);
  hipLaunchKernelGGL(k_eegc_grads_final, dim3(1), dim3(512), 0, s, ep, gpart, nsplit, Ch, w1, wd, mean1, inv1, sc1, RS, (double)B * Ch * T,
                     d_gamma, d_beta, d_w1);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
