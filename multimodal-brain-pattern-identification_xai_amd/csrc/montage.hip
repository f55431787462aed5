// Native-pipeline EEG montage stacker: raw frames [B, L, Craw] -> [B, R, out_len] fp32.
// Replaces the per-sample numpy/scipy chain of the reference's CombinedDataset.process_eeg
// (root/jupyter_notebooks/XAI_Multimodality.py:1148-1164 with helpers :1211-1276):
//   band-pass (order 5 Butterworth, scipy.signal.lfilter = direct form II transposed in fp64) on every raw channel
//   -> NaN -> the row's nanmean -> bipolar differences -> band-pass (order 6) -> mean of 4 consecutive samples
//   -> keep every 4th column of [0, L-1) -> per-row z-score (population std, eps) -> row selection -> pad / truncate.
// Three launches per batch:
//   stage 1: one thread per (sample, raw channel): IIR-1 over time, u[b][t][c] (fp64, lanes = neighbouring channels, so
//            wave loads/stores touch contiguous records), non-NaN count and sum per row;
//   stage 2: one thread per (sample, output row): NaN substitution, optional difference of two u rows, IIR-2, the
//            4-sample mean at every 4th step -> yk[b][k][r] (fp64), then mean and 1/(std+eps) in two more passes;
//   stage 3: tiled transpose + normalisation + padding to out[b][r][0..out_len) (coalesced both ways through LDS).
// The IIR recurrences are sequential in time; samples are loaded eight at a time so that the loads of a trip are in
// flight together.
#include "bx_common.h"

#define MONT_MAX_ORDER 12
struct MontCoef { double b[MONT_MAX_ORDER + 1]; double a[MONT_MAX_ORDER + 1]; };

// One direct-form-II-transposed step in exactly scipy's operation order (scipy/signal/_lfilter.c.in: y = Z0 + b0*x;
// Z[n] = Z[n+1] + x*b[n+1] - y*a[n+1]; last Z = x*b - y*a), every product and sum rounded separately: the 10th/12th-order
// transfer-function form of a 0.5-20 Hz band-pass at 200 Hz is badly conditioned, and a fused multiply-add here shows
// up as 6e-3 relative error in the output.
__device__ __forceinline__ double iir_step(double x, double (&z)[MONT_MAX_ORDER], const MontCoef& k, int order) {
#pragma clang fp contract(off)
  const double y = z[0] + k.b[0] * x;
#pragma unroll
  for (int i = 0; i < MONT_MAX_ORDER - 1; ++i)
    if (i < order - 1) z[i] = (z[i + 1] + x * k.b[i + 1]) - y * k.a[i + 1];
  z[order - 1] = x * k.b[order] - y * k.a[order];
  return y;
}

// numpy's float64 add.reduce order (np.nansum inside np.nanmean, NB:1218): the axis is cut into 8192-element buffer
// chunks; a chunk is summed pairwise (halves rounded down to a multiple of 8, leaves of <= 128 elements with 8 strided
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n%8 tail); chunk sums are added left to right.
// The substitute value feeds the second band-pass, whose transfer-function form amplifies a 1e-15 relative change of
// this mean to ~1e-4 of the output, so the order is reproduced exactly.  Elements at index >= valid count as 0 (NaN -> 0).
__device__ double np_leaf_sum(const double* p, size_t stride, int n, int valid) {
  auto at = [&](int i) { return i < valid ? p[(size_t)i * stride] : 0.0; };
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += at(i);
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = at(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += at(i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += at(i);
  return res;
}
__device__ double np_pairwise_sum(const double* p, size_t stride, int n, int valid) {
  int off[16], len[16], ph[16];
  double left[16];
  int sp = 0;
  off[0] = 0; len[0] = n; ph[0] = 0;
  double ret = 0.0;
  while (sp >= 0) {
    if (len[sp] <= 128) {
      ret = np_leaf_sum(p + (size_t)off[sp] * stride, stride, len[sp], valid - off[sp]);
      --sp;
      while (sp >= 0) {                       // hand the value to the parent
        if (ph[sp] == 1) {                    // it was the left half: descend into the right half
          left[sp] = ret;
          int n2 = len[sp] / 2; n2 -= n2 % 8;
          off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; ph[sp + 1] = 0;
          ph[sp] = 2; ++sp;
          break;
        }
        ret = left[sp] + ret;                 // right half done
        --sp;
      }
      continue;
    }
    int n2 = len[sp] / 2; n2 -= n2 % 8;
    ph[sp] = 1;
    off[sp + 1] = off[sp]; len[sp + 1] = n2; ph[sp + 1] = 0;
    ++sp;
  }
  return ret;
}
__device__ double np_sum(const double* p, size_t stride, int n, int valid) {
  double tot = 0.0;
  for (int c0 = 0; c0 < n; c0 += 8192) {
    const int cn = n - c0 < 8192 ? n - c0 : 8192;
    const double part = np_pairwise_sum(p + (size_t)c0 * stride, stride, cn, valid - c0);
    tot = c0 == 0 ? part : tot + part;
  }
  return tot;
}

__global__ __launch_bounds__(64) void k_montage_stage1(const float* __restrict__ raw, double* __restrict__ u, double* __restrict__ rowmean,
                                                       int* __restrict__ flags, int nrows, int L, int Craw, MontCoef k, int order) {
  const int row = blockIdx.x * 64 + threadIdx.x;
  if (row >= nrows) return;
  const int b = row / Craw, c = row % Craw;
  const float* src = raw + (size_t)b * L * Craw + c;
  double* dst = u + (size_t)b * L * Craw + c;
  double z[MONT_MAX_ORDER];
#pragma unroll
  for (int i = 0; i < MONT_MAX_ORDER; ++i) z[i] = 0.0;
  double sum = 0.0;
  int cnt = 0;
  for (int n0 = 0; n0 < L; n0 += 8) {
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = n0 + j < L ? src[(size_t)(n0 + j) * Craw] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (n0 + j < L) {
        const double y = iir_step((double)xv[j], z, k, order);
        dst[(size_t)(n0 + j) * Craw] = y;
        if (y == y) { sum += y; ++cnt; }
      }
    }
  }
  // lfilter turns the first NaN into a NaN tail, so the non-NaN samples are exactly the first cnt; only rows with a NaN use the mean
  if (cnt > 0 && cnt < L) sum = np_sum(dst, (size_t)Craw, L, cnt);
  rowmean[row] = cnt ? sum / (double)cnt : 0.0;
  if (cnt == 0) atomicOr(flags, 1);               // a row that is NaN from its first sample: the reference drops it (and mis-indexes)
}

__global__ __launch_bounds__(64) void k_montage_stage2(const double* __restrict__ u, const double* __restrict__ rowmean, const int* __restrict__ row_a,
                                                       const int* __restrict__ row_b, double* __restrict__ yk, double* __restrict__ stats,
                                                       int nout, int L, int Craw, int R, int K, MontCoef k, int order, double eps) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= nout) return;
  const int b = idx / R, r = idx % R;
  const int ca = row_a[r], cb = row_b[r];
  const double ma = rowmean[b * Craw + ca], mb = cb >= 0 ? rowmean[b * Craw + cb] : 0.0;
  const double* ua = u + (size_t)b * L * Craw + ca;
  const double* ub = u + (size_t)b * L * Craw + (cb >= 0 ? cb : ca);
  double* yrow = yk + (size_t)b * K * R + r;        // yk[b][k][r]
  double z[MONT_MAX_ORDER];
#pragma unroll
  for (int i = 0; i < MONT_MAX_ORDER; ++i) z[i] = 0.0;
  double s4 = 0.0, sum = 0.0;
  for (int n0 = 0; n0 < L; n0 += 8) {
    double va[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool in = n0 + j < L;
      va[j] = in ? ua[(size_t)(n0 + j) * Craw] : 0.0;
      vb[j] = (in && cb >= 0) ? ub[(size_t)(n0 + j) * Craw] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + j;
      if (n < L) {
        double x = va[j] == va[j] ? va[j] : ma;
        if (cb >= 0) x -= (vb[j] == vb[j] ? vb[j] : mb);
        const double w = iir_step(x, z, k, order);
        s4 = (n & 3) == 0 ? w : s4 + w;               // ((y[c] + y[c+1]) + y[c+2]) + y[c+3], the reference's order
        if ((n & 3) == 3 && (n >> 2) < K) {
          const double yv = s4 / 4.0;
          yrow[(size_t)(n >> 2) * R] = yv;
          sum += yv;
        }
      }
    }
  }
  const double mean = sum / (double)K;
  double var = 0.0;
  for (int kk = 0; kk < K; ++kk) { const double d = yrow[(size_t)kk * R] - mean; var += d * d; }
  stats[(size_t)idx * 2] = mean;
  stats[(size_t)idx * 2 + 1] = 1.0 / (sqrt(var / (double)K) + eps);
}

// out[b][r][k0..k0+63] = (yk[b][k][r] - mean) * inv for k < K, 0 for K <= k < out_len.   grid (ceil(out_len/64), B)
__global__ __launch_bounds__(256) void k_montage_stage3(const double* __restrict__ yk, const double* __restrict__ stats, float* __restrict__ out,
                                                        int R, int K, int out_len) {
  extern __shared__ float tile[];                    // [64][R + 1]
  const int b = blockIdx.y, k0 = blockIdx.x * 64, RS = R + 1;
  for (int i = threadIdx.x; i < 64 * R; i += 256) {
    const int kk = i / R, r = i % R;
    float v = 0.f;
    if (k0 + kk < K) {
      const double m = stats[((size_t)b * R + r) * 2], inv = stats[((size_t)b * R + r) * 2 + 1];
      v = (float)((yk[((size_t)b * K + k0 + kk) * R + r] - m) * inv);
    }
    tile[kk * RS + r] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * R; i += 256) {
    const int r = i / 64, kk = i % 64;
    if (k0 + kk < out_len) out[((size_t)b * R + r) * out_len + k0 + kk] = tile[kk * RS + r];
  }
}

static size_t mont_ws(int B, int L, int Craw, int R, size_t* off_mean, size_t* off_yk, size_t* off_stats, size_t* off_flags) {
  const int K = (L - 1 + 3) / 4;
  size_t o = 0;
  o += bx_align_up((size_t)B * L * Craw * sizeof(double), 256);
  *off_mean = o;  o += bx_align_up((size_t)B * Craw * sizeof(double), 256);
  *off_yk = o;    o += bx_align_up((size_t)B * K * R * sizeof(double), 256);
  *off_stats = o; o += bx_align_up((size_t)B * R * 2 * sizeof(double), 256);
  *off_flags = o; o += 256;
  return o;
}
extern "C" size_t bx_eeg_montage_workspace(int B, int L, int Craw, int R) {
  if (B <= 0 || L < 5 || Craw <= 0 || R <= 0) return 0;
  size_t a, b, c, d;
  return mont_ws(B, L, Craw, R, &a, &b, &c, &d);
}
extern "C" int bx_eeg_montage_stack(const float* raw, const int* row_a, const int* row_b, float* out, int B, int L, int Craw, int R,
                                    int out_len, const double* b1, const double* a1, int order1, const double* b2, const double* a2,
                                    int order2, float eps, int* status, void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(raw && row_a && row_b && out && b1 && a1 && b2 && a2 && status, "bx_eeg_montage_stack: null pointer");
  BX_REQUIRE(B > 0 && L >= 5 && Craw > 0 && R > 0 && out_len > 0, "bx_eeg_montage_stack: bad sizes");
  BX_REQUIRE(order1 >= 1 && order1 <= MONT_MAX_ORDER && order2 >= 1 && order2 <= MONT_MAX_ORDER, "bx_eeg_montage_stack: filter order must be 1..%d (tf coefficients: order+1 values)", MONT_MAX_ORDER);
  // the 4-sample mean reads columns c..c+3 of the kept columns c = 0,4,.. < L-1; for L % 4 in {2,3} the reference's
  // np.roll (over the flattened array) would pull the last mean across the row boundary: not reproduced here
  BX_REQUIRE(L % 4 == 0 || L % 4 == 1, "bx_eeg_montage_stack: L %% 4 must be 0 or 1 (got L=%d)", L);
  BX_REQUIRE(a1[0] != 0.0 && a2[0] != 0.0, "bx_eeg_montage_stack: a[0] == 0");
  size_t om, oy, os, of;
  const size_t need = mont_ws(B, L, Craw, R, &om, &oy, &os, &of);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_eeg_montage_stack: workspace %zu < %zu", workspace_bytes, need);
  MontCoef k1, k2;
  for (int i = 0; i <= MONT_MAX_ORDER; ++i) {
    k1.b[i] = i <= order1 ? b1[i] / a1[0] : 0.0; k1.a[i] = i <= order1 ? a1[i] / a1[0] : 0.0;
    k2.b[i] = i <= order2 ? b2[i] / a2[0] : 0.0; k2.a[i] = i <= order2 ? a2[i] / a2[0] : 0.0;
  }
  hipStream_t s = (hipStream_t)stream;
  char* W = (char*)workspace;
  double* u = (double*)W;
  double* rowmean = (double*)(W + om);
  double* yk = (double*)(W + oy);
  double* stats = (double*)(W + os);
  const int K = (L - 1 + 3) / 4;
  if (hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess) BX_FAIL(BX_EHIP, "bx_eeg_montage_stack: memset failed");
  hipLaunchKernelGGL(k_montage_stage1, dim3(bx_ceil_div(B * Craw, 64)), dim3(64), 0, s, raw, u, rowmean, status, B * Craw, L, Craw, k1, order1);
  BX_CHECK_LAUNCH("bx_eeg_montage_stack(stage 1)");
  hipLaunchKernelGGL(k_montage_stage2, dim3(bx_ceil_div(B * R, 64)), dim3(64), 0, s, u, rowmean, row_a, row_b, yk, stats, B * R, L, Craw, R, K,
                     k2, order2, (double)eps);
  BX_CHECK_LAUNCH("bx_eeg_montage_stack(stage 2)");
  hipLaunchKernelGGL(k_montage_stage3, dim3(bx_ceil_div(out_len, 64), B), dim3(256), (size_t)64 * (R + 1) * sizeof(float), s, yk, stats, out, R, K, out_len);
  BX_CHECK_LAUNCH("bx_eeg_montage_stack(stage 3)");
  return BX_OK;
}
