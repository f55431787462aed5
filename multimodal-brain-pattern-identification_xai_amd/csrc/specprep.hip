// Native-pipeline spectrogram pre-processing: parquet values [B, Trows, C] -> [B, 3, R, W] fp32.
// Replaces the per-sample numpy/scipy chain of the reference's CombinedDataset.process_spectrogram
// (root/jupyter_notebooks/XAI_Multimodality.py:1166-1204 with helpers :1211-1243,1288-1307):
//   (offset//2 column window of `sel` columns, zero padded) -> transpose -> pad/truncate to R x W (400 x 300) ->
//   NaN -> the row's nanmean -> subtract column means -> notch filtfilt along the rows (scipy.signal.filtfilt: odd
//   extension by 3*max(len(a),len(b)) samples, lfilter_zi initial state, forward and backward pass, fp64) ->
//   gaussian_filter sigma=1 (separable 9-tap correlate1d, 'reflect' boundary, rows then columns) -> min-max to [0,1) ->
//   resize to its own shape (identity) -> three identical channels.
// One workgroup per sample walks the phases over an fp64 plane in the caller's workspace (L2-resident, 2.9 MB per
// sample); phases that need other threads' results are separated by workgroup barriers.
#include "bx_common.h"

struct SpecPrepCoef { double b[3], a[3], zi[2], gw[5]; };     // gw[d] = gaussian weight at distance d (radius 4)
#define SP_PAD 9

__device__ __forceinline__ int sp_reflect(int i, int n) {     // scipy.ndimage 'reflect': (d c b a | a b c d | d c b a)
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  return i;
}

__global__ __launch_bounds__(256) void k_spec_preprocess(const float* __restrict__ raw, const int* __restrict__ offsets, float* __restrict__ out,
                                                         double* __restrict__ scratch, int* __restrict__ status, int Trows, int Ccols,
                                                         int R, int W, int sel, SpecPrepCoef k, double eps) {
  __shared__ double smin[256], smax[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t plane = (size_t)R * W;
  double* A = scratch + (size_t)b * (2 * plane + (size_t)(R + 2 * SP_PAD) * W);
  double* Bp = A + plane;
  double* E = Bp + plane;                                      // [R + 18][W] forward-pass output of filtfilt
  const float* src = raw + (size_t)b * Trows * Ccols;
  const int o = offsets ? offsets[b] / 2 : 0;
  const int nrows_src = offsets ? sel : Ccols;                 // rows of basic.T before padding to R
  // ---- phase 1: gather + transpose + pad, NaN -> row nanmean
  for (int i = tid; i < R; i += 256) {
    const bool row_live = i < nrows_src && o + i < Ccols && o + i >= 0;
    const int jn = Trows < W ? Trows : W;
    double sum = 0.0;
    int cnt = 0, nnan = 0;
    for (int j0 = 0; j0 < W; j0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (row_live && j0 + u < jn) ? src[(size_t)(j0 + u) * Ccols + o + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < W) {
          const double d = (double)v[u];
          A[(size_t)i * W + j0 + u] = d;
          if (d == d) { sum += d; ++cnt; } else ++nnan;
        }
    }
    if (nnan) {
      if (cnt == 0) atomicOr(status, 1);                       // an all-NaN row: the reference drops it and really resamples
      const double m = cnt ? sum / (double)cnt : 0.0;
      for (int j = 0; j < W; ++j) { const double d = A[(size_t)i * W + j]; if (d != d) A[(size_t)i * W + j] = m; }
    }
  }
  __syncthreads();
  // ---- phase 2 + 3: per column: subtract the column mean, then filtfilt along the rows
  for (int j = tid; j < W; j += 256) {
    double s = 0.0;
    for (int i = 0; i < R; ++i) s += A[(size_t)i * W + j];
    const double mean = s / (double)R;
    for (int i = 0; i < R; ++i) A[(size_t)i * W + j] -= mean;
    auto ext = [&](int n) -> double {                          // odd extension by SP_PAD samples at both ends
      if (n < SP_PAD) return 2.0 * A[j] - A[(size_t)(SP_PAD - n) * W + j];
      if (n < SP_PAD + R) return A[(size_t)(n - SP_PAD) * W + j];
      return 2.0 * A[(size_t)(R - 1) * W + j] - A[(size_t)(R - 2 - (n - SP_PAD - R)) * W + j];
    };
    const int N = R + 2 * SP_PAD;
    const double x0 = ext(0);
    double z0 = k.zi[0] * x0, z1 = k.zi[1] * x0, y = 0.0;
    for (int n = 0; n < N; ++n) {
      const double x = ext(n);
      y = z0 + k.b[0] * x;
      z0 = (z1 + x * k.b[1]) - y * k.a[1];
      z1 = x * k.b[2] - y * k.a[2];
      E[(size_t)n * W + j] = y;
    }
    z0 = k.zi[0] * y; z1 = k.zi[1] * y;                        // y = last forward output
    for (int n = N - 1; n >= 0; --n) {
      const double x = E[(size_t)n * W + j];
      const double w = z0 + k.b[0] * x;
      z0 = (z1 + x * k.b[1]) - w * k.a[1];
      z1 = x * k.b[2] - w * k.a[2];
      if (n >= SP_PAD && n < SP_PAD + R) A[(size_t)(n - SP_PAD) * W + j] = w;
    }
  }
  __syncthreads();
  // ---- phase 4: gaussian along the rows (axis 0), A -> Bp; scipy's symmetric correlate1d order: centre, then pairs far -> near
  for (size_t e = tid; e < plane; e += 256) {
    const int i = (int)(e / W), j = (int)(e % W);
    double t = A[e] * k.gw[0];
#pragma unroll
    for (int d = 4; d >= 1; --d) t += (A[(size_t)sp_reflect(i - d, R) * W + j] + A[(size_t)sp_reflect(i + d, R) * W + j]) * k.gw[d];
    Bp[e] = t;
  }
  __syncthreads();
  // ---- phase 5: gaussian along the columns (axis 1), Bp -> A, and the plane's min / max
  double mn = INFINITY, mx = -INFINITY;
  bool has_nan = false;
  for (size_t e = tid; e < plane; e += 256) {
    const int i = (int)(e / W), j = (int)(e % W);
    const double* row = Bp + (size_t)i * W;
    double t = row[j] * k.gw[0];
#pragma unroll
    for (int d = 4; d >= 1; --d) t += (row[sp_reflect(j - d, W)] + row[sp_reflect(j + d, W)]) * k.gw[d];
    A[e] = t;
    if (t != t) has_nan = true;
    mn = t < mn ? t : mn; mx = t > mx ? t : mx;
  }
  if (has_nan) atomicOr(status, 2);                            // nan_to_num(nan=nanmean) would be needed: not expected after the fill
  smin[tid] = mn; smax[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { smin[tid] = smin[tid + s] < smin[tid] ? smin[tid + s] : smin[tid]; smax[tid] = smax[tid + s] > smax[tid] ? smax[tid + s] : smax[tid]; }
    __syncthreads();
  }
  mn = smin[0]; mx = smax[0];
  const double den = (mx - mn) + eps;
  float* dst = out + (size_t)b * 3 * plane;
  for (size_t e = tid; e < plane; e += 256) {
    const float v = (float)((A[e] - mn) / den);
    dst[e] = v; dst[plane + e] = v; dst[2 * plane + e] = v;
  }
}

extern "C" size_t bx_spec_preprocess_workspace(int B, int R, int W) {
  if (B <= 0 || R <= SP_PAD || W <= 0) return 0;
  return (size_t)B * (2 * (size_t)R * W + (size_t)(R + 2 * SP_PAD) * W) * sizeof(double);
}
extern "C" int bx_spec_preprocess(const float* raw, const int* offsets, float* out, int B, int Trows, int Ccols, int R, int W, int sel,
                                  const double* notch_b, const double* notch_a, const double* notch_zi, const double* gauss_w,
                                  float eps, int* status, void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(raw && out && notch_b && notch_a && notch_zi && gauss_w && status, "bx_spec_preprocess: null pointer");
  BX_REQUIRE(B > 0 && Trows > 0 && Ccols > 0 && W > 4 && sel > 0, "bx_spec_preprocess: bad sizes");
  BX_REQUIRE(R > SP_PAD + 1, "bx_spec_preprocess: R must exceed filtfilt's pad length (%d)", SP_PAD);
  BX_REQUIRE(notch_a[0] != 0.0, "bx_spec_preprocess: a[0] == 0");
  const size_t need = bx_spec_preprocess_workspace(B, R, W);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_spec_preprocess: workspace %zu < %zu", workspace_bytes, need);
  SpecPrepCoef k;
  for (int i = 0; i < 3; ++i) { k.b[i] = notch_b[i] / notch_a[0]; k.a[i] = notch_a[i] / notch_a[0]; }
  k.zi[0] = notch_zi[0]; k.zi[1] = notch_zi[1];
  for (int d = 0; d < 5; ++d) k.gw[d] = gauss_w[4 + d];        // gauss_w: the 9 symmetric weights, centre at index 4
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(status, 0, sizeof(int), s) != hipSuccess) BX_FAIL(BX_EHIP, "bx_spec_preprocess: memset failed");
  hipLaunchKernelGGL(k_spec_preprocess, dim3(B), dim3(256), 0, s, raw, offsets, out, (double*)workspace, status, Trows, Ccols, R, W, sel, k, (double)eps);
  BX_CHECK_LAUNCH("bx_spec_preprocess");
  return BX_OK;
}

// ================================================================================================
// Benchmark-variant spectrogram stacker (spectrogram half): parquet values [B, Trows, C] (C = regions x
// bins: the four 100-bin regions of a Kaggle HMS spectrogram) -> [B, regions, Ho, Wo] fp32 region planes.
//   window of `win` time rows from offset // 2, zero padded (process_spectrogram, XAI_Multimodality.py:1178-1183) -> transpose
//   to [C, win] -> normalize_signal (root/src/utils/data_utils.py:133-136: NaN -> nanmean of the sample, min-max with eps) ->
//   per region resample_spectrogram (data_utils.py:145-147: skimage.transform.resize, mode='reflect', anti_aliasing=True).
// skimage 0.24's resize for 2-D input is gaussian_filter(sigma = max(0, (s-1)/2), 'mirror') + ndi.zoom(order 1, 'mirror',
// grid_mode=True) + clip to the input range (a no-op for convex weights): out(oy, ox) = bilinear sample of the filtered plane at
// ((oy + 0.5) h/Ho - 0.5, (ox + 0.5) w/Wo - 0.5), indices mirrored about the edge samples.  Normalisation is affine and the
// resize weights sum to one, so the min-max is applied to the resized value.  All arithmetic in fp64 (scipy's type).
#define SR_MAX_RADIUS 8
struct SpecRegCoef { double gy[2 * SR_MAX_RADIUS + 1], gx[2 * SR_MAX_RADIUS + 1]; int ry, rx; };

__device__ __forceinline__ int sr_mirror(int i, int n) {     // numpy 'reflect' / scipy 'mirror': (d c b | a b c d | c b a)
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i < 0 ? -i : i;
  i %= period;
  return i > n - 1 ? period - i : i;
}
// Statistics of the windowed sample (zero padding included, NaNs set aside): SR_SPLIT workgroups per sample each take a band of
// the window's rows and leave {sum, count, min, max} of its finite values in part[(b * SR_SPLIT + p) * 4 ..]; the resize kernel's
// workgroups combine the SR_SPLIT partials of their sample in band order (specreg_stats_of).  Round 2 ran ONE workgroup per sample
// with a 64-bit division per element: 120 us per B=64 batch for 31 MB -- longer than the resize itself and, in the overlapped
// end-to-end step (brainxai.StagingRing), time the GPU-bound training step has to share.
#define SR_SPLIT 8
__global__ __launch_bounds__(256) void k_specreg_stats(const float* __restrict__ raw, const int* __restrict__ offsets, double* __restrict__ part,
                                                       int Trows, int C, int win) {
  __shared__ double ssum[256], smin[256], smax[256], scnt[256];
  const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int o = offsets ? offsets[b] / 2 : 0;
  const float* src = raw + (size_t)b * Trows * C;
  const int j0 = (int)((long long)win * p / SR_SPLIT), j1 = (int)((long long)win * (p + 1) / SR_SPLIT);
  double sum = 0.0, mn = INFINITY, mx = -INFINITY, cnt = 0.0;
  for (int c = tid; c < C; c += 256) {
    for (int ja = j0; ja < j1; ja += 8) {                 // eight rows in flight (clamped, unconditional loads)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int t = o + ja + u;
        t = t < 0 ? 0 : (t >= Trows ? Trows - 1 : t);
        v[u] = src[(size_t)t * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = ja + u, t = o + j;
        if (j >= j1) continue;
        const double d = (t >= 0 && t < Trows) ? (double)v[u] : 0.0;        // rows outside the recording are zero padding
        if (d == d) { sum += d; cnt += 1.0; mn = fmin(mn, d); mx = fmax(mx, d); }
      }
    }
  }
  ssum[tid] = sum; scnt[tid] = cnt; smin[tid] = mn; smax[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { ssum[tid] += ssum[tid + s]; scnt[tid] += scnt[tid + s]; smin[tid] = fmin(smin[tid], smin[tid + s]); smax[tid] = fmax(smax[tid], smax[tid + s]); }
    __syncthreads();
  }
  if (tid == 0) {
    double* dst = part + ((size_t)b * SR_SPLIT + p) * 4;
    dst[0] = ssum[0]; dst[1] = scnt[0]; dst[2] = smin[0]; dst[3] = smax[0];
  }
}
// {nanmean, min, max} of sample b from its SR_SPLIT partials, in band order (every thread of every resize workgroup recomputes it:
// 32 doubles from L2)
__device__ __forceinline__ void specreg_stats_of(const double* __restrict__ part, int b, double n, double& mean, double& lo, double& hi) {
  double v[SR_SPLIT][4];
#pragma unroll
  for (int p = 0; p < SR_SPLIT; ++p)
#pragma unroll
    for (int k = 0; k < 4; ++k) v[p][k] = part[((size_t)b * SR_SPLIT + p) * 4 + k];
  double sum = 0.0, cnt = 0.0;
  lo = INFINITY; hi = -INFINITY;
#pragma unroll
  for (int p = 0; p < SR_SPLIT; ++p) { sum += v[p][0]; cnt += v[p][1]; lo = fmin(lo, v[p][2]); hi = fmax(hi, v[p][3]); }
  mean = cnt > 0.0 ? sum / cnt : 0.0;
  if (cnt < n) { lo = fmin(lo, mean); hi = fmax(hi, mean); }          // the fill value takes part in the min-max
  if (!(cnt > 0.0)) { lo = hi = 0.0; }
}
__global__ __launch_bounds__(256) void k_specreg_resize(const float* __restrict__ raw, const int* __restrict__ offsets, const double* __restrict__ stats,
                                                        float* __restrict__ out, int Trows, int C, int win, int bins, int Ho, int Wo,
                                                        SpecRegCoef k, double eps) {
  // 16 (frequency) x 16 (time) output tile per workgroup: neighbouring threads read neighbouring frequency bins (contiguous in `raw`)
  const int b = blockIdx.z, r = blockIdx.y;
  const int tiles_x = (Wo + 15) / 16;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy = ty * 16 + (threadIdx.x & 15), ox = tx * 16 + (threadIdx.x >> 4);
  if (oy >= Ho || ox >= Wo) return;
  const int o = offsets ? offsets[b] / 2 : 0;
  const float* src = raw + (size_t)b * Trows * C + (size_t)r * bins;
  double mean, lo, hi;
  specreg_stats_of(stats, b, (double)win * (double)C, mean, lo, hi);
  auto at = [&](int f, int j) -> double {                      // filled, un-normalised sample: frequency bin f of region r, time j of the window
    const int t = o + j;
    if (t < 0 || t >= Trows) return 0.0;
    const double d = (double)src[(size_t)t * C + f];
    return d == d ? d : mean;
  };
  auto filtered = [&](int f, int j) -> double {                // anti-aliasing gaussian (separable, mirrored), identity when both radii are 0
    if (k.ry == 0 && k.rx == 0) return at(f, j);
    double acc = 0.0;
    for (int dy = -k.ry; dy <= k.ry; ++dy) {
      const int ff = sr_mirror(f + dy, bins);
      double row = 0.0;
      for (int dx = -k.rx; dx <= k.rx; ++dx) row += k.gx[dx + k.rx] * at(ff, sr_mirror(j + dx, win));
      acc += k.gy[dy + k.ry] * row;
    }
    return acc;
  };
  const double cy = ((double)oy + 0.5) * (double)bins / (double)Ho - 0.5, cx = ((double)ox + 0.5) * (double)win / (double)Wo - 0.5;
  const int y0 = (int)floor(cy), x0 = (int)floor(cx);
  const double wy = cy - (double)y0, wx = cx - (double)x0;
  const int ya = sr_mirror(y0, bins), yb = sr_mirror(y0 + 1, bins), xa = sr_mirror(x0, win), xb = sr_mirror(x0 + 1, win);
  const double v = (1.0 - wy) * ((1.0 - wx) * filtered(ya, xa) + wx * filtered(ya, xb)) + wy * ((1.0 - wx) * filtered(yb, xa) + wx * filtered(yb, xb));
  out[(((size_t)b * gridDim.y + r) * Ho + oy) * Wo + ox] = (float)((v - lo) / (hi - lo + eps));
}

extern "C" size_t bx_spec_regions_workspace(int B) { return (size_t)B * SR_SPLIT * 4 * sizeof(double) + 256; }
extern "C" int bx_spec_regions(const float* raw, const int* offsets, float* out, int B, int Trows, int C, int regions, int win, int Ho, int Wo,
                               const double* gauss_y, int radius_y, const double* gauss_x, int radius_x, float eps, void* workspace,
                               size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(raw && out && workspace && B > 0 && B <= 65535 && Trows > 0 && C > 0 && regions > 0 && regions <= 65535 && C % regions == 0 && win > 0 && Ho > 0 && Wo > 0,
             "bx_spec_regions: bad arguments");
  BX_REQUIRE(radius_y >= 0 && radius_y <= SR_MAX_RADIUS && radius_x >= 0 && radius_x <= SR_MAX_RADIUS && (radius_y == 0 || gauss_y) && (radius_x == 0 || gauss_x),
             "bx_spec_regions: anti-aliasing radius must be 0..%d", SR_MAX_RADIUS);
  BX_REQUIRE(workspace_bytes >= bx_spec_regions_workspace(B), "bx_spec_regions: workspace too small");
  SpecRegCoef k;
  k.ry = radius_y; k.rx = radius_x;
  for (int i = 0; i < 2 * SR_MAX_RADIUS + 1; ++i) { k.gy[i] = radius_y && i <= 2 * radius_y ? gauss_y[i] : (i == 0 ? 1.0 : 0.0); k.gx[i] = radius_x && i <= 2 * radius_x ? gauss_x[i] : (i == 0 ? 1.0 : 0.0); }
  double* stats = reinterpret_cast<double*>(bx_align_up((uintptr_t)workspace, 8));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_specreg_stats, dim3(SR_SPLIT, B), dim3(256), 0, s, raw, offsets, stats, Trows, C, win);
  const int bins = C / regions;
  const int tiles = ((Ho + 15) / 16) * ((Wo + 15) / 16);
  hipLaunchKernelGGL(k_specreg_resize, dim3(tiles, regions, B), dim3(256), 0, s, raw, offsets, (const double*)stats, out, Trows, C, win, bins, Ho, Wo, k,
                     (double)eps);
  BX_CHECK_LAUNCH("bx_spec_regions");
  return BX_OK;
}
