// 3x3 / stride 1 / pad 1 convolution over channels-last activations: weight packing, the direct
// (VALU, any dtype) forward / data-gradient kernel, the direct weight-gradient kernel, and the
// dispatch to the MFMA implicit-GEMM kernels (conv3x3_mfma.hip).
// Replaces nn.Conv2d(k=3,p=1)+F.relu of Block.forward (reference root/src/models/models.py:49-51,64-66)
// and their autograd backward.
#include "bx_common.h"

// implemented in conv3x3_mfma.hip
int bx_conv3x3_mfma_launch(const void* x, const void* packed_mfma, const float* bias, const void* relu_mask_src,
                           const void* addend, void* y, int B, int H, int W, int Ci, int Co, int flags, hipStream_t s, bxWgradPending* carry);
int bx_conv3x3_mfma_supported(int Ci, int Co, int dtype);
void bx_conv3x3_mfma_pack_launch(const float* w_oihw, void* packed, int Cout, int Cin, int I_p, int O_p, int tf, hipStream_t s);
size_t bx_wgrad_mfma_workspace(int B, int H, int W, int Ci_p, int Co);
int bx_wgrad_mfma_supported(int Ci_p, int Co, int dtype);
int bx_wgrad_mfma_launch(const void* x, const void* dz, float* dw, float* db, int B, int H, int W, int Cin, int Ci_p,
                         int Co, void* ws, size_t ws_bytes, bxWgradPending* pending, hipStream_t s);
int bx_wgrad_mfma_finish(bxWgradPending* pending, hipStream_t s);
// implemented in conv3x3_split.hip: fp32 storage on the bf16 matrix cores (operands split hi + lo, three MFMAs per product)
int bx_conv3x3_split_supported(int Ci, int Co);
int bx_conv3x3_split_launch(const void* x, const void* packed_split, const float* bias, const void* relu_mask_src, const void* addend, void* y,
                            int B, int H, int W, int Ci, int Co, int flags, hipStream_t s);
int bx_wgrad_split_supported(int Ci_p, int Co);
size_t bx_wgrad_split_workspace(int B, int H, int W, int Ci_p, int Co);
int bx_wgrad_split_launch(const void* x, const void* dz, float* dw, float* db, int B, int H, int W, int Cin, int Ci_p, int Co, void* ws,
                          size_t ws_bytes, bxWgradPending* pending, hipStream_t s);
static inline int conv_mfma_ok(int Ci, int Co, int dtype) { return dtype == BX_BF16 ? bx_conv3x3_mfma_supported(Ci, Co, dtype) : bx_conv3x3_split_supported(Ci, Co); }
static inline int wgrad_mfma_ok(int Ci_p, int Co, int dtype) { return dtype == BX_BF16 ? bx_wgrad_mfma_supported(Ci_p, Co, dtype) : bx_wgrad_split_supported(Ci_p, Co); }

// ------------------------------------------------------------------------------------------------
// Packing: fp32 [9][I_p][O_p]
//   forward  (tf=0): I = Cin,  O = Cout, Wp[t][i][o] = W[o][i][t]
//   dgrad    (tf=1): I = Cout, O = Cin,  Wp[t][i][o] = W[i][o][8-t]
__global__ void k_pack3x3(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int I_p, int O_p, int tf) {
  const int n = 9 * I_p * O_p;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int o = idx % O_p, i = (idx / O_p) % I_p, t = idx / (O_p * I_p);
    float v = 0.f;
    if (!tf) { if (i < Cin && o < Cout) v = w[((size_t)o * Cin + i) * 9 + t]; }
    else     { if (i < Cout && o < Cin) v = w[((size_t)i * Cin + o) * 9 + (8 - t)]; }
    wp[idx] = v;
  }
}

extern "C" int bx_conv3x3_pack(const float* w_oihw, float* packed_f32, void* packed_mfma, int Cout, int Cin,
                               int I_p, int O_p, int transpose_flip, bxStream stream) {
  BX_REQUIRE(w_oihw && (packed_f32 || packed_mfma) && Cout > 0 && Cin > 0, "bx_conv3x3_pack: bad arguments");
  const int I = transpose_flip ? Cout : Cin, O = transpose_flip ? Cin : Cout;
  BX_REQUIRE(I_p >= I && O_p >= O && I_p % 8 == 0 && O_p % 8 == 0, "bx_conv3x3_pack: padded dims I_p=%d O_p=%d must cover %d/%d and be multiples of 8", I_p, O_p, I, O);
  hipStream_t s = (hipStream_t)stream;
  const int n = 9 * I_p * O_p;
  if (packed_f32) {
    hipLaunchKernelGGL(k_pack3x3, dim3(bx_ceil_div(n, 256) > 1024 ? 1024 : bx_ceil_div(n, 256)), dim3(256), 0, s,
                       w_oihw, packed_f32, Cout, Cin, I_p, O_p, transpose_flip);
    BX_CHECK_LAUNCH("bx_conv3x3_pack");
  }
  if (packed_mfma) {
    if (!bx_conv3x3_packed_mfma_bytes(I_p, O_p)) BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3_pack: no MFMA operand layout for I_p=%d O_p=%d", I_p, O_p);
    bx_conv3x3_mfma_pack_launch(w_oihw, packed_mfma, Cout, Cin, I_p, O_p, transpose_flip, s);
    BX_CHECK_LAUNCH("bx_conv3x3_pack(mfma)");
  }
  return BX_OK;
}

// ------------------------------------------------------------------------------------------------
// Direct kernel.  One thread = 4 consecutive x pixels x 8 output channels (32 fp32 accumulators).
// Input pixels come straight from global memory as 8-channel vectors (neighbouring threads overlap
// in L1); the weight slice is wave-uniform, so hipcc keeps it in SGPRs (s_load) and the inner loop
// is pure v_fma with a scalar operand.
template <typename T>
__global__ __launch_bounds__(256) void k_conv3x3_direct(const T* __restrict__ x, const float* __restrict__ wp,
    const float* __restrict__ bias, const T* __restrict__ mask_src, const T* __restrict__ addend, T* __restrict__ y,
    int B, int H, int W, int Ci, int Co, int relu) {
  const int W4 = (W + 3) >> 2;
  const long long ngroups = (long long)B * H * W4;
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int xg = (int)(g % W4);
  const long long t = g / W4;
  const int yy = (int)(t % H), b = (int)(t / H);
  const int x0 = xg * 4, co0 = blockIdx.y * 8;

  float acc[4][8];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[p][c] = 0.f;

  for (int ci0 = 0; ci0 < Ci; ci0 += 8) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = yy + ky - 1;
      if (iy < 0 || iy >= H) continue;
      float in[6][8];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int ix = x0 + j - 1;
        if (ix >= 0 && ix < W) ld8(x, (((size_t)b * H + iy) * W + ix) * Ci + ci0, in[j]);
        else {
#pragma unroll
          for (int c = 0; c < 8; ++c) in[j][c] = 0.f;
        }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float* wt = wp + ((size_t)(ky * 3 + kx) * Ci + ci0) * Co + co0;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          float w8[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) w8[c] = wt[(size_t)ci * Co + c];
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[p][c] = fmaf(in[p + kx][ci], w8[c], acc[p][c]);
        }
      }
    }
  }
  float bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bv[c] = bias ? bias[co0 + c] : 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (x0 + p >= W) break;
    const size_t o = (((size_t)b * H + yy) * W + x0 + p) * Co + co0;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] = acc[p][c] + bv[c]; if (relu) v[c] = fmaxf(v[c], 0.f); }
    if (mask_src) {
      float m[8];
      ld8(mask_src, o, m);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = m[c] > 0.f ? v[c] : 0.f;
    }
    if (addend) {
      float a[8];
      ld8(addend, o, a);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] += a[c];
    }
    st8(y, o, v);
  }
}

static int conv3x3_impl(const void* x, const float* packed_f32, const void* packed_mfma, const float* bias,
                        const void* relu_mask_src, const void* addend, void* y,
                        int B, int H, int W, int Ci, int Co, int dtype, int flags, int algo, bxWgradPending* carry, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "bx_conv3x3: bad arguments");
  BX_REQUIRE(Ci % 8 == 0 && Co % 8 == 0, "bx_conv3x3: Ci (%d) and Co (%d) must be multiples of 8 (pad the tensors)", Ci, Co);
  hipStream_t s = (hipStream_t)stream;
  if (algo == BX_ALGO_AUTO)
    algo = (packed_mfma && conv_mfma_ok(Ci, Co, dtype)) ? BX_ALGO_MFMA : BX_ALGO_DIRECT;
  if (algo == BX_ALGO_MFMA) {
    if (!packed_mfma || !conv_mfma_ok(Ci, Co, dtype))
      BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3: MFMA path needs packed_mfma in the layout of `dtype` and Ci%%8==0, Co%%16==0 (Ci=%d Co=%d dtype=%d)", Ci, Co, dtype);
    if (dtype == BX_F32) {                                       // fp32 storage: split-bf16 operands (conv3x3_split.hip); cannot carry a reduce
      if (carry && carry->valid) {
        const int rc = bx_wgrad_mfma_finish(carry, s);
        if (rc) return rc;
      }
      return bx_conv3x3_split_launch(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, flags, s);
    }
    return bx_conv3x3_mfma_launch(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, flags, s, carry);
  }
  BX_REQUIRE(!(flags & BX_EPI_MASK_BITS), "bx_conv3x3: BX_EPI_MASK_BITS needs the MFMA path");
  if (carry && carry->valid) {                                 // the direct kernels cannot carry a reduce: finish the chain first
    const int rc = bx_wgrad_mfma_finish(carry, s);
    if (rc) return rc;
  }
  BX_REQUIRE(packed_f32, "bx_conv3x3: direct path needs packed_f32");
  const long long ngroups = (long long)B * H * ((W + 3) / 4);
  dim3 grid(bx_ceil_div(ngroups, 256), Co / 8);
  BX_DISPATCH_DTYPE(dtype, T,
    hipLaunchKernelGGL((k_conv3x3_direct<T>), grid, dim3(256), 0, s, (const T*)x, packed_f32, bias,
                       (const T*)relu_mask_src, (const T*)addend, (T*)y, B, H, W, Ci, Co, (flags & BX_EPI_RELU) ? 1 : 0));
  BX_CHECK_LAUNCH("bx_conv3x3(direct)");
  return BX_OK;
}
extern "C" int bx_conv3x3(const void* x, const float* packed_f32, const void* packed_mfma, const float* bias,
                          const void* relu_mask_src, const void* addend, void* y,
                          int B, int H, int W, int Ci, int Co, int dtype, int flags, int algo, bxStream stream) {
  return conv3x3_impl(x, packed_f32, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, dtype, flags, algo, nullptr, stream);
}
extern "C" int bx_conv3x3_carry(const void* x, const float* packed_f32, const void* packed_mfma, const float* bias,
                                const void* relu_mask_src, const void* addend, void* y,
                                int B, int H, int W, int Ci, int Co, int dtype, int flags, int algo, bxWgradPending* pending, bxStream stream) {
  BX_REQUIRE(pending, "bx_conv3x3_carry: pending is NULL");
  return conv3x3_impl(x, packed_f32, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, dtype, flags, algo, pending, stream);
}

// ------------------------------------------------------------------------------------------------
// Direct weight gradient.  A workgroup owns a 16(ci) x 16(co) output tile for all 9 taps and walks a
// contiguous range of row segments (<=64 pixels of one image row); per segment the three input rows
// (with halo) and the dZ row are staged in LDS as fp32; thread (ci,co) slides a 3x3 register window.
// Partials per chunk go to the workspace; k_wgrad_reduce sums them in fixed order (deterministic)
// and writes OIHW.
#define WG_SEG 64
template <typename T>
__global__ __launch_bounds__(256) void k_wgrad_direct(const T* __restrict__ x, const T* __restrict__ dz, float* __restrict__ partial,
    int B, int H, int W, int Ci_p, int Co, int seg, int nseg_x, long long nitems, int items_per_chunk) {
  __shared__ float sx[3][WG_SEG + 2][16];
  __shared__ float sdz[WG_SEG][16];
  const int ci = threadIdx.x & 15, co = threadIdx.x >> 4;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16;
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  float bsum = 0.f;
  const long long it0 = (long long)blockIdx.x * items_per_chunk;
  long long it1 = it0 + items_per_chunk;
  if (it1 > nitems) it1 = nitems;
  for (long long it = it0; it < it1; ++it) {
    const int sgi = (int)(it % nseg_x);
    const long long r = it / nseg_x;
    const int yy = (int)(r % H), b = (int)(r / H);
    const int x0 = sgi * seg;
    __syncthreads();
    // stage x rows: units of 8 channels; 3 rows x (seg+2) px x 2 halves
    for (int u = threadIdx.x; u < 3 * (seg + 2) * 2; u += 256) {
      const int half = u & 1, px = (u >> 1) % (seg + 2), ky = (u >> 1) / (seg + 2);
      const int iy = yy + ky - 1, ix = x0 + px - 1, c = ci0 + half * 8;
      float v[8];
      if (iy >= 0 && iy < H && ix >= 0 && ix < W && c < Ci_p) ld8(x, (((size_t)b * H + iy) * W + ix) * Ci_p + c, v);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sx[ky][px][half * 8 + j] = v[j];
    }
    for (int u = threadIdx.x; u < seg * 2; u += 256) {
      const int half = u & 1, px = u >> 1, ix = x0 + px;
      float v[8];
      if (ix < W) ld8(dz, (((size_t)b * H + yy) * W + ix) * Co + co0 + half * 8, v);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sdz[px][half * 8 + j] = v[j];
    }
    __syncthreads();
    float win[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) { win[ky][1] = sx[ky][0][ci]; win[ky][2] = sx[ky][1][ci]; }
    for (int xx = 0; xx < seg; ++xx) {
      const float d = sdz[xx][co];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2]; win[ky][2] = sx[ky][xx + 2][ci];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(win[ky][kx], d, acc[ky * 3 + kx]);
      }
      bsum += d;
    }
  }
  const size_t per_chunk = (size_t)9 * Ci_p * Co + Co;
  float* out = partial + (size_t)blockIdx.x * per_chunk;
  if (ci0 + ci < Ci_p) {
#pragma unroll
    for (int k = 0; k < 9; ++k) out[((size_t)k * Ci_p + ci0 + ci) * Co + co0 + co] = acc[k];
  }
  if (blockIdx.y == 0 && ci == 0) out[(size_t)9 * Ci_p * Co + co0 + co] = bsum;
}

__global__ void k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                               int nchunk, int Cin, int Ci_p, int Co) {
  const size_t per_chunk = (size_t)9 * Ci_p * Co + Co;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_chunk) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += partial[(size_t)c * per_chunk + idx];
  if (idx < (size_t)9 * Ci_p * Co) {
    const int co = (int)(idx % Co), ci = (int)((idx / Co) % Ci_p), t = (int)(idx / ((size_t)Co * Ci_p));
    if (ci < Cin) dw[((size_t)co * Cin + ci) * 9 + t] = s;
  } else if (db) {
    db[idx - (size_t)9 * Ci_p * Co] = s;
  }
}

static int wgrad_direct_chunks(int B, int H, int W, int Ci_p, int Co, int* seg, int* nseg_x, long long* nitems, int* ipc) {
  *seg = W < WG_SEG ? W : WG_SEG;
  *nseg_x = (W + *seg - 1) / *seg;
  *nitems = (long long)B * H * *nseg_x;
  const int tiles = ((Ci_p + 15) / 16) * (Co / 16);
  long long want = 2048 / tiles;
  if (want < 1) want = 1;
  if (want > *nitems) want = *nitems;
  *ipc = (int)((*nitems + want - 1) / want);
  return (int)((*nitems + *ipc - 1) / *ipc);
}

extern "C" size_t bx_conv3x3_wgrad_workspace(int B, int H, int W, int Ci_p, int Co, int dtype, int algo) {
  if (algo == BX_ALGO_AUTO) algo = wgrad_mfma_ok(Ci_p, Co, dtype) ? BX_ALGO_MFMA : BX_ALGO_DIRECT;
  if (algo == BX_ALGO_MFMA) return dtype == BX_F32 ? bx_wgrad_split_workspace(B, H, W, Ci_p, Co) : bx_wgrad_mfma_workspace(B, H, W, Ci_p, Co);
  int seg, nseg_x, ipc; long long nitems;
  const int nchunk = wgrad_direct_chunks(B, H, W, Ci_p, Co, &seg, &nseg_x, &nitems, &ipc);
  return (size_t)nchunk * ((size_t)9 * Ci_p * Co + Co) * sizeof(float);
}

static int wgrad_impl(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W, int Cin, int Ci_p, int Co, int dtype,
                      int algo, void* workspace, size_t workspace_bytes, bxWgradPending* pending, bxStream stream);
extern "C" int bx_conv3x3_wgrad(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W,
                                int Cin, int Ci_p, int Co, int dtype, int algo, void* workspace, size_t workspace_bytes,
                                bxStream stream) {
  return wgrad_impl(x, dz, dw_oihw, dbias, B, H, W, Cin, Ci_p, Co, dtype, algo, workspace, workspace_bytes, nullptr, stream);
}
extern "C" int bx_conv3x3_wgrad_chained(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W,
                                        int Cin, int Ci_p, int Co, int dtype, int algo, void* workspace, size_t workspace_bytes,
                                        bxWgradPending* pending, bxStream stream) {
  BX_REQUIRE(pending, "bx_conv3x3_wgrad_chained: pending is NULL");
  return wgrad_impl(x, dz, dw_oihw, dbias, B, H, W, Cin, Ci_p, Co, dtype, algo, workspace, workspace_bytes, pending, stream);
}
extern "C" int bx_conv3x3_wgrad_finish(bxWgradPending* pending, bxStream stream) {
  BX_REQUIRE(pending, "bx_conv3x3_wgrad_finish: pending is NULL");
  return bx_wgrad_mfma_finish(pending, (hipStream_t)stream);
}
static int wgrad_impl(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W, int Cin, int Ci_p, int Co, int dtype,
                      int algo, void* workspace, size_t workspace_bytes, bxWgradPending* pending, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(x && dz && dw_oihw && B > 0 && H > 0 && W > 0, "bx_conv3x3_wgrad: bad arguments");
  BX_REQUIRE(Ci_p % 8 == 0 && Co % 16 == 0 && Cin <= Ci_p, "bx_conv3x3_wgrad: need Ci_p%%8==0, Co%%16==0 (Ci_p=%d Co=%d)", Ci_p, Co);
  hipStream_t s = (hipStream_t)stream;
  if (algo == BX_ALGO_AUTO) algo = wgrad_mfma_ok(Ci_p, Co, dtype) ? BX_ALGO_MFMA : BX_ALGO_DIRECT;
  const size_t need = bx_conv3x3_wgrad_workspace(B, H, W, Ci_p, Co, dtype, algo);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_conv3x3_wgrad: workspace %zu < %zu", workspace_bytes, need);
  if (algo == BX_ALGO_MFMA) {
    if (!wgrad_mfma_ok(Ci_p, Co, dtype)) BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3_wgrad: MFMA path unsupported for Ci_p=%d Co=%d dtype=%d", Ci_p, Co, dtype);
    if (dtype == BX_F32) return bx_wgrad_split_launch(x, dz, dw_oihw, dbias, B, H, W, Cin, Ci_p, Co, workspace, workspace_bytes, pending, s);
    return bx_wgrad_mfma_launch(x, dz, dw_oihw, dbias, B, H, W, Cin, Ci_p, Co, workspace, workspace_bytes, pending, s);
  }
  if (pending && pending->valid) {                             // the direct kernels cannot carry a reduce: finish the chain first
    const int rc = bx_wgrad_mfma_finish(pending, s);
    if (rc) return rc;
  }
  int seg, nseg_x, ipc; long long nitems;
  const int nchunk = wgrad_direct_chunks(B, H, W, Ci_p, Co, &seg, &nseg_x, &nitems, &ipc);
  dim3 grid(nchunk, (Ci_p + 15) / 16, Co / 16);
  BX_DISPATCH_DTYPE(dtype, T,
    hipLaunchKernelGGL((k_wgrad_direct<T>), grid, dim3(256), 0, s, (const T*)x, (const T*)dz, (float*)workspace,
                       B, H, W, Ci_p, Co, seg, nseg_x, nitems, ipc));
  BX_CHECK_LAUNCH("bx_conv3x3_wgrad(direct)");
  const size_t per_chunk = (size_t)9 * Ci_p * Co + Co;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(bx_ceil_div(per_chunk, 256)), dim3(256), 0, s, (const float*)workspace,
                     dw_oihw, dbias, nchunk, Cin, Ci_p, Co);
  BX_CHECK_LAUNCH("bx_conv3x3_wgrad(reduce)");
  return BX_OK;
}
