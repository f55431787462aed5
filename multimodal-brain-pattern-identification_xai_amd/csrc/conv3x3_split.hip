// fp32-STORAGE 3x3 convolution on the bf16 matrix cores (round 3): forward / data gradient (k_conv_split) and weight gradient
// (k_wgrad_split).  Replaces nn.Conv2d(k=3,p=1)+F.relu of Block.forward (reference root/src/models/models.py:49-51,64-66) and
// their autograd backward for compute_dtype=float32 -- the storage type that meets north_star's 1e-3 parity bound and that
// until now ran on the VALU kernels of conv3x3.hip (18.7 ms per benchmark step).
//
// Arithmetic: every fp32 operand is split into THREE bf16 values, x = h + m + l with h = rne_bf16(x), m = rne_bf16(x - h),
// l = rne_bf16(x - h - m) (both differences are exact in fp32; three 8-bit significands cover fp32's 24: |x - h - m - l| <= 2^-26 |x|),
// and a product runs as SIX v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator -- every partial product down to 2^-18 of the
// largest:  x*w ~= (m_x*m_w + h_x*l_w + l_x*h_w) + (h_x*m_w + m_x*h_w) + h_x*h_w ;  the dropped m*l, l*m, l*l terms are <= 2^-26 |x*w|.
// The result is fp32-grade: outputs sit 1e-7 .. 1e-6 from an fp64 convolution relative to the output scale, like the VALU fp32
// kernels they replace (accumulation order is what remains; tests/test_gpu_split.py measures both against fp64).
// Round 3 first built the two-term form (hi + lo, three MFMAs): 2e-6 rms / 4-6e-6 worst per output, 500 x better than bf16 storage
// but 20 x worse than fp32 -- enough to flip ReLU / max-pool decisions whose margins |z|/S are 1e-6 (the strict fixtures are chosen
// flip-free at fp32 accuracy) and, amplified by BatchNorm's backward cancellation, to push some weight gradients past the 1e-3
// parity bound (config0: 1.14e-3).  Parity is the gate, so the path pays six MFMAs.
// Activations stay fp32 in HBM (true fp32 storage: every other kernel of the path reads / writes them as before); the split
// happens on the way into LDS, weights are split once per step by the packing launch (h image, then m, then l).
//
// k_conv_split is k_conv_mfma's structure (conv3x3_mfma.hip: 8 x TW pixel tile x 16*NC output channels per workgroup, halo tile of
// a <= 64-channel chunk staged once and reused by all 9 taps, XOR-swizzled 16-byte LDS chunks, weights in A-fragment order from
// L2, lane = one pixel x 4 consecutive output channels in the epilogue) with three LDS images and 16-byte fp32 epilogue accesses.
// k_wgrad_split: a workgroup owns a (16*MA cin) x (16*NB cout) slice of dW for all 9 taps over a contiguous range of 8 x 16 pixel
// tiles; wave w owns the 16 x 16 quadrant w % (MA*NB) (9 accumulator tiles = 36 registers) and, when MA*NB < 4, one of the
// 4/(MA*NB) interleaved K-step subsets (summed through LDS in a fixed order at the end); operands via ds_read_b64_tr_b16 from
// pixel-major swizzled tiles; partials in k_wgrad_mfma's fragment order, so k_wgrad_reduce3 and the chaining machinery apply.
#include <stdlib.h>
#include <string.h>
#include "bx_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static inline int split_ck(int Ci) { return Ci < 64 ? Ci : 64; }
static inline int split_ks(int ck) { return (9 * ck + 31) / 32; }

int bx_conv3x3_split_supported(int Ci, int Co) {
  if (!(Ci == 8 || Ci == 16 || Ci == 32 || (Ci >= 64 && Ci % 64 == 0))) return 0;
  return Co >= 16 && Co % 16 == 0;
}
// elements of ONE image (h, m or l) of the packed operand
static inline size_t split_image_elems(int I_p, int O_p) {
  const int ck = split_ck(I_p);
  return (size_t)(I_p / ck) * split_ks(ck) * O_p * 32;
}
extern "C" size_t bx_conv3x3_packed_split_bytes(int I_p, int O_p) {
  if (!bx_conv3x3_split_supported(I_p, O_p)) return 0;
  return 3 * split_image_elems(I_p, O_p) * sizeof(bf16_t);
}

// x = h + m + l (see the header comment); two values per packed conversion
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack2bf(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pack2bf(ra, rb);
  l = pack2bf(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void split8(const u32x4& a, const u32x4& b, uint4& h, uint4& m, uint4& l) {
  split2(__uint_as_float(a.x), __uint_as_float(a.y), h.x, m.x, l.x);
  split2(__uint_as_float(a.z), __uint_as_float(a.w), h.y, m.y, l.y);
  split2(__uint_as_float(b.x), __uint_as_float(b.y), h.z, m.z, l.z);
  split2(__uint_as_float(b.z), __uint_as_float(b.w), h.w, m.w, l.w);
}
// the six partial products of the header comment, smallest first: (weight image, activation image)
#define BX_SPLIT_TERMS(F) F(1, 1) F(0, 2) F(2, 0) F(0, 1) F(1, 0) F(0, 0)

// Wp[img][chunk][s][o][kk]: the index map of k_pack_mfma (conv3x3_mfma.hip); img 0 = h, 1 = m, 2 = l
__global__ __launch_bounds__(256) void k_pack_split(const float* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int I_p, int O_p, int tf,
                                                    int ck, int ks, unsigned n) {
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
    const int kk = (int)(idx & 31);
    const unsigned row = idx >> 5, so = row / (unsigned)O_p;
    const int o = (int)(row - so * (unsigned)O_p);
    const int chunk = (int)(so / (unsigned)ks), s = (int)(so - (unsigned)chunk * (unsigned)ks);
    const int q = s * 32 + kk, tap = q / ck, i = chunk * ck + q % ck;
    float v = 0.f;
    if (tap < 9) {
      if (!tf) { if (i < Cin && o < Cout) v = w[((size_t)o * Cin + i) * 9 + tap]; }
      else     { if (i < Cout && o < Cin) v = w[((size_t)i * Cin + o) * 9 + (8 - tap)]; }
    }
    uint32_t h, m, l;
    split2(v, 0.f, h, m, l);
    wp[idx] = (bf16_t)(h & 0xffffu);
    wp[(size_t)n + idx] = (bf16_t)(m & 0xffffu);
    wp[2 * (size_t)n + idx] = (bf16_t)(l & 0xffffu);
  }
}
extern "C" int bx_conv3x3_pack_split(const float* w_oihw, void* packed_split, int Cout, int Cin, int I_p, int O_p, int transpose_flip,
                                     bxStream stream) {
  BX_REQUIRE(w_oihw && packed_split && Cout > 0 && Cin > 0, "bx_conv3x3_pack_split: bad arguments");
  const int I = transpose_flip ? Cout : Cin, O = transpose_flip ? Cin : Cout;
  BX_REQUIRE(I_p >= I && O_p >= O && I_p % 8 == 0 && O_p % 8 == 0, "bx_conv3x3_pack_split: padded dims I_p=%d O_p=%d must cover %d/%d and be multiples of 8", I_p, O_p, I, O);
  if (!bx_conv3x3_split_supported(I_p, O_p)) BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3_pack_split: no split-MFMA operand layout for I_p=%d O_p=%d", I_p, O_p);
  const size_t n = split_image_elems(I_p, O_p);
  BX_REQUIRE(n < ((size_t)1 << 31), "bx_conv3x3_pack_split: operand too large");
  const int ck = split_ck(I_p);
  const int grid = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(k_pack_split, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16_t*)packed_split, Cout, Cin, I_p, O_p,
                     transpose_flip ? 1 : 0, ck, split_ks(ck), (unsigned)n);
  BX_CHECK_LAUNCH("bx_conv3x3_pack_split");
  return BX_OK;
}

template <int CK>
__device__ __forceinline__ int split_lds_chunk(int c, int p) {
  if (CK == 32) return c ^ ((p >> 1) & 3);
  if (CK == 64) return c ^ (p & 7);
  return c;
}

// ================================================================================================
// forward / data gradient
// SB (64-channel chunks): the weight fragments are SINGLE-buffered -- image t of K-step s+1 is requested into the registers of image
// t of step s right after that image's last sweep (sweeps ordered h, h, h, m, m, l on the weight side), 48 instead of 96 registers, so
// that two workgroups fit a CU (<= 256 registers) and one's MFMA sweeps cover the other's fragment and halo latencies.  Without it
// <64,4,16> took 256 + 82 registers: one wave per SIMD with every LDS / L2 latency exposed (in place it still spills at NC = 4: the
// launcher gives 64-channel-chunk layers 32 output channels per workgroup).
// W22: the four waves form a 2 x 2 grid -- wave (wp, wc) owns HALF of the workgroup's pixel tiles and HALF of its output channels
// (MPw = 2 MP tiles x NCw = NC / 2 channel tiles: the same accumulators) instead of a quarter of the pixels and all channels.  A
// weight fragment is then fetched by two waves instead of four: the pixel-split mapping pulls 3 NC KB per K-step per wave through the
// CU's 64 B/clk vector-memory path for 768 cycles of MFMA (NC = 4: 16 B/clk per wave, i.e. the whole path at one wave per SIMD --
// in-kernel the MFMA pipe sat at ~50 %); 2 x 2 halves that and doubles the (cheap) LDS pixel-fragment reads.  Measured at B = 64
// (forward / data gradient, us): 64->64 66.4 / 69.8 -> 56.6 / 57.9, 128->128 60.5 / 62.5 -> 48.4 / 50.3, 256->256 56.9 / 56.5 -> 45.8 / 48.3;
// the HBM-bound early stages (chunks of 8-32 channels) do not gain and keep the pixel-split mapping.
template <int CK, int NC, int TW, bool W22 = (CK == 64 && NC % 2 == 0), bool SB = false>
__global__ __launch_bounds__(256, SB ? 2 : 1) void k_conv_split(const float* __restrict__ x, const bf16_t* __restrict__ wp, uint32_t lo_off,
    const float* __restrict__ bias, const float* __restrict__ mask_src, const float* __restrict__ addend, float* __restrict__ y,
    int H, int W, int Ci, int Co, int relu, int tiles_x, int tiles_y, uint32_t x_bytes, uint32_t y_bytes) {
  constexpr int TH = 8, HWID = TW + 2, HH = TH + 2, CKB = CK * 2, NCH = CK / 8, KS = (9 * CK + 31) / 32;
  constexpr int MP = (W22 ? 2 : 1) * TH * TW / 64;          // 16-pixel tiles per wave
  constexpr int NCW = W22 ? NC / 2 : NC;    // 16-channel tiles per wave
  constexpr int TPR = TW / 16;              // 16-pixel tiles per tile row
  constexpr int IMG = HH * HWID * CKB;      // bytes of one LDS image (h; the m and l images follow)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int bid = (int)blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int wpix = W22 ? wave >> 1 : wave;  // which group of pixel tiles this wave owns
  const int y0 = ty * TH, x0 = tx * TW, co_base = blockIdx.y * (NC * 16) + (W22 ? (wave & 1) * NCW * 16 : 0);

  f32x4 acc[MP][NCW];
#pragma unroll
  for (int i = 0; i < MP; ++i)
#pragma unroll
    for (int n = 0; n < NCW; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = Ci / CK;
  constexpr int NU = HH * HWID * NCH, NR = (NU + 255) / 256;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  uint32_t hoff[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / NCH, c = u % NCH;
    const int iy = y0 + p / HWID - 1, ix = x0 + p % HWID - 1;
    const bool ok = u < NU && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hoff[k] = ok ? (uint32_t)((((b * H + iy) * W + ix) * Ci + c * 8) * 4) : 0x80000000u;     // out of the image: reads zeros
  }
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    u32x4 ra[NR], rb[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {          // all of a thread's loads before the first LDS write
      ra[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, hoff[k], 0, 0);
      rb[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, hoff[k] + 16u, 0, 0);
      hoff[k] += (uint32_t)(CK * 4);
    }
    const bf16_t* wchunk = wp + (size_t)chunk * KS * Co * 32;
    bf16x8 a[SB ? 1 : 2][3][NCW];           // weight fragments (h, m, l): one K-step ahead in a second set, or (SB) in place
    auto load_a1 = [&](int s, int t, bf16x8 (&d)[NCW]) {
#pragma unroll
      for (int n = 0; n < NCW; ++n)
        d[n] = *reinterpret_cast<const bf16x8*>(wchunk + ((size_t)s * Co + co_base + n * 16 + li) * 32 + 8 * g + (size_t)t * lo_off);
    };
    auto load_a = [&](int s, bf16x8 (&d)[3][NCW]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) load_a1(s, t, d[t]);
    };
    load_a(0, a[0]);
    __syncthreads();                          // previous chunk's fragment reads are done
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NU) {
        const int p = u / NCH, c = u % NCH;
        uint4 h, m, l;
        split8(ra[k], rb[k], h, m, l);
        char* dst = lds + p * CKB + 16 * split_lds_chunk<CK>(c, p);
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + IMG) = m;
        *reinterpret_cast<uint4*>(dst + 2 * IMG) = l;
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (!SB) {
        if (s + 1 < KS) load_a(s + 1, a[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);    // keep the prefetch up here (hipcc otherwise sinks the loads to their use)
      }
      const int q0 = s * 32 + 8 * g;
      int tap = q0 / CK;
      const int c = (q0 % CK) / 8;
      const bool valid = tap < 9;
      if (!valid) tap = 0;
      const int dy = tap / 3, dx = tap - 3 * dy;
      bf16x8 bv[3][MP];
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        const int t = wpix * MP + i;
        const int p = (t / TPR + dy) * HWID + (t % TPR) * 16 + li + dx;
        const char* src = lds + p * CKB + 16 * split_lds_chunk<CK>(c, p);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          bv[u][i] = *reinterpret_cast<const bf16x8*>(src + u * IMG);
          if (!valid) bv[u][i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      // six sweeps over the accumulators, smallest partial products first: consecutive MFMAs never write the same registers
#define BX_SWEEP(TA, TB) \
      _Pragma("unroll") for (int i = 0; i < MP; ++i) \
      _Pragma("unroll") for (int n = 0; n < NCW; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[SB ? 0 : (s & 1)][TA][n], bv[TB][i], acc[i][n], 0, 0, 0);
      if constexpr (SB) {                     // weight image by weight image, each re-requested for the next K-step after its last sweep
        BX_SWEEP(0, 2) BX_SWEEP(0, 1) BX_SWEEP(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) load_a1(s + 1, 0, a[0][0]);
        __builtin_amdgcn_sched_barrier(0);
        BX_SWEEP(1, 1) BX_SWEEP(1, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) load_a1(s + 1, 1, a[0][1]);
        __builtin_amdgcn_sched_barrier(0);
        BX_SWEEP(2, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) load_a1(s + 1, 2, a[0][2]);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        BX_SPLIT_TERMS(BX_SWEEP)
      }
#undef BX_SWEEP
    }
  }
  // epilogue: lane = (pixel li of tile t, output channels co_base + n*16 + 4g .. +3): one 16-byte access per operand
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)(mask_src ? mask_src : y), 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : y), 0, y_bytes, 0x00020000);
  float4 bz[NCW];
#pragma unroll
  for (int n = 0; n < NCW; ++n)
    bz[n] = bias ? *reinterpret_cast<const float4*>(bias + co_base + n * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t offs[MP];
#pragma unroll
  for (int i = 0; i < MP; ++i) {
    const int t = wpix * MP + i;
    const int oy = y0 + t / TPR, ox = x0 + (t % TPR) * 16 + li;
    offs[i] = (oy < H && ox < W) ? (uint32_t)((((b * H + oy) * W + ox) * Co + co_base + 4 * g) * 4) : 0x80000000u;
  }
  u32x4 mk[MP][NCW], ad[MP][NCW];
  if (mask_src) {
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) mk[i][n] = __builtin_amdgcn_raw_buffer_load_b128(mres, offs[i] + (uint32_t)(n * 64), 0, 0);
  }
  if (addend) {
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) ad[i][n] = __builtin_amdgcn_raw_buffer_load_b128(ares, offs[i] + (uint32_t)(n * 64), 0, 0);
  }
#pragma unroll
  for (int i = 0; i < MP; ++i) {
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
      float v[4] = {acc[i][n][0] + bz[n].x, acc[i][n][1] + bz[n].y, acc[i][n][2] + bz[n].z, acc[i][n][3] + bz[n].w};
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (mask_src) {
        const float m[4] = {__uint_as_float(mk[i][n].x), __uint_as_float(mk[i][n].y), __uint_as_float(mk[i][n].z), __uint_as_float(mk[i][n].w)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = m[r] > 0.f ? v[r] : 0.f;
      }
      if (addend) {
        v[0] += __uint_as_float(ad[i][n].x); v[1] += __uint_as_float(ad[i][n].y);
        v[2] += __uint_as_float(ad[i][n].z); v[3] += __uint_as_float(ad[i][n].w);
      }
      const u32x4 out = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      __builtin_amdgcn_raw_buffer_store_b128(out, yres, offs[i] + (uint32_t)(n * 64), 0, 0);
    }
  }
}

template <int CK, int NC, int TW>
static int launch_split(const float* x, const bf16_t* wp, const float* bias, const float* mask, const float* addend, float* y,
                        int B, int H, int W, int Ci, int Co, int relu, hipStream_t s) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 7) / 8;
  const size_t lds = (size_t)3 * 10 * (TW + 2) * CK * 2;
  static_assert((size_t)3 * 10 * (TW + 2) * CK * 2 <= 64 * 1024 + 8 * 1024 || true, "");
  if (lds > 64 * 1024) {
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute((const void*)k_conv_split<CK, NC, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)k_conv_split<CK, NC, TW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        BX_FAIL(BX_EHIP, "bx_conv3x3(split): cannot reserve %zu bytes of LDS", lds);
      attr_done = true;
    }
  }
  static const bool w22 = !(getenv("BX_SPLIT_W22") && atoi(getenv("BX_SPLIT_W22")) == 0);      // 0: the pixel-split wave mapping (A/B sweeps)
  dim3 grid((unsigned)(tiles_x * tiles_y * B), (unsigned)(Co / (16 * NC)));
  if (w22)
    hipLaunchKernelGGL((k_conv_split<CK, NC, TW>), grid, dim3(256), lds, s, x, wp, (uint32_t)split_image_elems(Ci, Co), bias, mask, addend, y,
                       H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 4), (uint32_t)((size_t)B * H * W * Co * 4));
  else
    hipLaunchKernelGGL((k_conv_split<CK, NC, TW, false>), grid, dim3(256), lds, s, x, wp, (uint32_t)split_image_elems(Ci, Co), bias, mask, addend, y,
                       H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 4), (uint32_t)((size_t)B * H * W * Co * 4));
  BX_CHECK_LAUNCH("bx_conv3x3(split mfma)");
  return BX_OK;
}
template <int CK, int TW>
static int launch_split_nc(const float* x, const bf16_t* wp, const float* bias, const float* mask, const float* addend, float* y,
                           int B, int H, int W, int Ci, int Co, int relu, hipStream_t s) {
  // output channels per workgroup: 64 while that still launches >= 512 workgroups, else 32, else 16
  const long long tiles = (long long)((W + TW - 1) / TW) * ((H + 7) / 8) * B;
  static const long long nc4_min = getenv("BX_SPLIT_NC4_MIN") ? atoll(getenv("BX_SPLIT_NC4_MIN")) : 512;
  if (Co % 64 == 0 && tiles * (Co / 64) >= nc4_min) return launch_split<CK, 4, TW>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s);
  if (Co % 32 == 0) return launch_split<CK, 2, TW>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s);
  return launch_split<CK, 1, TW>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s);
}
// fp32 storage, split-bf16 operand `packed_split` (bx_conv3x3_pack_split / pack_many with the split bit)
int bx_conv3x3_split_launch(const void* x, const void* packed_split, const float* bias, const void* relu_mask_src, const void* addend, void* y,
                            int B, int H, int W, int Ci, int Co, int flags, hipStream_t s) {
  BX_REQUIRE(bx_conv3x3_split_supported(Ci, Co), "bx_conv3x3(split): unsupported channel counts Ci=%d Co=%d", Ci, Co);
  BX_REQUIRE(!(flags & BX_EPI_MASK_BITS), "bx_conv3x3(split): BX_EPI_MASK_BITS belongs to the bf16 pair kernels");
  BX_REQUIRE((size_t)B * H * W * (Ci > Co ? Ci : Co) * 4 < ((size_t)1 << 31), "bx_conv3x3(split): an activation tensor of 2 GiB or more is not supported (B=%d H=%d W=%d)", B, H, W);
  const int relu = (flags & BX_EPI_RELU) ? 1 : 0;
  const float* xf = (const float*)x; const bf16_t* wp = (const bf16_t*)packed_split;
  const float* mk = (const float*)relu_mask_src; const float* ad = (const float*)addend; float* yf = (float*)y;
  const bool narrow = W <= 16;
  switch (split_ck(Ci)) {
    case 8:  return narrow ? launch_split_nc<8, 16>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s) : launch_split_nc<8, 32>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s);
    case 16: return narrow ? launch_split_nc<16, 16>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s) : launch_split_nc<16, 32>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s);
    case 32: return narrow ? launch_split_nc<32, 16>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s) : launch_split_nc<32, 32>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s);
    default: return launch_split_nc<64, 16>(xf, wp, bias, mk, ad, yf, B, H, W, Ci, Co, relu, s);      // 64-channel chunks: 8 x 16 tiles (staging registers)
  }
}

// ================================================================================================
// weight gradient
__device__ __forceinline__ bf16x8 split_tr_read8(const char* lds, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// byte offset of 16-byte unit c of the record of pixel (row, col) in a pixel-major tile of ROWLEN pixels per row and RB bytes per
// record.  RB = 64: the two 32-byte halves are swapped where bit 3 of col is set (k_wgrad_own's swizzle); RB = 32: the records of
// columns 8..15 of a row are permuted (col ^ 4): either way the 8 records a 32-lane group of a transposing read touches
// (columns {x..x+3, x+8..x+11}) cover all 64 banks exactly once.
template <int RB, int ROWLEN>
__device__ __forceinline__ int wsplit_unit_off(int row, int col, int c) {
  if (RB == 64) return (row * ROWLEN + col) * 64 + ((((c >> 1) ^ (col >> 3)) & 1) << 5) + (c & 1) * 16;
  return (row * ROWLEN + (col ^ (((col >> 3) & 1) << 2))) * 32 + c * 16;
}
// byte offset a lane supplies to a transposing read: 16-channel group m of pixel (row, col), 4-channel piece pc
template <int RB, int ROWLEN>
__device__ __forceinline__ int wsplit_read_off(int row, int col, int m, int pc) {
  if (RB == 64) return (row * ROWLEN + col) * 64 + (((m ^ (col >> 3)) & 1) << 5) + pc * 8;
  return (row * ROWLEN + (col ^ (((col >> 3) & 1) << 2))) * 32 + pc * 8;
}

// fixed-order sum of a layer's partials + scatter to OIHW: the arithmetic of wgrad_reduce3_body (conv3x3_mfma.hip) on the same partial
// layout ([split][slice][9*MA*NB tiles][64 lanes][4] fragments, then Co bias sums); S slices of the splits per output, combined in
// slice order through LDS.
struct WsplitRedJob { const float* partial; float* dw; float* db; int nsplit, Cin, Co, S, MA, NB, ztiles, nfrag4, nblocks; };
__device__ __forceinline__ void wsplit_reduce_body(const WsplitRedJob& jb, int bid, float4* sm) {
  const size_t per_split = (size_t)jb.nfrag4 * 4 + jb.Co;
  const int S = jb.S, NO = 256 / S;
  const int o = threadIdx.x % NO, part = threadIdx.x / NO;
  const size_t e4 = (size_t)bid * NO + o;
  const bool live = e4 * 4 < per_split;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* src = jb.partial + e4 * 4;
    for (int c = part; c < jb.nsplit; c += S) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * per_split);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && live) {
    float4 r = sm[o];
    for (int k = 1; k < S; ++k) { const float4 v = sm[k * NO + o]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
    const float rr[4] = {r.x, r.y, r.z, r.w};
    if (e4 < (size_t)jb.nfrag4) {
      const int NT = 9 * jb.MA * jb.NB;
      const int lane = (int)(e4 & 63), tl = (int)(e4 >> 6);
      const int i = tl % NT, yz = tl / NT, z = yz % jb.ztiles, y = yz / jb.ztiles;
      const int n = i % jb.NB, m = (i / jb.NB) % jb.MA, t = i / (jb.NB * jb.MA);
      const int cout = (z * jb.NB + n) * 16 + (lane & 15), cin0 = (y * jb.MA + m) * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (cin0 + j < jb.Cin) jb.dw[((size_t)cout * jb.Cin + cin0 + j) * 9 + t] = rr[j];
    } else if (jb.db) {
      const size_t b0 = (e4 - jb.nfrag4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) jb.db[b0 + j] = rr[j];
    }
  }
}

template <int MA, int NB>
__global__ __launch_bounds__(256, 2) void k_wgrad_split(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ partial,
    int H, int W, int Ci_p, int Co, int tiles_x, int tiles_y, int ntiles, int tiles_per_split, WsplitRedJob prev, int zextra) {
  constexpr int TH = 8, TW = 16, HWID = TW + 2, HH = TH + 2, CIT = 16 * MA, COT = 16 * NB, XB = CIT * 2, ZB = COT * 2;
  constexpr int Q = MA * NB, KW = 4 / Q, KSTEPS = TH * TW / 32;        // 4 K-steps of 32 pixels (two tile rows each) per tile
  constexpr int XIMG = HH * HWID * XB, ZIMG = TH * TW * ZB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // chained mode (as k_wgrad_mfma): the first `zextra` z-slices of the grid sum the PREVIOUS layer's partials, so that layer needs no
  // reduce launch of its own; these light workgroups are dispatched first and overlap with the heavy ones behind them
  if ((int)blockIdx.z < zextra) {
    const int rid = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    if (rid < prev.nblocks) wsplit_reduce_body(prev, rid, reinterpret_cast<float4*>(lds));
    return;
  }
  char* xs = lds;                            // X halo tile: images h, m, l (XIMG bytes each), then the dZ tile's three images
  char* zs = lds + 3 * XIMG;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pc = li & 3;
  const int quad = wave % Q, kpart = wave / Q, m = quad / NB, n = quad % NB;
  const int by = blockIdx.y, bz = (int)blockIdx.z - zextra;
  const int ci0 = by * CIT, co0 = bz * COT;
  const bool want_bias = by == 0;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  const int t_begin = blockIdx.x * tiles_per_split;
  const int t_end = t_begin + tiles_per_split < ntiles ? t_begin + tiles_per_split : ntiles;
  constexpr int NXU = HH * HWID * (CIT / 8), NZU = TH * TW * (COT / 8);
  constexpr int NX = (NXU + 255) / 256, NZ = (NZU + 255) / 256;
  u32x4 rxa[NX], rxb[NX], rza[NZ], rzb[NZ];
  const int nimg = ntiles / (tiles_x * tiles_y);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (uint32_t)((size_t)nimg * H * W * Ci_p * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (uint32_t)((size_t)nimg * H * W * Co * 4), 0x00020000);
  int xpp[NX], zpp[NZ];                                          // (row << 16) | (column & 0xffff) of the unit inside the tile
  uint32_t xrel[NX], zrel[NZ];
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (CIT / 8), c = u % (CIT / 8);
    const bool live = u < NXU && ci0 + c * 8 < Ci_p;
    const int px = p % HWID;
    xpp[k] = ((live ? p / HWID - 1 : -20000) << 16) | ((px - 1) & 0xffff);
    xrel[k] = (uint32_t)((((p / HWID - 1) * W + (px - 1)) * Ci_p + ci0 + c * 8) * 4);
  }
#pragma unroll
  for (int k = 0; k < NZ; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (COT / 8), c = u % (COT / 8);
    const int px = p % TW;
    zpp[k] = ((u < NZU ? p / TW : 20000) << 16) | px;
    zrel[k] = (uint32_t)((((p / TW) * W + px) * Co + co0 + c * 8) * 4);
  }
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t pix0 = (uint32_t)((b * H + y0) * W + x0);
    const uint32_t xb = pix0 * (uint32_t)(Ci_p * 4), zb = pix0 * (uint32_t)(Co * 4);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool ok = (unsigned)(y0 + (xpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (int)(short)(xpp[k] & 0xffff)) < (unsigned)W;
      const uint32_t o = ok ? xb + xrel[k] : 0x80000000u;
      rxa[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, o, 0, 0);
      rxb[k] = __builtin_amdgcn_raw_buffer_load_b128(xres, o + 16u, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = (unsigned)(y0 + (zpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (zpp[k] & 0xffff)) < (unsigned)W;
      const uint32_t o = ok ? zb + zrel[k] : 0x80000000u;
      rza[k] = __builtin_amdgcn_raw_buffer_load_b128(zres, o, 0, 0);
      rzb[k] = __builtin_amdgcn_raw_buffer_load_b128(zres, o + 16u, 0, 0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int u = threadIdx.x + k * 256, p = u / (CIT / 8), c = u % (CIT / 8);
      if (u < NXU) {
        uint4 h, m, l;
        split8(rxa[k], rxb[k], h, m, l);
        const int off = wsplit_unit_off<XB, HWID>(p / HWID, p % HWID, c);
        *reinterpret_cast<uint4*>(xs + off) = h;
        *reinterpret_cast<uint4*>(xs + XIMG + off) = m;
        *reinterpret_cast<uint4*>(xs + 2 * XIMG + off) = l;
      }
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const int u = threadIdx.x + k * 256, p = u / (COT / 8), c = u % (COT / 8);
      if (u < NZU) {
        if (want_bias) {                    // the bias gradient from the exact fp32 values (out-of-image pixels loaded zeros)
          bsum[0] += __uint_as_float(rza[k].x); bsum[1] += __uint_as_float(rza[k].y); bsum[2] += __uint_as_float(rza[k].z); bsum[3] += __uint_as_float(rza[k].w);
          bsum[4] += __uint_as_float(rzb[k].x); bsum[5] += __uint_as_float(rzb[k].y); bsum[6] += __uint_as_float(rzb[k].z); bsum[7] += __uint_as_float(rzb[k].w);
        }
        uint4 h, m, l;
        split8(rza[k], rzb[k], h, m, l);
        const int off = wsplit_unit_off<ZB, TW>(p / TW, p % TW, c);
        *reinterpret_cast<uint4*>(zs + off) = h;
        *reinterpret_cast<uint4*>(zs + ZIMG + off) = m;
        *reinterpret_cast<uint4*>(zs + 2 * ZIMG + off) = l;
      }
    }
  };
  // per-lane read offsets: pixel kp = 8g + 4h + q of a K-step (h = the two 4-row halves of a transposing read pair); a K-step spans
  // tile rows 2ks and 2ks + 1
  int xoff[2][3], zoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kp = 8 * g + 4 * h + q, r = kp >> 4, c = kp & 15;
    zoff[h] = wsplit_read_off<ZB, TW>(r, c, n, pc);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) xoff[h][dx] = wsplit_read_off<XB, HWID>(r, c + dx, m, pc);
  }
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
    stage();
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
#pragma unroll
    for (int it = 0; it < KSTEPS / KW; ++it) {
      const int r0 = 2 * (kpart + it * KW);
      bf16x8 bz[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) bz[u] = split_tr_read8(zs + u * ZIMG, zoff[0] + r0 * TW * ZB, zoff[1] + r0 * TW * ZB);
      // A operand = X^T (image TA), B operand = dZ (image TB): the six partial products TA + TB <= 2, grouped by X image so that its
      // nine tap fragments (9 x 4 registers) are read once; the nine taps of a term are independent accumulators
#pragma unroll
      for (int ta = 2; ta >= 0; --ta) {
        bf16x8 ax[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3, dx = tap % 3;
          ax[tap] = split_tr_read8(xs + ta * XIMG, xoff[0][dx] + (r0 + dy) * HWID * XB, xoff[1][dx] + (r0 + dy) * HWID * XB);
        }
#pragma unroll
        for (int tb = 2 - ta; tb >= 0; --tb)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[tap], bz[tb], acc[tap], 0, 0, 0);
      }
    }
  }
  // ---- waves that share a quadrant (KW > 1) are summed through LDS in a fixed order (K-step subset 0, 1, ...)
  if (KW > 1) {
    float4* red4 = reinterpret_cast<float4*>(lds);              // [wave - Q][9][64]
    __syncthreads();
    if (kpart > 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) red4[((wave - Q) * 9 + tap) * 64 + lane] = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
    }
    __syncthreads();
    if (kpart == 0) {
#pragma unroll
      for (int kp = 1; kp < KW; ++kp)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float4 v = red4[((kp * Q + quad - Q) * 9 + tap) * 64 + lane];
          acc[tap][0] += v.x; acc[tap][1] += v.y; acc[tap][2] += v.z; acc[tap][3] += v.w;
        }
    }
  }
  // ---- the partial: k_wgrad_mfma's fragment order [slice][tile = tap*MA*NB + m*NB + n][lane][reg], then Co bias sums
  constexpr int NT = 9 * Q;
  const int gy = (int)gridDim.y, gz = (int)gridDim.z - zextra;
  const size_t nfrag = (size_t)gy * gz * NT * 256;
  const size_t per_split = nfrag + Co;
  float* out = partial + (size_t)blockIdx.x * per_split;
  if (kpart == 0) {
    float4* out4 = reinterpret_cast<float4*>(out) + ((size_t)by * gz + bz) * NT * 64;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) out4[(tap * Q + quad) * 64 + lane] = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
  }
  if (want_bias) {                         // thread t staged channel group t % (COT/8) of every unit it handled
    float* red = reinterpret_cast<float*>(lds);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = bsum[j];
    __syncthreads();
    if ((int)threadIdx.x < COT) {
      constexpr int NG = COT / 8;
      const int cg = threadIdx.x >> 3, j = threadIdx.x & 7;
      float s = 0.f;
      for (int t = cg; t < 256; t += NG) s += red[t * 8 + j];
      out[nfrag + co0 + threadIdx.x] = s;
    }
  }
}

struct WsplitPlan { int ma, nb, tiles_x, tiles_y, ntiles, ytiles, ztiles, nsplit, tps; size_t lds; };
static WsplitPlan wsplit_plan(int B, int H, int W, int Ci_p, int Co) {
  WsplitPlan p;
  p.ma = Ci_p >= 32 ? 2 : 1;
  p.nb = Co >= 32 ? 2 : 1;
  p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8; p.ntiles = p.tiles_x * p.tiles_y * B;
  p.ytiles = (Ci_p + 16 * p.ma - 1) / (16 * p.ma); p.ztiles = Co / (16 * p.nb);
  static const int want_env = getenv("BX_WSPLIT_WANT") ? atoi(getenv("BX_WSPLIT_WANT")) : 0;
  int want = (want_env ? want_env : 1024) / (p.ytiles * p.ztiles);
  if (want < 1) want = 1;
  if (want > p.ntiles) want = p.ntiles;
  p.tps = (p.ntiles + want - 1) / want;
  p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
  const size_t stage = (size_t)3 * (10 * 18 * 32 * p.ma + 8 * 16 * 32 * p.nb);
  const size_t red = (size_t)(4 - p.ma * p.nb) * 9 * 64 * sizeof(float4);
  const size_t bias = 256 * 8 * sizeof(float);
  p.lds = stage > red ? stage : red;
  if (p.lds < bias) p.lds = bias;
  return p;
}
int bx_wgrad_split_supported(int Ci_p, int Co) {
  return Ci_p % 8 == 0 && Co % 16 == 0 && (Co < 32 || Co % 32 == 0) && (Ci_p < 32 || Ci_p % 32 == 0);
}
size_t bx_wgrad_split_workspace(int B, int H, int W, int Ci_p, int Co) {
  const WsplitPlan p = wsplit_plan(B, H, W, Ci_p, Co);
  return (size_t)p.nsplit * ((size_t)p.ytiles * p.ztiles * 9 * p.ma * p.nb * 256 + Co) * sizeof(float);
}
int bx_wgrad_mfma_finish(bxWgradPending* pd, hipStream_t s);      // conv3x3_mfma.hip: k_wgrad_reduce3 over a pending descriptor
// pending == NULL: partials are reduced right away.  Otherwise a valid *pending (previous layer) is summed by extra workgroups of THIS
// launch and *pending is overwritten with this layer's unreduced partials (bx_conv3x3_wgrad_chained's contract).
int bx_wgrad_split_launch(const void* x, const void* dz, float* dw, float* db, int B, int H, int W, int Cin, int Ci_p, int Co, void* ws,
                          size_t ws_bytes, bxWgradPending* pending, hipStream_t s) {
  const WsplitPlan p = wsplit_plan(B, H, W, Ci_p, Co);
  (void)ws_bytes;
  BX_REQUIRE(bx_wgrad_split_supported(Ci_p, Co), "bx_conv3x3_wgrad(split): unsupported channel counts Ci_p=%d Co=%d", Ci_p, Co);
  BX_REQUIRE((size_t)B * H * W * (Ci_p > Co ? Ci_p : Co) * 4 < ((size_t)1 << 31), "bx_conv3x3_wgrad(split): an activation tensor of 2 GiB or more is not supported");
  BX_REQUIRE(p.ytiles <= 65535 && p.ztiles <= 65535, "bx_conv3x3_wgrad(split): too many channel slices");
  WsplitRedJob prev;
  memset(&prev, 0, sizeof(prev));
  int zextra = 0;
  if (pending && pending->valid) {                       // the previous layer's sum rides in this launch
    BX_REQUIRE(pending->partial != ws, "bx_conv3x3_wgrad(chained): the pending partials live in this call's workspace");
    prev.partial = (const float*)pending->partial; prev.dw = pending->dw; prev.db = pending->db; prev.nsplit = pending->nsplit; prev.Cin = pending->Cin;
    prev.Co = pending->Co; prev.MA = pending->ma; prev.NB = pending->nb; prev.ztiles = pending->ztiles; prev.nfrag4 = pending->nfrag4;
    int S = 1;
    while (S < 64 && pending->nsplit > 8 * S) S *= 2;      // few workgroups, short chains of loads per thread (wgrad_reduce_slices, chained form)
    prev.S = S;
    const size_t per_split = (size_t)pending->nfrag4 * 4 + pending->Co;
    prev.nblocks = (int)((per_split / 4 + (256 / S) - 1) / (256 / S));
    zextra = (prev.nblocks + p.nsplit * p.ytiles - 1) / (p.nsplit * p.ytiles);
    pending->valid = 0;
  }
  size_t lds = p.lds < 4096 ? 4096 : p.lds;              // the reduce role's float4[256]
  dim3 grid((unsigned)p.nsplit, (unsigned)p.ytiles, (unsigned)(p.ztiles + zextra));
#define BX_WS(MA_, NB_) hipLaunchKernelGGL((k_wgrad_split<MA_, NB_>), grid, dim3(256), lds, s, (const float*)x, (const float*)dz, (float*)ws, H, W, \
                                           Ci_p, Co, p.tiles_x, p.tiles_y, p.ntiles, p.tps, prev, zextra)
  if (p.ma == 1 && p.nb == 1) BX_WS(1, 1);
  else if (p.ma == 1 && p.nb == 2) BX_WS(1, 2);
  else if (p.ma == 2 && p.nb == 1) BX_WS(2, 1);
  else BX_WS(2, 2);
#undef BX_WS
  BX_CHECK_LAUNCH("bx_conv3x3_wgrad(split mfma)");
  bxWgradPending cur;
  cur.partial = ws; cur.dw = dw; cur.db = db; cur.nsplit = p.nsplit; cur.Cin = Cin; cur.Co = Co; cur.ma = p.ma; cur.nb = p.nb;
  cur.ztiles = p.ztiles; cur.nfrag4 = p.ytiles * p.ztiles * 9 * p.ma * p.nb * 64; cur.valid = 1;
  if (pending) { *pending = cur; return BX_OK; }
  return bx_wgrad_mfma_finish(&cur, s);
}
