// MFMA implicit-GEMM 3x3 convolution for gfx950: bf16 operands, fp32 accumulate.
//
// Forward / data-gradient (k_conv_mfma):  D[cout][pixel] = sum_{tap,cin} Wm[cout][(tap,cin)] * X[(tap,cin)][pixel]
//   v_mfma_f32_16x16x32_bf16:  A = weights  (row = cout = lane&15, k = 8*(lane>>4)+j)
//                              B = input    (k = 8*(lane>>4)+j,    col = pixel = lane&15)
//                              D            (col = pixel = lane&15, row = cout = 4*(lane>>4)+reg)
//   A workgroup (4 waves) owns an 8 x TW pixel tile and 16*NC output channels.  Input channels are walked
//   in chunks of CK = min(Ci,64): the (8+2) x (TW+2) halo tile of the chunk is staged once in LDS
//   ([pixel][CK] bf16, 16-byte chunks XOR-swizzled so that every ds_read_b128 of a B fragment is
//   conflict-free) and re-read by all 9 taps -- the 9x input reuse never touches L2.  The K dimension of a
//   chunk is (tap, cin) flattened, 32 per MFMA; for CK < 32 one MFMA spans 2 or 4 taps (each lane group
//   reads its own tap's pixel).  Weights are pre-packed by k_pack_mfma in exactly the A-fragment order
//   [chunk][kstep][cout][32] and come straight from L2 (1 KiB contiguous per wave load).
//   Each lane ends up with 4 consecutive output channels of one pixel -> 8-byte NHWC stores after the fused
//   epilogue (bias, ReLU | ReLU-mask of the producing layer, skip-gradient addend).
#include <stdlib.h>
#include "bx_common.h"
#include <string.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

static inline int mfma_ck(int Ci) { return Ci < 64 ? Ci : 64; }
static inline int mfma_ks(int ck) { return (9 * ck + 31) / 32; }

int bx_conv3x3_mfma_supported(int Ci, int Co, int dtype) {
  if (dtype != BX_BF16) return 0;
  if (!(Ci == 8 || Ci == 16 || Ci == 32 || (Ci >= 64 && Ci % 64 == 0))) return 0;
  // Co == 8: the data gradient of a stage's first convolution when the network input has <= 8 (padded) channels -- only saliency /
  // integrated gradients ask for it.  The operand is packed with 16 rows (8 of them zero), the kernels store the 8 real channels.
  return (Co >= 16 && Co % 16 == 0) || (Co == 8 && (Ci == 16 || Ci == 32));
}
static inline int mfma_ow(int O_p) { return O_p < 16 ? 16 : O_p; }        // weight rows per K-step of the packed operand
extern "C" size_t bx_conv3x3_packed_mfma_bytes(int I_p, int O_p) {
  if (!bx_conv3x3_mfma_supported(I_p, O_p, BX_BF16)) return 0;
  const int ck = mfma_ck(I_p);
  return (size_t)(I_p / ck) * mfma_ks(ck) * mfma_ow(O_p) * 32 * sizeof(bf16_t);
}

// Wp[chunk][s][o][kk]:  q = s*32+kk, tap = q / CK, i = chunk*CK + q % CK
//   forward (tf=0): W[o][i][tap];  data-gradient (tf=1): W[i][o][8-tap]   (zero outside the logical ranges)
__global__ void k_pack_mfma(const float* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int I_p, int O_p, int tf,
                            int ck, int ks, size_t n) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(idx & 31);
    const int o = (int)((idx >> 5) % O_p);
    const int s = (int)((idx / ((size_t)32 * O_p)) % ks);
    const int chunk = (int)(idx / ((size_t)32 * O_p * ks));
    const int q = s * 32 + kk, tap = q / ck, i = chunk * ck + q % ck;
    float v = 0.f;
    if (tap < 9) {
      if (!tf) { if (i < Cin && o < Cout) v = w[((size_t)o * Cin + i) * 9 + tap]; }
      else     { if (i < Cout && o < Cin) v = w[((size_t)i * Cin + o) * 9 + (8 - tap)]; }
    }
    wp[idx] = f2bf(v);
  }
}
void bx_conv3x3_mfma_pack_launch(const float* w_oihw, void* packed, int Cout, int Cin, int I_p, int O_p, int tf, hipStream_t s) {
  const size_t bytes = bx_conv3x3_packed_mfma_bytes(I_p, O_p);
  if (!bytes) return;
  const size_t n = bytes / sizeof(bf16_t);
  const int ck = mfma_ck(I_p);
  const int grid = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(k_pack_mfma, dim3(grid), dim3(256), 0, s, w_oihw, (bf16_t*)packed, Cout, Cin, I_p, mfma_ow(O_p), tf, ck, mfma_ks(ck), n);
}

// Many pack jobs in ONE launch (all 3x3 convolutions of the model, forward and data-gradient operands): the job
// table lives in device memory and is built once by the host (parameter pointers are stable under FlatAdamW).
// Optional second role (workgroups >= npack): the batch's fp32 NCHW -> bf16 NHWC(Cp) conversion, which is independent of
// the packing and otherwise a launch of its own at the start of every step (lsrc == nullptr: no such workgroups).  njobs == 0
// (npack == 0): only that role -- evaluation sweeps whose weights did not change since the last pack (bx_conv3x3_pack_layout_ex).
// Optional third role (round 2): the dropout counters of the two branches advance here (thread 0 / 1 of workgroup 0; sa == nullptr:
// not wanted) -- this launch opens the training step, every consumer of the seeds is a later launch.
__global__ __launch_bounds__(256) void k_pack_mfma_many(const bxPackJob* __restrict__ jobs, int njobs, int npack, const float* __restrict__ lsrc,
                                                        bf16_t* __restrict__ ldst, int C, int Cp, int HW, int nbx,
                                                        uint64_t* sa, uint64_t* oa, uint64_t* sb, uint64_t* ob, const float* const* lslot = nullptr) {
  if (lslot) lsrc = *lslot;                                // the batch's address comes from a device slot (graph replays on caller buffers)
  if (sa && blockIdx.x == 0 && threadIdx.x < 2) {
    uint64_t* st = threadIdx.x ? sb : sa;
    uint64_t* ou = threadIdx.x ? ob : oa;
    const uint64_t v = st[0] + 1; st[0] = v; ou[0] = v;
  }
  if ((int)blockIdx.x >= npack) {
    const int lb = (int)blockIdx.x - npack, b = lb / nbx, r = (lb - b * nbx) * 256 + (int)threadIdx.x;
    if (r >= HW) return;
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int c = c0 + j < C ? c0 + j : C - 1; v[j] = lsrc[((size_t)b * C + c) * HW + r]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = c0 + j < C ? v[j] : 0.f;
      st8(ldst, ((size_t)b * HW + r) * Cp + c0, v);
    }
    return;
  }
  // job lookup in one memory round trip: lanes compare their job's first block with this block, ballot counts them
  __shared__ int sj;
  if (threadIdx.x < 64) {
    int cnt = 0;
    for (int j0 = 0; j0 < njobs; j0 += 64) {
      const int j = j0 + (int)threadIdx.x;
      const bool le = j < njobs && jobs[j].block_begin <= (int)blockIdx.x;
      cnt += __popcll(__ballot(le));
    }
    if (threadIdx.x == 0) sj = cnt - 1;
  }
  __syncthreads();
  const int j = sj;
  const bxPackJob jb = jobs[j];
  const int ck = jb.I_p < 64 ? jb.I_p : 64, ks = (9 * ck + 31) / 32;
  const int Ow = jb.O_p < 16 ? 16 : jb.O_p;               // mfma_ow
  const size_t n = (size_t)(jb.I_p / ck) * ks * Ow * 32;
  const float* __restrict__ w = (const float*)jb.w_oihw;
  bf16_t* __restrict__ wp = (bf16_t*)jb.packed_mfma;
  const int nblk = (j + 1 < njobs ? jobs[j + 1].block_begin : npack) - jb.block_begin;
  for (size_t i0 = (size_t)(blockIdx.x - jb.block_begin) * 2048 + threadIdx.x; i0 < n; i0 += (size_t)nblk * 2048) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                 // eight gathers in flight
      const size_t idx = i0 + u * 256;
      v[u] = 0.f;
      if (idx < n) {
        const unsigned i32 = (unsigned)idx, row = i32 >> 5;          // n < 2^31: 32-bit divisions (64-bit ones cost ~70 instructions each)
        const int kk = (int)(i32 & 31);
        const unsigned so = row / (unsigned)Ow;
        const int o = (int)(row - so * (unsigned)Ow);
        const int chunk = (int)(so / (unsigned)ks);
        const int s = (int)(so - (unsigned)chunk * (unsigned)ks);
        const int q = s * 32 + kk, tap = q / ck, i = chunk * ck + q % ck;
        if (tap < 9) {
          if (!(jb.transpose_flip & 1)) { if (i < jb.Cin && o < jb.Cout) v[u] = w[((size_t)o * jb.Cin + i) * 9 + tap]; }
          else                    { if (i < jb.Cout && o < jb.Cin) v[u] = w[((size_t)i * jb.Cin + o) * 9 + (8 - tap)]; }
        }
      }
    }
    if (jb.transpose_flip & 2) {              // fp32-storage operand (conv3x3_split.hip): images h, m, l with w = h + m + l
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t idx = i0 + u * 256;
        if (idx < n) {
          const bf16_t h = f2bf(v[u]);
          const float r = v[u] - bf2f(h);
          const bf16_t m = f2bf(r);
          wp[idx] = h; wp[n + idx] = m; wp[2 * n + idx] = f2bf(r - bf2f(m));
        }
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t idx = i0 + u * 256;
      if (idx < n) wp[idx] = f2bf(v[u]);
    }
  }
}
extern "C" int bx_conv3x3_pack_many(const bxPackJob* jobs_device, int njobs, int total_blocks, bxStream stream) {
  BX_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0, "bx_conv3x3_pack_many: bad arguments");
  hipLaunchKernelGGL(k_pack_mfma_many, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs, total_blocks,
                     (const float*)nullptr, (bf16_t*)nullptr, 0, 0, 0, 1, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr);
  BX_CHECK_LAUNCH("bx_conv3x3_pack_many");
  return BX_OK;
}
extern "C" int bx_conv3x3_pack_many_layout(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw, void* dst_nhwc_bf16,
                                           int B, int C, int H, int W, int Cp, bxStream stream) {
  BX_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0 && src_nchw && dst_nhwc_bf16, "bx_conv3x3_pack_many_layout: bad arguments");
  BX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && C <= Cp, "bx_conv3x3_pack_many_layout: Cp must be a multiple of 8 and >= C");
  const long long HW = (long long)H * W;
  const long long nbx = (HW + 255) / 256;
  BX_REQUIRE(HW < (1ll << 31) && nbx * B + total_blocks < (1ll << 31), "bx_conv3x3_pack_many_layout: input too large");
  hipLaunchKernelGGL(k_pack_mfma_many, dim3((unsigned)(total_blocks + nbx * B)), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs,
                     total_blocks, src_nchw, (bf16_t*)dst_nhwc_bf16, C, Cp, (int)HW, (int)nbx, (uint64_t*)nullptr, (uint64_t*)nullptr,
                     (uint64_t*)nullptr, (uint64_t*)nullptr);
  BX_CHECK_LAUNCH("bx_conv3x3_pack_many_layout");
  return BX_OK;
}
extern "C" int bx_conv3x3_pack_layout_ex(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw,
                                         const float* const* src_slot, void* dst_nhwc_bf16, int B, int C, int H, int W, int Cp, bxStream stream) {
  BX_REQUIRE(njobs >= 0 && (njobs == 0 ? total_blocks == 0 : (jobs_device && total_blocks > 0)), "bx_conv3x3_pack_layout_ex: bad job table");
  BX_REQUIRE((src_nchw || src_slot) && dst_nhwc_bf16, "bx_conv3x3_pack_layout_ex: needs a source (pointer or device slot) and a destination");
  BX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && C <= Cp, "bx_conv3x3_pack_layout_ex: Cp must be a multiple of 8 and >= C");
  const long long HW = (long long)H * W;
  const long long nbx = (HW + 255) / 256;
  BX_REQUIRE(HW < (1ll << 31) && nbx * B + total_blocks < (1ll << 31), "bx_conv3x3_pack_layout_ex: input too large");
  hipLaunchKernelGGL(k_pack_mfma_many, dim3((unsigned)(total_blocks + nbx * B)), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs,
                     total_blocks, src_slot ? (const float*)dst_nhwc_bf16 /* non-null marker, replaced in-kernel */ : src_nchw,
                     (bf16_t*)dst_nhwc_bf16, C, Cp, (int)HW, (int)nbx, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr,
                     (uint64_t*)nullptr, src_slot);
  BX_CHECK_LAUNCH("bx_conv3x3_pack_layout_ex");
  return BX_OK;
}
extern "C" int bx_conv3x3_pack_many_step(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw, void* dst_nhwc_bf16,
                                         int B, int C, int H, int W, int Cp, uint64_t* state_a, uint64_t* out_a, uint64_t* state_b,
                                         uint64_t* out_b, bxStream stream) {
  BX_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0, "bx_conv3x3_pack_many_step: bad arguments");
  BX_REQUIRE((src_nchw == nullptr) == (dst_nhwc_bf16 == nullptr), "bx_conv3x3_pack_many_step: src and dst go together");
  BX_REQUIRE(state_a && out_a && state_b && out_b && state_a != state_b, "bx_conv3x3_pack_many_step: four seed pointers, two distinct states");
  long long HW = 0, nbx = 1, extra = 0;
  if (src_nchw) {
    BX_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cp % 8 == 0 && C <= Cp, "bx_conv3x3_pack_many_step: Cp must be a multiple of 8 and >= C");
    HW = (long long)H * W;
    nbx = (HW + 255) / 256;
    extra = nbx * B;
    BX_REQUIRE(HW < (1ll << 31) && extra + total_blocks < (1ll << 31), "bx_conv3x3_pack_many_step: input too large");
  }
  hipLaunchKernelGGL(k_pack_mfma_many, dim3((unsigned)(total_blocks + extra)), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs,
                     total_blocks, src_nchw, (bf16_t*)dst_nhwc_bf16, C, Cp, (int)HW, (int)nbx, state_a, out_a, state_b, out_b);
  BX_CHECK_LAUNCH("bx_conv3x3_pack_many_step");
  return BX_OK;
}

// A pending weight-gradient partial sum (see k_wgrad_reduce3 below) can ride in a convolution launch as `nred` extra workgroups at
// the front of grid.x (channel group 0 does the work, the others leave): the last weight gradient of a Block's backward is
// followed by that layer's data gradient, whose launch then carries the sum instead of a k_wgrad_reduce3 launch of its own.
struct WgradRedJob { const float* partial; float* dw; float* db; int nsplit, Cin, Co, S, MA, NB, ztiles, nfrag4, nblocks; };
__device__ __forceinline__ void wgrad_reduce3_body(const WgradRedJob& jb, int bid, float4* sm);
static WgradRedJob wgrad_job_from(const bxWgradPending* pd, bool chained);

template <int CK>
__device__ __forceinline__ int lds_chunk(int c, int p) {
  if (CK == 32) return c ^ ((p >> 1) & 3);
  if (CK == 64) return c ^ (p & 7);
  return c;
}

// ReLU decisions as BITS (round 2): the data gradient of a layer multiplies by (y > 0) of the layer below, and reading the bf16
// activation for that costs as many bytes as the gradient itself in the HBM-bound early stages.  The forward pair kernels also write
// one byte per (pixel, group of 4 channels) -- bit r = channel 4q + r is positive -- and the data-gradient epilogues read that byte
// instead of 8 bytes of activations (BX_EPI_MASK_BITS).  Byte offset = bf16 byte offset / 8.
__device__ __forceinline__ unsigned char relu_bits(u32x2 v) {
  const uint32_t h[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
  unsigned b = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) b |= (h[r] != 0u && h[r] < 0x8000u) ? (1u << r) : 0u;
  return (unsigned char)b;
}
// lane exchanges inside a row of 16 lanes as DPP modifiers (no LDS crossbar trip): quad_perm [1,0,3,2] / [2,3,0,1], row_ror:4 / :8
template <int CTRL> __device__ __forceinline__ float dpp16(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {       // every lane of the row gets the sum of its 16 lanes (fixed order)
  v += dpp16<0xB1>(v); v += dpp16<0x4E>(v); v += dpp16<0x124>(v); v += dpp16<0x128>(v);
  return v;
}
// ---- pooled epilogue (conv3 of a Block): 2x2 pool of the values AS STORED + per-channel (sum, sum of squares) of the pooled map -------
// Accumulator layout: lane (li = pixel of the 16-pixel tile, g) holds channels co_base + n*16 + 4g .. +3 of tile t = wave*MP + i; the tile
// one image row below is i + TPR, the horizontal neighbour is lane li ^ 1.  Even-li lanes of even tile rows own a pooled pixel.
// The average adds in k_pool_stats' order ((y,x) + (y,x+1) + (y+1,x) + (y+1,x+1)) so both forms store the same bits.
template <int MP, int NC, int TPR, typename PixFn>
__device__ __forceinline__ void conv_pool_tiles(const f32x4 (&q)[MP][NC], const BxConvPoolEpi& pe, const __amdgpu_buffer_rsrc_t& pres, int Co,
                                                int co_base, PixFn pix, float (&st)[2][NC][4]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(pe.route, 0, pe.route ? pe.route_bytes : 0u, 0x00020000);
#pragma unroll
  for (int i = 0; i < MP; ++i) {
    if ((i / TPR) % 2) continue;                       // bottom rows are consumed by the tile above (MP = 2*TPR*IMGS)
    int bimg, oy, ox;
    pix(i, bimg, oy, ox);                              // image, row and column of this lane's pixel in tile i
    const bool own = !(li & 1) && (oy >> 1) < pe.Ho && (ox >> 1) < pe.Wo;
    const uint32_t po = own ? (uint32_t)(((((bimg * pe.Ho + (oy >> 1)) * pe.Wo + (ox >> 1)) * Co) + co_base + 4 * g) * 2) : 0x80000000u;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      float v[4];
      uint32_t code = 0;                                // four route nibbles (this lane's four channels), see BxConvPoolEpi.route
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t0 = q[i][n][r], b0 = q[i + TPR][n][r];
        const float t1 = dpp16<0xB1>(t0), b1 = dpp16<0xB1>(b0);          // the horizontal neighbour (lane ^ 1)
        v[r] = pe.pool == BX_POOL_MAX ? fmaxf(fmaxf(t0, t1), fmaxf(b0, b1)) : 0.25f * (t0 + t1 + b0 + b1);
        if (pe.route) {                                // (workgroup-uniform: evaluation sweeps and inference skip the nibbles)
          uint32_t nib;
          if (pe.pool == BX_POOL_MAX) {                // first maximum in row-major window order (ATen), and only a positive one passes ReLU
            nib = t0 == v[r] ? 1u : t1 == v[r] ? 2u : b0 == v[r] ? 4u : 8u;
            nib = v[r] > 0.f ? nib : 0u;
          } else {
            nib = (t0 > 0.f ? 1u : 0u) | (t1 > 0.f ? 2u : 0u) | (b0 > 0.f ? 4u : 0u) | (b1 > 0.f ? 8u : 0u);
          }
          code |= nib << (4 * r);
        }
      }
      const uint32_t lo = pack2bf(v[0], v[1]), hi = pack2bf(v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, pres, po + (uint32_t)(n * 32), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16((short)code, rres, own ? (po >> 2) + (uint32_t)(n * 8) : 0x80000000u, 0, 0);   // (zero-sized resource when not wanted)
      if (own) {                                       // statistics of the pooled values as stored
        const float w[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[0][n][r] += w[r]; st[1][n][r] += w[r] * w[r]; }
      }
    }
  }
}
// workgroup sums of st -> one row of the reduction tree; the last workgroup of the channel group finalizes its 16*NC channels
template <int NC>
__device__ __forceinline__ void conv_pool_finish(float (&st)[2][NC][4], const BxConvPoolEpi& pe, int co_base, char* lds) {
  if (!pe.want_stats) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  constexpr int CW = NC * 16;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[k][n][r] = row16_sum(st[k][n][r]);
      }
  __syncthreads();                                     // every wave is done with the halo tiles in LDS
  float* stage = reinterpret_cast<float*>(lds + BX_STAT_TREE_LDS(2));          // [4 waves][2][CW]
  if (li == 0) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int n = 0; n < NC; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(wave * 2 + k) * CW + n * 16 + 4 * g + r] = st[k][n][r];
  }
  __syncthreads();
  float val[2] = {0.f, 0.f};
  if ((int)threadIdx.x < CW) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      val[k] = ((stage[(0 + k) * CW + threadIdx.x] + stage[(2 + k) * CW + threadIdx.x]) + stage[(4 + k) * CW + threadIdx.x]) + stage[(6 + k) * CW + threadIdx.x];
  }
  double tot[2];
  if (bx_stat_tree_arrive<2>(pe.tree, blockIdx.y, co_base, blockIdx.x, val, tot, lds) && (int)threadIdx.x < CW)
    bx_bn_finalize_channel(co_base + threadIdx.x, tot[0], tot[1], pe.fin);
}
// side job of the pooled form: each workgroup transposes a slice of the block's 1x1 weights for the tail's apply kernel
__device__ __forceinline__ void conv_pool_transpose_w1x1(const BxConvPoolEpi& pe, int C) {
  const int nw = pe.Cin1_p * C, nwg = (int)(gridDim.x * gridDim.y), wg = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int per = (nw + nwg - 1) / nwg, lo = wg * per, hi = lo + per < nw ? lo + per : nw;
  for (int i = lo + (int)threadIdx.x; i < hi; i += 256) {
    const int ci = i / C, c = i - ci * C;
    pe.wT[i] = ci < pe.Cin1 ? pe.w1x1[(size_t)c * pe.Cin1 + ci] : 0.f;
  }
}

#ifdef BX_CONV_STAMPS
// Diagnostic build only (hipcc -DBX_CONV_STAMPS, tools/conv_stamps.py): shader-clock stamps of one workgroup in four.
// Slots per workgroup: [start, first halo image staged, end of chunk 0..3, epilogue stores issued].
__device__ unsigned long long bx_conv_stamps[64 * 8];
extern "C" int bx_debug_conv_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bx_conv_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -1;
}
#define BX_CSTAMP(i) do { if (cstamp_wg >= 0 && threadIdx.x == 0) bx_conv_stamps[cstamp_wg * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BX_CSTAMP(i) do { } while (0)
#endif
// IMGS (round 2): a workgroup's pixel tile may span the SAME 8 x TW window of IMGS consecutive images (their halo tiles sit one
// after the other in LDS).  The weight fragments a wave fetches per K-step then feed IMGS times as many MFMAs: the late stages
// (8x16 and 16x32 maps, 64-channel chunks) ran 4-8 MFMAs per K-step per wave and spent 470-830 cycles on each (in-kernel stamps).
// C8 (round 2): the output tensor has 8 channels while the packed operand has 16 rows (see bx_conv3x3_mfma_supported); a template
// flag, so that the ordinary instantiations keep their register allocation (as a run-time test it cost <32,2,32> one wave of occupancy)
// W22 (round 3; plain forward / data-gradient launches of the 64-channel-chunk layers): the four waves as a 2 x 2 grid -- wave (wpix, wc)
// owns HALF of the pixel tiles and HALF of the NC channel tiles (same accumulator count) instead of a quarter of the pixels and all
// channels, so a weight fragment is fetched by two waves instead of four.  The pixel-split mapping pulls NC KB per K-step per wave
// through the CU's 64 B/clk vector-memory path for 16 NC cycles of MFMA each (in-kernel stamps: 800 cycles per K-step for 2 x 256
// cycles of MFMA at two waves per SIMD); the fp32-storage kernels (conv3x3_split.hip), where it was found, gained 20 %.
template <int CK, int NC, int TW, int IMGS = 1, bool POOL = false, bool C8 = false, bool W22 = false>
__global__ __launch_bounds__(256) void k_conv_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const bf16_t* __restrict__ mask_src, const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
    int H, int W, int Ci, int Co, int relu, int tiles_x, int tiles_y, uint32_t x_bytes, BxConvPoolEpi pe, WgradRedJob red, int nred) {
  constexpr int TH = 8, HWID = TW + 2, HH = TH + 2, CKB = CK * 2, NCH = CK / 8, KS = (9 * CK + 31) / 32;
  constexpr int MP = (W22 ? 2 : 1) * IMGS * TH * TW / 64;   // 16-pixel tiles per wave
  constexpr int NCW = W22 ? NC / 2 : NC;    // 16-channel tiles per wave
  static_assert(!W22 || (NC % 2 == 0 && !POOL && !C8), "2 x 2 wave grid: even NC, plain epilogue");
  constexpr int TPI = TH * TW / 16;         // 16-pixel tiles per image
  constexpr int TPR = TW / 16;              // 16-pixel tiles per tile row
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if ((int)blockIdx.x < nred) {             // carried weight-gradient sum (nred = 0 in ordinary launches)
    if (blockIdx.y == 0 && (int)blockIdx.x < red.nblocks) wgrad_reduce3_body(red, (int)blockIdx.x, reinterpret_cast<float4*>(lds));
    return;
  }
  const int bid = (int)blockIdx.x - nred;
  const int gx = (int)gridDim.x - nred;
#ifdef BX_CONV_STAMPS
  const int cstamp_lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int cstamp_wg = (cstamp_lin % 4 == 1 && cstamp_lin / 4 < 64) ? cstamp_lin / 4 : -1;
#endif
  BX_CSTAMP(0);
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = (bid / (tiles_x * tiles_y)) * IMGS;      // first image of the tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int wpix = W22 ? wave >> 1 : wave;  // which group of pixel tiles this wave owns
  const int y0 = ty * TH, x0 = tx * TW, co_base = blockIdx.y * (NC * 16) + (W22 ? (wave & 1) * NCW * 16 : 0);
  if (POOL) conv_pool_transpose_w1x1(pe, Co);

  f32x4 acc[MP][NCW];
#pragma unroll
  for (int i = 0; i < MP; ++i)
#pragma unroll
    for (int n = 0; n < NCW; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = Ci / CK;
  const int Cw = C8 ? 16 : Co;              // rows of the packed operand (C8: rows 8..15 are zero and their results are not stored)
  // staging: every thread issues ALL of its 16-byte global loads before the first LDS write (a rolled
  // load->wait->write loop serialises one HBM round trip per iteration)
  constexpr int NU = IMGS * HH * HWID * NCH, NR = (NU + 255) / 256;
  // halo loads through a raw buffer resource (out-of-image lanes point past the end and read zeros): no divergent
  // branches, 32-bit offsets, per-thread invariants hoisted out of the chunk loop (see k_conv_mfma_p)
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  uint32_t hoff[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / NCH, c = u % NCH;
    const int img = p / (HH * HWID), pl = p - img * (HH * HWID);
    const int iy = y0 + pl / HWID - 1, ix = x0 + pl % HWID - 1;
    const bool ok = u < NU && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hoff[k] = ok ? (uint32_t)(((((b + img) * H + iy) * W + ix) * Ci + c * 8) * 2) : 0x80000000u;
  }
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    uint4 rv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, hoff[k], 0, 0);
      rv[k] = make_uint4(v.x, v.y, v.z, v.w);
      hoff[k] += (uint32_t)(CK * 2);                          // next chunk of input channels (an invalid lane stays >= 2^31)
    }
    const bf16_t* wchunk = wp + (size_t)chunk * KS * Cw * 32;
    // weight fragments run two K-steps ahead of the MFMAs that consume them (L2 latency >> one K-step)
    bf16x8 a[3][NCW];
    auto load_a = [&](int s, bf16x8 (&dst)[NCW]) {
#pragma unroll
      for (int n = 0; n < NCW; ++n)
        dst[n] = *reinterpret_cast<const bf16x8*>(wchunk + ((size_t)s * Cw + co_base + n * 16 + li) * 32 + 8 * g);
    };
    load_a(0, a[0]);
    if (KS > 1) load_a(1, a[1]);
    __syncthreads();                          // previous chunk's fragment reads are done
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NU) {
        const int p = u / NCH, c = u % NCH;
        *reinterpret_cast<uint4*>(lds + p * CKB + 16 * lds_chunk<CK>(c, p)) = rv[k];
      }
    }
    __syncthreads();
    if (chunk == 0) BX_CSTAMP(1);
    // Measured here and dropped (round 2): a pinned read-ahead ring for the pixel fragments as in k_conv_mfma_c (every shape 3-5 %
    // slower: with NC >= 2 MFMAs per fragment and 2-4 waves per SIMD the scheduler's own order already hides the LDS latency), and
    // requesting the bias before the chunk loop instead of in the epilogue (no change; 4-60 more registers).
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 2 < KS) load_a(s + 2, a[(s + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch loads up here: hipcc otherwise sinks them to their use
      const int q0 = s * 32 + 8 * g;
      int tap = q0 / CK;
      const int c = (q0 % CK) / 8;
      const bool valid = tap < 9;
      if (!valid) tap = 0;
      const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        const int t = wpix * MP + i, img = t / TPI, tl = t % TPI;
        const int p = (img * HH + tl / TPR + dy) * HWID + (tl % TPR) * 16 + li + dx;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(lds + p * CKB + 16 * lds_chunk<CK>(c, p));
        if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < NCW; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s % 3][n], bv, acc[i][n], 0, 0, 0);
      }
    }
    if (chunk < 4) BX_CSTAMP(2 + chunk);
  }
  // epilogue: lane = (pixel li of tile t, output channels co_base + n*16 + 4g .. +3).  y / mask / addend share one shape:
  // 32-bit byte offsets through buffer resources, out-of-image lanes point past the end (loads read 0, stores are dropped)
  const uint32_t y_bytes = (uint32_t)((size_t)gx / (tiles_x * tiles_y) * IMGS * H * W * Co * 2);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, y_bytes, 0x00020000);
  const bool mbits = CK <= 32 && (relu & 2);      // BX_EPI_MASK_BITS: mask_src is the byte-per-4-channels form (early stages only)
  const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)(mask_src ? mask_src : y), 0, mbits ? y_bytes / 8 : y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : y), 0, y_bytes, 0x00020000);
  const bool ch_ok = C8 ? g < 2 : true;           // false only for the zero rows of an 8-channel output
  float4 bz[NCW];
#pragma unroll
  for (int n = 0; n < NCW; ++n)
    bz[n] = bias && ch_ok ? *reinterpret_cast<const float4*>(bias + co_base + n * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t lane_off = (uint32_t)((((b * H + y0) * W + x0 + li) * Co + co_base + 4 * g) * 2);
  // Epilogue operands (ReLU mask of the data gradient, residual addend): ALL of a thread's loads are issued before the first
  // one is consumed.  Loaded inside the store loop each of the MP*NC iterations was its own memory round trip (the ISA had
  // one full `s_waitcnt vmcnt(0)` per load: 32 serial trips in the <64,4,32> data-gradient launches).  The main loop's
  // fragment registers are dead here, so the batch does not raise the kernel's register peak.
  uint32_t offs[MP][NCW];
  u32x2 mk[MP][NCW], ad[MP][NCW];
#pragma unroll
  for (int i = 0; i < MP; ++i) {
    const int t = wpix * MP + i, img = t / TPI, tl = t % TPI;
    const int oy = y0 + tl / TPR, ox = x0 + (tl % TPR) * 16 + li;
    const bool inb = oy < H && ox < W && ch_ok;
    const uint32_t orow = lane_off + (uint32_t)((((img * H + tl / TPR) * W + (tl % TPR) * 16) * Co) * 2);
#pragma unroll
    for (int n = 0; n < NCW; ++n) offs[i][n] = inb ? orow + (uint32_t)(n * 32) : 0x80000000u;
  }
  if constexpr (POOL) {          // conv3 of a Block: y = relu(conv + bias) stored, then pool + statistics from the stored values
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const float v[4] = {fmaxf(acc[i][n][0] + bz[n].x, 0.f), fmaxf(acc[i][n][1] + bz[n].y, 0.f), fmaxf(acc[i][n][2] + bz[n].z, 0.f),
                            fmaxf(acc[i][n][3] + bz[n].w, 0.f)};
        const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        if (pe.store_y) __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i][n], 0, 0);
        acc[i][n] = (f32x4){__uint_as_float(out.x << 16), __uint_as_float(out.x & 0xffff0000u), __uint_as_float(out.y << 16),
                            __uint_as_float(out.y & 0xffff0000u)};
      }
    const uint32_t p_bytes = (uint32_t)((size_t)gx / (tiles_x * tiles_y) * IMGS * pe.Ho * pe.Wo * Co * 2);
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(pe.pooled, 0, p_bytes, 0x00020000);
    float st[2][NC][4];
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[0][n][r] = st[1][n][r] = 0.f;
    conv_pool_tiles<MP, NC, TPR>(acc, pe, pres, Co, co_base, [&](int i, int& bimg, int& oy, int& ox) {
      const int t = wave * MP + i, img = t / TPI, tl = t % TPI;
      bimg = b + img; oy = y0 + tl / TPR; ox = x0 + (tl % TPR) * 16 + li;
    }, st);
    conv_pool_finish<NC>(st, pe, co_base, lds);
    return;
  }
  if (mask_src) {
    if (mbits) {
#pragma unroll
      for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int n = 0; n < NCW; ++n) mk[i][n].x = __builtin_amdgcn_raw_buffer_load_b8(mres, offs[i][n] >> 3, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int n = 0; n < NCW; ++n) mk[i][n] = __builtin_amdgcn_raw_buffer_load_b64(mres, offs[i][n], 0, 0);
    }
  }
  if (addend) {
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) ad[i][n] = __builtin_amdgcn_raw_buffer_load_b64(ares, offs[i][n], 0, 0);
  }
#pragma unroll
  for (int i = 0; i < MP; ++i) {
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
      float v[4] = {acc[i][n][0] + bz[n].x, acc[i][n][1] + bz[n].y, acc[i][n][2] + bz[n].z, acc[i][n][3] + bz[n].w};
      if (relu & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (mask_src && mbits) {
        const uint32_t m = mk[i][n].x;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (m >> r) & 1u ? v[r] : 0.f;
      } else if (mask_src) {
        const u32x2 m = mk[i][n];
        // bf16 > 0  <=>  sign bit clear and magnitude non-zero
        const uint32_t mm[4] = {m.x & 0xffffu, m.x >> 16, m.y & 0xffffu, m.y >> 16};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (mm[r] != 0u && mm[r] < 0x8000u) ? v[r] : 0.f;
      }
      if (addend) {
        const u32x2 a2 = ad[i][n];
        v[0] += __uint_as_float(a2.x << 16); v[1] += __uint_as_float(a2.x & 0xffff0000u);
        v[2] += __uint_as_float(a2.y << 16); v[3] += __uint_as_float(a2.y & 0xffff0000u);
      }
      const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i][n], 0, 0);
    }
  }
  BX_CSTAMP(6);
}

// Channel-split variant for the 64-channel-chunk layers (stages 3-5, Ci % 64 == 0, Co % 64 == 0).
// k_conv_mfma gives each wave a quarter of the workgroup's PIXELS and all of its NC*16 output channels, so the four waves fetch the
// same weight fragments (4 x NC KB per K-step through the CU's 64 B/clk vector-memory path: 128-256 cycles for 128 cycles of MFMA)
// and a weight prefetch ring costs NC*4 registers per stage.  Here a wave owns 16*NCW output channels and ALL pixel tiles of the
// workgroup: every weight fragment is fetched by exactly one wave (4x fewer bytes through the L1), the ring is NCW*4 registers per
// stage (depth 6, five K-steps ahead, across chunk boundaries), and each pixel fragment read from LDS feeds NCW MFMAs.
// LDS image: the halo rows are padded to a multiple of 8 pixels so that the bank swizzle (16-byte chunk c of pixel p at c ^ (p & 7))
// depends on (lane, dx) only -- every fragment read of a K-step is one of six per-lane base addresses plus an immediate offset.
// The next chunk's halo is fetched into registers while the current chunk is multiplied.
template <int NCW, int TW, int IMGS, bool POOL>
__global__ __launch_bounds__(256) void k_conv_mfma_c(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const bf16_t* __restrict__ mask_src, const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
    int H, int W, int Ci, int Co, int relu, int tiles_x, int tiles_y, uint32_t x_bytes, BxConvPoolEpi pe) {
  constexpr int CK = 64, TH = 8, HWR = TW + 2, HWID = (TW + 2 + 7) / 8 * 8, HH = TH + 2, CKB = CK * 2, NCH = CK / 8, KS = 18, D = 6;
  constexpr int MPT = IMGS * TH * TW / 16;  // 16-pixel tiles of the workgroup (every wave computes all of them)
  constexpr int TPI = TH * TW / 16, TPR = TW / 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = (bid / (tiles_x * tiles_y)) * IMGS;
  const int y0 = ty * TH, x0 = tx * TW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int co_base = blockIdx.y * (4 * NCW * 16) + wave * (NCW * 16);          // this WAVE's first output channel
#ifdef BX_CONV_STAMPS
  const int cstamp_lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int cstamp_wg = (cstamp_lin % 4 == 1 && cstamp_lin / 4 < 64) ? cstamp_lin / 4 : -1;
#endif
  BX_CSTAMP(0);
  if (POOL) conv_pool_transpose_w1x1(pe, Co);

  f32x4 acc[MPT][NCW];
#pragma unroll
  for (int i = 0; i < MPT; ++i)
#pragma unroll
    for (int n = 0; n < NCW; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = Ci / CK;
  constexpr int NU = IMGS * HH * HWR * NCH, NR = (NU + 255) / 256;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  uint32_t hoff[NR], lpos[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = threadIdx.x + k * 256;
    const int pr = u / NCH, c = u % NCH;
    const int img = pr / (HH * HWR), pl = pr - img * (HH * HWR);
    const int row = pl / HWR, col = pl - row * HWR;
    const int iy = y0 + row - 1, ix = x0 + col - 1;
    const bool ok = u < NU && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hoff[k] = ok ? (uint32_t)(((((b + img) * H + iy) * W + ix) * Ci + c * 8) * 2) : 0x80000000u;
    const int P = (img * HH + row) * HWID + col;
    lpos[k] = (uint32_t)(P * CKB + 16 * (c ^ (P & 7)));
  }
  // per-lane fragment base addresses: [dx][half of the chunk]
  uint32_t a0[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int h = 0; h < 2; ++h) a0[dx][h] = (uint32_t)((li + dx) * CKB + 16 * ((h * 4 + g) ^ ((li + dx) & 7)));

  uint4 rv[NR];
  auto fetch = [&]() {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, hoff[k], 0, 0);
      rv[k] = make_uint4(v.x, v.y, v.z, v.w);
      hoff[k] += (uint32_t)(CK * 2);                          // next chunk of input channels (an invalid lane stays >= 2^31)
    }
  };
  fetch();
  // weight ring: slot of global K-step gs is gs % D (KS % D == 0, so slots are compile-time inside the unrolled chunk)
  bf16x8 a[D][NCW];
  const bf16_t* wlane = wp + ((size_t)(co_base + li)) * 32 + 8 * g;
  auto load_a = [&](int gs, bf16x8 (&dst)[NCW]) {
#pragma unroll
    for (int n = 0; n < NCW; ++n) dst[n] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)gs * Co + n * 16) * 32);
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) load_a(s, a[s]);

  // Two LDS images: chunk c is multiplied out of image c & 1 while chunk c + 1's halo, fetched at the start of chunk c, is written
  // into the other image in the middle of chunk c (its loads have landed by then and the wait counts only the loads older than the
  // weight prefetches issued since).  A single image needed `s_waitcnt vmcnt(0)` at every chunk boundary, i.e. it also waited for
  // the five weight fragments just requested for the next chunk -- one exposed L2 round trip per chunk.
  constexpr int IMG_BYTES = IMGS * HH * HWID * CKB;
  auto stage_write = [&](char* img) {
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if ((int)threadIdx.x + k * 256 < NU) *reinterpret_cast<uint4*>(img + lpos[k]) = rv[k];
  };
  stage_write(lds);
  __syncthreads();
  BX_CSTAMP(1);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const char* cur = lds + (chunk & 1) * IMG_BYTES;
    char* nxt = lds + ((chunk + 1) & 1) * IMG_BYTES;
    const bool more = chunk + 1 < nchunk;
    if (more) fetch();                        // next chunk's halo rides under this chunk's MFMAs
    // Pixel fragments run R MFMA slots ahead of their use in a register ring, the order pinned with scheduling barriers: left to
    // itself hipcc sinks every ds_read to its MFMA (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, one LDS round trip per MFMA).
    constexpr int R = MPT < 8 ? MPT : 8, NIT = KS * MPT;
    bf16x8 ring[R];
    auto frag = [&](int idx) -> bf16x8 {      // idx = s * MPT + i, compile-time after unrolling
      const int s = idx / MPT, i = idx % MPT;
      const int tap = s / 2, dy = tap / 3, dx = tap - 3 * dy, half = s & 1;
      const int img = i / TPI, tl = i % TPI;
      const int off = ((img * HH + tl / TPR + dy) * HWID + (tl % TPR) * 16) * CKB;
      return *reinterpret_cast<const bf16x8*>(cur + a0[dx][half] + off);
    };
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = frag(j);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (more || s + D - 1 < KS) load_a(chunk * KS + s + D - 1, a[(s + D - 1) % D]);
      if (s == KS / 2 && more) stage_write(nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MPT; ++i) {
        const int idx = s * MPT + i;
#pragma unroll
        for (int n = 0; n < NCW; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s % D][n], ring[idx % R], acc[i][n], 0, 0, 0);
        if (idx + R < NIT) ring[idx % R] = frag(idx + R);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                          // this image may be overwritten, the other one is complete
    if (chunk < 4) BX_CSTAMP(2 + chunk);
  }
  // epilogue: lane = (pixel li of tile i, output channels co_base + n*16 + 4g .. +3), as in k_conv_mfma
  const uint32_t y_bytes = (uint32_t)((size_t)gridDim.x / (tiles_x * tiles_y) * IMGS * H * W * Co * 2);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)(mask_src ? mask_src : y), 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : y), 0, y_bytes, 0x00020000);
  float4 bz[NCW];
#pragma unroll
  for (int n = 0; n < NCW; ++n)
    bz[n] = bias ? *reinterpret_cast<const float4*>(bias + co_base + n * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t lane_off = (uint32_t)((((b * H + y0) * W + x0 + li) * Co + co_base + 4 * g) * 2);
  uint32_t offs[MPT];
#pragma unroll
  for (int i = 0; i < MPT; ++i) {
    const int img = i / TPI, tl = i % TPI;
    const int oy = y0 + tl / TPR, ox = x0 + (tl % TPR) * 16 + li;
    const bool inb = oy < H && ox < W;
    offs[i] = inb ? lane_off + (uint32_t)((((img * H + tl / TPR) * W + (tl % TPR) * 16) * Co) * 2) : 0x80000000u;
  }
  if constexpr (POOL) {
#pragma unroll
    for (int i = 0; i < MPT; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) {
        const float v[4] = {fmaxf(acc[i][n][0] + bz[n].x, 0.f), fmaxf(acc[i][n][1] + bz[n].y, 0.f), fmaxf(acc[i][n][2] + bz[n].z, 0.f),
                            fmaxf(acc[i][n][3] + bz[n].w, 0.f)};
        const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        if (pe.store_y) __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i] + (uint32_t)(n * 32), 0, 0);
        acc[i][n] = (f32x4){__uint_as_float(out.x << 16), __uint_as_float(out.x & 0xffff0000u), __uint_as_float(out.y << 16),
                            __uint_as_float(out.y & 0xffff0000u)};
      }
    const uint32_t p_bytes = (uint32_t)((size_t)gridDim.x / (tiles_x * tiles_y) * IMGS * pe.Ho * pe.Wo * Co * 2);
    const __amdgpu_buffer_rsrc_t pres = __builtin_amdgcn_make_buffer_rsrc(pe.pooled, 0, p_bytes, 0x00020000);
    float st[2][NCW][4];
#pragma unroll
    for (int n = 0; n < NCW; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[0][n][r] = st[1][n][r] = 0.f;
    conv_pool_tiles<MPT, NCW, TPR>(acc, pe, pres, Co, co_base, [&](int i, int& bimg, int& oy, int& ox) {
      const int img = i / TPI, tl = i % TPI;
      bimg = b + img; oy = y0 + tl / TPR; ox = x0 + (tl % TPR) * 16 + li;
    }, st);
    // a wave holds ALL pixels of the workgroup for its channels: the row of partial sums needs no exchange between waves
    if (pe.want_stats) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < NCW; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[k][n][r] = row16_sum(st[k][n][r]);
      if (li == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int n = 0; n < NCW; ++n)
            *reinterpret_cast<float4*>(pe.tree.rows + ((size_t)blockIdx.x * 2 + k) * pe.tree.ld + co_base + n * 16 + 4 * g) =
                make_float4(st[k][n][0], st[k][n][1], st[k][n][2], st[k][n][3]);
      }
    }
    return;
  }
  u32x2 mk[MPT][NCW], ad[MPT][NCW];
  if (mask_src) {
#pragma unroll
    for (int i = 0; i < MPT; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) mk[i][n] = __builtin_amdgcn_raw_buffer_load_b64(mres, offs[i] + (uint32_t)(n * 32), 0, 0);
  }
  if (addend) {
#pragma unroll
    for (int i = 0; i < MPT; ++i)
#pragma unroll
      for (int n = 0; n < NCW; ++n) ad[i][n] = __builtin_amdgcn_raw_buffer_load_b64(ares, offs[i] + (uint32_t)(n * 32), 0, 0);
  }
#pragma unroll
  for (int i = 0; i < MPT; ++i) {
#pragma unroll
    for (int n = 0; n < NCW; ++n) {
      float v[4] = {acc[i][n][0] + bz[n].x, acc[i][n][1] + bz[n].y, acc[i][n][2] + bz[n].z, acc[i][n][3] + bz[n].w};
      if (relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (mask_src) {
        const u32x2 m = mk[i][n];
        const uint32_t mm[4] = {m.x & 0xffffu, m.x >> 16, m.y & 0xffffu, m.y >> 16};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (mm[r] != 0u && mm[r] < 0x8000u) ? v[r] : 0.f;
      }
      if (addend) {
        const u32x2 a2 = ad[i][n];
        v[0] += __uint_as_float(a2.x << 16); v[1] += __uint_as_float(a2.x & 0xffff0000u);
        v[2] += __uint_as_float(a2.y << 16); v[3] += __uint_as_float(a2.y & 0xffff0000u);
      }
      const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i] + (uint32_t)(n * 32), 0, 0);
    }
  }
  BX_CSTAMP(6);
}

// Persistent variant for single-chunk layers (Ci == CK <= 32: the HBM-bound early stages).  A workgroup walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...; the NEXT tile's halo is fetched into registers before the current tile's
// MFMAs and epilogue, so every CU always has input loads in flight (the one-shot kernel exposes one HBM round trip per
// workgroup and relies on occupancy alone to hide it).
// Launch bound: the block-1 instantiation <16,1,32> (HBM-bound, five launches per training step) allocated 108 + 24 registers, four
// over the limit for four workgroups per CU; bounded, the compiler finds 122 without spilling: 1.560 -> 1.547 ms/step (same box).
// DL (round 2, CK <= 16 where the staged tile is linear in the thread index): halos go from HBM straight into one of THREE LDS
// images, two tiles ahead (buffer_load ... lds: no registers hold them -- 95 instead of 122 -- no LDS write instructions, one barrier
// per tile instead of two); out-of-image lanes load zeros through the same out-of-range offsets.  Not faster (see the launcher).
template <int CK, int NC, int TW, bool POOL = false, bool C8 = false, bool DL = false>
__global__ __launch_bounds__(256, (CK == 16 && NC == 1 && TW == 32 && !POOL) ? 4 : 1) void k_conv_mfma_p(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const bf16_t* __restrict__ mask_src, const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
    int H, int W, int Co, int relu, int tiles_x, int tiles_y, int ntiles, uint32_t x_bytes, BxConvPoolEpi pe, WgradRedJob red, int nred) {
  constexpr int TH = 8, HWID = TW + 2, HH = TH + 2, CKB = CK * 2, NCH = CK / 8, KS = (9 * CK + 31) / 32;
  constexpr int MP = TH * TW / 64, TPR = TW / 16;
  constexpr int NU = HH * HWID * NCH, NR = (NU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if ((int)blockIdx.x < nred) {             // carried weight-gradient sum (see k_conv_mfma)
    if (blockIdx.y == 0 && (int)blockIdx.x < red.nblocks) wgrad_reduce3_body(red, (int)blockIdx.x, reinterpret_cast<float4*>(lds));
    return;
  }
  const int first_tile = (int)blockIdx.x - nred, tile_stride = (int)gridDim.x - nred;
  const int co_base = blockIdx.y * (NC * 16);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  uint4 rv[NR];
  // Halo loads go through a raw buffer resource over x: a lane whose pixel is outside the image sets its byte offset past
  // the end and the load returns zeros -- no divergent branches and no 64-bit address arithmetic per load (the flat-pointer
  // form cost ~25 VALU instructions and two exec-mask branches per load; these kernels sit at 40-50 % VALU issue utilisation).
  // Per-thread invariants (pixel of the halo tile, byte offset relative to the tile origin) are computed once.
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  int hpy[NR], hpx[NR];
  uint32_t hrel[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / NCH, c = u % NCH;
    hpy[k] = u < NU ? p / HWID - 1 : -100000;                 // outside every image -> always masked
    hpx[k] = p % HWID - 1;
    hrel[k] = (uint32_t)(((p / HWID - 1) * W + (p % HWID - 1)) * CK + c * 8) * 2u;
  }
  constexpr int IMG_BYTES = NR * 256 * 16;        // DL: one LDS image (thread-linear, padded to whole waves)
  auto fetch = [&](int tile, int img = 0) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t tbase = (uint32_t)(((b * H + y0) * W + x0) * CK) * 2u;        // bytes; < 2^31 (checked by the launcher)
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const bool ok = (unsigned)(y0 + hpy[k]) < (unsigned)H && (unsigned)(x0 + hpx[k]) < (unsigned)W;
      const uint32_t off = ok ? tbase + hrel[k] : 0x80000000u;
      if constexpr (DL) {      // lane l of this wave lands at (wave base of trip k) + 16 l
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)(lds + img * IMG_BYTES + (k * 256 + wave * 64) * 16), 16,
                                                 (int)off, 0, 0, 0);
      } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0);
        rv[k] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  };
  const uint32_t y_bytes = (uint32_t)((size_t)(ntiles / (tiles_x * tiles_y)) * H * W * Co * 2);
  const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, y_bytes, 0x00020000);
  const bool mbits = (relu & 2) != 0;             // BX_EPI_MASK_BITS (see k_conv_mfma)
  const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)(mask_src ? mask_src : y), 0, mbits ? y_bytes / 8 : y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend : y), 0, y_bytes, 0x00020000);
  const uint32_t lane_rel = (uint32_t)((li * Co + co_base + 4 * g) * 2);
  const int Cw = C8 ? 16 : Co;                    // rows of the packed operand (see k_conv_mfma)
  const bool ch_ok = C8 ? g < 2 : true;
  float4 bz[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n)
    bz[n] = bias && ch_ok ? *reinterpret_cast<const float4*>(bias + co_base + n * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (first_tile < ntiles) fetch(first_tile);
  if constexpr (DL) { if (first_tile + tile_stride < ntiles) fetch(first_tile + tile_stride, 1); }      // two tiles ahead, three images
  // pooled form: statistics accumulate over the workgroup's tiles in registers; one reduction-tree row per workgroup at the end
  float st[2][NC][4];
  __amdgpu_buffer_rsrc_t pres = yres;
  if constexpr (POOL) {
    conv_pool_transpose_w1x1(pe, Co);
    pres = __builtin_amdgcn_make_buffer_rsrc(pe.pooled, 0, (uint32_t)((size_t)(ntiles / (tiles_x * tiles_y)) * pe.Ho * pe.Wo * Co * 2), 0x00020000);
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[0][n][r] = st[1][n][r] = 0.f;
  }
  int it = 0;
  for (int tile = first_tile; tile < ntiles; tile += tile_stride, ++it) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const char* cur = lds;
    if constexpr (DL) {
      // this wave's loads of the tile have landed (vmcnt counts in issue order: only the previous tile's MP*NC output stores, issued
      // after them, may still be in flight), then everybody's; the other image was last read before this barrier
      static_assert(MP * NC + NR <= 15, "vmcnt immediate");
      if (tile + tile_stride < ntiles) __builtin_amdgcn_s_waitcnt(0x0F70 | (MP * NC + NR));    // + the NR loads of the tile after this one
      else __builtin_amdgcn_s_waitcnt(0x0F70 | (MP * NC));
      __syncthreads();
      cur = lds + (it % 3) * IMG_BYTES;
      if (tile + 2 * tile_stride < ntiles) fetch(tile + 2 * tile_stride, (it + 2) % 3);
    } else {
    __syncthreads();                          // previous tile's fragment reads are done
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NU) {
        const int p = u / NCH, c = u % NCH;
        *reinterpret_cast<uint4*>(lds + p * CKB + 16 * lds_chunk<CK>(c, p)) = rv[k];
      }
    }
    __syncthreads();
    if (tile + tile_stride < ntiles) fetch(tile + tile_stride);
    }
    f32x4 acc[MP][NC];
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[3][NC];
    auto load_a = [&](int s, bf16x8 (&dst)[NC]) {
#pragma unroll
      for (int n = 0; n < NC; ++n)
        dst[n] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)s * Cw + co_base + n * 16 + li) * 32 + 8 * g);
    };
    load_a(0, a[0]);
    if (KS > 1) load_a(1, a[1]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 2 < KS) load_a(s + 2, a[(s + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      const int q0 = s * 32 + 8 * g;
      int tap = q0 / CK;
      const int c = (q0 % CK) / 8;
      const bool valid = tap < 9;
      if (!valid) tap = 0;
      const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        const int t = wave * MP + i;
        const int p = (t / TPR + dy) * HWID + (t % TPR) * 16 + li + dx;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(cur + p * CKB + 16 * lds_chunk<CK>(c, p));
        if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s % 3][n], bv, acc[i][n], 0, 0, 0);
      }
    }
    const uint32_t tile_o = (uint32_t)((((b * H + y0) * W + x0) * Co) * 2) + lane_rel;
    // epilogue operands: all loads of the tile before the first use (see k_conv_mfma)
    uint32_t offs[MP][NC];
    u32x2 mk[MP][NC], ad[MP][NC];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int t = wave * MP + i;
      const int oy = y0 + t / TPR, ox = x0 + (t % TPR) * 16 + li;
      const bool inb = oy < H && ox < W && ch_ok;
      const uint32_t orow = tile_o + (uint32_t)((((t / TPR) * W + (t % TPR) * 16) * Co) * 2);
#pragma unroll
      for (int n = 0; n < NC; ++n) offs[i][n] = inb ? orow + (uint32_t)(n * 32) : 0x80000000u;
    }
    if constexpr (POOL) {
#pragma unroll
      for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
          const float v[4] = {fmaxf(acc[i][n][0] + bz[n].x, 0.f), fmaxf(acc[i][n][1] + bz[n].y, 0.f), fmaxf(acc[i][n][2] + bz[n].z, 0.f),
                              fmaxf(acc[i][n][3] + bz[n].w, 0.f)};
          const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          if (pe.store_y) __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i][n], 0, 0);
          acc[i][n] = (f32x4){__uint_as_float(out.x << 16), __uint_as_float(out.x & 0xffff0000u), __uint_as_float(out.y << 16),
                              __uint_as_float(out.y & 0xffff0000u)};
        }
      conv_pool_tiles<MP, NC, TPR>(acc, pe, pres, Co, co_base, [&](int i, int& bimg, int& oy, int& ox) {
        const int t = wave * MP + i;
        bimg = b; oy = y0 + t / TPR; ox = x0 + (t % TPR) * 16 + li;
      }, st);
      continue;
    }
    if (mask_src) {
      if (mbits) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
          for (int n = 0; n < NC; ++n) mk[i][n].x = __builtin_amdgcn_raw_buffer_load_b8(mres, offs[i][n] >> 3, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
          for (int n = 0; n < NC; ++n) mk[i][n] = __builtin_amdgcn_raw_buffer_load_b64(mres, offs[i][n], 0, 0);
      }
    }
    if (addend) {
#pragma unroll
      for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int n = 0; n < NC; ++n) ad[i][n] = __builtin_amdgcn_raw_buffer_load_b64(ares, offs[i][n], 0, 0);
    }
#pragma unroll
    for (int i = 0; i < MP; ++i) {
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        float v[4] = {acc[i][n][0] + bz[n].x, acc[i][n][1] + bz[n].y, acc[i][n][2] + bz[n].z, acc[i][n][3] + bz[n].w};
        if (relu & 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (mask_src && mbits) {
          const uint32_t m = mk[i][n].x;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (m >> r) & 1u ? v[r] : 0.f;
        } else if (mask_src) {
          const u32x2 m = mk[i][n];
          const uint32_t mm[4] = {m.x & 0xffffu, m.x >> 16, m.y & 0xffffu, m.y >> 16};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (mm[r] != 0u && mm[r] < 0x8000u) ? v[r] : 0.f;
        }
        if (addend) {
          const u32x2 a2 = ad[i][n];
          v[0] += __uint_as_float(a2.x << 16); v[1] += __uint_as_float(a2.x & 0xffff0000u);
          v[2] += __uint_as_float(a2.y << 16); v[3] += __uint_as_float(a2.y & 0xffff0000u);
        }
        const u32x2 out = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(out, yres, offs[i][n], 0, 0);
      }
    }
  }
  if constexpr (POOL) conv_pool_finish<NC>(st, pe, co_base, lds);
}

// ------------------------------------------------------------------------------------------------
// Stage 1 of the spectrogram CNN: conv1 (8 padded input channels -> 16) and conv2 (16 -> 16) in ONE launch (round 2).
// Both layers are HBM-bound (153 FLOP/B ridge vs 9 / 72 FLOP/B here), and conv2's only input is conv1's output: the workgroup
// that owns an 8 x 32 tile of conv2 recomputes conv1 on the tile's (8+2) x (32+2) halo from the (8+4) x (32+4) input halo --
// 1.33 x of conv1's (negligible) MFMA work -- and keeps that result in LDS in exactly the layout k_conv_mfma_p stages its halo in,
// so conv2's main loop is the one above.  conv1's output makes no round trip through HBM: it is written once when the backward
// pass will need it (STORE1; interior pixels only, every pixel belongs to one tile) and not at all in evaluation-mode passes
// without gradients (Grad-CAM sweeps, inference).  Per image 67 MB (bf16, B=64) less to read, another 67 MB less to write when
// STORE1 is off.  Arithmetic is operation-for-operation that of the two separate launches: results are bit-identical.
template <bool STORE1>
__global__ __launch_bounds__(256, 4) void k_conv12_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp1, const float* __restrict__ bias1,
    const bf16_t* __restrict__ wp2, const float* __restrict__ bias2, bf16_t* __restrict__ y1, bf16_t* __restrict__ y2,
    unsigned char* __restrict__ m1, unsigned char* __restrict__ m2, int H, int W, int tiles_x, int tiles_y, int ntiles, uint32_t x_bytes) {
  constexpr int TH = 8, TW = 32, XW = TW + 4, XH = TH + 4, YW = TW + 2, YH = TH + 2, NX = XH * XW, NY = YH * YW;
  constexpr int NR = (NX + 255) / 256, NT1 = (NY + 15) / 16, MP = TH * TW / 64, TPR = TW / 16, KS1 = 3, KS2 = 5;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* xs = lds;                           // [NX] pixels x 16 bytes (8 channels)
  char* y1s = lds + NX * 16;                // [NY] pixels x 32 bytes (16 channels): conv2's halo tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  int hpy[NR], hpx[NR];
  uint32_t hrel[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int u = threadIdx.x + k * 256;
    hpy[k] = u < NX ? u / XW - 2 : -100000;
    hpx[k] = u % XW - 2;
    hrel[k] = (uint32_t)(((u / XW - 2) * W + (u % XW - 2)) * 8) * 2u;
  }
  uint4 rv[NR];
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t tbase = (uint32_t)(((b * H + y0) * W + x0) * 8) * 2u;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const bool ok = (unsigned)(y0 + hpy[k]) < (unsigned)H && (unsigned)(x0 + hpx[k]) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? tbase + hrel[k] : 0x80000000u, 0, 0);
      rv[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  const uint32_t y_bytes = (uint32_t)((size_t)(ntiles / (tiles_x * tiles_y)) * H * W * 16 * 2);
  const __amdgpu_buffer_rsrc_t y1res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 ? y1 : y2), 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2res = __builtin_amdgcn_make_buffer_rsrc((void*)y2, 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t m1res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 && m1 ? (void*)m1 : (void*)y2), 0, STORE1 && m1 ? y_bytes / 8 : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t m2res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 && m2 ? (void*)m2 : (void*)y2), 0, STORE1 && m2 ? y_bytes / 8 : 0u, 0x00020000);
  // both layers' weight fragments are loop-invariant: 3 + 5 K-steps x 16 bytes per lane
  bf16x8 a1[KS1], a2[KS2];
#pragma unroll
  for (int s = 0; s < KS1; ++s) a1[s] = *reinterpret_cast<const bf16x8*>(wp1 + ((size_t)s * 16 + li) * 32 + 8 * g);
#pragma unroll
  for (int s = 0; s < KS2; ++s) a2[s] = *reinterpret_cast<const bf16x8*>(wp2 + ((size_t)s * 16 + li) * 32 + 8 * g);
  const float4 b1z = *reinterpret_cast<const float4*>(bias1 + 4 * g), b2z = *reinterpret_cast<const float4*>(bias2 + 4 * g);
  const int first_tile = (int)blockIdx.x, tile_stride = (int)gridDim.x;
  if (first_tile < ntiles) fetch(first_tile);
  for (int tile = first_tile; tile < ntiles; tile += tile_stride) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    __syncthreads();                          // previous tile's fragment reads (both phases) are done
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NX) *reinterpret_cast<uint4*>(xs + u * 16) = rv[k];
    }
    __syncthreads();
    if (tile + tile_stride < ntiles) fetch(tile + tile_stride);
    // ---- conv1 on the (8+2) x (32+2) halo: 16-pixel tiles of the FLATTENED halo, tile t to wave t % 4
#pragma unroll 2
    for (int it = 0; it < (NT1 + 3) / 4; ++it) {         // two rounds at a time: their LDS reads and MFMA chains interleave (all six: spills)
      const int t = wave + 4 * it;
      if (t >= NT1) break;
      const int p = 16 * t + li, pc = p < NY ? p : NY - 1;
      const int r = pc / YW, c = pc - r * YW;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        int tap = 4 * s + g;                   // K index 8g + j of K-step s = (tap 4s + g, channel j)
        const bool valid = tap < 9;
        if (!valid) tap = 0;
        const int dy = tap / 3, dx = tap - 3 * dy;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(xs + ((r + dy) * XW + c + dx) * 16);
        if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s], bv, acc, 0, 0, 0);
      }
      const int iy = y0 - 1 + r, ix = x0 - 1 + c;
      const bool inside = p < NY && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;   // outside the image: conv2's zero padding
      const float v0 = inside ? fmaxf(acc[0] + b1z.x, 0.f) : 0.f, v1 = inside ? fmaxf(acc[1] + b1z.y, 0.f) : 0.f;
      const float v2 = inside ? fmaxf(acc[2] + b1z.z, 0.f) : 0.f, v3 = inside ? fmaxf(acc[3] + b1z.w, 0.f) : 0.f;
      const u32x2 out = {pack2bf(v0, v1), pack2bf(v2, v3)};
      if (p < NY) *reinterpret_cast<u32x2*>(y1s + p * 32 + 8 * g) = out;
      if constexpr (STORE1) {
        const bool own = inside && r >= 1 && r <= TH && c >= 1 && c <= TW;
        const uint32_t off = own ? (uint32_t)((((b * H + iy) * W + ix) * 16 + 4 * g) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(out, y1res, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8(relu_bits(out), m1res, off >> 3, 0, 0);     // (zero-sized resource when no mask is wanted: dropped)
      }
    }
    __syncthreads();
    // ---- conv2 on the 8 x 32 tile: k_conv_mfma_p<16, 1, 32>'s loop over the LDS tile just written
    f32x4 acc2[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const int q0 = s * 32 + 8 * g;
      int tap = q0 / 16;
      const int c8 = (q0 % 16) / 8;
      const bool valid = tap < 9;
      if (!valid) tap = 0;
      const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
      for (int i = 0; i < MP; ++i) {
        const int t = wave * MP + i;
        const int p = (t / TPR + dy) * YW + (t % TPR) * 16 + li + dx;
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(y1s + p * 32 + 16 * c8);
        if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s], bv, acc2[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int t = wave * MP + i;
      const int oy = y0 + t / TPR, ox = x0 + (t % TPR) * 16 + li;
      const bool inb = oy < H && ox < W;
      const uint32_t off = inb ? (uint32_t)((((b * H + oy) * W + ox) * 16 + 4 * g) * 2) : 0x80000000u;
      const u32x2 out = {pack2bf(fmaxf(acc2[i][0] + b2z.x, 0.f), fmaxf(acc2[i][1] + b2z.y, 0.f)),
                         pack2bf(fmaxf(acc2[i][2] + b2z.z, 0.f), fmaxf(acc2[i][3] + b2z.w, 0.f))};
      __builtin_amdgcn_raw_buffer_store_b64(out, y2res, off, 0, 0);
      if constexpr (STORE1) __builtin_amdgcn_raw_buffer_store_b8(relu_bits(out), m2res, off >> 3, 0, 0);
    }
  }
}
// The same for stages 2 and 3 (16 -> 32 -> 32 and 32 -> 64 -> 64 channels): both tiles sit in LDS in the one-shot kernels' swizzled
// layouts (lds_chunk<C>), weight fragments come from L2 per K-step, two K-steps ahead of their use (5 x 2 + 9 x 2 fragments and more do
// not fit the register budget as loop invariants).  ONE tile per workgroup: see the note at the tile below.
template <int C0, int C1, bool STORE1>
__global__ __launch_bounds__(256, C1 <= 32 ? 4 : 2) void k_conv12b_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp1, const float* __restrict__ bias1,
    const bf16_t* __restrict__ wp2, const float* __restrict__ bias2, bf16_t* __restrict__ y1, bf16_t* __restrict__ y2,
    unsigned char* __restrict__ m1, unsigned char* __restrict__ m2, int H, int W, int tiles_x, int tiles_y, int ntiles, uint32_t x_bytes) {
  constexpr int TH = 8, TW = 32, XW = TW + 4, XH = TH + 4, YW = TW + 2, YH = TH + 2, NX = XH * XW, NY = YH * YW;
  constexpr int NCH0 = C0 / 8, XB = C0 * 2, YB = C1 * 2, NU = NX * NCH0, NR = (NU + 255) / 256, NT1 = (NY + 15) / 16, MP = TH * TW / 64, TPR = TW / 16;
  constexpr int KS1 = (9 * C0 + 31) / 32, KS2 = (9 * C1 + 31) / 32, NC = C1 / 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* xs = lds;                           // [NX] pixels x XB bytes, 16-byte chunks swizzled as k_conv_mfma<C0> does
  char* y1s = lds + NX * XB;                // [NY] pixels x YB bytes, swizzled as k_conv_mfma<C1> does
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, x_bytes, 0x00020000);
  const uint32_t y_bytes = (uint32_t)((size_t)(ntiles / (tiles_x * tiles_y)) * H * W * C1 * 2);
  const __amdgpu_buffer_rsrc_t y1res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 ? y1 : y2), 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2res = __builtin_amdgcn_make_buffer_rsrc((void*)y2, 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t m1res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 && m1 ? (void*)m1 : (void*)y2), 0, STORE1 && m1 ? y_bytes / 8 : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t m2res = __builtin_amdgcn_make_buffer_rsrc((void*)(STORE1 && m2 ? (void*)m2 : (void*)y2), 0, STORE1 && m2 ? y_bytes / 8 : 0u, 0x00020000);
  // one tile per workgroup (no persistent loop: with one, the compiler hoists the fragment addresses of both phases out of it as loop
  // invariants and the 16 -> 32 -> 32 kernel needed 276 registers -- k_conv_mfma_p<32,2,32> has the same problem)
  const int tile = (int)blockIdx.x;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  {
    const uint32_t tbase = (uint32_t)(((b * H + y0) * W + x0) * C0) * 2u;
    uint4 rv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256, px = u / NCH0, c = u % NCH0;
      const int hy = px / XW - 2, hx = px % XW - 2;
      const bool ok = u < NU && (unsigned)(y0 + hy) < (unsigned)H && (unsigned)(x0 + hx) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? tbase + (uint32_t)((hy * W + hx) * C0 + c * 8) * 2u : 0x80000000u, 0, 0);
      rv[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256, px = u / NCH0, c = u % NCH0;
      if (u < NU) *reinterpret_cast<uint4*>(xs + px * XB + 16 * lds_chunk<C0>(c, px)) = rv[k];
    }
  }
  float4 b1z[NC], b2z[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    b1z[n] = *reinterpret_cast<const float4*>(bias1 + n * 16 + 4 * g);
    b2z[n] = *reinterpret_cast<const float4*>(bias2 + n * 16 + 4 * g);
  }
  __syncthreads();
  // ---- conv1 (C0 -> C1) on the flattened (8+2) x (32+2) halo
  for (int t = wave; t < NT1; t += 4) {
    const int p = 16 * t + li, pc = p < NY ? p : NY - 1;
    const int r = pc / YW, c = pc - r * YW;
    f32x4 acc[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a1[3][NC];
    auto load_a1 = [&](int s_, bf16x8 (&dst)[NC]) {
#pragma unroll
      for (int n = 0; n < NC; ++n) dst[n] = *reinterpret_cast<const bf16x8*>(wp1 + ((size_t)s_ * C1 + n * 16 + li) * 32 + 8 * g);
    };
    load_a1(0, a1[0]);
    if (KS1 > 1) load_a1(1, a1[1]);
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      if (s + 2 < KS1) load_a1(s + 2, a1[(s + 2) % 3]);
      const int q0 = s * 32 + 8 * g;
      int tap = q0 / C0;
      const int c8 = (q0 % C0) / 8;
      const bool valid = tap < 9;
      if (!valid) tap = 0;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int px = (r + dy) * XW + c + dx;
      bf16x8 bv = *reinterpret_cast<const bf16x8*>(xs + px * XB + 16 * lds_chunk<C0>(c8, px));
      if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[s % 3][n], bv, acc[n], 0, 0, 0);
    }
    const int iy = y0 - 1 + r, ix = x0 - 1 + c;
    const bool inside = p < NY && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const bool own = inside && r >= 1 && r <= TH && c >= 1 && c <= TW;
    const uint32_t goff = own ? (uint32_t)((((b * H + iy) * W + ix) * C1 + 4 * g) * 2) : 0x80000000u;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const float v0 = inside ? fmaxf(acc[n][0] + b1z[n].x, 0.f) : 0.f, v1 = inside ? fmaxf(acc[n][1] + b1z[n].y, 0.f) : 0.f;
      const float v2 = inside ? fmaxf(acc[n][2] + b1z[n].z, 0.f) : 0.f, v3 = inside ? fmaxf(acc[n][3] + b1z[n].w, 0.f) : 0.f;
      const u32x2 out = {pack2bf(v0, v1), pack2bf(v2, v3)};
      // channels n*16 + 4g .. +3 = 16-byte chunk 2n + g/2, second half of it when g is odd
      if (p < NY) *reinterpret_cast<u32x2*>(y1s + p * YB + 16 * lds_chunk<C1>(2 * n + (g >> 1), p) + 8 * (g & 1)) = out;
      if constexpr (STORE1) {
        __builtin_amdgcn_raw_buffer_store_b64(out, y1res, own ? goff + (uint32_t)(n * 32) : 0x80000000u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8(relu_bits(out), m1res, own ? (goff + (uint32_t)(n * 32)) >> 3 : 0x80000000u, 0, 0);
      }
    }
  }
  __syncthreads();
  // ---- conv2 (C1 -> C1) on the 8 x 32 tile: k_conv_mfma<C1, NC, 32>'s loop over the LDS tile just written
  f32x4 acc2[MP][NC];
#pragma unroll
  for (int i = 0; i < MP; ++i)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc2[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a2[3][NC];
  auto load_a2 = [&](int s_, bf16x8 (&dst)[NC]) {
#pragma unroll
    for (int n = 0; n < NC; ++n) dst[n] = *reinterpret_cast<const bf16x8*>(wp2 + ((size_t)s_ * C1 + n * 16 + li) * 32 + 8 * g);
  };
  load_a2(0, a2[0]);
  load_a2(1, a2[1]);
#pragma unroll
  for (int s = 0; s < KS2; ++s) {
    if (s + 2 < KS2) load_a2(s + 2, a2[(s + 2) % 3]);
    __builtin_amdgcn_sched_barrier(0);
    const int q0 = s * 32 + 8 * g;
    int tap = q0 / C1;
    const int c8 = (q0 % C1) / 8;
    const bool valid = tap < 9;
    if (!valid) tap = 0;
    const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
      const int t = wave * MP + i;
      const int p = (t / TPR + dy) * YW + (t % TPR) * 16 + li + dx;
      bf16x8 bv = *reinterpret_cast<const bf16x8*>(y1s + p * YB + 16 * lds_chunk<C1>(c8, p));
      if (!valid) bv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < NC; ++n) acc2[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[s % 3][n], bv, acc2[i][n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MP; ++i) {
    const int t = wave * MP + i;
    const int oy = y0 + t / TPR, ox = x0 + (t % TPR) * 16 + li;
    const bool inb = oy < H && ox < W;
    const uint32_t off = inb ? (uint32_t)((((b * H + oy) * W + ox) * C1 + 4 * g) * 2) : 0x80000000u;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const u32x2 out = {pack2bf(fmaxf(acc2[i][n][0] + b2z[n].x, 0.f), fmaxf(acc2[i][n][1] + b2z[n].y, 0.f)),
                         pack2bf(fmaxf(acc2[i][n][2] + b2z[n].z, 0.f), fmaxf(acc2[i][n][3] + b2z[n].w, 0.f))};
      __builtin_amdgcn_raw_buffer_store_b64(out, y2res, inb ? off + (uint32_t)(n * 32) : 0x80000000u, 0, 0);
      if constexpr (STORE1) __builtin_amdgcn_raw_buffer_store_b8(relu_bits(out), m2res, inb ? (off + (uint32_t)(n * 32)) >> 3 : 0x80000000u, 0, 0);
    }
  }
}
extern "C" int bx_conv3x3_pair_supported(int C0_p, int C1, int C2, int dtype) {
  static const int second = getenv("BX_CONV_PAIR2") ? atoi(getenv("BX_CONV_PAIR2")) : 2;       // 0 = stage 1 only, 1 = stages 1-2, 2 = stages 1-3
  return dtype == BX_BF16 && ((C0_p == 8 && C1 == 16 && C2 == 16) || (second >= 1 && C0_p == 16 && C1 == 32 && C2 == 32) ||
                              (second >= 2 && C0_p == 32 && C1 == 64 && C2 == 64));
}
extern "C" int bx_conv3x3_pair(const void* x, const void* packed1_mfma, const float* bias1, const void* packed2_mfma, const float* bias2,
                               void* y1, void* y2, unsigned char* mask1, unsigned char* mask2, int B, int H, int W, int C0_p, int C1, int C2,
                               int dtype, bxStream stream) {
  BX_REQUIRE(y1 || (!mask1 && !mask2), "bx_conv3x3_pair: the ReLU bit masks are written together with y1 (training passes)");
  BX_REQUIRE(x && packed1_mfma && bias1 && packed2_mfma && bias2 && y2 && B > 0 && H > 0 && W > 0, "bx_conv3x3_pair: bad arguments");
  BX_REQUIRE(bx_conv3x3_pair_supported(C0_p, C1, C2, dtype), "bx_conv3x3_pair: built for bf16 storage, 8 (padded) -> 16 -> 16, 16 -> 32 -> 32 or 32 -> 64 -> 64 channels (got %d -> %d -> %d)",
             C0_p, C1, C2);
  BX_REQUIRE((size_t)B * H * W * C1 * 2 < ((size_t)1 << 31), "bx_conv3x3_pair: an activation tensor of 2 GiB or more is not supported");
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  const long long ntiles = (long long)tiles_x * tiles_y * B;
  BX_REQUIRE(ntiles < (1ll << 31), "bx_conv3x3_pair: too many tiles");
  static const int gx_cap = getenv("BX_CONV12_GX") ? atoi(getenv("BX_CONV12_GX")) : 2048;      // sweep knob (workgroups of the stage-1 pair kernel)
  const int gx = ntiles < gx_cap ? (int)ntiles : gx_cap;
  if (C1 >= 32) {
    const size_t lds2 = (size_t)12 * 36 * C0_p * 2 + (size_t)10 * 34 * C1 * 2;
    const uint32_t xb2 = (uint32_t)((size_t)B * H * W * C0_p * 2);
    auto go = [&](auto kern) -> int {
      static bool attr_done = false;                       // (one static per instantiation of this lambda's call operator)
      if (lds2 > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
          BX_FAIL(BX_EHIP, "bx_conv3x3_pair: cannot reserve %zu bytes of LDS", lds2);
        attr_done = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), lds2, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)packed1_mfma, bias1,
                         (const bf16_t*)packed2_mfma, bias2, (bf16_t*)y1, (bf16_t*)y2, y1 ? mask1 : (unsigned char*)nullptr,
                         y1 ? mask2 : (unsigned char*)nullptr, H, W, tiles_x, tiles_y, (int)ntiles, xb2);
      BX_CHECK_LAUNCH("bx_conv3x3_pair (stages 2-3)");
      return BX_OK;
    };
    if (C1 == 32) return y1 ? go(k_conv12b_mfma<16, 32, true>) : go(k_conv12b_mfma<16, 32, false>);
    return y1 ? go(k_conv12b_mfma<32, 64, true>) : go(k_conv12b_mfma<32, 64, false>);
  }
  const size_t lds = (size_t)12 * 36 * 16 + (size_t)10 * 34 * 32;
  const uint32_t xb = (uint32_t)((size_t)B * H * W * 8 * 2);
  if (y1)
    hipLaunchKernelGGL((k_conv12_mfma<true>), dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)packed1_mfma, bias1,
                       (const bf16_t*)packed2_mfma, bias2, (bf16_t*)y1, (bf16_t*)y2, mask1, mask2, H, W, tiles_x, tiles_y, (int)ntiles, xb);
  else
    hipLaunchKernelGGL((k_conv12_mfma<false>), dim3((unsigned)gx), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)packed1_mfma, bias1,
                       (const bf16_t*)packed2_mfma, bias2, (bf16_t*)nullptr, (bf16_t*)y2, (unsigned char*)nullptr, (unsigned char*)nullptr, H, W, tiles_x, tiles_y, (int)ntiles, xb);
  BX_CHECK_LAUNCH("bx_conv3x3_pair");
  return BX_OK;
}

// K-split variant for Ci >= 64 (the MFMA-bound late stages, where maps are small and a workgroup-per-tile kernel is a
// serial chain stage -> 18*nchunk K-steps -> epilogue on too few workgroups).  A workgroup owns 4 x 16 pixels x 64 output
// channels; ALL input chunks of its (4+2) x (16+2) halo sit in LDS at once and wave w takes K-steps w, w+4, w+8, ...
// (a K-step = 32 of the 9*Ci reduction elements), so each wave's dependency chain is a quarter as long and every weight
// fragment is loaded by exactly one wave.  The four partial accumulators meet in LDS (one barrier); wave r then finishes
// pixel row r (bias / ReLU / mask / addend epilogue, 8-byte NHWC stores).
template <int NCHUNK>
__global__ __launch_bounds__(256) void k_conv_mfma_ks(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
    const bf16_t* __restrict__ mask_src, const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
    int H, int W, int Co, int relu, int tiles_x, int tiles_y) {
  constexpr int CK = 64, TH = 4, TW = 16, HWID = TW + 2, HH = TH + 2, NPIX = HH * HWID, CKB = CK * 2, NCH = CK / 8, KS = 18, NC = 4;
  constexpr int Ci = CK * NCHUNK, KTOT = KS * NCHUNK;
  constexpr int NU = NPIX * NCH * NCHUNK, NR = (NU + 255) / 256;
  constexpr int CHUNK_BYTES = NPIX * CKB;
  extern __shared__ __attribute__((aligned(16))) char lds[];      // max(NCHUNK * CHUNK_BYTES, 64 KiB reduce buffer)
  const int bid = blockIdx.x;
  const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * TW, co_base = blockIdx.y * (NC * 16);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;

  {  // stage every chunk of the halo tile: all loads first, then the LDS writes
    uint4 rv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      const int c = u % NCH, p = (u / NCH) % NPIX, ch = u / (NCH * NPIX);
      const int iy = y0 + p / HWID - 1, ix = x0 + p % HWID - 1;
      rv[k] = make_uint4(0u, 0u, 0u, 0u);
      if (u < NU && iy >= 0 && iy < H && ix >= 0 && ix < W)
        rv[k] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * Ci + ch * CK + c * 8);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NU) {
        const int c = u % NCH, p = (u / NCH) % NPIX, ch = u / (NCH * NPIX);
        *reinterpret_cast<uint4*>(lds + ch * CHUNK_BYTES + p * CKB + 16 * lds_chunk<CK>(c, p)) = rv[k];
      }
    }
  }
  f32x4 acc[TH][NC];
#pragma unroll
  for (int i = 0; i < TH; ++i)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3][NC];
  auto load_a = [&](int ks, bf16x8 (&dst)[NC]) {      // packed layout [chunk][s][Co][32] == [ks][Co][32]
#pragma unroll
    for (int n = 0; n < NC; ++n)
      dst[n] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)ks * Co + co_base + n * 16 + li) * 32 + 8 * g);
  };
  constexpr int NIT = (KTOT + 3) / 4;                  // K-steps per wave (the last one may be empty for some waves)
  load_a(wave, a[0]);
  if (NIT > 1 && wave + 4 < KTOT) load_a(wave + 4, a[1]);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ks = wave + 4 * it;
    if (it + 2 < NIT && ks + 8 < KTOT) load_a(ks + 8, a[(it + 2) % 3]);
    __builtin_amdgcn_sched_barrier(0);
    if (ks < KTOT) {
      const int chunk = ks / KS, s = ks % KS;
      const int tap = s >> 1, c = (s & 1) * 4 + g;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const char* base = lds + chunk * CHUNK_BYTES;
#pragma unroll
      for (int i = 0; i < TH; ++i) {
        const int p = (i + dy) * HWID + li + dx;
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(base + p * CKB + 16 * lds_chunk<CK>(c, p));
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[it % 3][n], bv, acc[i][n], 0, 0, 0);
      }
    }
  }
  // cross-wave reduction: red[wave][row i][n][reg r][lane]
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int i = 0; i < TH; ++i)
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * TH + i) * NC + n) * 4 + r) * 64 + lane] = acc[i][n][r];
  __syncthreads();
  const int i = wave;                                   // this wave finishes pixel row i
  const int oy = y0 + i, ox = x0 + li;
  if (oy >= H || ox >= W) return;
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int co = co_base + n * 16 + 4 * g;
    const size_t o = (((size_t)b * H + oy) * W + ox) * Co + co;
    const float4 bz = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    float v[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += red[(((w * TH + i) * NC + n) * 4 + r) * 64 + lane];
    if (relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (mask_src) {
      const uint2 mk = *reinterpret_cast<const uint2*>(mask_src + o);
      const uint32_t mm[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (mm[r] != 0u && mm[r] < 0x8000u) ? v[r] : 0.f;
    }
    if (addend) {
      const uint2 a2 = *reinterpret_cast<const uint2*>(addend + o);
      v[0] += __uint_as_float(a2.x << 16); v[1] += __uint_as_float(a2.x & 0xffff0000u);
      v[2] += __uint_as_float(a2.y << 16); v[3] += __uint_as_float(a2.y & 0xffff0000u);
    }
    uint2 out;
    out.x = pack2bf(v[0], v[1]);
    out.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(y + o) = out;
  }
}
template <int NCHUNK>
static int launch_conv_ks(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                          int B, int H, int W, int Co, int relu, hipStream_t s) {
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 3) / 4;
  size_t lds = (size_t)NCHUNK * 6 * 18 * 128;
  if (lds < 65536) lds = 65536;
  dim3 grid((unsigned)(tiles_x * tiles_y * B), (unsigned)(Co / 64));
  hipLaunchKernelGGL((k_conv_mfma_ks<NCHUNK>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)mask,
                     (const bf16_t*)addend, (bf16_t*)y, H, W, Co, relu, tiles_x, tiles_y);
  BX_CHECK_LAUNCH("bx_conv3x3(mfma k-split)");
  return BX_OK;
}

// the pooled form's reduction tree: one row per workgroup of a channel group; checks the caller's buffers against the grid chosen here
template <int NC>
static int pool_tree_shape(BxConvPoolEpi* pe, int grid_x, int ygroups, int Co) {
  pe->tree.nrows = grid_x; pe->tree.Cw = NC * 16; pe->tree.ld = Co;
  bx_stat_tree_shape(grid_x, &pe->tree.G, &pe->tree.ngroups);
  if (!pe->want_stats) return BX_OK;
  if (grid_x > pe->tree_max_rows) pe->tree.cnt = nullptr;
  BX_REQUIRE((size_t)grid_x * 2 * Co <= pe->rows_cap_floats, "bx_conv3x3(pooled): %d partial rows exceed the workspace", grid_x);
  BX_REQUIRE(!pe->tree.cnt || ygroups * (pe->tree.ngroups + 1) <= pe->cnt_cap_words, "bx_conv3x3(pooled): %d x %d reduction groups exceed the sync words",
             ygroups, pe->tree.ngroups);
  return BX_OK;
}
template <int CK, int NC, int TW>
static int launch_conv(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                       int B, int H, int W, int Ci, int Co, int relu, hipStream_t s, BxConvPoolEpi* pe = nullptr, const WgradRedJob* red = nullptr) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 7) / 8;
  size_t lds = (size_t)10 * (TW + 2) * CK * 2;
  const BxConvPoolEpi none = {};
  const WgradRedJob nored = {};
  const int nred = red && !pe ? red->nblocks : 0;          // workgroups in front of grid.x that sum a pending weight gradient
  const WgradRedJob& rj = nred ? *red : nored;
  if (nred && lds < 4096) lds = 4096;
  if (pe) {
    const size_t need = BX_STAT_TREE_LDS(2) + (size_t)4 * 2 * NC * 16 * sizeof(float);       // conv_pool_finish reuses the halo tile's LDS
    if (lds < need) lds = need;
  }
  if (Co < 16) {                                   // 8-channel output (bx_conv3x3_mfma_supported: Ci 16 or 32, so NC == 1 and one chunk)
    if constexpr (NC == 1 && (CK == 16 || CK == 32)) {
      BX_REQUIRE(Co == 8 && Ci == CK && !pe && !bias, "bx_conv3x3(mfma): the 8-channel output form is a plain data gradient (Ci=%d Co=%d)", Ci, Co);
      const int ntiles = tiles_x * tiles_y * B;
      const int gx = ntiles < 2048 ? ntiles : 2048;
      if (ntiles >= 4 * gx && (size_t)B * H * W * CK * 2 < ((size_t)1 << 31)) {
        hipLaunchKernelGGL((k_conv_mfma_p<CK, 1, TW, false, true>), dim3((unsigned)(gx + nred), 1u), dim3(256), lds, s, (const bf16_t*)x,
                           (const bf16_t*)wp, bias, (const bf16_t*)mask, (const bf16_t*)addend, (bf16_t*)y, H, W, Co, relu, tiles_x, tiles_y, ntiles,
                           (uint32_t)((size_t)B * H * W * CK * 2), none, rj, nred);
        BX_CHECK_LAUNCH("bx_conv3x3(mfma persistent, 8 output channels)");
        return BX_OK;
      }
      hipLaunchKernelGGL((k_conv_mfma<CK, 1, TW, 1, false, true>), dim3((unsigned)(ntiles + nred), 1u), dim3(256), lds, s, (const bf16_t*)x,
                         (const bf16_t*)wp, bias, (const bf16_t*)mask, (const bf16_t*)addend, (bf16_t*)y, H, W, Ci, Co, relu, tiles_x, tiles_y,
                         (uint32_t)((size_t)B * H * W * Ci * 2), none, rj, nred);
      BX_CHECK_LAUNCH("bx_conv3x3(mfma, 8 output channels)");
      return BX_OK;
    }
    BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3(mfma): %d output channels with Ci=%d", Co, Ci);
  }
  if constexpr (CK <= 32) if (Ci == CK) {
    const int ntiles = tiles_x * tiles_y * B, ygroups = Co / (16 * NC);
    static const int pgx = getenv("BX_CONV_PGX") ? atoi(getenv("BX_CONV_PGX")) : 2048;
    static const int pmin = getenv("BX_CONV_PMIN") ? atoi(getenv("BX_CONV_PMIN")) : 4;
    int gx = pgx / ygroups;                        // ~8 workgroups per CU in total, each walking ntiles/gx tiles
    if (gx > ntiles) gx = ntiles;
    if (ntiles >= pmin * gx && (size_t)B * H * W * CK * 2 < ((size_t)1 << 31)) {       // 32-bit byte offsets in the halo fetch
      if (pe) {
        if constexpr (CK >= 16) {
          const int rc = pool_tree_shape<NC>(pe, gx, ygroups, Co);
          if (rc != BX_OK) return rc;
          hipLaunchKernelGGL((k_conv_mfma_p<CK, NC, TW, true>), dim3((unsigned)gx, (unsigned)ygroups), dim3(256), lds, s, (const bf16_t*)x,
                             (const bf16_t*)wp, bias, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (bf16_t*)y, H, W, Co, 1, tiles_x, tiles_y, ntiles,
                             (uint32_t)((size_t)B * H * W * CK * 2), *pe, nored, 0);
          BX_CHECK_LAUNCH("bx_conv3x3(mfma persistent, pooled)");
          return BX_OK;
        }
      }
      if constexpr (CK <= 16 && NC == 1) {
        // Measured (same box, training step): 1.534-1.538 ms with it (two images, one tile ahead: 1.539-1.542) against 1.532-1.536
        // without -- the block-1 kernels are bound by instruction issue (LDS reads + MFMA + epilogue arithmetic of four waves per
        // SIMD), not by the latency of their input loads; opt-in (BX_CONV_DLDS=1), results bit-identical.
        static const bool use_dl = getenv("BX_CONV_DLDS") != nullptr;
        if (use_dl) {                          // three thread-linear LDS images filled by direct-to-LDS loads, two tiles ahead
          const size_t lds_dl = (size_t)3 * ((10 * (TW + 2) * (CK / 8) + 255) / 256) * 256 * 16;
          hipLaunchKernelGGL((k_conv_mfma_p<CK, NC, TW, false, false, true>), dim3((unsigned)(gx + nred), (unsigned)ygroups), dim3(256),
                             lds_dl < 4096 ? 4096 : lds_dl, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)mask, (const bf16_t*)addend,
                             (bf16_t*)y, H, W, Co, relu, tiles_x, tiles_y, ntiles, (uint32_t)((size_t)B * H * W * CK * 2), none, rj, nred);
          BX_CHECK_LAUNCH("bx_conv3x3(mfma persistent, direct-to-LDS)");
          return BX_OK;
        }
      }
      hipLaunchKernelGGL((k_conv_mfma_p<CK, NC, TW>), dim3((unsigned)(gx + nred), (unsigned)ygroups), dim3(256), lds, s, (const bf16_t*)x,
                         (const bf16_t*)wp, bias, (const bf16_t*)mask, (const bf16_t*)addend, (bf16_t*)y, H, W, Co, relu, tiles_x, tiles_y, ntiles,
                         (uint32_t)((size_t)B * H * W * CK * 2), none, rj, nred);
      BX_CHECK_LAUNCH("bx_conv3x3(mfma persistent)");
      return BX_OK;
    }
  }
  // two-image tiles where one 8 x 16 tile IS the image and the layer is wide (stage 5, 256 output channels): measured at B=64
  // 21.0 -> 15.6 us (256->256 forward), 21.7 -> 16.5 (its dgrad), 11.9 -> 9.9 (128->256 forward); every other shape got slower
  // with IMGS = 2 or 4 (fewer, longer workgroups), see DESIGN section 6.  BX_CONV_IMGS=1|2 overrides for sweeps.
  if constexpr (CK == 64) {
    static const int imgs_env = getenv("BX_CONV_IMGS") ? atoi(getenv("BX_CONV_IMGS")) : 0;
    const bool two = imgs_env ? imgs_env == 2 : (tiles_x * tiles_y == 1 && Co >= 256);
    if (two && B % 2 == 0) {
      const size_t lds2 = lds * 2;
      static bool attr_done = false;
      if (lds2 > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute((const void*)k_conv_mfma<CK, NC, TW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
          BX_FAIL(BX_EHIP, "bx_conv3x3(mfma): cannot reserve %zu bytes of LDS", lds2);
        attr_done = true;
      }
      dim3 grid2((unsigned)(tiles_x * tiles_y * (B / 2) + nred), (unsigned)(Co / (16 * NC)));
      if (pe) {
        const int rc = pool_tree_shape<NC>(pe, (int)grid2.x, (int)grid2.y, Co);      // pe => nred == 0
        if (rc != BX_OK) return rc;
        static bool attr_done_p = false;
        if (lds2 > 64 * 1024 && !attr_done_p) {
          if (hipFuncSetAttribute((const void*)k_conv_mfma<CK, NC, TW, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
            BX_FAIL(BX_EHIP, "bx_conv3x3(mfma): cannot reserve %zu bytes of LDS", lds2);
          attr_done_p = true;
        }
        hipLaunchKernelGGL((k_conv_mfma<CK, NC, TW, 2, true>), grid2, dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)nullptr,
                           (const bf16_t*)nullptr, (bf16_t*)y, H, W, Ci, Co, 1, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), *pe, nored, 0);
        BX_CHECK_LAUNCH("bx_conv3x3(mfma, two-image tiles, pooled)");
        return BX_OK;
      }
      hipLaunchKernelGGL((k_conv_mfma<CK, NC, TW, 2>), grid2, dim3(256), lds2, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)mask,
                         (const bf16_t*)addend, (bf16_t*)y, H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), none, rj, nred);
      BX_CHECK_LAUNCH("bx_conv3x3(mfma, two-image tiles)");
      return BX_OK;
    }
  }
  dim3 grid((unsigned)(tiles_x * tiles_y * B + nred), (unsigned)(Co / (16 * NC)));
  if (pe) {
    if constexpr (CK >= 16) {
      const int rc = pool_tree_shape<NC>(pe, (int)grid.x, (int)grid.y, Co);
      if (rc != BX_OK) return rc;
      hipLaunchKernelGGL((k_conv_mfma<CK, NC, TW, 1, true>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)nullptr,
                         (const bf16_t*)nullptr, (bf16_t*)y, H, W, Ci, Co, 1, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), *pe, nored, 0);
      BX_CHECK_LAUNCH("bx_conv3x3(mfma, pooled)");
      return BX_OK;
    }
    BX_FAIL(BX_EUNSUPPORTED, "bx_conv3x3(pooled): needs at least 16 input channels");
  }
  if constexpr (CK == 64 && NC % 2 == 0) {
    static const bool w22 = !(getenv("BX_CONV_W22") && atoi(getenv("BX_CONV_W22")) == 0);     // 2 x 2 wave grid (see k_conv_mfma)
    if (w22) {
      hipLaunchKernelGGL((k_conv_mfma<CK, NC, TW, 1, false, false, true>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias,
                         (const bf16_t*)mask, (const bf16_t*)addend, (bf16_t*)y, H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), none, rj, nred);
      BX_CHECK_LAUNCH("bx_conv3x3(mfma, 2 x 2 waves)");
      return BX_OK;
    }
  }
  hipLaunchKernelGGL((k_conv_mfma<CK, NC, TW>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias,
                     (const bf16_t*)mask, (const bf16_t*)addend, (bf16_t*)y, H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), none, rj, nred);
  BX_CHECK_LAUNCH("bx_conv3x3(mfma)");
  return BX_OK;
}
// channel-split kernel for Ci % 64 == 0, Co % 64 == 0 (BX_CONV_C=0 keeps the pixel-split kernels)
template <int NCW, int TW, int IMGS>
static int launch_conv_c(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                         int B, int H, int W, int Ci, int Co, int relu, hipStream_t s, BxConvPoolEpi* pe) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 7) / 8;
  const size_t lds = (size_t)2 * IMGS * 10 * ((TW + 2 + 7) / 8 * 8) * 128;          // two halo images (double buffer)
  dim3 grid((unsigned)(tiles_x * tiles_y * (B / IMGS)), (unsigned)(Co / (64 * NCW)));
  const BxConvPoolEpi none = {};
  auto attr = [&](const void* fn, bool& done) -> int {
    if (lds > 64 * 1024 && !done) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        BX_FAIL(BX_EHIP, "bx_conv3x3(mfma): cannot reserve %zu bytes of LDS", lds);
      done = true;
    }
    return BX_OK;
  };
  if (pe) {
    pe->tree.nrows = (int)grid.x; pe->tree.Cw = NCW * 16; pe->tree.ld = Co;
    bx_stat_tree_shape((int)grid.x, &pe->tree.G, &pe->tree.ngroups);
    pe->tree.cnt = nullptr;                                   // rows only: the caller launches the finalize
    BX_REQUIRE(!pe->want_stats || (size_t)grid.x * 2 * Co <= pe->rows_cap_floats, "bx_conv3x3(pooled): %u partial rows exceed the workspace", grid.x);
    static bool done_p = false;
    const int rc = attr((const void*)k_conv_mfma_c<NCW, TW, IMGS, true>, done_p);
    if (rc != BX_OK) return rc;
    hipLaunchKernelGGL((k_conv_mfma_c<NCW, TW, IMGS, true>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, (bf16_t*)y, H, W, Ci, Co, 1, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), *pe);
    BX_CHECK_LAUNCH("bx_conv3x3(mfma channel-split, pooled)");
    return BX_OK;
  }
  static bool done = false;
  const int rc = attr((const void*)k_conv_mfma_c<NCW, TW, IMGS, false>, done);
  if (rc != BX_OK) return rc;
  hipLaunchKernelGGL((k_conv_mfma_c<NCW, TW, IMGS, false>), grid, dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (const bf16_t*)mask,
                     (const bf16_t*)addend, (bf16_t*)y, H, W, Ci, Co, relu, tiles_x, tiles_y, (uint32_t)((size_t)B * H * W * Ci * 2), none);
  BX_CHECK_LAUNCH("bx_conv3x3(mfma channel-split)");
  return BX_OK;
}
// returns -1 when the channel-split kernel does not apply.  Measured at B=64 against the pixel-split kernels (us, forward / data gradient):
//   16x32 128->128  14.1 / 17.5  vs 15.3 / 17.1      8x16 128->256  8.5 / 13.2 vs 9.6 / 12.4      8x16 256->256  13.6 / 15.0 vs 15.4 / 16.3
//   32x64  64->64   18.3 / 24.8  vs 18.5 / 21.8     16x32  64->128  9.9 / 14.2 vs 10.2 / 10.5
// so: forward (and pooled) layers with Co >= 128, data gradients only at 256 -> 256.  BX_CONV_C=0 never, 2 wherever it applies.
static int try_conv_c(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                      int B, int H, int W, int Ci, int Co, int relu, hipStream_t s, BxConvPoolEpi* pe) {
  static const int mode = getenv("BX_CONV_C") ? atoi(getenv("BX_CONV_C")) : 1;
  if (!mode || Ci % 64 || Co % 64) return -1;
  const bool dgrad = mask || addend;
  if (mode == 1 && (dgrad ? (Ci < 256 || Co < 256) : Co < 128)) return -1;
  if (W <= 16) {
    const long long wgs = (long long)((H + 7) / 8) * B * (Co / 64);
    if (B % 2 == 0 && wgs / 2 >= 256) return launch_conv_c<1, 16, 2>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe);
    return launch_conv_c<1, 16, 1>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe);
  }
  return launch_conv_c<1, 32, 1>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe);
}
// Tile choice: small late-stage maps would launch fewer workgroups than the chip has CUs (each then serialises its
// own stage -> MFMA chain); prefer 8x16 pixel tiles and fewer output channels per workgroup until >= 512 are in flight.
template <int CK, int NC>
static int launch_conv_tw(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                          int B, int H, int W, int Ci, int Co, int relu, hipStream_t s, BxConvPoolEpi* pe = nullptr, const WgradRedJob* red = nullptr) {
  // 8x32 tiles wherever the map is wide enough: a workgroup re-reads its whole weight slab from L2 per pixel tile, so
  // twice the pixels per tile halves the dominant L2 traffic of the late stages (measured: 16x32 maps 22.4 -> 17.3 us)
  static const int tw16_ck = getenv("BX_CONV_TW16_CK") ? atoi(getenv("BX_CONV_TW16_CK")) : 0;     // sweep knob: 8 x 16 tiles for layers with this chunk size
  if (W <= 16 || (tw16_ck && (tw16_ck & CK))) return launch_conv<CK, NC, 16>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe, red);
  return launch_conv<CK, NC, 32>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe, red);
}
template <int CK>
static int launch_conv_nc(const void* x, const void* wp, const float* bias, const void* mask, const void* addend, void* y,
                          int B, int H, int W, int Ci, int Co, int relu, hipStream_t s, BxConvPoolEpi* pe = nullptr, const WgradRedJob* red = nullptr) {
  // output channels per workgroup: 64 while that still launches >= 512 workgroups, else 32 (keeps two per CU in flight)
  const int tw = W <= 16 ? 16 : 32;
  const long long tiles = (long long)((W + tw - 1) / tw) * ((H + 7) / 8) * B;
  static const long long nc4_min = getenv("BX_CONV_NC4_MIN") ? atoll(getenv("BX_CONV_NC4_MIN")) : 512;
  if (Co % 64 == 0 && tiles * (Co / 64) >= nc4_min) return launch_conv_tw<CK, 4>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe, red);
  if (Co % 32 == 0) return launch_conv_tw<CK, 2>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe, red);
  return launch_conv_tw<CK, 1>(x, wp, bias, mask, addend, y, B, H, W, Ci, Co, relu, s, pe, red);
}
int bx_conv3x3_mfma_launch(const void* x, const void* packed_mfma, const float* bias, const void* relu_mask_src,
                           const void* addend, void* y, int B, int H, int W, int Ci, int Co, int flags, hipStream_t s, bxWgradPending* carry) {
  const int relu = ((flags & BX_EPI_RELU) ? 1 : 0) | ((flags & BX_EPI_MASK_BITS) ? 2 : 0);
  BX_REQUIRE(!(flags & BX_EPI_MASK_BITS) || (relu_mask_src && Ci <= 32 && Co >= 16), "bx_conv3x3(mfma): BX_EPI_MASK_BITS applies to the early stages' data gradients (Ci <= 32)");
  // a pending weight-gradient sum rides in this launch (pixel-split kernels only); the job is built like a chained reduce's
  WgradRedJob job = {};
  const WgradRedJob* red = nullptr;
  if (carry && carry->valid) { job = wgrad_job_from(carry, true); red = &job; carry->valid = 0; }
  // the kernels address activations with 32-bit byte offsets through buffer resources
  BX_REQUIRE((size_t)B * H * W * (Ci > Co ? Ci : Co) * 2 < ((size_t)1 << 31), "bx_conv3x3(mfma): an activation tensor of 2 GiB or more is not supported (B=%d H=%d W=%d)", B, H, W);
  // K-split variant: measured equal to the tile-per-workgroup kernel on MI355X (round 1: fwd +8 %, dgrad -3 %), so it is
  // opt-in (BX_KSPLIT=1) until its main loop gets LDS double buffering
  if (!red && Co % 64 == 0 && getenv("BX_KSPLIT")) {
    if (Ci == 64)  return launch_conv_ks<1>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Co, relu, s);
    if (Ci == 128) return launch_conv_ks<2>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Co, relu, s);
    if (Ci == 256) return launch_conv_ks<4>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Co, relu, s);
  }
  if (!red) {
    const int rc = try_conv_c(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, relu, s, nullptr);
    if (rc >= 0) return rc;
  }
  switch (mfma_ck(Ci)) {
    case 8:  return launch_conv_nc<8>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, relu, s, nullptr, red);
    case 16: return launch_conv_nc<16>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, relu, s, nullptr, red);
    case 32: return launch_conv_nc<32>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, relu, s, nullptr, red);
    default: return launch_conv_nc<64>(x, packed_mfma, bias, relu_mask_src, addend, y, B, H, W, Ci, Co, relu, s, nullptr, red);
  }
}

// conv3 of a Block: y = relu(conv + bias), pooled = pool2x2(y), batch statistics of pooled (+ finalize) -- one launch
int bx_conv3x3_mfma_pool_launch(const void* x, const void* packed_mfma, const float* bias, void* y, int B, int H, int W, int Ci, int Co,
                                BxConvPoolEpi* pe, hipStream_t s) {
  BX_REQUIRE((size_t)B * H * W * (Ci > Co ? Ci : Co) * 2 < ((size_t)1 << 31), "bx_conv3x3(pooled): an activation tensor of 2 GiB or more is not supported");
  BX_REQUIRE(Ci >= 16 && pe && pe->pooled, "bx_conv3x3(pooled): needs at least 16 input channels and a pooled output");
  {
    const int rc = try_conv_c(x, packed_mfma, bias, nullptr, nullptr, y, B, H, W, Ci, Co, 1, s, pe);
    if (rc >= 0) return rc;
  }
  switch (mfma_ck(Ci)) {
    case 16: return launch_conv_nc<16>(x, packed_mfma, bias, nullptr, nullptr, y, B, H, W, Ci, Co, 1, s, pe);
    case 32: return launch_conv_nc<32>(x, packed_mfma, bias, nullptr, nullptr, y, B, H, W, Ci, Co, 1, s, pe);
    default: return launch_conv_nc<64>(x, packed_mfma, bias, nullptr, nullptr, y, B, H, W, Ci, Co, 1, s, pe);
  }
}

// sum the partials in a fixed order and scatter to OIHW.  A split is [ytiles][ztiles][9*MA*NB tiles][64 lanes][4 regs]
// accumulator fragments followed by Co bias sums.  A workgroup covers NO = 256/S float4 groups x S split slices
// (S = 4, 16 or 64: small outputs get many slices so that the chip is filled and no thread walks hundreds of splits);
// slices are combined through LDS in slice order.
__device__ __forceinline__ void wgrad_reduce3_body(const WgradRedJob& jb, int bid, float4* sm) {
  const float* __restrict__ partial = jb.partial;
  float* __restrict__ dw = jb.dw;
  float* __restrict__ db = jb.db;
  const int nsplit = jb.nsplit, Cin = jb.Cin, Co = jb.Co, S = jb.S, MA = jb.MA, NB = jb.NB, ztiles = jb.ztiles, nfrag4 = jb.nfrag4;
  const size_t per_split = (size_t)nfrag4 * 4 + Co;
  const int NO = 256 / S;
  const int o = threadIdx.x % NO, part = threadIdx.x / NO;
  const size_t e4 = (size_t)bid * NO + o;          // float4 index inside a split
  const bool live = e4 * 4 < per_split;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const float* src = partial + e4 * 4;
    int c = part;
    for (; c + 3 * S < nsplit; c += 4 * S) {      // four independent loads in flight; adds stay in split order
      const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)c * per_split);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(c + S) * per_split);
      const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(c + 2 * S) * per_split);
      const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)(c + 3 * S) * per_split);
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; c < nsplit; c += S) {
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
    if (e4 < (size_t)nfrag4) {
      const int NT = 9 * MA * NB;
      const int lane = (int)(e4 & 63), tl = (int)(e4 >> 6);
      const int i = tl % NT, yz = tl / NT, z = yz % ztiles, y = yz / ztiles;
      const int n = i % NB, m = (i / NB) % MA, t = i / (NB * MA);
      const int cout = (z * NB + n) * 16 + (lane & 15), cin0 = (y * MA + m) * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (cin0 + j < Cin) dw[((size_t)cout * Cin + cin0 + j) * 9 + t] = rr[j];
    } else if (db) {
      const size_t b0 = (e4 - nfrag4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) db[b0 + j] = rr[j];
    }
  }
}
__global__ __launch_bounds__(256) void k_wgrad_reduce3(WgradRedJob jb) {
  __shared__ float4 sm[256];
  wgrad_reduce3_body(jb, blockIdx.x, sm);
}

// ================================================================================================
// Weight gradient (k_wgrad_mfma):  dW[tap][cin][cout] = sum_pixels X[pixel+tap][cin] * dZ[pixel][cout]
//   v_mfma_f32_16x16x32_bf16 with the PIXEL index as K:  A = X^T (row = cin, k = 32 pixels), B = dZ (k = pixels,
//   col = cout), D[cin][cout] per tap.  Both operands need "8 consecutive pixels of one channel" per lane while
//   the tiles sit in LDS pixel-major ([pixel][channel], as they stream in from NHWC) -- exactly what
//   ds_read_b64_tr_b16 delivers (4 pixel rows x 16 channels per 16-lane group, transposed on the way to the
//   registers); every lane supplies its own row address, so the 9 tap shifts cost nothing.
//   A workgroup owns a (16*MA cin) x (16*NB cout) slice of dW for all 9 taps and walks a contiguous range of
//   8 x TW pixel tiles; its 4 waves split each tile's K-steps (rows) and keep 9*MA*NB accumulator tiles in
//   registers across the whole range.  At the end the waves are summed through LDS (reduce-scatter, fixed order) and
//   the slice goes to the workspace as one partial in MFMA fragment order; k_wgrad_reduce3 sums the partials in a
//   fixed order (deterministic, no atomics) and writes OIHW.  The bias gradient rides along on the dZ fragments.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_read8(const char* lds, int off0, int off1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int MA, int NB, int TW>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, float* __restrict__ partial,
    int H, int W, int Ci_p, int Co, int tiles_x, int tiles_y, int ntiles, int tiles_per_split, WgradRedJob prev, int zextra) {
  constexpr int TH = 8, HWID = TW + 2, HH = TH + 2, CIT = 16 * MA, COT = 16 * NB, XB = CIT * 2, ZB = COT * 2;
  constexpr int KPW = TH * TW / 32 / 4;                 // K-steps (32 pixels) per wave per tile
  constexpr int XS_BYTES = HH * HWID * XB;             // the dZ tile (TH * TW * ZB bytes) follows the halo tile
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // chained mode: the first `zextra` z-slices of the grid sum the PREVIOUS layer's partials (its reduce launch is saved; these
  // light workgroups are dispatched first and overlap with the heavy ones that follow)
  if ((int)blockIdx.z < zextra) {
    const int rid = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    if (rid < prev.nblocks) wgrad_reduce3_body(prev, rid, reinterpret_cast<float4*>(lds));
    return;
  }
  const int bz = (int)blockIdx.z - zextra, gz = (int)gridDim.z - zextra;
  char* xs = lds;
  char* zs = lds + XS_BYTES;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pc = li & 3;
  const int ci0 = blockIdx.y * CIT, co0 = bz * COT;
  const bool want_bias = blockIdx.y == 0;

  f32x4 acc[9][MA][NB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[t][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) bsum[n] = 0.f;

  const int t_begin = blockIdx.x * tiles_per_split;
  const int t_end = t_begin + tiles_per_split < ntiles ? t_begin + tiles_per_split : ntiles;
  // software pipeline: the next tile's global loads are issued into registers before the current tile's MFMAs
  constexpr int NXU = HH * HWID * (CIT / 8), NZU = TH * TW * (COT / 8);
  constexpr int NX = (NXU + 255) / 256, NZ = (NZU + 255) / 256;
  uint4 rx[NX], rz[NZ];
  // tile loads through raw buffer resources with 32-bit byte offsets; out-of-image lanes point past the end and read
  // zeros (no divergent branches, no 64-bit address arithmetic); per-thread invariants are computed once
  const int nimg = ntiles / (tiles_x * tiles_y);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (uint32_t)((size_t)nimg * H * W * Ci_p * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (uint32_t)((size_t)nimg * H * W * Co * 2), 0x00020000);
  int xpy[NX], xpx[NX], zpy[NZ], zpx[NZ];
  uint32_t xrel[NX], zrel[NZ];
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (CIT / 8), c = u % (CIT / 8);
    const bool live = u < NXU && ci0 + c * 8 < Ci_p;
    xpy[k] = live ? p / HWID - 1 : -100000;
    xpx[k] = p % HWID - 1;
    xrel[k] = (uint32_t)((((p / HWID - 1) * W + (p % HWID - 1)) * Ci_p + ci0 + c * 8) * 2);
  }
#pragma unroll
  for (int k = 0; k < NZ; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (COT / 8), c = u % (COT / 8);
    zpy[k] = u < NZU ? p / TW : 100000000;
    zpx[k] = p % TW;
    zrel[k] = (uint32_t)((((p / TW) * W + p % TW) * Co + co0 + c * 8) * 2);
  }
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t pix0 = (uint32_t)((b * H + y0) * W + x0);
    const uint32_t xb = pix0 * (uint32_t)(Ci_p * 2), zb = pix0 * (uint32_t)(Co * 2);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool ok = (unsigned)(y0 + xpy[k]) < (unsigned)H && (unsigned)(x0 + xpx[k]) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? xb + xrel[k] : 0x80000000u, 0, 0);
      rx[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = (unsigned)(y0 + zpy[k]) < (unsigned)H && (unsigned)(x0 + zpx[k]) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(zres, ok ? zb + zrel[k] : 0x80000000u, 0, 0);
      rz[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NXU) *reinterpret_cast<uint4*>(xs + (u / (CIT / 8)) * XB + (u % (CIT / 8)) * 16) = rx[k];
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < NZU) *reinterpret_cast<uint4*>(zs + (u / (COT / 8)) * ZB + (u % (COT / 8)) * 16) = rz[k];
    }
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
      const int step = wave * KPW + ks;
      int row[2], col[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kp = 8 * g + 4 * h + q;                       // pixel of this K-step whose row address the lane supplies
        if (TW == 32) { row[h] = step; col[h] = kp; }
        else          { row[h] = 2 * step + (kp >> 4); col[h] = kp & 15; }
      }
      bf16x8 bfr[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        bfr[n] = tr_read8(zs, (row[0] * TW + col[0]) * ZB + (n * 16 + 4 * pc) * 2, (row[1] * TW + col[1]) * ZB + (n * 16 + 4 * pc) * 2);
        if (want_bias) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[n] += __uint_as_float(((uint32_t)(unsigned short)bfr[n][j]) << 16);
        }
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int m = 0; m < MA; ++m) {
          const bf16x8 afr = tr_read8(xs, ((row[0] + dy) * HWID + col[0] + dx) * XB + (m * 16 + 4 * pc) * 2,
                                      ((row[1] + dy) * HWID + col[1] + dx) * XB + (m * 16 + 4 * pc) * 2);
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[tap][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[n], acc[tap][m][n], 0, 0, 0);
        }
      }
    }
  }
  // ---- cross-wave sum as a reduce-scatter through LDS: tile i of the 9*MA*NB accumulator tiles is owned by wave i%4.
  //      In round s every wave hands the tiles owned by wave+s to their owner (float4 per lane, conflict-free) and
  //      adds what wave-s handed to it; the order per tile is fixed (own, -1, -2, -3), so the result is deterministic.
  //      Owners then store their tiles in FRAGMENT order ([tile][lane][reg]: 1 KiB coalesced per wave store);
  //      k_wgrad_reduce3 maps fragments back to OIHW after summing the splits.
  constexpr int NT = 9 * MA * NB, SLOTS = (NT + 3) / 4;
  float4* red4 = reinterpret_cast<float4*>(lds);       // [4 writers][SLOTS][64]
  __syncthreads();
#pragma unroll
  for (int s = 1; s < 4; ++s) {
    const int dst = (wave + s) & 3, src = (wave - s) & 3;
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if ((i & 3) == dst) {
        const f32x4 v = acc[i / (MA * NB)][(i / NB) % MA][i % NB];
        red4[(wave * SLOTS + i / 4) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if ((i & 3) == wave) {
        const float4 v = red4[(src * SLOTS + i / 4) * 64 + lane];
        f32x4& a = acc[i / (MA * NB)][(i / NB) % MA][i % NB];
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
      }
    __syncthreads();
  }
  const size_t nfrag = (size_t)gridDim.y * gz * NT * 256;              // floats of accumulator fragments per split
  const size_t per_split = nfrag + Co;
  float* out = partial + (size_t)blockIdx.x * per_split;
  float4* out4 = reinterpret_cast<float4*>(out) + ((size_t)blockIdx.y * gz + bz) * NT * 64;
#pragma unroll
  for (int i = 0; i < NT; ++i)
    if ((i & 3) == wave) {
      const f32x4 v = acc[i / (MA * NB)][(i / NB) % MA][i % NB];
      out4[i * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
  float* red = reinterpret_cast<float*>(lds);
  if (want_bias) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float v = bsum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) red[(wave * NB + n) * 16 + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < COT) {
      const int n = threadIdx.x >> 4, c = threadIdx.x & 15;
      out[nfrag + co0 + threadIdx.x] =
          red[(0 * NB + n) * 16 + c] + red[(1 * NB + n) * 16 + c] + red[(2 * NB + n) * 16 + c] + red[(3 * NB + n) * 16 + c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_wgrad_own (round 2): the 32 x 32 (cin, cout) slice with its four waves OWNING output tiles instead of splitting K.
//   Round 1's kernel above gives every wave a full copy of the 36 accumulator tiles (144 registers, 2 workgroups per CU) and
//   splits each staged pixel tile's K-steps over the waves; the copies are then summed through LDS (three exchange rounds).
//   With 4 pixel tiles per workgroup the launch was neither MFMA- nor bandwidth-bound: every shape took 22-35 us (hbm bound
//   2-21 us, MFMA bound 2-4 us, tools/conv_bench.py) -- a workgroup's life was one exposed global-load latency per tile plus that
//   epilogue, with two tiles in flight per CU.
//   Here wave (m, n) owns the (16 cin) x (16 cout) quadrant for all 9 taps (9 accumulator tiles, 36 registers) and every wave
//   walks every K-step of the staged tile: no cross-wave sum at all (the partial is stored straight from the accumulators), and
//   at ~100 registers four workgroups fit a CU, i.e. twice the tiles in flight.  The price is LDS operand traffic (10 fragment
//   reads per 9 MFMAs instead of 22 per 36), which is why the tiles are SWIZZLED here: a transposing read takes its 32-lane
//   group's rows from pixels {x..x+3, x+8..x+11}, whose 64-byte records alias in the 64 banks; swapping the two 32-byte halves of
//   the records of pixels with bit 3 of x set makes every read conflict-free (brute-forced over both lane groups, all taps, both
//   tile widths).  Round 1 measured 86 % of the LDS-active cycles as bank-conflict cycles in the kernel above and had no
//   registers left for swizzled addresses.
//   Workgroup -> work mapping: 1-D grid; consecutive ids are dealt to the 8 XCDs round-robin, so id & 7 picks the pixel range and
//   id >> 3 walks the (cin, cout) slices of that range: all slices that re-read one range's X / dZ tiles sit on ONE XCD at the
//   same time and share them through its L2 (before: every slice pair fetched them from HBM again; a performance assumption
//   only, any placement is correct).
#ifdef BX_WGRAD_STAMPS
// Diagnostic build only (hipcc -DBX_WGRAD_STAMPS, tools/wgrad_stamps.py): shader-clock stamps of one workgroup in 16 go to a
// buffer of their own that nothing else reads.  Slot layout per workgroup: [start, first tile staged, end of tile 0..3, stored].
__device__ unsigned long long bx_wgrad_stamps[64 * 8];
extern "C" int bx_debug_wgrad_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bx_wgrad_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -1;
}
#define BX_STAMP(i) do { if (stamp_wg >= 0 && threadIdx.x == 0) bx_wgrad_stamps[stamp_wg * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BX_STAMP(i) do { } while (0)
#endif
// DL (round 3): the tiles go from HBM / L2 straight into one of THREE LDS images, two tiles ahead (buffer_load ... lds: no staging
// registers, no LDS write instructions, one barrier per tile).  The in-kernel stamps showed the register-staged loop at one exposed
// global-load latency per tile (2500 cycles per 8 x 16 tile for 4 x 144 cycles of MFMA, 4500 per 8 x 32 tile): with one tile ahead and
// two workgroups per CU there is nothing else to run while a load is in flight.  A lane's 16 bytes land at (wave base) + 16 lane, so
// the swizzle moves to the SOURCE side: LDS slot L = 4 p + c' of pixel p takes channel unit c = ((c' >> 1) ^ (px >> 3)) << 1 | (c' & 1)
// -- the image is byte for byte what stage() writes.  TH = 4 with 32-wide tiles keeps three images at 74 KB (two workgroups per CU).
// SHIFT (round 3): the loop is bound by LDS READ bandwidth, not by latency (which is why DL did not pay: 2 workgroups x 4 waves x 20
// ds_read_b64_tr_b16 x 512 bytes per K-step = 2560 cycles per pair of tiles at 128 bytes/clk -- exactly the stamps).  The nine tap
// fragments of a K-step are three shifted windows per tile row: a lane's 8 pixels of tap dx are pixels dx .. dx + 7 of the 10 it would
// read for dx = 0 .. 2 together.  So each row is read ONCE as 12 pixels (three transposing reads) and the dx = 1 / 2 fragments are
// built in registers (dword re-indexing and four v_alignbit): 9 + 2 reads per K-step instead of 18 + 2.
template <int TW, int OCC, int DEPTH, int TH = 8, bool DL = false, bool SHIFT = true>
__global__ __launch_bounds__(256, OCC) void k_wgrad_own(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, float* __restrict__ partial,
    int H, int W, int Ci_p, int Co, int tiles_x, int tiles_y, int ntiles, int tiles_per_split, int nsplit, int ytiles, int ztiles,
    WgradRedJob prev, int nred) {
  constexpr int HWID = TW + 2, HH = TH + 2, CIT = 32, COT = 32, XB = 64, ZB = 64;
  constexpr int KSTEPS = TH * TW / 32, ROWS_PER_STEP = 32 / TW;   // TW = 32: one tile row per K-step; TW = 16: two
  constexpr int XS_BYTES = HH * HWID * XB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if ((int)blockIdx.x < nred) {                                  // chained mode: sum the PREVIOUS layer's partials
    if ((int)blockIdx.x < prev.nblocks) wgrad_reduce3_body(prev, (int)blockIdx.x, reinterpret_cast<float4*>(lds));
    return;
  }
  const int L = (int)blockIdx.x - nred, YZ = ytiles * ztiles;     // nred is a multiple of 8: L & 7 == blockIdx.x & 7
  const int slot = L >> 3, yz = slot % YZ, split = (slot / YZ) * 8 + (L & 7);
  if (split >= nsplit) return;
  const int by = yz / ztiles, bz = yz - by * ztiles;
#ifdef BX_WGRAD_STAMPS
  const int stamp_wg = (L % 16 == 5 && L / 16 < 64) ? L / 16 : -1;
#endif
  BX_STAMP(0);
  char* xs = lds;
  char* zs = lds + XS_BYTES;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pc = li & 3;
  const int m = wave >> 1, n = wave & 1;
  const int ci0 = by * CIT, co0 = bz * COT;
  const bool want_bias = by == 0 && m == 0;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int t_begin = split * tiles_per_split;
  const int t_end = t_begin + tiles_per_split < ntiles ? t_begin + tiles_per_split : ntiles;
  constexpr int NXU = HH * HWID * (CIT / 8), NZU = TH * TW * (COT / 8);
  constexpr int NX = (NXU + 255) / 256, NZ = (NZU + 255) / 256;
  constexpr int IMG_BYTES = (NX + NZ) * 256 * 16;                // DL: one LDS image = x units padded to whole waves, then the dZ units
  if constexpr (DL) zs = lds + NX * 256 * 16;
  uint4 rxa[DL ? 1 : NX], rza[DL ? 1 : NZ], rxb[DL ? 1 : NX], rzb[DL ? 1 : NZ];        // two tiles in flight (OCC = 3 leaves the registers): fetch distance 2
  const int nimg = ntiles / (tiles_x * tiles_y);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (uint32_t)((size_t)nimg * H * W * Ci_p * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t zres = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (uint32_t)((size_t)nimg * H * W * Co * 2), 0x00020000);
  int xpp[NX], zpp[NZ];                                          // (row << 16) | (column & 0xffff) of the unit inside the tile
  uint32_t xrel[NX], zrel[NZ];
#pragma unroll
  for (int k = 0; k < NX; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (CIT / 8), px = p % HWID;
    const int cu = u % (CIT / 8), c = DL ? (((((cu >> 1) ^ (px >> 3)) & 1) << 1) | (cu & 1)) : cu;     // DL: u is the LDS slot
    const bool live = u < NXU && ci0 + c * 8 < Ci_p;
    xpp[k] = ((live ? p / HWID - 1 : -20000) << 16) | ((px - 1) & 0xffff);
    xrel[k] = (uint32_t)((((p / HWID - 1) * W + (px - 1)) * Ci_p + ci0 + c * 8) * 2);
  }
#pragma unroll
  for (int k = 0; k < NZ; ++k) {
    const int u = threadIdx.x + k * 256;
    const int p = u / (COT / 8), px = p % TW;
    const int cu = u % (COT / 8), c = DL ? (((((cu >> 1) ^ (px >> 3)) & 1) << 1) | (cu & 1)) : cu;
    zpp[k] = ((u < NZU ? p / TW : 20000) << 16) | px;
    zrel[k] = (uint32_t)((((p / TW) * W + px) * Co + co0 + c * 8) * 2);
  }
  auto fetch_lds = [&](int tile, int img) {                    // DL: lane l of wave w lands at image + (k * 256 + w * 64 + l) * 16
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t pix0 = (uint32_t)((b * H + y0) * W + x0);
    const uint32_t xb = pix0 * (uint32_t)(Ci_p * 2), zb = pix0 * (uint32_t)(Co * 2);
    char* base = lds + img * IMG_BYTES + wave * 64 * 16;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool ok = (unsigned)(y0 + (xpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (int)(short)(xpp[k] & 0xffff)) < (unsigned)W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) void*)(base + k * 256 * 16), 16,
                                               (int)(ok ? xb + xrel[k] : 0x80000000u), 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = (unsigned)(y0 + (zpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (zpp[k] & 0xffff)) < (unsigned)W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(zres, (__attribute__((address_space(3))) void*)(base + (NX + k) * 256 * 16), 16,
                                               (int)(ok ? zb + zrel[k] : 0x80000000u), 0, 0, 0);
    }
  };
  auto fetch = [&](int tile, uint4 (&rx)[NX], uint4 (&rz)[NZ]) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const uint32_t pix0 = (uint32_t)((b * H + y0) * W + x0);
    const uint32_t xb = pix0 * (uint32_t)(Ci_p * 2), zb = pix0 * (uint32_t)(Co * 2);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const bool ok = (unsigned)(y0 + (xpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (int)(short)(xpp[k] & 0xffff)) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? xb + xrel[k] : 0x80000000u, 0, 0);
      rx[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const bool ok = (unsigned)(y0 + (zpp[k] >> 16)) < (unsigned)H && (unsigned)(x0 + (zpp[k] & 0xffff)) < (unsigned)W;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(zres, ok ? zb + zrel[k] : 0x80000000u, 0, 0);
      rz[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  // per-lane read addresses: pixel kp = 8g + 4h + q of a K-step (h = the two 4-row halves of a transposing read pair)
  int xbase[2][3], zbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kp = 8 * g + 4 * h + q;
    const int r = TW == 32 ? 0 : (kp >> 4), c = TW == 32 ? kp : (kp & 15);
    zbase[h] = (r * TW + c) * ZB + (((n ^ (c >> 3)) & 1) << 5) + pc * 8;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) xbase[h][dx] = (r * HWID + c + dx) * XB + (((m ^ ((c + dx) >> 3)) & 1) << 5) + pc * 8;
  }
  auto stage = [&](const uint4 (&rx)[NX], const uint4 (&rz)[NZ]) {
    // LDS destinations are recomputed per tile (a handful of integer ops; keeping them would cost 10 registers):
    // record p, 16-byte unit c; the two 32-byte halves of a record are swapped where bit 3 of its x position is set (swizzle)
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int u = threadIdx.x + k * 256, p = u >> 2, c = u & 3, px = p % HWID;
      if (u < NXU) *reinterpret_cast<uint4*>(xs + p * XB + ((((c >> 1) ^ (px >> 3)) & 1) << 5) + (c & 1) * 16) = rx[k];
    }
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const int u = threadIdx.x + k * 256, p = u >> 2, c = u & 3, px = p % TW;
      if (u < NZU) *reinterpret_cast<uint4*>(zs + p * ZB + ((((c >> 1) ^ (px >> 3)) & 1) << 5) + (c & 1) * 16) = rz[k];
    }
  };
  // The K-steps of a staged tile with the operand fragments of step k+1 requested BEFORE the nine MFMAs of step k.  Written
  // as plain "read, then use" the compiler waits for each fragment right before its MFMA: the in-kernel stamps (tools/
  // wgrad_stamps.py) showed 560 cycles per K-step for 144 cycles of MFMA -- nine exposed LDS latencies.
  int img_off = 0;                                               // DL: byte offset of the image being consumed
  int xbase2;                                                    // SHIFT: pixels 8 .. 11 past the lane group's first (third transposing read)
  {
    const int kp = 8 * g + q;
    const int r = TW == 32 ? 0 : (kp >> 4), c = (TW == 32 ? kp : (kp & 15)) + 8;
    xbase2 = (r * HWID + c) * XB + (((m ^ (c >> 3)) & 1) << 5) + pc * 8;
  }
  auto load_frags = [&](int ks, bf16x8 (&a)[9], bf16x8& b) {
    const int r0 = ks * ROWS_PER_STEP;
    b = tr_read8(zs + img_off, zbase[0] + r0 * TW * ZB, zbase[1] + r0 * TW * ZB);
    if constexpr (SHIFT) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ro = (r0 + dy) * HWID * XB;
        const char* xi = xs + img_off;
        const u32x2 p0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xi + xbase[0][0] + ro)));
        const u32x2 p1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xi + xbase[1][0] + ro)));
        const u32x2 p2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xi + xbase2 + ro)));
        const uint32_t d0 = p0.x, d1 = p0.y, d2 = p1.x, d3 = p1.y, d4 = p2.x;       // pixel pairs (0,1) (2,3) (4,5) (6,7) (8,9)
        a[dy * 3 + 0] = __builtin_bit_cast(bf16x8, (u32x4){d0, d1, d2, d3});
        a[dy * 3 + 1] = __builtin_bit_cast(bf16x8, (u32x4){__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                                           __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)});
        a[dy * 3 + 2] = __builtin_bit_cast(bf16x8, (u32x4){d1, d2, d3, d4});
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        a[tap] = tr_read8(xs + img_off, xbase[0][dx] + (r0 + dy) * HWID * XB, xbase[1][dx] + (r0 + dy) * HWID * XB);
      }
    }
  };
  auto mma = [&](const bf16x8 (&a)[9], const bf16x8& b) {
    if (want_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum += __uint_as_float(((uint32_t)(unsigned short)b[j]) << 16);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tap], b, acc[tap], 0, 0, 0);
  };
  auto compute = [&]() {
    bf16x8 a0[9], a1[9], b0, b1;
    load_frags(0, a0, b0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      if (ks + 1 < KSTEPS) load_frags(ks + 1, a1, b1);
      mma(a0, b0);
      if (ks + 2 < KSTEPS) load_frags(ks + 2, a0, b0);
      if (ks + 1 < KSTEPS) mma(a1, b1);
    }
  };
  if constexpr (DL) {
    static_assert(NX + NZ <= 15, "vmcnt immediate");
    if (t_begin < t_end) fetch_lds(t_begin, 0);
    if (t_begin + 1 < t_end) fetch_lds(t_begin + 1, 1);
    int it = 0;
    for (int tile = t_begin; tile < t_end; ++tile, ++it) {
      // this wave's loads of `tile` have landed (vmcnt counts in issue order: only the NX + NZ loads of the tile after it may still be
      // in flight), then everybody's; the image about to be refilled was last read before this barrier
      if (tile + 1 < t_end) __builtin_amdgcn_s_waitcnt(0x0F70 | (NX + NZ));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (tile == t_begin) BX_STAMP(1);
      if (tile + 2 < t_end) fetch_lds(tile + 2, (it + 2) % 3);
      img_off = (it % 3) * IMG_BYTES;
      compute();
      if (tile - t_begin < 4) BX_STAMP(2 + tile - t_begin);
    }
  } else if (DEPTH == 2) {                                       // two tiles in flight (needs the 256-register budget of OCC = 2)
    if (t_begin < t_end) fetch(t_begin, rxa, rza);
    if (t_begin + 1 < t_end) fetch(t_begin + 1, rxb, rzb);
    for (int tile = t_begin; tile < t_end; tile += 2) {
      __syncthreads();
      stage(rxa, rza);
      __syncthreads();
      if (tile + 2 < t_end) fetch(tile + 2, rxa, rza);
      compute();
      if (tile + 1 < t_end) {
        __syncthreads();
        stage(rxb, rzb);
        __syncthreads();
        if (tile + 3 < t_end) fetch(tile + 3, rxb, rzb);
        compute();
      }
    }
  } else {
    if (t_begin < t_end) fetch(t_begin, rxa, rza);
    for (int tile = t_begin; tile < t_end; ++tile) {
      __syncthreads();
      stage(rxa, rza);
      __syncthreads();
      if (tile == t_begin) BX_STAMP(1);
      if (tile + 1 < t_end) fetch(tile + 1, rxa, rza);
      compute();
      if (tile - t_begin < 4) BX_STAMP(2 + tile - t_begin);
    }
  }
  // ---- the partial: fragment order [tile = tap*4 + m*2 + n][lane][reg], exactly what k_wgrad_mfma writes (same reduce kernel)
  constexpr int NT = 36;
  const size_t nfrag = (size_t)ytiles * ztiles * NT * 256;
  const size_t per_split = nfrag + Co;
  float* out = partial + (size_t)split * per_split;
  float4* out4 = reinterpret_cast<float4*>(out) + ((size_t)by * ztiles + bz) * NT * 64;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) out4[(tap * 4 + m * 2 + n) * 64 + lane] = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
  if (want_bias) {                                               // wave (0, n) saw every pixel of the range: its 16 couts' sums are complete
    float v = bsum;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) out[nfrag + co0 + n * 16 + lane] = v;
  }
  BX_STAMP(6);
}

struct WgradPlan { int ma, nb, tw, th, tiles_x, tiles_y, ntiles, ytiles, ztiles, nsplit, tps; size_t lds; };
// tile-owner kernel with direct-to-LDS tiles (k_wgrad_own<.., DL>): BX_WGRAD_DL=1; default is the register-staged loop
static bool wgrad_own_dl() {
  static const bool v = getenv("BX_WGRAD_DL") && atoi(getenv("BX_WGRAD_DL")) != 0;     // opt-in: measured slower (1.501 vs 1.476 ms/step)
  return v;
}
static bool wgrad_own_on() {
  static const bool v = !(getenv("BX_WGRAD_OWN") && atoi(getenv("BX_WGRAD_OWN")) == 0);
  return v;
}
static WgradPlan wgrad_plan(int B, int H, int W, int Ci_p, int Co) {
  WgradPlan p;
  p.ma = Ci_p >= 32 ? 2 : 1;
  p.nb = Co >= 32 ? 2 : 1;
  p.tw = W <= 16 ? 16 : 32;
  // 4-row tiles for the direct-to-LDS form of the tile-owner kernel at 32-wide tiles (three LDS images, two workgroups per CU)
  p.th = (p.ma == 2 && p.nb == 2 && p.tw == 32 && wgrad_own_on() && wgrad_own_dl()) ? 4 : 8;
  p.tiles_x = (W + p.tw - 1) / p.tw; p.tiles_y = (H + p.th - 1) / p.th; p.ntiles = p.tiles_x * p.tiles_y * B;
  p.ytiles = (Ci_p + 16 * p.ma - 1) / (16 * p.ma); p.ztiles = Co / (16 * p.nb);
  // two rounds of 256 workgroups for every tile shape: with the partial sum riding in the next layer's launch, more (smaller)
  // splits only add partial traffic (sweep of BX_WGRAD_WANT on the training step: 384 1.789, 512 1.733, 576 1.745, 768 1.870,
  // 1024 1.882 ms; the lighter tiles used to run at 1024)
  static const int want_env = getenv("BX_WGRAD_WANT") ? atoi(getenv("BX_WGRAD_WANT")) : 0;
  int want = (want_env ? want_env : 512) / (p.ytiles * p.ztiles);
  if (want < 1) want = 1;
  if (want > p.ntiles) want = p.ntiles;
  p.tps = (p.ntiles + want - 1) / want;
  p.nsplit = (p.ntiles + p.tps - 1) / p.tps;
  const size_t stage = (size_t)10 * (p.tw + 2) * 32 * p.ma + (size_t)8 * p.tw * 32 * p.nb;
  const size_t red = (size_t)((9 * p.ma * p.nb + 3) / 4) * 4 * 64 * sizeof(float4);
  p.lds = stage > red ? stage : red;
  return p;
}
int bx_wgrad_mfma_supported(int Ci_p, int Co, int dtype) {
  return dtype == BX_BF16 && Ci_p % 8 == 0 && Co % 16 == 0 && (Co < 32 || Co % 32 == 0) && (Ci_p < 32 || Ci_p % 32 == 0);
}
size_t bx_wgrad_mfma_workspace(int B, int H, int W, int Ci_p, int Co) {
  const WgradPlan p = wgrad_plan(B, H, W, Ci_p, Co);
  return (size_t)p.nsplit * ((size_t)p.ytiles * p.ztiles * 9 * p.ma * p.nb * 256 + Co) * sizeof(float);
}
template <int MA, int NB, int TW>
static void launch_wgrad(const WgradPlan& p, const void* x, const void* dz, float* ws, int H, int W, int Ci_p, int Co, const WgradRedJob& prev,
                         int zextra, hipStream_t s) {
  dim3 grid(p.nsplit, p.ytiles, p.ztiles + zextra);
  hipLaunchKernelGGL((k_wgrad_mfma<MA, NB, TW>), grid, dim3(256), p.lds, s, (const bf16_t*)x, (const bf16_t*)dz, ws, H, W, Ci_p, Co,
                     p.tiles_x, p.tiles_y, p.ntiles, p.tps, prev, zextra);
}
// slices per output of the partial sum: a stand-alone reduce wants many slices when there are few outputs (fills the chip);
// a reduce that rides in the next layer's launch wants few workgroups and a short chain of loads per thread
static int wgrad_reduce_slices(size_t per_split, int nsplit, bool chained) {
  if (!chained) return per_split >= 65536 ? 4 : per_split >= 8192 ? 16 : 64;
  int S = 1;
  while (S < 64 && nsplit > 8 * S) S *= 2;
  return S;
}
static WgradRedJob wgrad_job_from(const bxWgradPending* pd, bool chained) {
  WgradRedJob jb;
  jb.partial = (const float*)pd->partial; jb.dw = pd->dw; jb.db = pd->db; jb.nsplit = pd->nsplit; jb.Cin = pd->Cin; jb.Co = pd->Co;
  jb.MA = pd->ma; jb.NB = pd->nb; jb.ztiles = pd->ztiles; jb.nfrag4 = pd->nfrag4;
  const size_t per_split = (size_t)pd->nfrag4 * 4 + pd->Co;
  jb.S = wgrad_reduce_slices(per_split, pd->nsplit, chained);
  const int per_wg = 256 / jb.S;
  jb.nblocks = (int)((per_split / 4 + per_wg - 1) / per_wg);
  return jb;
}
int bx_wgrad_mfma_finish(bxWgradPending* pd, hipStream_t s) {
  if (!pd || !pd->valid) return BX_OK;
  const WgradRedJob jb = wgrad_job_from(pd, false);
  hipLaunchKernelGGL(k_wgrad_reduce3, dim3(jb.nblocks), dim3(256), 0, s, jb);
  pd->valid = 0;
  BX_CHECK_LAUNCH("bx_conv3x3_wgrad(mfma reduce)");
  return BX_OK;
}
// pending == NULL: partials are reduced right away.  Otherwise: a valid *pending (previous layer) is reduced inside this launch
// and *pending is overwritten with this layer's unreduced partials (the caller chains it on or calls bx_wgrad_mfma_finish).
int bx_wgrad_mfma_launch(const void* x, const void* dz, float* dw, float* db, int B, int H, int W, int Cin, int Ci_p,
                         int Co, void* ws, size_t ws_bytes, bxWgradPending* pending, hipStream_t s) {
  const WgradPlan p = wgrad_plan(B, H, W, Ci_p, Co);
  (void)ws_bytes;
  BX_REQUIRE((size_t)B * H * W * (Ci_p > Co ? Ci_p : Co) * 2 < ((size_t)1 << 31), "bx_conv3x3_wgrad(mfma): an activation tensor of 2 GiB or more is not supported");
  BX_REQUIRE(!pending || !pending->valid || pending->partial != ws, "bx_conv3x3_wgrad(chained): the pending partials live in this call's workspace");
  float* part = (float*)ws;
  WgradRedJob prev;
  memset(&prev, 0, sizeof(prev));
  int zextra = 0;
  if (pending && pending->valid) {
    prev = wgrad_job_from(pending, true);
    zextra = (prev.nblocks + p.nsplit * p.ytiles - 1) / (p.nsplit * p.ytiles);
  }
  const bool own = wgrad_own_on();
  if (own && p.ma == 2 && p.nb == 2) {
    // tile-owner kernel (k_wgrad_own): 1-D grid = [reduce-role workgroups, padded to a multiple of 8][8 XCD lanes x slots]
    const int nred = pending && pending->valid ? (prev.nblocks + 7) / 8 * 8 : 0;
    const int ranges8 = (p.nsplit + 7) / 8;                           // pixel ranges per XCD lane
    const int nwg = nred + ranges8 * 8 * p.ytiles * p.ztiles;
    const size_t lds = (size_t)10 * (p.tw + 2) * 64 + (size_t)8 * p.tw * 64;
    // two workgroups per CU (256-register budget: 36 accumulators + two fragment sets of 40 + 40 staging registers); BX_WGRAD_DEPTH=2
    // keeps two tiles in flight instead of one
    static const int depth = getenv("BX_WGRAD_DEPTH") ? atoi(getenv("BX_WGRAD_DEPTH")) : 1;
    static const bool shift = !(getenv("BX_WGRAD_SHIFT") && atoi(getenv("BX_WGRAD_SHIFT")) == 0);      // (the third read of the last halo row reaches 128 bytes past the x tile: lds + 256)
    if (wgrad_own_dl()) {
      // three LDS images of (x halo units + dZ units, each padded to whole 256-lane trips) x 16 bytes
      const int nxu = (p.th + 2) * (p.tw + 2) * 4, nzu = p.th * p.tw * 4;
      const size_t lds_dl = (size_t)3 * ((nxu + 255) / 256 + (nzu + 255) / 256) * 256 * 16;
      static bool attr16 = false, attr32 = false;
      if (p.tw == 16) {
        if (!attr16 && lds_dl > 64 * 1024) {
          if (hipFuncSetAttribute((const void*)k_wgrad_own<16, 2, 1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dl) != hipSuccess)
            BX_FAIL(BX_EHIP, "bx_conv3x3_wgrad(mfma): cannot reserve %zu bytes of LDS", lds_dl);
          attr16 = true;
        }
        hipLaunchKernelGGL((k_wgrad_own<16, 2, 1, 8, true>), dim3(nwg), dim3(256), lds_dl, s, (const bf16_t*)x, (const bf16_t*)dz, part, H, W,
                           Ci_p, Co, p.tiles_x, p.tiles_y, p.ntiles, p.tps, p.nsplit, p.ytiles, p.ztiles, prev, nred);
      } else {
        if (!attr32 && lds_dl > 64 * 1024) {
          if (hipFuncSetAttribute((const void*)k_wgrad_own<32, 2, 1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dl) != hipSuccess)
            BX_FAIL(BX_EHIP, "bx_conv3x3_wgrad(mfma): cannot reserve %zu bytes of LDS", lds_dl);
          attr32 = true;
        }
        hipLaunchKernelGGL((k_wgrad_own<32, 2, 1, 4, true>), dim3(nwg), dim3(256), lds_dl, s, (const bf16_t*)x, (const bf16_t*)dz, part, H, W,
                           Ci_p, Co, p.tiles_x, p.tiles_y, p.ntiles, p.tps, p.nsplit, p.ytiles, p.ztiles, prev, nred);
      }
    } else
#define BX_OWN(TW_, D_, SH_) hipLaunchKernelGGL((k_wgrad_own<TW_, 2, D_, 8, false, SH_>), dim3(nwg), dim3(256), lds + 256, s, (const bf16_t*)x, \
                                           (const bf16_t*)dz, part, H, W, Ci_p, Co, p.tiles_x, p.tiles_y, p.ntiles, p.tps, p.nsplit, p.ytiles, p.ztiles, prev, nred)
    if (p.tw == 16) { if (depth == 2) BX_OWN(16, 2, false); else if (shift) BX_OWN(16, 1, true); else BX_OWN(16, 1, false); }
    else            { if (depth == 2) BX_OWN(32, 2, false); else if (shift) BX_OWN(32, 1, true); else BX_OWN(32, 1, false); }
#undef BX_OWN
  } else
#define BX_WG(MA_, NB_) do { if (p.tw == 16) launch_wgrad<MA_, NB_, 16>(p, x, dz, part, H, W, Ci_p, Co, prev, zextra, s); \
                             else launch_wgrad<MA_, NB_, 32>(p, x, dz, part, H, W, Ci_p, Co, prev, zextra, s); } while (0)
  { if (p.ma == 1 && p.nb == 1) BX_WG(1, 1);
  else if (p.ma == 1 && p.nb == 2) BX_WG(1, 2);
  else if (p.ma == 2 && p.nb == 1) BX_WG(2, 1);
  else BX_WG(2, 2); }
#undef BX_WG
  BX_CHECK_LAUNCH("bx_conv3x3_wgrad(mfma)");
  bxWgradPending cur;
  cur.partial = part; cur.dw = dw; cur.db = db; cur.nsplit = p.nsplit; cur.Cin = Cin; cur.Co = Co; cur.ma = p.ma; cur.nb = p.nb;
  cur.ztiles = p.ztiles; cur.nfrag4 = p.ytiles * p.ztiles * 9 * p.ma * p.nb * 64; cur.valid = 1;
  if (pending) { *pending = cur; return BX_OK; }
  return bx_wgrad_mfma_finish(&cur, s);
}
