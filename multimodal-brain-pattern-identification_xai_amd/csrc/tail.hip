// Block tail: 2x2 pool -> BatchNorm2d -> Dropout -> + conv1x1(bilinear_half(x)), forward and backward,
// over channels-last activations.  Replaces reference root/src/models/models.py:67-76 (Block.forward after
// the three convolutions) and its autograd backward.
//
// Thread layout shared by the pixel-wise kernels: a thread owns 8 consecutive channels of one pooled
// pixel; a 256-thread workgroup covers `slots = 256 / (C/8)` pixels at a time and grid-strides.
// Per-channel reductions (BN statistics, BN/bias gradients) go thread -> LDS[slot][C] -> fixed-order
// column sums -> per-workgroup partials in the workspace -> a one-workgroup finalize kernel that sums
// the partials in double precision.  No atomics: results are run-to-run deterministic.
#include "bx_common.h"

#define TAIL_MAX_GROUPS 128         // BxStatTree: bx_stat_tree_shape never makes more groups
#define TAIL_MAX_BLOCKS 2048        // layout bound of the partial buffers; the launch cap is tail_block_cap()

struct TailGeom {
  int B, H, W, Ho, Wo, C, Cin_p, ncg, slots;
  long long npool;
  float sy, sx;  // bilinear scales H/Ho, W/Wo (float, as ATen computes them)
};

static int make_geom(const bxTailDesc* d, TailGeom* g) {
  g->B = d->B; g->H = d->H; g->W = d->W; g->Ho = d->H / 2; g->Wo = d->W / 2; g->C = d->C; g->Cin_p = d->Cin_p;
  if (d->B <= 0 || g->Ho <= 0 || g->Wo <= 0) return -1;
  if (d->C % 8 || d->C > 256 || 256 % (d->C / 8)) return -2;
  if (d->Cin_p % 8 || d->Cin_p <= 0) return -3;
  if ((long long)d->B * d->H * d->W * ((d->Cin_p > d->C ? d->Cin_p : d->C) / 8) >= (1ll << 31)) return -4;   // 32-bit pixel-group indices
  g->ncg = d->C / 8; g->slots = 256 / g->ncg;
  g->npool = (long long)d->B * g->Ho * g->Wo;
  g->sy = (float)d->H / (float)g->Ho; g->sx = (float)d->W / (float)g->Wo;
  return 0;
}
// kernels without per-workgroup partials take one pixel group per workgroup (no serial grid-stride trips)
static int tail_block_cap() {          // workgroups of the statistics / reduction passes (BX_TAIL_BLOCKS, default 512)
  static const int cap = [] { const char* e = getenv("BX_TAIL_BLOCKS"); int v = e ? atoi(e) : 512; return v < 1 ? 1 : v > TAIL_MAX_BLOCKS ? TAIL_MAX_BLOCKS : v; }();
  return cap;
}
static int tail_blocks_all(const TailGeom& g) {
  long long nb = (g.npool + g.slots - 1) / g.slots;
  return (int)(nb > 65535 ? 65535 : nb);
}
static int tail_blocks(const TailGeom& g) {
  long long nb = (g.npool + g.slots - 1) / g.slots;
  return (int)(nb > tail_block_cap() ? tail_block_cap() : nb);
}

// pooled-pixel index -> (sample, row, column) with 32-bit unsigned divisions (the launchers check B*H*W < 2^31): the
// 64-bit / and % these kernels used compile to ~70 VALU instructions each, four per pixel
__device__ __forceinline__ void px_decode(long long pp, int Wn, int Hn, int& xx, int& yy, int& b) {
  const unsigned p = (unsigned)pp;
  const unsigned r = p / (unsigned)Wn;
  xx = (int)(p - r * (unsigned)Wn);
  const unsigned bb = r / (unsigned)Hn;
  yy = (int)(r - bb * (unsigned)Hn);
  b = (int)bb;
}

// fixed-order column reduction of NV per-thread 8-vectors through LDS; result for channel c in thread c (< C)
template <int NV>
__device__ __forceinline__ void block_channel_reduce(float (*vals)[8], float* lds, int C, int ncg, int slots, float out[NV]) {
  const int cg = threadIdx.x % ncg, slot = threadIdx.x / ncg;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[slot * C + cg * 8 + j] = vals[k][j];
    __syncthreads();
    float s = 0.f;
    if ((int)threadIdx.x < C)
      for (int sl = 0; sl < slots; ++sl) s += lds[sl * C + threadIdx.x];
    out[k] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// forward 1: pooled = pool2x2(y3) (stored, needed again by backward) + per-workgroup (sum, sumsq)
template <typename T>
__global__ __launch_bounds__(256) void k_pool_stats(const T* __restrict__ y3, T* __restrict__ pooled, float* partials,
                                                     TailGeom g, int pool, int want_stats, const float* __restrict__ w1x1, int Cin,
                                                     float* __restrict__ wT, BxStatTree tree, BxBnFinalize fin) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  {                                                     // wT[ci][c] = W1x1[c][ci] (zero rows for padded inputs): read by k_tail_apply
    const int nw = g.Cin_p * g.C, per = (nw + (int)gridDim.x - 1) / (int)gridDim.x;     // each workgroup transposes a slice
    const int lo = (int)blockIdx.x * per, cnt = nw - lo < per ? nw - lo : per;
    if (cnt > 0)
      lds_fill<4>(wT + lo, cnt, [&](int k) { const int i = lo + k, ci = i / g.C, c = i % g.C; return ci < Cin ? w1x1[(size_t)c * Cin + ci] : 0.f; });
  }
  const int cg = threadIdx.x % g.ncg, slot = threadIdx.x / g.ncg;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  // two pooled pixels per trip: all eight window loads are issued before the first use (the trips of a grid-stride
  // loop are otherwise serial memory round trips)
  const long long stride = (long long)gridDim.x * g.slots;
  for (long long pp0 = (long long)blockIdx.x * g.slots + slot; pp0 < g.npool; pp0 += 2 * stride) {
    float w[2][4][8];
    bool ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long pp = pp0 + h * stride;
      ok[h] = pp < g.npool;
      if (ok[h]) {
        int ox, oy, b;
        px_decode(pp, g.Wo, g.Ho, ox, oy, b);
        const size_t base = (((size_t)b * g.H + 2 * oy) * g.W + 2 * ox) * g.C + cg * 8;
        ld8(y3, base, w[h][0]); ld8(y3, base + g.C, w[h][1]);
        ld8(y3, base + (size_t)g.W * g.C, w[h][2]); ld8(y3, base + (size_t)g.W * g.C + g.C, w[h][3]);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!ok[h]) continue;
      const long long pp = pp0 + h * stride;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = pool == BX_POOL_MAX ? fmaxf(fmaxf(w[h][0][j], w[h][1][j]), fmaxf(w[h][2][j], w[h][3][j]))
                                   : 0.25f * (w[h][0][j] + w[h][1][j] + w[h][2][j] + w[h][3][j]);
      st8(pooled, (size_t)pp * g.C + cg * 8, v);
      if (want_stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {          // statistics of the values as stored (bf16-rounded if bf16)
          const float q = round_as(pooled, v[j]);
          acc[0][j] += q; acc[1][j] += q * q;
        }
      }
    }
  }
  if (!want_stats) return;
  float red[2];
  block_channel_reduce<2>(acc, lds, g.C, g.ncg, g.slots, red);
  if (tree.cnt) {                                       // the last workgroup to arrive finalizes (no k_bn_finalize launch)
    double tot[2];
    __syncthreads();                                    // block_channel_reduce's LDS reads are done
    if (bx_stat_tree_arrive<2>(tree, 0, 0, blockIdx.x, red, tot, reinterpret_cast<char*>(lds)) && (int)threadIdx.x < g.C)
      bx_bn_finalize_channel(threadIdx.x, tot[0], tot[1], fin);
    return;
  }
  if ((int)threadIdx.x < g.C) {
    partials[((size_t)blockIdx.x * 2 + 0) * g.C + threadIdx.x] = red[0];
    partials[((size_t)blockIdx.x * 2 + 1) * g.C + threadIdx.x] = red[1];
  }
}

// 8 consecutive elements held in their storage form (a load issued early costs 4 registers for bf16, not 8)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 r;
  __device__ __forceinline__ void load(const bf16_t* p, size_t i) { r = *reinterpret_cast<const uint4*>(p + i); }
  __device__ __forceinline__ void get(float v[8]) const {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p, size_t i) { a = *reinterpret_cast<const float4*>(p + i); b = *reinterpret_cast<const float4*>(p + i + 4); }
  __device__ __forceinline__ void get(float v[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};

// bilinear sample of the block input at pooled pixel (oy, ox): 8 channels starting at c0
template <typename T>
__device__ __forceinline__ void skip_sample(const T* __restrict__ x, const TailGeom& g, int b, int oy, int ox, int c0, float out[8]) {
  int y0, y1, x0, x1; float ly, lx;
  bilinear_src(oy, g.sy, g.H, y0, y1, ly);
  bilinear_src(ox, g.sx, g.W, x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  float p00[8], p01[8], p10[8], p11[8];
  const size_t rb = (size_t)b * g.H;
  ld8(x, ((rb + y0) * g.W + x0) * g.Cin_p + c0, p00); ld8(x, ((rb + y0) * g.W + x1) * g.Cin_p + c0, p01);
  ld8(x, ((rb + y1) * g.W + x0) * g.Cin_p + c0, p10); ld8(x, ((rb + y1) * g.W + x1) * g.Cin_p + c0, p11);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = hy * (hx * p00[j] + lx * p01[j]) + ly * (hx * p10[j] + lx * p11[j]);
}

// forward 3: out = dropout(pooled*scale + shift) + b1x1 + W1x1 . bilinear(x)
// RPT = weight rows per trip of the 1x1 loop (2 RPT loads in flight).  The late stages take ~12 us here whatever Cin_p is
// (32, 64 or 128 rows): neither 16 rows per trip nor 4 pixels per thread (4x fewer weight loads, 14 us) moved it -- the time is
// the fixed chain of cold round trips after the kernel boundary (scale/shift, skip input, first weight rows), not the loop.
// Folded finalize (round 3, TailFwdPro.rows != NULL): no k_bn_finalize launch between the statistics pass and this kernel -- every
// workgroup sums the producer's partial rows itself (bx_rows_total) and derives (scale, shift); workgroup 0 also writes the running
// statistics and the saved mean / invstd the backward reads.
struct TailFwdPro { const float* rows; int nrows; BxBnFinalize fin; };
template <typename T, int RPT>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 3 : 4) void k_tail_apply(const T* __restrict__ pooled, const T* __restrict__ x, const float* __restrict__ wT,
    int Cin, const float* __restrict__ b1x1, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint64_t* __restrict__ seed, float dropout_p, uint32_t salt, T* __restrict__ out, TailGeom g,
    const float* __restrict__ ev_gamma, const float* __restrict__ ev_beta, const float* __restrict__ ev_rmean,
    const float* __restrict__ ev_rvar, float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd, TailFwdPro pro) {
  extern __shared__ float xs[];  // [slots][Cin_p + 1]
  const int cg = threadIdx.x % g.ncg, slot = threadIdx.x / g.ncg;
  const int xstride = g.Cin_p + 1, nci8 = g.Cin_p / 8;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  float sc[8], sh[8], bb[8];
  if (pro.rows) {
    __shared__ __attribute__((aligned(8))) double sp[BX_ROWS_TOTAL_LDS(2) / 8];
    __shared__ float ssc[2][256];
    bx_rows_total<2>(pro.rows, pro.nrows, g.C, sp);
    if ((int)threadIdx.x < g.C) {
      const int c = threadIdx.x;
      const BxBnFinalize& f = pro.fin;
      float mean, invstd; double unbiased;
      bx_bn_batch_stats(sp[c], sp[g.C + c], f.count, f.eps, mean, invstd, unbiased);
      const float s_ = f.gamma[c] * invstd, h_ = f.beta[c] - mean * s_;
      ssc[0][c] = s_; ssc[1][c] = h_;
      if (blockIdx.x == 0) {                              // the one writer of the module's state and of what the backward reads
        f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * mean;
        f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unbiased;
        if (c == 0 && f.nbt) f.nbt[0] += 1;
        f.scale[c] = s_; f.shift[c] = h_; f.save_mean[c] = mean; f.save_invstd[c] = invstd;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = ssc[0][cg * 8 + j]; sh[j] = ssc[1][cg * 8 + j]; bb[j] = b1x1[cg * 8 + j]; }
  } else if (ev_gamma) {
    // eval mode: scale / shift come straight from the running statistics (no finalize launch); workgroup 0 also
    // publishes them as the saved mean / invstd an eval-mode backward (saliency, IG, inner Grad-CAM targets) reads
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      const float is = 1.0f / sqrtf(ev_rvar[c] + eps), s_ = ev_gamma[c] * is;
      sc[j] = s_; sh[j] = ev_beta[c] - ev_rmean[c] * s_; bb[j] = b1x1[c];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < g.C) {
      save_mean[threadIdx.x] = ev_rmean[threadIdx.x];
      save_invstd[threadIdx.x] = 1.0f / sqrtf(ev_rvar[threadIdx.x] + eps);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; bb[j] = b1x1[cg * 8 + j]; }
  }
  const long long nsteps = (g.npool + (long long)gridDim.x * g.slots - 1) / ((long long)gridDim.x * g.slots);
  for (long long st = 0; st < nsteps; ++st) {
    const long long p0 = (st * gridDim.x + blockIdx.x) * g.slots;
    // this thread's pooled values do not depend on the staged skip input: request them before the staging round trip
    Raw8<T> pvr;                                             // kept packed (4 registers for bf16) until it is used
    {
      const long long ppq = p0 + slot;
      pvr.load(pooled, (size_t)(ppq < g.npool ? ppq : 0) * g.C + cg * 8);
    }
    __syncthreads();
    for (int u = threadIdx.x; u < g.slots * nci8; u += 256) {
      const int sl = u / nci8, c8 = u % nci8;
      const long long pp = p0 + sl;
      if (pp < g.npool) {
        int ox, oy, b;
        px_decode(pp, g.Wo, g.Ho, ox, oy, b);
        float v[8];
        skip_sample(x, g, b, oy, ox, c8 * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[sl * xstride + c8 * 8 + j] = v[j];
      }
    }
    __syncthreads();
    const long long pp = p0 + slot;
    if (pp >= g.npool) continue;
    float pv[8], acc[8];
    pvr.get(pv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = pv[j] * sc[j] + sh[j];
      if (dropout_p > 0.f) v *= bx_dropout_scale(sd, salt, (uint64_t)pp * g.C + cg * 8 + j, dropout_p, inv_keep);
      acc[j] = v + bb[j];
    }
    // wT has Cin_p rows (zero beyond Cin) and Cin_p % RPT == 0.  The late stages (64 / 128 rows, one wave per SIMD) are a chain of
    // L2 round trips: the rows of trip k + 1 are requested before the FMAs of trip k (half-size trips, two register sets).  Written
    // as plain "load RPT rows, use them" the scalar-FMA build left every trip's latency exposed (stage 5: 16 -> 26 us).
    constexpr int H = RPT / 4;                       // (two register sets of RPT / 4 rows: RPT / 2 spilled 66 registers at the 128-register bound)
    float4 wa[2][H], wb[2][H];
    auto load_rows = [&](int c0, float4 (&a)[H], float4 (&b)[H]) {
#pragma unroll
      for (int u = 0; u < H; ++u) {
        a[u] = *reinterpret_cast<const float4*>(wT + (size_t)(c0 + u) * g.C + cg * 8);
        b[u] = *reinterpret_cast<const float4*>(wT + (size_t)(c0 + u) * g.C + cg * 8 + 4);
      }
    };
    auto use_rows = [&](int c0, const float4 (&a)[H], const float4 (&b)[H]) {
#pragma unroll
      for (int u = 0; u < H; ++u) {
        const float xv = xs[slot * xstride + c0 + u];
        acc[0] = fmaf(a[u].x, xv, acc[0]); acc[1] = fmaf(a[u].y, xv, acc[1]); acc[2] = fmaf(a[u].z, xv, acc[2]); acc[3] = fmaf(a[u].w, xv, acc[3]);
        acc[4] = fmaf(b[u].x, xv, acc[4]); acc[5] = fmaf(b[u].y, xv, acc[5]); acc[6] = fmaf(b[u].z, xv, acc[6]); acc[7] = fmaf(b[u].w, xv, acc[7]);
      }
    };
    load_rows(0, wa[0], wb[0]);
    for (int c0 = 0; c0 < g.Cin_p; c0 += 2 * H) {        // Cin_p % RPT == 0: an even number of half-trips
      load_rows(c0 + H, wa[1], wb[1]);
      use_rows(c0, wa[0], wb[0]);
      if (c0 + 2 * H < g.Cin_p) load_rows(c0 + 2 * H, wa[0], wb[0]);
      use_rows(c0 + H, wa[1], wb[1]);
    }
    st8(out, (size_t)pp * g.C + cg * 8, acc);
  }
}

static int g_tree_max_rows = -1;
int bx_tree_max_rows() {
  if (g_tree_max_rows < 0) { const char* e = getenv("BX_TREE_MAX_ROWS"); g_tree_max_rows = e ? atoi(e) : 0; }
  return g_tree_max_rows;
}
extern "C" int bx_set_tree_max_rows(int rows) {
  BX_REQUIRE(rows >= 0, "bx_set_tree_max_rows: rows must be >= 0");
  g_tree_max_rows = rows;
  return BX_OK;
}
// Folded finalizes (consumer-side sums, bx_rows_total): BX_TAIL_FOLD bit 0 = backward (no k_tail_bwd_mid), bit 1 = forward (no
// k_bn_finalize); default 3.  BX_TAIL_FOLD_RC bounds rows x C of a folded pass (the backward reduction role launches that many rows,
// the forward folds only when its producer wrote no more), BX_TAIL_FOLD_GRID caps the consuming kernel's workgroups (each re-reads the rows).
static int g_tail_fold = -1;
static int tail_fold() {
  if (g_tail_fold < 0) { const char* e = getenv("BX_TAIL_FOLD"); g_tail_fold = e ? atoi(e) & 3 : 3; }
  return g_tail_fold;
}
int bx_tail_fold_mask() { return tail_fold(); }            // (the EEG branch's forward finalizes follow bit 1 too)
extern "C" int bx_set_tail_fold(int mask) {
  BX_REQUIRE(mask >= 0 && mask <= 3, "bx_set_tail_fold: mask is a combination of 1 (backward) and 2 (forward)");
  g_tail_fold = mask;
  return BX_OK;
}
static int tail_fold_rc(bool fwd) {
  static const int rc_b = getenv("BX_TAIL_FOLD_RC") ? atoi(getenv("BX_TAIL_FOLD_RC")) : 8192;
  static const int rc_f = getenv("BX_TAIL_FOLD_RC_FWD") ? atoi(getenv("BX_TAIL_FOLD_RC_FWD")) : 32768;   // (16384 and 32768 measure the same; 32768 also folds stage 3: one launch fewer)
  return fwd ? rc_f : rc_b;
}
static int tail_fold_grid() {
  static const int v = getenv("BX_TAIL_FOLD_GRID") ? atoi(getenv("BX_TAIL_FOLD_GRID")) : 512;       // measured: 512 1.498 ms, 1024 1.500, 4096 1.514
  return v < 1 ? 1 : v;
}
static bool tail_fold_ok(const TailGeom& g) { return g.C <= 256 && bx_rows_total_ok(g.C); }
static int tail_fold_rows(int C) { const int r = tail_fold_rc(false) / C; return r < 16 ? 16 : r > TAIL_MAX_BLOCKS ? TAIL_MAX_BLOCKS : r; }
// rows of forward partial sums: the pooling kernel writes at most TAIL_MAX_BLOCKS, conv3's pooled epilogue one per 8 x 16 (or larger) tile
static size_t tail_rows_cap(const bxTailDesc* d) {
  const size_t conv_rows = (size_t)d->B * ((d->H + 7) / 8) * ((d->W + 15) / 16);
  return conv_rows > TAIL_MAX_BLOCKS ? conv_rows : TAIL_MAX_BLOCKS;
}
extern "C" size_t bx_block_tail_route_bytes(const bxTailDesc* d) {
  TailGeom g;
  if (!d || make_geom(d, &g)) return 0;
  return (size_t)g.npool * g.C / 2;                       // C % 8 == 0: a thread's 8 channels are one 32-bit word
}
extern "C" size_t bx_block_tail_workspace(const bxTailDesc* d) {
  TailGeom g;
  if (!d || make_geom(d, &g)) return 0;
  const size_t Ho = d->H / 2, Wo = d->W / 2;
  size_t fwd = ((size_t)tail_rows_cap(d) * 2 + 2) * d->C * sizeof(float) + (size_t)d->Cin_p * d->C * sizeof(float)
             + (size_t)TAIL_MAX_GROUPS * 2 * d->C * sizeof(double);                                  // reduction-tree group sums
  size_t bwd = ((size_t)TAIL_MAX_BLOCKS * 3 + 3) * d->C * sizeof(float)                 // partials + coefficients
             + bx_align_up((size_t)d->B * Ho * Wo * d->Cin_p * sizeof(float), 256)       // dXs (half-res, fp32)
             + (size_t)2048 * 256 * sizeof(float) + (size_t)65 * d->C * d->Cin_p * sizeof(float);   // conv1x1 weight-grad partials (+1 slab for compaction)
  return bx_align_up(fwd > bwd ? fwd : bwd, 256) + (size_t)TAIL_MAX_GROUPS * 3 * d->C * sizeof(double);    // + backward reduction-tree group sums
}

extern "C" int bx_block_tail_fwd(const bxTailDesc* d, const void* y3, const void* x, const float* w1x1, int Cin,
                                 const float* b1x1, const float* bn_weight, const float* bn_bias,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 const uint64_t* seed, void* pooled, void* out, float* save_mean, float* save_invstd,
                                 void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(d && y3 && x && w1x1 && b1x1 && bn_weight && bn_bias && running_mean && running_var && pooled && out && save_mean && save_invstd,
             "bx_block_tail_fwd: null pointer");
  BX_DTYPE_OK(d->dtype);
  TailGeom g;
  const int ge = make_geom(d, &g);
  BX_REQUIRE(ge == 0, "bx_block_tail_fwd: unsupported geometry (code %d): need C%%8==0, C<=256, 256%%(C/8)==0, Cin_p%%8==0, H,W>=2", ge);
  BX_REQUIRE(Cin > 0 && Cin <= d->Cin_p, "bx_block_tail_fwd: Cin=%d exceeds Cin_p=%d", Cin, d->Cin_p);
  BX_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "bx_block_tail_fwd: dropout_p must be in [0,1)");
  BX_REQUIRE(d->dropout_p == 0.f || !d->training || seed, "bx_block_tail_fwd: dropout needs a device seed");
  const size_t need = bx_block_tail_workspace(d);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_block_tail_fwd: workspace %zu < %zu", workspace_bytes, need);
  const size_t xs_bytes = (size_t)g.slots * (g.Cin_p + 1) * sizeof(float);
  BX_REQUIRE(xs_bytes <= 60 * 1024, "bx_block_tail_fwd: Cin_p too large for the LDS tile");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = tail_blocks(g);
  float* partials = (float*)workspace;
  float* scale = partials + tail_rows_cap(d) * 2 * g.C;
  float* shift = scale + g.C;
  float* wT = shift + g.C;
  const float p = d->training ? d->dropout_p : 0.f;
  // with d->sync the statistics finalize rides in the pooling kernel (last workgroup to arrive, see BxStatTree)
  const bool in_launch = d->training && d->sync && 256 % g.C == 0 && nblk <= bx_tree_max_rows();
  BxStatTree tree = {};
  BxBnFinalize fin = {};
  if (in_launch) {
    tree.rows = partials; tree.cnt = d->sync + BX_TAIL_SYNC_FWD; tree.nrows = nblk; tree.Cw = tree.ld = g.C;
    bx_stat_tree_shape(nblk, &tree.G, &tree.ngroups);
    tree.mid = (double*)(wT + (size_t)g.Cin_p * g.C);                     // ngroups x 2 x C doubles (workspace formula)
    BX_REQUIRE(tree.ngroups + 1 <= BX_TAIL_SYNC_BWD - BX_TAIL_SYNC_FWD, "bx_block_tail_fwd: %d reduction groups exceed the sync block", tree.ngroups);
    fin = BxBnFinalize{bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, d->momentum, d->eps, scale, shift, save_mean, save_invstd,
                       (double)g.npool};
  }
  BX_DISPATCH_DTYPE(d->dtype, T,
    hipLaunchKernelGGL((k_pool_stats<T>), dim3(nblk), dim3(256), 0, s, (const T*)y3, (T*)pooled, partials, g, d->pool, d->training,
                       w1x1, Cin, wT, tree, fin));
  BX_CHECK_LAUNCH("bx_block_tail_fwd(pool)");
  // folded finalize: the apply kernel's workgroups sum the partial rows themselves (no k_bn_finalize launch)
  const bool fold = d->training && !in_launch && (tail_fold() & 2) && tail_fold_ok(g) && (long long)nblk * g.C <= tail_fold_rc(true);
  TailFwdPro pro = {};
  if (fold)
    pro = TailFwdPro{partials, nblk, BxBnFinalize{bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, d->momentum, d->eps, scale, shift,
                                                 save_mean, save_invstd, (double)g.npool}};
  if (d->training && !in_launch && !fold) {
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.C)), dim3(1024), 0, s, partials, nblk, (double)g.npool, g.C, d->training, bn_weight, bn_bias,
                       running_mean, running_var, num_batches_tracked, d->momentum, d->eps, scale, shift, save_mean, save_invstd);
    BX_CHECK_LAUNCH("bx_block_tail_fwd(finalize)");
  }
  const bool ev = !d->training;
  const int napply = fold && tail_blocks_all(g) > tail_fold_grid() ? tail_fold_grid() : tail_blocks_all(g);
  BX_DISPATCH_DTYPE(d->dtype, T,
    hipLaunchKernelGGL((k_tail_apply<T, 8>), dim3(napply), dim3(256), xs_bytes, s, (const T*)pooled, (const T*)x, wT, Cin, b1x1,
                       scale, shift, seed, p, d->salt, (T*)out, g, ev ? bn_weight : (const float*)nullptr, bn_bias, (const float*)running_mean,
                       (const float*)running_var, d->eps, save_mean, save_invstd, pro));
  BX_CHECK_LAUNCH("bx_block_tail_fwd(apply)");
  return BX_OK;
}

// conv3 + tail forward in two launches (bf16 MFMA path): conv3's epilogue pools the tile it holds in registers, sums the batch
// statistics and (last workgroup per channel group) finalizes them; the apply kernel follows.  Replaces conv3 -> k_pool_stats ->
// k_bn_finalize -> k_tail_apply: the 2x2 pool no longer re-reads conv3's output from HBM.
extern "C" int bx_block_conv3_tail_fwd(const bxTailDesc* d, const void* y2, const void* w3_mfma, const float* b3, void* y3,
                                       const void* x, const float* w1x1, int Cin, const float* b1x1, const float* bn_weight, const float* bn_bias,
                                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                       const uint64_t* seed, void* pooled, void* out, float* save_mean, float* save_invstd,
                                       void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(d && y2 && w3_mfma && b3 && x && w1x1 && b1x1 && bn_weight && bn_bias && running_mean && running_var && pooled && out && save_mean
             && save_invstd, "bx_block_conv3_tail_fwd: null pointer");            // (y3 may be NULL: conv3's output is then not stored)
  if (d->dtype != BX_BF16 || !bx_conv3x3_mfma_supported(d->C, d->C, d->dtype) || d->C < 16)
    BX_FAIL(BX_EUNSUPPORTED, "bx_block_conv3_tail_fwd: needs bf16 storage and an MFMA-capable channel count (C=%d dtype=%d)", d->C, d->dtype);
  TailGeom g;
  const int ge = make_geom(d, &g);
  BX_REQUIRE(ge == 0, "bx_block_conv3_tail_fwd: unsupported geometry (code %d)", ge);
  BX_REQUIRE(Cin > 0 && Cin <= d->Cin_p, "bx_block_conv3_tail_fwd: Cin=%d exceeds Cin_p=%d", Cin, d->Cin_p);
  BX_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "bx_block_conv3_tail_fwd: dropout_p must be in [0,1)");
  BX_REQUIRE(d->dropout_p == 0.f || !d->training || seed, "bx_block_conv3_tail_fwd: dropout needs a device seed");
  const size_t need = bx_block_tail_workspace(d);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_block_conv3_tail_fwd: workspace %zu < %zu", workspace_bytes, need);
  const size_t xs_bytes = (size_t)g.slots * (g.Cin_p + 1) * sizeof(float);
  BX_REQUIRE(xs_bytes <= 60 * 1024, "bx_block_conv3_tail_fwd: Cin_p too large for the LDS tile");
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  float* scale = partials + tail_rows_cap(d) * 2 * g.C;
  float* shift = scale + g.C;
  float* wT = shift + g.C;
  BxConvPoolEpi pe = {};
  pe.pooled = pooled; pe.pool = d->pool; pe.want_stats = d->training; pe.Ho = g.Ho; pe.Wo = g.Wo;
  pe.w1x1 = w1x1; pe.wT = wT; pe.Cin1 = Cin; pe.Cin1_p = g.Cin_p;
  pe.route = d->route; pe.route_bytes = (uint32_t)bx_block_tail_route_bytes(d); pe.store_y = y3 != nullptr;
  if (d->training) {
    pe.tree.rows = partials; pe.tree.mid = (double*)(wT + (size_t)g.Cin_p * g.C); pe.tree.cnt = d->sync ? d->sync + BX_TAIL_SYNC_FWD : nullptr;
    pe.rows_cap_floats = tail_rows_cap(d) * 2 * g.C; pe.cnt_cap_words = BX_TAIL_SYNC_BWD - BX_TAIL_SYNC_FWD; pe.tree_max_rows = bx_tree_max_rows();
    pe.fin = BxBnFinalize{bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, d->momentum, d->eps, scale, shift, save_mean, save_invstd,
                          (double)g.npool};
  }
  if (g_bx_prof_ev[0]) (void)hipEventRecord(g_bx_prof_ev[0], s);          // bx_profile_next_conv3: bracket the convolution kernel only
  const int rc = bx_conv3x3_mfma_pool_launch(y2, w3_mfma, b3, y3, d->B, d->H, d->W, d->C, d->C, &pe, s);
  if (g_bx_prof_ev[1]) (void)hipEventRecord(g_bx_prof_ev[1], s);
  g_bx_prof_ev[0] = g_bx_prof_ev[1] = nullptr;
  if (rc != BX_OK) return rc;
  // few partial rows (the late stages: one per 8 x 32 tile or per image): the apply kernel's workgroups sum them themselves
  // (and at most 1 024 rows: stage 1's 2 048 rows x 16 channels fit the rows x C bound, but its apply kernel -- HBM-bound, 4 096 one-trip
  // workgroups -- loses more under the 512-workgroup cap of a folded consumer (23 -> 33 us) than the 8 us finalize launch costs)
  const bool fold = d->training && !pe.tree.cnt && (tail_fold() & 2) && tail_fold_ok(g) && (long long)pe.tree.nrows * g.C <= tail_fold_rc(true)
                    && pe.tree.nrows <= 1024;
  TailFwdPro pro = {};
  if (fold) pro = TailFwdPro{partials, pe.tree.nrows, pe.fin};
  if (d->training && !pe.tree.cnt && !fold) {           // many partial rows: the separate finalize launch (its workgroups split the channels)
    hipLaunchKernelGGL(k_bn_finalize, dim3(bx_finalize_grid(g.C)), dim3(1024), 0, s, partials, pe.tree.nrows, (double)g.npool, g.C, 1, bn_weight, bn_bias,
                       running_mean, running_var, num_batches_tracked, d->momentum, d->eps, scale, shift, save_mean, save_invstd);
    BX_CHECK_LAUNCH("bx_block_conv3_tail_fwd(finalize)");
  }
  const float p = d->training ? d->dropout_p : 0.f;
  const bool ev = !d->training;
  const int napply = fold && tail_blocks_all(g) > tail_fold_grid() ? tail_fold_grid() : tail_blocks_all(g);
  hipLaunchKernelGGL((k_tail_apply<bf16_t, 8>), dim3(napply), dim3(256), xs_bytes, s, (const bf16_t*)pooled, (const bf16_t*)x, wT, Cin, b1x1,
                     scale, shift, seed, p, d->salt, (bf16_t*)out, g, ev ? bn_weight : (const float*)nullptr, bn_bias, (const float*)running_mean,
                     (const float*)running_var, d->eps, save_mean, save_invstd, pro);
  BX_CHECK_LAUNCH("bx_block_conv3_tail_fwd(apply)");
  return BX_OK;
}

// ================================================================================================
// backward
// reduce: per channel  s1 = sum dD,  s2 = sum dD * xhat,  s3 = sum dOut   (dD = dOut * dropout multiplier)
// what the BatchNorm-backward finalize needs besides the three sums
struct TailBwdFin { const float* gamma; const float* invstd; float* coef; float* dgamma; float* dbeta; float* db1x1; double count; int training; };
__device__ __forceinline__ void tail_bwd_finalize_channel(int c, int C, const double (&s)[3], const TailBwdFin& f) {
  if (f.dbeta) f.dbeta[c] = (float)s[0];
  if (f.dgamma) f.dgamma[c] = (float)s[1];
  if (f.db1x1) f.db1x1[c] = (float)s[2];
  f.coef[c] = f.gamma[c] * f.invstd[c];
  f.coef[C + c] = f.training ? (float)(s[0] / f.count) : 0.f;
  f.coef[2 * C + c] = f.training ? (float)(s[1] / f.count) : 0.f;
}
template <typename T, int NPT = 2>
__device__ __forceinline__ void tail_bwd_reduce_body(const T* __restrict__ dout, const T* __restrict__ pooled,
    const float* __restrict__ mean, const float* __restrict__ invstd, const uint64_t* __restrict__ seed, float dropout_p,
    uint32_t salt, float* partials, const TailGeom& g, int bid, int nblocks, const BxStatTree& tree, const TailBwdFin& fin) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  const int cg = threadIdx.x % g.ncg, slot = threadIdx.x / g.ncg;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  float mu[8], is[8], acc[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mu[j] = mean[cg * 8 + j]; is[j] = invstd[cg * 8 + j]; acc[0][j] = acc[1][j] = acc[2][j] = 0.f; }
  const long long stride = (long long)nblocks * g.slots;
  // NPT pixels per trip: 4 in the first stage's launch (MF = 0: with 2 the role moved 34 MB in 16.9 us; front kernel 33.8 -> 31.9 us),
  // 2 elsewhere (4 costs the MFMA instantiations a workgroup per CU: stage 2 20.5 -> 23.2 us)
  for (long long pp0 = (long long)bid * g.slots + slot; pp0 < g.npool; pp0 += NPT * stride) {
    float go[NPT][8], pv[NPT][8];
    bool ok[NPT];
#pragma unroll
    for (int h = 0; h < NPT; ++h) {
      const long long pp = pp0 + h * stride;
      ok[h] = pp < g.npool;
      if (ok[h]) { ld8(dout, (size_t)pp * g.C + cg * 8, go[h]); ld8(pooled, (size_t)pp * g.C + cg * 8, pv[h]); }
    }
#pragma unroll
    for (int h = 0; h < NPT; ++h) {
      if (!ok[h]) continue;
      const long long pp = pp0 + h * stride;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float dd = go[h][j];
        if (dropout_p > 0.f) dd *= bx_dropout_scale(sd, salt, (uint64_t)pp * g.C + cg * 8 + j, dropout_p, inv_keep);
        acc[0][j] += dd;
        acc[1][j] += dd * (pv[h][j] - mu[j]) * is[j];
        acc[2][j] += go[h][j];
      }
    }
  }
  float red[3];
  block_channel_reduce<3>(acc, lds, g.C, g.ncg, g.slots, red);
  if (tree.cnt) {                                       // the last reduction workgroup to arrive finalizes (no mid launch)
    double tot[3];
    __syncthreads();
    if (bx_stat_tree_arrive<3>(tree, 0, 0, bid, red, tot, reinterpret_cast<char*>(lds)) && (int)threadIdx.x < g.C)
      tail_bwd_finalize_channel(threadIdx.x, g.C, tot, fin);
    return;
  }
  if ((int)threadIdx.x < g.C)
#pragma unroll
    for (int k = 0; k < 3; ++k) partials[((size_t)bid * 3 + k) * g.C + threadIdx.x] = red[k];
}

__device__ __forceinline__ void tail_bwd_finalize_body(const float* __restrict__ partials, int nblk, double count, int C, int training,
                                    const float* __restrict__ gamma, const float* __restrict__ invstd, float* __restrict__ coef,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db1x1, int bid) {
  double s[3] = {0.0, 0.0, 0.0};
  const int CG = C < 16 ? C : 16, c0 = bid * CG;                   // bx_finalize_grid(C) workgroups
  sum_partials_256<3>(partials, nblk, C, c0, CG, s);
  const int c = c0 + threadIdx.x;
  if ((int)threadIdx.x >= CG || c >= C) return;
  tail_bwd_finalize_channel(c, C, s, TailBwdFin{gamma, invstd, coef, dgamma, dbeta, db1x1, count, training});
}

// fixed-order sum of the 1x1 weight-gradient partials: one 256-thread group covers NO = 256/S outputs x S chunk slices;
// (c, ci) of the padded layout goes to dw[c][ci < Cin].  Runs as a role of k_tail_bwd_mid (four groups per workgroup) or, when
// the finalize rides in the front kernel, of k_tail_bwd_apply (one group per extra workgroup) -- the sums are the same.
__device__ __forceinline__ void w1x1_sum_group(const float* __restrict__ wpart, float* __restrict__ dw, int nchunk, int C, int Cin, int Cin_p,
                                               int S, int group, int t, float* sm) {
  const int n = C * Cin_p, NO = 256 / S;
  const int o = t % NO, sl = t / NO;
  const int i = group * NO + o;
  float s = 0.f;
  if (i < n) {
    int k = sl;
    for (; k + 3 * S < nchunk; k += 4 * S) {
      const float v0 = wpart[(size_t)k * n + i], v1 = wpart[(size_t)(k + S) * n + i];
      const float v2 = wpart[(size_t)(k + 2 * S) * n + i], v3 = wpart[(size_t)(k + 3 * S) * n + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nchunk; k += S) s += wpart[(size_t)k * n + i];
  }
  sm[t] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    float r = sm[o];
    for (int k = 1; k < S; ++k) r += sm[k * NO + o];
    const int c = i / Cin_p, ci = i - c * Cin_p;
    if (ci < Cin) dw[(size_t)c * Cin + ci] = r;
  }
}
struct TailWsum { const float* wpart; float* dw; int nchunk, Cin, S, n_apply; };     // n_apply = workgroups of the apply role (0: no sum role)
// folded finalize (round 3): rows != NULL -- no k_tail_bwd_mid launch; every apply workgroup sums the reduction role's partial rows
// itself (bx_rows_total) and derives the three coefficients, workgroup 0 writes the parameter gradients
struct TailBwdPro { const float* rows; int nrows; TailBwdFin fin; };

// apply: dP = a*(dD - k1 - xhat*k2); route through the 2x2 pool and conv3's ReLU to full resolution
template <typename T>
__global__ __launch_bounds__(256) void k_tail_bwd_apply(const T* __restrict__ dout, const T* __restrict__ pooled, const T* __restrict__ y3,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
    const uint64_t* __restrict__ seed, float dropout_p, uint32_t salt, int pool, T* __restrict__ dz3, TailGeom g, TailWsum ws, TailBwdPro pro,
    const uint32_t* __restrict__ route) {
  if (ws.n_apply && (int)blockIdx.x >= ws.n_apply) {            // extra workgroups: the 1x1 weight-gradient sum rides here
    __shared__ float sm[256];
    w1x1_sum_group(ws.wpart, ws.dw, ws.nchunk, g.C, ws.Cin, g.Cin_p, ws.S, (int)blockIdx.x - ws.n_apply, threadIdx.x, sm);
    return;
  }
  const int cg = threadIdx.x % g.ncg, slot = threadIdx.x / g.ncg;
  const uint64_t sd = (dropout_p > 0.f && seed) ? seed[0] : 0;
  const float inv_keep = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  float mu[8], is[8], a[8], k1[8], k2[8];
  // Software pipeline: a pixel group's loads (dOut, pooled, route word) are issued one trip ahead -- the first group's before the
  // folded finalize below, so its round trip hides under the row sums (the capped grid leaves only two workgroups per CU).
  const long long napply = ws.n_apply ? ws.n_apply : (int)gridDim.x;
  const long long pstep = napply * g.slots;
  long long pp = (long long)blockIdx.x * g.slots + slot;
  Raw8<T> gor, pvr;
  uint32_t code = 0;
  {
    const size_t e = (size_t)(pp < g.npool ? pp : 0) * g.C + cg * 8;
    gor.load(dout, e); pvr.load(pooled, e);
    if (route) code = route[e >> 3];
  }
  if (pro.rows) {
    __shared__ __attribute__((aligned(8))) double sp[BX_ROWS_TOTAL_LDS(3) / 8];
    __shared__ float cf[3][256];
    bx_rows_total<3>(pro.rows, pro.nrows, g.C, sp);
    if ((int)threadIdx.x < g.C) {
      const int c = threadIdx.x;
      const double st[3] = {sp[c], sp[g.C + c], sp[2 * g.C + c]};
      const TailBwdFin& f = pro.fin;
      cf[0][c] = f.gamma[c] * f.invstd[c];
      cf[1][c] = f.training ? (float)(st[0] / f.count) : 0.f;
      cf[2][c] = f.training ? (float)(st[1] / f.count) : 0.f;
      if (blockIdx.x == 0) tail_bwd_finalize_channel(c, g.C, st, f);       // parameter gradients (+ the coefficients, for inspection)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      mu[j] = mean[c]; is[j] = invstd[c]; a[j] = cf[0][c]; k1[j] = cf[1][c]; k2[j] = cf[2][c];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      mu[j] = mean[c]; is[j] = invstd[c]; a[j] = coef[c]; k1[j] = coef[g.C + c]; k2[j] = coef[2 * g.C + c];
    }
  }
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // (two trips ahead with two register sets: 28.1 -> 30.9 us at stage 1, occupancy 3 -- one trip ahead it stays)
  for (; pp < g.npool; pp += pstep) {
    int ox, oy, b;
    px_decode(pp, g.Wo, g.Ho, ox, oy, b);
    float go[8], pv[8], dp[8];
    gor.get(go); pvr.get(pv);
    const uint32_t code_now = code;
    {                                                   // the next trip's loads (clamped index: the values are unused past the end)
      const long long pn = pp + pstep;
      const size_t e = (size_t)(pn < g.npool ? pn : pp) * g.C + cg * 8;
      gor.load(dout, e); pvr.load(pooled, e);
      if (route) code = route[e >> 3];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dd = go[j];
      if (dropout_p > 0.f) dd *= bx_dropout_scale(sd, salt, (uint64_t)pp * g.C + cg * 8 + j, dropout_p, inv_keep);
      dp[j] = a[j] * (dd - k1[j] - (pv[j] - mu[j]) * is[j] * k2[j]);
    }
    const size_t base = (((size_t)b * g.H + 2 * oy) * g.W + 2 * ox) * g.C + cg * 8;
    const size_t off[4] = {0, (size_t)g.C, (size_t)g.W * g.C, (size_t)g.W * g.C + g.C};
    float v[4][8], o[4][8];
    if (route) {                                        // bxTailDesc.route: the forward wrote where each pooled element's gradient goes
      const float share = pool == BX_POOL_MAX ? 1.f : 0.25f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][j] = (code_now >> (4 * j + q)) & 1u ? share * dp[j] : 0.f;
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) ld8(y3, base + off[q], v[q]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (pool == BX_POOL_MAX) {
        int arg = 0; float m = v[0][j];                       // first maximum in row-major window order (ATen)
#pragma unroll
        for (int q = 1; q < 4; ++q) if (v[q][j] > m) { m = v[q][j]; arg = q; }
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][j] = (q == arg && m > 0.f) ? dp[j] : 0.f;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][j] = v[q][j] > 0.f ? 0.25f * dp[j] : 0.f;
      }
    }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) st8(dz3, base + off[q], o[q]);
    // rows / columns the floor-pool never covers get zero gradient
    const bool last_x = (ox == g.Wo - 1) && (g.W & 1), last_y = (oy == g.Ho - 1) && (g.H & 1);
    if (last_x) { st8(dz3, base + 2 * (size_t)g.C, zero8); st8(dz3, base + (size_t)g.W * g.C + 2 * (size_t)g.C, zero8); }
    if (last_y) { st8(dz3, base + 2 * (size_t)g.W * g.C, zero8); st8(dz3, base + 2 * (size_t)g.W * g.C + g.C, zero8); }
    if (last_x && last_y) st8(dz3, base + 2 * (size_t)g.W * g.C + 2 * (size_t)g.C, zero8);
  }
}

// dXs[p][ci] = sum_c W1x1[c][ci] * dOut[p][c]   (half resolution, fp32).  One (pixel, CIV input channels) unit per
// thread (CIV = 8, or 4 when the map is so small that 8 would leave most CUs without a workgroup); the weight matrix
// streams through LDS in 64-output-channel slabs ([c][Cin_p], zero padded), dOut comes straight from global as 16-byte vectors.
template <typename T, int CIV>
__device__ __forceinline__ void skip_dxs_body(const T* __restrict__ dout, const float* __restrict__ w1x1, int Cin,
                                              float* __restrict__ dxs, T* __restrict__ dx_even, const TailGeom& g, int bid) {
  extern __shared__ __attribute__((aligned(16))) float swt[];   // [64][Cin_p]
  const int ncv = g.Cin_p / CIV;
  const long long n = g.npool * ncv;
  const long long u = (long long)bid * 256 + threadIdx.x;
  const bool live = u < n;
  const unsigned u32 = live ? (unsigned)u : 0u;
  const long long pp = (long long)(u32 / (unsigned)ncv);
  const int cv = (int)(u32 - (unsigned)pp * (unsigned)ncv);
  float acc[CIV];
#pragma unroll
  for (int j = 0; j < CIV; ++j) acc[j] = 0.f;
  for (int cb = 0; cb < g.C; cb += 64) {
    const int nc = g.C - cb < 64 ? g.C - cb : 64;
    float go[8][8];                                       // this pixel's dOut over the slab: all loads issued up front
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (live && q * 8 < nc) ld8(dout, (size_t)pp * g.C + cb + q * 8, go[q]);
    __syncthreads();
    if (Cin == g.Cin_p) {
      // contiguous slab: 16-byte copies, 8 in flight per thread
      const float4* src = reinterpret_cast<const float4*>(w1x1 + (size_t)cb * Cin);
      float4* dst = reinterpret_cast<float4*>(swt);
      const int n4 = nc * g.Cin_p / 4;
      for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; v[u] = src[i < n4 ? i : 0]; }   // (conditional fill would put v[] in scratch)
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; if (i < n4) dst[i] = v[u]; }
      }
    } else {
      const int cinp = g.Cin_p;
      lds_fill<8>(swt, nc * cinp, [&](int i) { const int c = i / cinp, ci = i % cinp; return ci < Cin ? w1x1[(size_t)(cb + c) * Cin + ci] : 0.f; });
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q * 8 >= nc) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gv = go[q][k];
#pragma unroll
          for (int h = 0; h < CIV / 4; ++h) {
            const float4 wv = *reinterpret_cast<const float4*>(swt + (q * 8 + k) * g.Cin_p + cv * CIV + 4 * h);
            acc[4 * h] = fmaf(wv.x, gv, acc[4 * h]); acc[4 * h + 1] = fmaf(wv.y, gv, acc[4 * h + 1]);
            acc[4 * h + 2] = fmaf(wv.z, gv, acc[4 * h + 2]); acc[4 * h + 3] = fmaf(wv.w, gv, acc[4 * h + 3]);
          }
        }
      }
    }
  }
  if (live) {
    if (dx_even) {
      // H and W even: bilinear-to-half is the 2x2 mean, its transpose spreads dXs/4 over the 2x2 footprint
      int ox, oy, b;
      px_decode(pp, g.Wo, g.Ho, ox, oy, b);
      const size_t base = (((size_t)b * g.H + 2 * oy) * g.W + 2 * ox) * g.Cin_p + cv * CIV;
      const size_t foot[4] = {0, (size_t)g.Cin_p, (size_t)g.W * g.Cin_p, (size_t)g.W * g.Cin_p + g.Cin_p};
      if (CIV == 8) {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = 0.25f * acc[j % CIV];
#pragma unroll
        for (int f = 0; f < 4; ++f) st8(dx_even, base + foot[f], q);
      } else {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < CIV; ++j) stf(dx_even, base + foot[f] + j, 0.25f * acc[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CIV; ++j) dxs[(size_t)pp * g.Cin_p + cv * CIV + j] = acc[j];
    }
  }
}

// transposed bilinear: full-resolution gradient of the skip path gathered from dXs
__device__ __forceinline__ int bilinear_T_taps(int i, float scale, int in_size, int out_size, int idx[4], float wt[4]) {
  int n = 0;
  int lo = (int)floorf(((float)i - 0.5f) / scale - 0.5f) - 1;
  for (int o = lo; o <= lo + 4; ++o) {
    if (o < 0 || o >= out_size) continue;
    int i0, i1; float l1;
    bilinear_src(o, scale, in_size, i0, i1, l1);
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i1 == i) w += l1;
    if (w != 0.f && n < 4) { idx[n] = o; wt[n] = w; ++n; }
  }
  return n;
}
template <typename T>
__global__ __launch_bounds__(256) void k_skip_scatter(const float* __restrict__ dxs, T* __restrict__ dx, TailGeom g) {
  const int nci8 = g.Cin_p / 8;
  const long long n = (long long)g.B * g.H * g.W * nci8;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < n; u += (long long)gridDim.x * 256) {
    const unsigned u32 = (unsigned)u, p = u32 / (unsigned)nci8;
    const int c8 = (int)(u32 - p * (unsigned)nci8);
    int xx, yy, b;
    px_decode((long long)p, g.W, g.H, xx, yy, b);
    int iy[4], ix[4]; float wy[4], wx[4];
    const int ny = bilinear_T_taps(yy, g.sy, g.H, g.Ho, iy, wy), nx = bilinear_T_taps(xx, g.sx, g.W, g.Wo, ix, wx);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
      for (int c = 0; c < nx; ++c) {
        const float w = wy[a] * wx[c];
        const float* src = dxs + (((size_t)b * g.Ho + iy[a]) * g.Wo + ix[c]) * g.Cin_p + c8 * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, src[j], acc[j]);
      }
    st8(dx, (size_t)p * g.Cin_p + c8 * 8, acc);
  }
}

// conv1x1 weight gradient: dW[c][ci] = sum_p dOut[p][c] * xs[p][ci]; 16x16 output tile per workgroup,
// pixel chunks over blockIdx.x, partials reduced in fixed order.
template <typename T>
__device__ __forceinline__ void w1x1_grad_body(const T* __restrict__ dout, const T* __restrict__ x, float* __restrict__ partial,
                                               int pix_per_chunk, const TailGeom& g, int bx, int by, int bz) {
  // Operand tiles TRANSPOSED in LDS ([channel][pixel], round 3): a thread reads four pixels of its dOut channel and of its input
  // channel as two 16-byte reads per four FMAs.  The [pixel][channel] form took two 4-byte reads per FMA and the role was bound by
  // LDS instruction issue (stage 1 alone: 28 us for 51 MB).  Pitch 68: 16-byte aligned rows, four banks apart.
  __shared__ __attribute__((aligned(16))) float sdT[16][68], sxsT[16][68];
  // With at most 8 input channels (stage 1) the two halves of a 16-lane row would compute padding: they split the trip's pixels instead
  // (lane bit 3 = pixel half) and the halves are added at the end.
  const bool ksplit = g.Cin_p <= 8;
  const int c = threadIdx.x >> 4, ci = ksplit ? (threadIdx.x & 7) : (threadIdx.x & 15), kh = ksplit ? ((threadIdx.x >> 3) & 1) : 0;
  const int q_lo = ksplit ? 32 * kh : 0, q_n = ksplit ? 32 : 64;
  const int c0 = by * 16, ci0 = bz * 16;
  float acc = 0.f;
  const long long p_begin = (long long)bx * pix_per_chunk;
  long long p_end = p_begin + pix_per_chunk;
  if (p_end > g.npool) p_end = g.npool;
  // threads 0..127 stage 64 pixels x 2 halves of dOut, threads 128..255 the bilinear-sampled input; the next trip's
  // values are fetched into registers while the current trip is multiplied
  const int st = threadIdx.x & 127, px = st >> 1, half = st & 1;
  const bool is_d = threadIdx.x < 128;
  float v[8];
  auto fetch = [&](long long p0) {
    const long long pp = p0 + px;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (pp < p_end) {
      if (is_d) {
        ld8(dout, (size_t)pp * g.C + c0 + half * 8, v);
      } else if (ci0 + half * 8 < g.Cin_p) {
        int ox, oy, b;
        px_decode(pp, g.Wo, g.Ho, ox, oy, b);
        skip_sample(x, g, b, oy, ox, ci0 + half * 8, v);
      }
    }
  };
  if (p_begin < p_end) fetch(p_begin);
  for (long long p0 = p_begin; p0 < p_end; p0 += 64) {
    __syncthreads();
    float (*dst)[68] = is_d ? sdT : sxsT;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[half * 8 + j][px] = v[j];
    __syncthreads();
    if (p0 + 64 < p_end) fetch(p0 + 64);
    for (int q = q_lo; q < q_lo + q_n; q += 16) {           // 4 x (2 reads, 4 FMAs) per step, pixels in ascending order
      float4 dv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { dv[u] = *reinterpret_cast<const float4*>(&sdT[c][q + 4 * u]); xv[u] = *reinterpret_cast<const float4*>(&sxsT[ci][q + 4 * u]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = fmaf(dv[u].x, xv[u].x, acc); acc = fmaf(dv[u].y, xv[u].y, acc); acc = fmaf(dv[u].z, xv[u].z, acc); acc = fmaf(dv[u].w, xv[u].w, acc);
      }
    }
  }
  if (ksplit) acc += __shfl_xor(acc, 8, 64);                // the other pixel half of the same output (lanes l, l ^ 8)
  if (ci0 + ci < g.Cin_p && kh == 0) partial[((size_t)bx * g.C + c0 + c) * g.Cin_p + ci0 + ci] = acc;
}

// Late-stage variant (C and Cin_p multiples of 64): 64 x 64 output tile per workgroup, a thread owns 4 x 4 outputs and one
// trip covers 64 pixels (two 16-byte LDS reads feed 16 FMAs).  The 16 x 16 tiling above re-reads every pixel C*Cin_p/256
// times from L2 (128 x for the last stage: 40 MB for 3 MB of operands); this one 16 x less.
template <typename T>
__device__ __forceinline__ void w1x1_grad_big_body(const T* __restrict__ dout, const T* __restrict__ x, float* __restrict__ partial,
                                                   int pix_per_chunk, const TailGeom& g, int bx, int by, int bz) {
  __shared__ __attribute__((aligned(16))) float sd[64][64], sxs[64][64];
  const int tc = threadIdx.x & 15, tci = threadIdx.x >> 4;
  const int c0 = by * 64, ci0 = bz * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long long p_begin = (long long)bx * pix_per_chunk;
  long long p_end = p_begin + pix_per_chunk;
  if (p_end > g.npool) p_end = g.npool;
  for (long long p0 = p_begin; p0 < p_end; p0 += 64) {
    float vd[2][8], vx[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                             // unit = (pixel, 8-channel group): 512 units per operand, 2 per thread
      const int u = threadIdx.x + 256 * h, px = u >> 3, c8 = u & 7;
      const long long pp = p0 + px;
#pragma unroll
      for (int j = 0; j < 8; ++j) vd[h][j] = vx[h][j] = 0.f;
      if (pp < p_end) {
        ld8(dout, (size_t)pp * g.C + c0 + c8 * 8, vd[h]);
        int ox, oy, b;
        px_decode(pp, g.Wo, g.Ho, ox, oy, b);
        skip_sample(x, g, b, oy, ox, ci0 + c8 * 8, vx[h]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = threadIdx.x + 256 * h, px = u >> 3, c8 = u & 7;
      *reinterpret_cast<float4*>(&sd[px][c8 * 8]) = make_float4(vd[h][0], vd[h][1], vd[h][2], vd[h][3]);
      *reinterpret_cast<float4*>(&sd[px][c8 * 8 + 4]) = make_float4(vd[h][4], vd[h][5], vd[h][6], vd[h][7]);
      *reinterpret_cast<float4*>(&sxs[px][c8 * 8]) = make_float4(vx[h][0], vx[h][1], vx[h][2], vx[h][3]);
      *reinterpret_cast<float4*>(&sxs[px][c8 * 8 + 4]) = make_float4(vx[h][4], vx[h][5], vx[h][6], vx[h][7]);
    }
    __syncthreads();
#pragma unroll 8
    for (int q = 0; q < 64; ++q) {
      const float4 d4 = *reinterpret_cast<const float4*>(&sd[q][4 * tc]);
      const float4 x4 = *reinterpret_cast<const float4*>(&sxs[q][4 * tci]);
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], xv[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<float4*>(partial + ((size_t)bx * g.C + c0 + 4 * tc + i) * g.Cin_p + ci0 + 4 * tci) =
        make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
}

// dXs on the matrix cores (bf16 storage, Cin_p in {16,32,64,128}, C % 32 == 0): the VALU kernel above re-stages the whole
// weight matrix through LDS for 8-32 pixels per workgroup (31 us for the 2048-pixel last stage).  Here D^T = W^T dOut^T:
// A = W^T [ci][c] built from the fp32 weights as a bf16 hi + lo pair (two MFMAs, ~fp32-accurate weights), B = dOut rows read
// straight from global memory (8 consecutive channels of one pixel = one 16-byte load).  A lane ends up with 4 consecutive
// input channels of one pixel: one 8-byte bf16 store per footprint position.  Waves split (channel tiles) x (pixel tiles).
typedef __attribute__((ext_vector_type(8))) short tl_bf16x8;
typedef __attribute__((ext_vector_type(4))) float tl_f32x4;
template <int NTW, int WN, int PTW>
__device__ __forceinline__ void skip_dxs_mfma_body(const bf16_t* __restrict__ dout, const float* __restrict__ w1x1, int Cin,
                                                   float* __restrict__ dxs, bf16_t* __restrict__ dx_even, const TailGeom& g, int bid) {
  constexpr int PW = (4 / WN) * PTW * 16;                      // pixels per workgroup
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gq = lane >> 4, li = lane & 15;
  const int wn_ = wave % WN, wp = wave / WN;
  const long long p0 = (long long)bid * PW + wp * PTW * 16;
  tl_f32x4 acc[PTW][NTW];
#pragma unroll
  for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[pt][nt] = (tl_f32x4){0.f, 0.f, 0.f, 0.f};
  // software pipeline with explicit current / next registers (a [2][..] array indexed by a runtime toggle lives in scratch memory)
  float wv[NTW][8], wn[NTW][8];
  uint4 bv[PTW], bn[PTW];
  auto fetch = [&](int kb, float (&w_)[NTW][8], uint4 (&b_)[PTW]) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int ci = (wn_ * NTW + nt) * 16 + li;
#pragma unroll
      for (int j = 0; j < 8; ++j) w_[nt][j] = ci < Cin ? w1x1[(size_t)(kb + 8 * gq + j) * Cin + ci] : 0.f;
    }
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
      const long long pix = p0 + pt * 16 + li;
      b_[pt] = pix < g.npool ? *reinterpret_cast<const uint4*>(dout + (size_t)pix * g.C + kb + 8 * gq) : make_uint4(0, 0, 0, 0);
    }
  };
  fetch(0, wv, bv);
  for (int kb = 0; kb < g.C; kb += 32) {
    const bool more = kb + 32 < g.C;
    if (more) fetch(kb + 32, wn, bn);
    tl_bf16x8 ahi[NTW], alo[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      uint32_t h[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = wv[nt][2 * j], b = wv[nt][2 * j + 1];
        h[j] = pack2bf(a, b);
        l[j] = pack2bf(a - __uint_as_float(h[j] << 16), b - __uint_as_float(h[j] & 0xffff0000u));
      }
      ahi[nt] = __builtin_bit_cast(tl_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
      alo[nt] = __builtin_bit_cast(tl_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    }
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
      const tl_bf16x8 bfr = __builtin_bit_cast(tl_bf16x8, bv[pt]);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[nt], bfr, acc[pt][nt], 0, 0, 0);
        acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[nt], bfr, acc[pt][nt], 0, 0, 0);
      }
    }
    if (more) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[nt][j] = wn[nt][j];
#pragma unroll
      for (int pt = 0; pt < PTW; ++pt) bv[pt] = bn[pt];
    }
  }
#pragma unroll
  for (int pt = 0; pt < PTW; ++pt) {
    const long long pix = p0 + pt * 16 + li;
    if (pix >= g.npool) continue;
    if (dx_even) {
      int ox, oy, b;
      px_decode(pix, g.Wo, g.Ho, ox, oy, b);
      const size_t base = (((size_t)b * g.H + 2 * oy) * g.W + 2 * ox) * g.Cin_p;
      const size_t foot[4] = {0, (size_t)g.Cin_p, (size_t)g.W * g.Cin_p, (size_t)g.W * g.Cin_p + g.Cin_p};
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int ci = (wn_ * NTW + nt) * 16 + 4 * gq;
        const uint2 v = make_uint2(pack2bf(0.25f * acc[pt][nt][0], 0.25f * acc[pt][nt][1]), pack2bf(0.25f * acc[pt][nt][2], 0.25f * acc[pt][nt][3]));
#pragma unroll
        for (int f = 0; f < 4; ++f) *reinterpret_cast<uint2*>(dx_even + base + foot[f] + ci) = v;
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int ci = (wn_ * NTW + nt) * 16 + 4 * gq;
        *reinterpret_cast<float4*>(dxs + (size_t)pix * g.Cin_p + ci) = make_float4(acc[pt][nt][0], acc[pt][nt][1], acc[pt][nt][2], acc[pt][nt][3]);
      }
    }
  }
}
template <typename T, int MF> struct SkipMfma {
  static __device__ __forceinline__ void run(const T*, const float*, int, float*, T*, const TailGeom&, int) {}
};
template <> struct SkipMfma<bf16_t, 16>  { static __device__ __forceinline__ void run(const bf16_t* d, const float* w, int Cin, float* x, bf16_t* e, const TailGeom& g, int bid) { skip_dxs_mfma_body<1, 1, 2>(d, w, Cin, x, e, g, bid); } };
template <> struct SkipMfma<bf16_t, 32>  { static __device__ __forceinline__ void run(const bf16_t* d, const float* w, int Cin, float* x, bf16_t* e, const TailGeom& g, int bid) { skip_dxs_mfma_body<1, 2, 2>(d, w, Cin, x, e, g, bid); } };
template <> struct SkipMfma<bf16_t, 64>  { static __device__ __forceinline__ void run(const bf16_t* d, const float* w, int Cin, float* x, bf16_t* e, const TailGeom& g, int bid) { skip_dxs_mfma_body<1, 4, 4>(d, w, Cin, x, e, g, bid); } };
template <> struct SkipMfma<bf16_t, 128> { static __device__ __forceinline__ void run(const bf16_t* d, const float* w, int Cin, float* x, bf16_t* e, const TailGeom& g, int bid) { skip_dxs_mfma_body<2, 4, 4>(d, w, Cin, x, e, g, bid); } };
static inline int skip_mfma_pixels(int cin_p) { return cin_p == 16 ? 128 : 64; }      // pixels per workgroup of the configurations above

// ---- merged launches of the block-tail backward ---------------------------------------------------------------------
// A kernel boundary costs ~5 us here (drain, L2 write-back, next dispatch) and the late stages' tail kernels are a few
// microseconds of latency-bound work each, so independent pieces share a launch, the role chosen by workgroup index:
//   front: [BN/dropout reduction | 1x1-skip weight-gradient partials | 1x1-skip input gradient]   (all read only dOut)
//   mid:   [BN-backward finalize | fixed-order sum of the weight-gradient partials (+ un-padding of the first stage)]
// followed by the apply kernel.  6-7 launches per block become 3.
template <typename T>
struct TailFrontArgs {
  const T *dout, *pooled, *x;
  const float *mean, *invstd, *w1x1;
  const uint64_t* seed;
  float *partials, *wpart, *dxs;
  T* dx_even;
  float dropout_p;
  uint32_t salt;
  int n_red, n_w, wg_x, wg_y, ppc, Cin;
  BxStatTree tree;          // tree.cnt != NULL: the reduction role finalizes in-launch
  TailBwdFin fin;
};
template <typename T, int CIV, int MF>      // MF = Cin_p of the MFMA input-gradient role, 0 = VALU role with CIV channels per thread
__global__ __launch_bounds__(256) void k_tail_bwd_front(TailFrontArgs<T> a, TailGeom g) {
  int bid = blockIdx.x;
  if (bid < a.n_red) { tail_bwd_reduce_body<T, MF == 0 ? 4 : 2>(a.dout, a.pooled, a.mean, a.invstd, a.seed, a.dropout_p, a.salt, a.partials, g, bid, a.n_red, a.tree, a.fin); return; }
  bid -= a.n_red;
  if (bid < a.n_w) {
    const int bx = bid % a.wg_x, r = bid / a.wg_x;
    if (MF >= 64) w1x1_grad_big_body<T>(a.dout, a.x, a.wpart, a.ppc, g, bx, r % a.wg_y, r / a.wg_y);   // MF >= 64 <=> late stage (launcher)
    else w1x1_grad_body<T>(a.dout, a.x, a.wpart, a.ppc, g, bx, r % a.wg_y, r / a.wg_y);
    return;
  }
  if (MF) SkipMfma<T, MF>::run(a.dout, a.w1x1, a.Cin, a.dxs, a.dx_even, g, bid - a.n_w);
  else skip_dxs_body<T, CIV>(a.dout, a.w1x1, a.Cin, a.dxs, a.dx_even, g, bid - a.n_w);
}
// mid: 1024 threads.  Workgroups [0, n_fin) finalize 16 channels each; the others sum weight-gradient partials, four
// 256-thread groups per workgroup, each group as in k_sum_partials; (c, ci) of the padded layout goes to dw[c][ci < Cin].
__global__ __launch_bounds__(1024) void k_tail_bwd_mid(const float* __restrict__ partials, int nblk, double count, int C, int training,
                                                       const float* __restrict__ gamma, const float* __restrict__ invstd, float* __restrict__ coef,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db1x1, int n_fin,
                                                       const float* __restrict__ wpart, float* __restrict__ dw, int nchunk, int Cin, int Cin_p, int S) {
  if ((int)blockIdx.x < n_fin) {
    tail_bwd_finalize_body(partials, nblk, count, C, training, gamma, invstd, coef, dgamma, dbeta, db1x1, blockIdx.x);
    return;
  }
  __shared__ float sm[1024];
  w1x1_sum_group(wpart, dw, nchunk, C, Cin, Cin_p, S, ((int)blockIdx.x - n_fin) * 4 + (threadIdx.x >> 8), threadIdx.x & 255, sm + (threadIdx.x >> 8) * 256);
}

extern "C" int bx_block_tail_bwd(const bxTailDesc* d, const void* dout, const void* y3, const void* x, const void* pooled,
                                 const float* w1x1, int Cin, const float* bn_weight, const float* save_mean,
                                 const float* save_invstd, const uint64_t* seed, void* dz3, void* dx_skip,
                                 float* d_bn_weight, float* d_bn_bias, float* d_w1x1, float* d_b1x1,
                                 void* workspace, size_t workspace_bytes, bxStream stream) {
  BX_REQUIRE(d && dout && (y3 || d->route) && x && pooled && w1x1 && bn_weight && save_mean && save_invstd && dz3, "bx_block_tail_bwd: null pointer");
  BX_DTYPE_OK(d->dtype);
  TailGeom g;
  const int ge = make_geom(d, &g);
  BX_REQUIRE(ge == 0, "bx_block_tail_bwd: unsupported geometry (code %d)", ge);
  BX_REQUIRE(Cin > 0 && Cin <= d->Cin_p, "bx_block_tail_bwd: Cin=%d exceeds Cin_p=%d", Cin, d->Cin_p);
  BX_REQUIRE(d->C % 16 == 0, "bx_block_tail_bwd: C must be a multiple of 16");
  const size_t need = bx_block_tail_workspace(d);
  if (!workspace || workspace_bytes < need) BX_FAIL(BX_EWORKSPACE, "bx_block_tail_bwd: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = tail_blocks(g);
  const float p = d->training ? d->dropout_p : 0.f;
  float* partials = (float*)workspace;
  float* coef = partials + (size_t)TAIL_MAX_BLOCKS * 3 * g.C;
  float* dxs = coef + 3 * g.C;
  float* wpart = (float*)((char*)dxs + bx_align_up((size_t)g.npool * g.Cin_p * sizeof(float), 256));

  // ---- front: reduction | w1x1 weight-gradient partials | skip input gradient
  const bool even = (g.H % 2 == 0) && (g.W % 2 == 0);
  static const int no_mfma = getenv("BX_SKIP_NO_MFMA") ? atoi(getenv("BX_SKIP_NO_MFMA")) : 0;
  const int mf = (dx_skip && !no_mfma && d->dtype == BX_BF16 && g.C % 32 == 0 && (g.Cin_p == 16 || g.Cin_p == 32 || g.Cin_p == 64 || g.Cin_p == 128))
                 ? g.Cin_p : 0;
  const bool wbig = mf >= 64 && g.C % 64 == 0;                 // 64 x 64 weight-gradient tiles (kernel: MF >= 64)
  BX_REQUIRE(mf < 64 || wbig, "bx_block_tail_bwd: C must be a multiple of 64 when Cin_p >= 64");
  int nchunk = 0, ppc = 0, wg_y = wbig ? g.C / 64 : g.C / 16, wg_z = wbig ? g.Cin_p / 64 : (g.Cin_p + 15) / 16;
  if (d_w1x1) {
    // up to ~4096 workgroups in total; partial buffer = nchunk * C * Cin_p floats <= 2048*256 + 64*C*Cin_p (workspace formula)
    const int otiles = wg_y * wg_z;
    nchunk = (int)((g.npool + 63) / 64);
    // one output tile (first stage): 1 024 chunks of eight 64-pixel trips; more, shorter chunks do not help (BX_TAIL_W1_CAP sweep:
    // 1024 1.472, 2048 1.474, 4096 1.479 ms/step) -- the role is bound by its per-trip work, not by the chain of trips.  Role times at
    // stage 1, measured by launching the front kernel with one role at a time: BN reduction 16.9 us, this role 28.4 -> 23.5 us with
    // the transposed tiles, together 36.8 -> 33.8 us.
    static const int cap1 = getenv("BX_TAIL_W1_CAP") ? atoi(getenv("BX_TAIL_W1_CAP")) : 1024;
    int cap = otiles == 1 ? cap1 : 4096 / otiles;
    if (cap < 64) cap = 64;
    while ((size_t)cap * g.C * g.Cin_p > (size_t)2048 * 256 + (size_t)64 * g.C * g.Cin_p) cap /= 2;
    if (nchunk > cap) nchunk = cap;
    ppc = (int)((g.npool + nchunk - 1) / nchunk);
    ppc = (ppc + 63) / 64 * 64;
    nchunk = (int)((g.npool + ppc - 1) / ppc);
  }
  const bool narrow = g.npool * (g.Cin_p / 8) < 256 * 256;          // fewer than one workgroup per CU at 8 channels per thread
  const int n_dxs = !dx_skip ? 0 : mf ? bx_ceil_div(g.npool, skip_mfma_pixels(mf)) : bx_ceil_div(g.npool * (g.Cin_p / (narrow ? 4 : 8)), 256);
  const int n_w = nchunk * wg_y * wg_z;
  const size_t front_lds = (dx_skip && !mf) ? (size_t)64 * g.Cin_p * sizeof(float) : 0;
  const bool in_launch = d->sync && 256 % g.C == 0 && nblk <= bx_tree_max_rows();
  // folded finalize: the reduction role writes few rows (rows x C <= BX_TAIL_FOLD_RC), every apply workgroup sums them itself
  const bool fold = !in_launch && (tail_fold() & 1) && tail_fold_ok(g);
  const int nred = fold && nblk > tail_fold_rows(g.C) ? tail_fold_rows(g.C) : nblk;
  BxStatTree tree = {};
  TailBwdFin fin = {};
  if (in_launch) {
    tree.rows = partials; tree.cnt = d->sync + BX_TAIL_SYNC_BWD; tree.nrows = nblk; tree.Cw = tree.ld = g.C;
    bx_stat_tree_shape(nblk, &tree.G, &tree.ngroups);
    tree.mid = (double*)((char*)workspace + bx_align_up(bx_block_tail_workspace(d) - (size_t)TAIL_MAX_GROUPS * 3 * g.C * sizeof(double), 8));
    BX_REQUIRE(tree.ngroups + 1 <= BX_TAIL_SYNC_WORDS - BX_TAIL_SYNC_BWD, "bx_block_tail_bwd: %d reduction groups exceed the sync block", tree.ngroups);
    fin = TailBwdFin{bn_weight, save_invstd, coef, d_bn_weight, d_bn_bias, d_b1x1, (double)g.npool, d->training};
  }
  BX_DISPATCH_DTYPE(d->dtype, T, {
    TailFrontArgs<T> a;
    a.dout = (const T*)dout; a.pooled = (const T*)pooled; a.x = (const T*)x; a.mean = save_mean; a.invstd = save_invstd; a.w1x1 = w1x1;
    a.seed = seed; a.partials = partials; a.wpart = wpart; a.dxs = dxs; a.dx_even = even ? (T*)dx_skip : (T*)nullptr;
    a.dropout_p = p; a.salt = d->salt; a.n_red = nred; a.n_w = n_w;
 a.wg_x = nchunk > 0 ? nchunk : 1; a.wg_y = wg_y; a.ppc = ppc; a.Cin = Cin;
    a.tree = tree; a.fin = fin;
    const dim3 grid(nred + n_w + n_dxs);
    if (mf == 16) hipLaunchKernelGGL((k_tail_bwd_front<T, 8, 16>), grid, dim3(256), front_lds, s, a, g);
    else if (mf == 32) hipLaunchKernelGGL((k_tail_bwd_front<T, 8, 32>), grid, dim3(256), front_lds, s, a, g);
    else if (mf == 64) hipLaunchKernelGGL((k_tail_bwd_front<T, 8, 64>), grid, dim3(256), front_lds, s, a, g);
    else if (mf == 128) hipLaunchKernelGGL((k_tail_bwd_front<T, 8, 128>), grid, dim3(256), front_lds, s, a, g);
    else if (narrow) hipLaunchKernelGGL((k_tail_bwd_front<T, 4, 0>), grid, dim3(256), front_lds, s, a, g);
    else hipLaunchKernelGGL((k_tail_bwd_front<T, 8, 0>), grid, dim3(256), front_lds, s, a, g);
  });
  BX_CHECK_LAUNCH("bx_block_tail_bwd(front)");
  // ---- mid: finalize | weight-gradient sum.  In-launch form: the finalize was done by the front kernel's last reduction workgroup
  // and the sum rides as extra workgroups of the apply kernel (it only feeds the optimizer).
  const int n_sum_elems = g.C * g.Cin_p;
  const int S = d_w1x1 ? bx_partial_slices(n_sum_elems, nchunk) : 4;
  TailWsum wsum = {};
  TailBwdPro pro = {};
  int n_sum256 = 0;
  const int napply = fold && tail_blocks_all(g) > tail_fold_grid() ? tail_fold_grid() : tail_blocks_all(g);
  if (fold) {
    pro = TailBwdPro{partials, nred, TailBwdFin{bn_weight, save_invstd, coef, d_bn_weight, d_bn_bias, d_b1x1, (double)g.npool, d->training}};
    if (d_w1x1) {
      n_sum256 = bx_ceil_div(n_sum_elems, 256 / S);
      wsum = TailWsum{wpart, d_w1x1, nchunk, Cin, S, napply};
    }
  } else if (!in_launch) {
    const int n_fin = bx_finalize_grid(g.C);
    const int n_sum = d_w1x1 ? bx_ceil_div(n_sum_elems, 4 * (256 / S)) : 0;
    hipLaunchKernelGGL(k_tail_bwd_mid, dim3(n_fin + n_sum), dim3(1024), 0, s, partials, nblk, (double)g.npool, g.C, d->training, bn_weight,
                       save_invstd, coef, d_bn_weight, d_bn_bias, d_b1x1, n_fin, wpart, d_w1x1, nchunk, Cin, g.Cin_p, S);
    BX_CHECK_LAUNCH("bx_block_tail_bwd(mid)");
  } else if (d_w1x1) {
    n_sum256 = bx_ceil_div(n_sum_elems, 256 / S);
    wsum = TailWsum{wpart, d_w1x1, nchunk, Cin, S, napply};
  }
  BX_DISPATCH_DTYPE(d->dtype, T,
    hipLaunchKernelGGL((k_tail_bwd_apply<T>), dim3(napply + n_sum256), dim3(256), 0, s, (const T*)dout, (const T*)pooled, (const T*)y3,
                       save_mean, save_invstd, coef, seed, p, d->salt, d->pool, (T*)dz3, g, wsum, pro, (const uint32_t*)d->route));
  BX_CHECK_LAUNCH("bx_block_tail_bwd(apply)");
  if (dx_skip) {
    if (!even) {
      const long long n2 = (long long)g.B * g.H * g.W * (g.Cin_p / 8);
      BX_DISPATCH_DTYPE(d->dtype, T,
        hipLaunchKernelGGL((k_skip_scatter<T>), dim3(bx_ceil_div(n2, 256) > 4096 ? 4096 : bx_ceil_div(n2, 256)), dim3(256), 0, s,
                           dxs, (T*)dx_skip, g));
      BX_CHECK_LAUNCH("bx_block_tail_bwd(scatter)");
    }
  }
  return BX_OK;
}
