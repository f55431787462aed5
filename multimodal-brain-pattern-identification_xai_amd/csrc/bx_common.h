// Shared device/host helpers for libbrainxai (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/brainxai.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

// ---- error plumbing ---------------------------------------------------------------------------
void bx_set_error(const char* fmt, ...);
#define BX_FAIL(code, ...) do { bx_set_error(__VA_ARGS__); return (code); } while (0)
#define BX_REQUIRE(cond, ...) do { if (!(cond)) BX_FAIL(BX_EINVAL, __VA_ARGS__); } while (0)
#define BX_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
  if (e_ != hipSuccess) BX_FAIL(BX_EHIP, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)
#define BX_DTYPE_OK(dt) do { if ((dt) != BX_F32 && (dt) != BX_BF16) BX_FAIL(BX_EDTYPE, "unsupported dtype %d", (dt)); } while (0)

// measurement hook (bx_profile_next_conv3): event pair the next pooled-epilogue convolution launch of this thread records around
// its kernel, so that bench.py can time the FUSED launch the training step really runs (HIP events cannot reach inside a C call)
extern thread_local hipEvent_t g_bx_prof_ev[2];

static inline int bx_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t bx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline size_t bx_esize(int dtype) { return dtype == BX_BF16 ? 2 : 4; }

// ---- scalar conversion ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even (NaN stays NaN): gfx950 has v_cvt_pk_bf16_f32, two conversions in one instruction
// (the bit-twiddled form costs ~6 VALU instructions per value and made several epilogues VALU-bound)
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const bx_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bx_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const bf16_t* p, size_t i) { return bf2f(p[i]); }
__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(bf16_t* p, size_t i, float v) { p[i] = f2bf(v); }

// value as it will read back after being stored as T (identity for fp32, round-to-nearest-even for bf16)
__device__ __forceinline__ float round_as(const float*, float v) { return v; }
__device__ __forceinline__ float round_as(const bf16_t*, float v) { return bf2f(f2bf(v)); }

// ---- 8-channel vector access (16 B for bf16, 2 x 16 B for fp32); p + i must be 8-element aligned ----
__device__ __forceinline__ void ld8(const float* p, size_t i, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p + i);
  const float4 b = *reinterpret_cast<const float4*>(p + i + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, size_t i, float v[8]) {
  const uint4 r = *reinterpret_cast<const uint4*>(p + i);
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(w[k] << 16);
    v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
}
// 4-element vector load (p + i must be 4-element aligned)
__device__ __forceinline__ void ld4(const float* p, size_t i, float v[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p + i);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, size_t i, float v[4]) {
  const uint2 r = *reinterpret_cast<const uint2*>(p + i);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, size_t i, const float v[8]) {
  *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, size_t i, const float v[8]) {
  uint32_t w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = pack2bf(v[2 * k], v[2 * k + 1]);
  *reinterpret_cast<uint4*>(p + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- counter-based dropout mask: keep iff hash(seed, salt, idx) maps above p --------------------
// 32-bit mixing (murmur3 finaliser over a Weyl sequence of the element index keyed by seed and salt): ~10 VALU
// instructions per element.  The previous 64-bit splitmix hash cost ~45 (three 64-bit multiplies) and made the pooled-map
// kernels of the block tail VALU-bound; mask quality for dropout is the same.
__device__ __forceinline__ uint32_t bx_hash(uint64_t seed, uint32_t salt, uint64_t idx) {
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= ((uint32_t)(seed >> 32) + salt * 0x85EBCA77u) ^ ((uint32_t)(idx >> 32) * 0x27D4EB2Fu);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// returns the multiplier applied to an element: 0 or 1/(1-p)
__device__ __forceinline__ float bx_dropout_scale(uint64_t seed, uint32_t salt, uint64_t idx, float p, float inv_keep) {
  const float u = (float)(bx_hash(seed, salt, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.0f;
}

// ---- guarded loads through a raw buffer resource -----------------------------------------------------------------------
// `ok ? p[i] : 0.f` compiles to an exec-mask branch around the load, usually with its own s_waitcnt: a batch of such loads
// becomes a chain of serial memory round trips (tools/isa_wait_audit.py).  A raw buffer load whose offset is past the end of
// the resource returns zero instead, needs no branch and pipelines like any other load.  The resource covers `bytes` (< 4 GiB).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bx_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bx_ldf_or0(__amdgpu_buffer_rsrc_t r, int elem, bool ok) {     // fp32 element `elem`, or 0 when !ok
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? (uint32_t)elem * 4u : 0x80000000u, 0, 0));
}

// ---- wave / block reductions (wave = 64 lanes) ----------------------------------------------------
// Every lane gets the result.  Inside a row of 16 lanes the exchange is a DPP modifier on the add itself (quad_perm [1,0,3,2] and
// [2,3,0,1], row_ror:4, row_ror:8); the four row totals meet through v_readlane.  The __shfl_xor butterfly this replaces compiles
// to ds_bpermute_b32 -- an LDS-crossbar round trip per step, six dependent ones per value: a kernel that ends in 65 wave sums
// (k_eegc_stats) spent 16 us in them.  Fixed order, so results are reproducible (they differ from the butterfly's in the last bit).
template <int CTRL> __device__ __forceinline__ float bx_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float bx_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += bx_dpp<0xB1>(v); v += bx_dpp<0x4E>(v); v += bx_dpp<0x124>(v); v += bx_dpp<0x128>(v);
  return (bx_lane(v, 0) + bx_lane(v, 16)) + (bx_lane(v, 32) + bx_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, bx_dpp<0xB1>(v)); v = fmaxf(v, bx_dpp<0x4E>(v)); v = fmaxf(v, bx_dpp<0x124>(v)); v = fmaxf(v, bx_dpp<0x128>(v));
  return fmaxf(fmaxf(bx_lane(v, 0), bx_lane(v, 16)), fmaxf(bx_lane(v, 32), bx_lane(v, 48)));
}

// LogSoftmax backward for one row, dz_n = dy_n - softmax_n * sum_j dy_j, evaluated WITHOUT its cancellation:
//     dz_n = dy_n * (sum_{j != n} p_j)  -  p_n * (sum_{j != n} dy_j),      p = exp(logp).
// The textbook form computes (1 - p_n) by subtraction; with a one-hot seed on a confident sample (saliency / Grad-CAM of the
// arg-max class: p_n -> 1) that turns an absolute 1e-6 error of the log-prob into a relative 1e-6 / (1 - p_n) error of every
// gradient behind it (measured: 1.1e-3 on the saliency maps at fp32, against 4e-5 for the cancellation-free form).
// Plain pointers (global or LDS), runtime N, O(N^2) for the N <= 32 classes the heads support (N = 6 in the reference).
__device__ __forceinline__ void bx_lsm_bwd(const float* dy, const float* logp, int N, float* dz) {
  for (int n = 0; n < N; ++n) {
    float ps = 0.f, ds = 0.f;
    for (int j = 0; j < N; ++j)
      if (j != n) { ps += expf(logp[j]); ds += dy[j]; }
    dz[n] = dy[n] * ps - expf(logp[n]) * ds;
  }
}

// Cooperative LDS fill: dst[i] = load(i) for i < n.  U loads per thread are issued before the first LDS store, so a fill
// costs ceil(n / (U * blockDim)) memory round trips instead of one per element (a plain `for (i...) dst[i] = src[..]`
// loop is compiled into load -> wait -> store per trip).
template <int U, typename LoadF>
__device__ __forceinline__ void lds_fill(float* dst, int n, LoadF load) {
  const int step = (int)blockDim.x;
  for (int i0 = (int)threadIdx.x; i0 < n; i0 += step * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * step; v[u] = load(i < n ? i : n - 1); }   // clamped, unconditional: `i < n ? load(i) : 0`
                                                                                                  // compiles to a branch (and often a full wait) per load
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * step; if (i < n) dst[i] = v[u]; }
  }
}

// PyTorch's area_pixel_compute_source_index for bilinear, align_corners=False.
__device__ __forceinline__ void bilinear_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// Sum NV per-workgroup partial vectors in double, fixed order: partial k, vector v, channel c lives at
// partials[(k*NV + v)*C + c].  The calling workgroup owns channels c0 .. c0+CG-1 (CG divides blockDim): thread =
// (slice, channel); slices are combined through LDS in slice order, so the result is deterministic.  Returns the totals
// in out[] for threads < CG (channel c0 + threadIdx.x).  One workgroup per 16 channels keeps a thread's chain of partial
// loads short (a single 1024-thread workgroup over 256 channels walked 128 partials per thread).
template <int NV>
__device__ __forceinline__ void sum_partials_256(const float* __restrict__ partials, int nblk, int C, int c0, int CG, double out[NV]) {
  __shared__ double sp[NV][1024];
  const int nslice = (int)blockDim.x / CG;
  const int cc = threadIdx.x % CG, slice = threadIdx.x / CG;
  const int c = c0 + cc;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  if (slice < nslice && c < C) {
    int k = slice;
    for (; k + 3 * nslice < nblk; k += 4 * nslice) {        // 4*NV independent loads in flight; adds stay in k order
      float t[4][NV];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) t[u][v] = partials[((size_t)(k + u * nslice) * NV + v) * C + c];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += (double)t[u][v];
    }
    for (; k < nblk; k += nslice)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += (double)partials[((size_t)k * NV + v) * C + c];
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) sp[v][threadIdx.x] = acc[v];
  __syncthreads();
  if ((int)threadIdx.x < CG) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      double s = 0.0;
      for (int sl = 0; sl < nslice; ++sl) s += sp[v][sl * CG + threadIdx.x];
      out[v] = s;
    }
  }
}
// ---- in-launch reduction tree: the LAST workgroup to arrive finishes the sum -------------------------------------------
// A statistics pass used to end in its own finalize launch (k_bn_finalize, k_tail_bwd_mid: ~6 us each, a dozen per training
// step, for a few KB of arithmetic).  Here every workgroup of the producing kernel publishes its row of per-channel partial
// sums, draws a ticket, and the workgroup that draws the last ticket of its GROUP of G rows sums that group in double; the
// last group to finish sums the groups and returns true: its threads < Cw hold the channel totals and run the finalize.
// Sums are taken in row order whatever the arrival order, so results do not depend on scheduling.
//
// Cross-XCD visibility (gfx950: eight private L2s): rows are written with agent-scope (sc1, write-through) stores, every
// storing wave drains them (s_waitcnt vmcnt(0)) before the workgroup barrier, one lane then adds to the agent-scope counter;
// the reducer executes ONE agent-scope acquire after its ticket and reads the rows with plain vector loads
// (cdna_hip_programming.md, guideline 16 recipe R1 in its counter form).  Counters: ngroups + 1 words per domain, zero
// before the first launch; the reducers put them back to zero, so graph replays and later launches need no memset.
struct BxStatTree {
  float* rows;          // [nrows][NV][ld]: a domain (workgroups that share channels c0 .. c0+Cw-1) writes columns c0 .. c0+Cw-1 of every row
  double* mid;          // [ngroups][NV][ld]
  unsigned* cnt;        // [ndom][ngroups + 1]; NULL = rows only (a separate launch sums them)
  int nrows, G, ngroups, Cw, ld;
};
static inline void bx_stat_tree_shape(int nrows, int* G, int* ngroups) {
  int g = nrows <= 96 ? nrows : 64;
  while ((nrows + g - 1) / g > 128) g *= 2;
  *G = g < 1 ? 1 : g; *ngroups = nrows > 0 ? (nrows + *G - 1) / *G : 1;
}
#define BX_STAT_TREE_LDS(NV) ((NV) * 256 * 8 + 16)
__device__ __forceinline__ void bx_store_agent(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bx_store_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one level of the tree: publish done by the caller; returns (workgroup-uniform) whether this workgroup drew the last ticket
__device__ __forceinline__ bool bx_last_ticket(unsigned* cnt, unsigned expected, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // EVERY storing wave drains its sc1 stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == expected - 1u;
    if (last) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}
// fixed-order column sums of rows [r0, r1) of src[(r*NV + k)*ld + c] (Src = float or double) -> tot[k] in threads c < Cw
template <int NV, typename Src>
__device__ __forceinline__ void bx_sum_rows(const Src* src, int r0, int r1, int Cw, int ld, double (&tot)[NV], double* sp) {
  const int S = 256 / Cw, c = threadIdx.x % Cw, sl = threadIdx.x / Cw;
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.0;
  int r = r0 + sl;
  for (; r + 3 * S < r1; r += 4 * S) {             // 4*NV independent loads in flight; adds stay in row order
    Src t[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < NV; ++k) t[u][k] = src[((size_t)(r + u * S) * NV + k) * ld + c];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < NV; ++k) acc[k] += (double)t[u][k];
  }
  for (; r < r1; r += S)
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] += (double)src[((size_t)r * NV + k) * ld + c];
  __syncthreads();                                  // sp may still be read by a previous level
#pragma unroll
  for (int k = 0; k < NV; ++k) sp[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  if ((int)threadIdx.x < Cw) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double s = 0.0;
      for (int q = 0; q < S; ++q) s += sp[k * 256 + q * Cw + threadIdx.x];
      tot[k] = s;
    }
  }
}
// Called by all 256 threads of a workgroup; threads c < Cw pass the workgroup's partial sums val[k] of channel c0 + c.
// lds: BX_STAT_TREE_LDS(NV) bytes, 8-byte aligned, not in use by any wave of the workgroup.
template <int NV>
__device__ __forceinline__ bool bx_stat_tree_arrive(const BxStatTree& t, int dom, int c0, int row, const float (&val)[NV], double (&tot)[NV], char* lds) {
  double* sp = reinterpret_cast<double*>(lds);
  unsigned* flag = reinterpret_cast<unsigned*>(lds + NV * 256 * 8);
  float* rows = t.rows + c0;
  if (!t.cnt) {                                       // rows only: plain stores, the kernel boundary publishes them
    if ((int)threadIdx.x < t.Cw) {
#pragma unroll
      for (int k = 0; k < NV; ++k) rows[((size_t)row * NV + k) * t.ld + threadIdx.x] = val[k];
    }
    return false;
  }
  unsigned* cnt = t.cnt + (size_t)dom * (t.ngroups + 1);
  if ((int)threadIdx.x < t.Cw) {
#pragma unroll
    for (int k = 0; k < NV; ++k) bx_store_agent(rows + ((size_t)row * NV + k) * t.ld + threadIdx.x, val[k]);
  }
  const int grp = row / t.G, r0 = grp * t.G, r1 = r0 + t.G < t.nrows ? r0 + t.G : t.nrows;
  if (!bx_last_ticket(cnt + grp, (unsigned)(r1 - r0), flag)) return false;
  bx_sum_rows<NV, float>(rows, r0, r1, t.Cw, t.ld, tot, sp);
  if (t.ngroups == 1) return true;
  double* mid = t.mid + c0;
  if ((int)threadIdx.x < t.Cw) {
#pragma unroll
    for (int k = 0; k < NV; ++k) bx_store_agent(mid + ((size_t)grp * NV + k) * t.ld + threadIdx.x, tot[k]);
  }
  if (!bx_last_ticket(cnt + t.ngroups, (unsigned)t.ngroups, flag)) return false;
  bx_sum_rows<NV, double>(mid, 0, t.ngroups, t.Cw, t.ld, tot, sp);
  return true;
}

// ---- consumer-side finish of a statistics pass (round 3) -------------------------------------------------------------------
// The in-launch tree above puts the cross-workgroup sum at the END of the producer (tickets: every workgroup waits for a write-through
// store and an atomic round trip before it retires -- measured slower than the finalize launch it removes).  This form puts it at the
// START of the consumer instead: the producer writes its rows with plain stores (the kernel boundary publishes them), and EVERY
// workgroup of the consuming kernel sums the rows itself, in one fixed order, while its first data loads are in flight -- no atomics,
// no finalize launch, the same totals in every workgroup.  It pays when rows x C is small (<= 8192: at most 8 NV 16-byte loads per
// thread, rows x NV x C x 4 bytes of L2 reads per consumer workgroup); the launchers choose the producer's row count accordingly.
// rows[r][k][C] floats; all 256 threads call; C % 4 == 0 and (C / 4) divides 256.  On return (after its closing barrier) the totals
// are sp[k * C + c] as doubles.  sp: BX_ROWS_TOTAL_LDS(NV) bytes of LDS, 8-byte aligned.
#define BX_ROWS_TOTAL_LDS(NV) ((NV) * 1024 * 8)
template <int NV>
__device__ __forceinline__ void bx_rows_total(const float* __restrict__ rows, int nrows, int C, double* sp) {
  const int q = C >> 2, nsl = 256 / q, c4 = threadIdx.x % q, sl = threadIdx.x / q;
  double acc[NV][4];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
  for (int r0 = sl; r0 < nrows; r0 += 4 * nsl) {            // 4 NV 16-byte loads in flight; adds stay in row order
    float4 t[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nsl < nrows ? r0 + u * nsl : r0;
#pragma unroll
      for (int k = 0; k < NV; ++k) t[u][k] = *reinterpret_cast<const float4*>(rows + ((size_t)r * NV + k) * C + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r0 + u * nsl < nrows) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { acc[k][0] += (double)t[u][k].x; acc[k][1] += (double)t[u][k].y; acc[k][2] += (double)t[u][k].z; acc[k][3] += (double)t[u][k].w; }
      }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) sp[((size_t)sl * NV + k) * C + c4 * 4 + i] = acc[k][i];
  __syncthreads();
  for (int idx = threadIdx.x; idx < NV * C; idx += 256) {   // slice 0's entry idx is read by this thread only: the total goes there
    double s = sp[idx];
    for (int j = 1; j < nsl; ++j) s += sp[(size_t)j * NV * C + idx];
    sp[idx] = s;
  }
  __syncthreads();
}
static inline bool bx_rows_total_ok(int C) { return C % 4 == 0 && C >= 4 && C <= 1024 && 256 % (C / 4) == 0; }

static inline int bx_finalize_cg(int C) { return C < 16 ? C : 16; }      // channels per finalize workgroup
static inline int bx_finalize_grid(int C) { const int cg = bx_finalize_cg(C); return (C + cg - 1) / cg; }

// (sum, sum of squares) of one channel -> batch mean, 1/sqrt(var + eps) and the unbiased variance of the running-stat update
__device__ __forceinline__ void bx_bn_batch_stats(double sum, double sumsq, double count, float eps, float& mean, float& invstd, double& unbiased) {
  const double m = sum / count;
  double var = sumsq / count - m * m;
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
}
// batch statistics of one channel -> (scale, shift), saved statistics, running-stat update (unbiased variance, momentum)
__device__ __forceinline__ void bx_bn_finalize_channel(int c, bool training, double sum, double sumsq, double count, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                       int64_t* __restrict__ nbt, float momentum, float eps, float* __restrict__ scale,
                                                       float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  float mean, invstd;
  if (training) {
    double unbiased;
    bx_bn_batch_stats(sum, sumsq, count, eps, mean, invstd, unbiased);
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    if (c == 0 && nbt) nbt[0] += 1;
  } else {
    mean = rmean[c];
    invstd = 1.0f / sqrtf(rvar[c] + eps);
  }
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  save_mean[c] = mean;
  save_invstd[c] = invstd;
}
// what a training-mode finalize needs besides the sums (the in-launch form carries it as one kernel argument)
struct BxBnFinalize {
  const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt; float momentum, eps;
  float* scale; float* shift; float* save_mean; float* save_invstd; double count;
};
__device__ __forceinline__ void bx_bn_finalize_channel(int c, double sum, double sumsq, const BxBnFinalize& f) {
  bx_bn_finalize_channel(c, true, sum, sumsq, f.count, f.gamma, f.beta, f.rmean, f.rvar, f.nbt, f.momentum, f.eps, f.scale, f.shift, f.save_mean, f.save_invstd);
}
// conv3 of a Block with the 2x2 pool and the BatchNorm batch statistics in its epilogue (bf16 MFMA kernels): the launcher fills the
// tree's shape from the grid it chooses; rows / mid / cnt and their capacities come from the caller (the tail's workspace and sync words)
struct BxConvPoolEpi {
  void* pooled;                   // [B, H/2, W/2, Co] bf16
  int pool, want_stats, Ho, Wo;
  BxStatTree tree;                // domain = output-channel group of the grid (blockIdx.y), row = blockIdx.x
  BxBnFinalize fin;
  const float* w1x1; float* wT;   // side job: wT[ci][c] = W1x1[c][ci] for the tail's apply kernel (zero rows beyond Cin1)
  int Cin1, Cin1_p;
  size_t rows_cap_floats; int cnt_cap_words;
  int tree_max_rows;              // more partial rows than this: rows only (tree.cnt is cleared), the caller launches the finalize
  // round 3: what the backward needs of conv3's output is only WHERE each pooled element's gradient goes -- one nibble per pooled
  // element, bit q = position q of the 2x2 window (row-major) receives it (max pool: the arg-max if it is positive; average pool: the
  // positive ones).  route != NULL: the nibbles are written ([B,Ho,Wo,C] nibbles, channel pairs per byte, low nibble = even channel);
  // store_y = 0: the full-resolution output is not stored at all (1/16 of its bytes are).
  void* route; uint32_t route_bytes; int store_y;
};
// in-launch finalize only for trees of at most this many rows (BX_TREE_MAX_ROWS; 0 = always a separate finalize launch)
int bx_tree_max_rows();
int bx_tail_fold_mask();          // folded finalizes: bit 0 backward, bit 1 forward (bx_set_tail_fold / BX_TAIL_FOLD)
int bx_conv3x3_mfma_supported(int Ci, int Co, int dtype);
int bx_conv3x3_mfma_pool_launch(const void* x, const void* packed_mfma, const float* bias, void* y, int B, int H, int W, int Ci, int Co,
                                BxConvPoolEpi* pe, hipStream_t s);

// forward 2 (separate-launch form): statistics -> (scale, shift).  Launch with 1024 threads, grid = bx_finalize_grid(C).
static __global__ void k_bn_finalize(const float* __restrict__ partials, int nblk, double count, int C, int training,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                              float* __restrict__ rvar, int64_t* __restrict__ nbt, float momentum, float eps,
                              float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                              float* __restrict__ save_invstd) {
  double tot[2] = {0.0, 0.0};
  const int CG = C < 16 ? C : 16, c0 = blockIdx.x * CG;            // grid = bx_finalize_grid(C)
  if (training) sum_partials_256<2>(partials, nblk, C, c0, CG, tot);
  const int c = c0 + threadIdx.x;
  if ((int)threadIdx.x >= CG || c >= C) return;
  bx_bn_finalize_channel(c, training != 0, tot[0], tot[1], count, gamma, beta, rmean, rvar, nbt, momentum, eps, scale, shift, save_mean, save_invstd);
}

// out[i] = sum_k part[k*n + i], fixed order.  A 256-thread workgroup covers NO = 256/S outputs x S chunk slices
// (S = 4, 16 or 64, chosen by bx_partial_slices so that small outputs still fill the chip).
static __global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, float* __restrict__ out, int nchunk, int n, int S) {
  __shared__ float sm[256];
  const int NO = 256 / S;
  const int o = threadIdx.x % NO, sl = threadIdx.x / NO;
  const int i = blockIdx.x * NO + o;
  float s = 0.f;
  if (i < n) {
    int k = sl;
    for (; k + 3 * S < nchunk; k += 4 * S) {
      const float v0 = part[(size_t)k * n + i], v1 = part[(size_t)(k + S) * n + i];
      const float v2 = part[(size_t)(k + 2 * S) * n + i], v3 = part[(size_t)(k + 3 * S) * n + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < nchunk; k += S) s += part[(size_t)k * n + i];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    float r = sm[o];
    for (int k = 1; k < S; ++k) r += sm[k * NO + o];
    out[i] = r;
  }
}
static inline int bx_partial_slices(int n, int nchunk) { return (n >= 16384 || nchunk <= 8) ? 4 : (n >= 2048 || nchunk <= 64) ? 16 : 64; }
#define BX_SUM_PARTIALS(part, out, nchunk, n, stream) do { const int S_ = bx_partial_slices((n), (nchunk)); \
  hipLaunchKernelGGL(k_sum_partials, dim3(bx_ceil_div((n), 256 / S_)), dim3(256), 0, (stream), (part), (out), (nchunk), (n), S_); } while (0)

#define BX_DISPATCH_DTYPE(dtype, T, ...) \
  do { if ((dtype) == BX_F32) { typedef float T; __VA_ARGS__; } else { typedef bf16_t T; __VA_ARGS__; } } while (0)
