// libbrainxai core: error plumbing, layout conversion, elementwise helpers, AdamW, seed.
#include <stdarg.h>
#include <string.h>
#include "bx_common.h"

static thread_local char g_err[512] = "";
void bx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" int bx_version(void) { return BX_VERSION; }
thread_local hipEvent_t g_bx_prof_ev[2] = {nullptr, nullptr};
extern "C" int bx_profile_next_conv3(void* ev_start, void* ev_stop) {
  g_bx_prof_ev[0] = (hipEvent_t)ev_start; g_bx_prof_ev[1] = (hipEvent_t)ev_stop;
  return BX_OK;
}
extern "C" const char* bx_last_error_string(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// NCHW fp32 -> NHWC(Cp) T.  One thread per pixel; reads are coalesced along x per channel plane,
// each 8-channel group of the pixel is written with 16-byte stores; channels C..Cp-1 are zero.
template <typename T>
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, T* __restrict__ dst, int C, int Cp, long long HW, long long npix) {
  // grid (ceil(HW/256), B): the sample index comes from blockIdx.y, no 64-bit division per thread
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (r >= HW) return;
  const long long p = b * HW + r;
  (void)npix;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c0 + j) < C ? src[(b * C + c0 + j) * HW + r] : 0.f;
    st8(dst, (size_t)p * Cp + c0, v);
  }
}

template <typename T>
__global__ void k_nhwc_to_nchw(const T* __restrict__ src, float* __restrict__ dst, int C, int Cs, long long HW, long long npix) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (r >= HW) return;
  const long long p = b * HW + r;
  (void)npix;
  for (int c0 = 0; c0 < C; c0 += 8) {
    float v[8];
    ld8(src, (size_t)p * Cs + c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < C) dst[(b * C + c0 + j) * HW + r] = v[j];
  }
}

extern "C" int bx_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "bx_nchw_to_nhwc: bad arguments");
  BX_REQUIRE(Cp % 8 == 0 && C <= Cp, "bx_nchw_to_nhwc: Cp must be a multiple of 8 and >= C (C=%d Cp=%d)", C, Cp);
  const long long HW = (long long)H * W, npix = HW * B;
  BX_REQUIRE(B <= 65535, "bx_nchw_to_nhwc: batch %d exceeds the grid's y extent", B);
  hipStream_t s = (hipStream_t)stream;
  BX_DISPATCH_DTYPE(dtype, T,
    hipLaunchKernelGGL((k_nchw_to_nhwc<T>), dim3(bx_ceil_div(HW, 256), B), dim3(256), 0, s, src, (T*)dst, C, Cp, HW, npix));
  BX_CHECK_LAUNCH("bx_nchw_to_nhwc");
  return BX_OK;
}

extern "C" int bx_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cs, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(src && dst && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && Cs % 8 == 0 && C <= Cs, "bx_nhwc_to_nchw: bad arguments");
  const long long HW = (long long)H * W, npix = HW * B;
  hipStream_t s = (hipStream_t)stream;
  BX_DISPATCH_DTYPE(dtype, T,
    hipLaunchKernelGGL((k_nhwc_to_nchw<T>), dim3(bx_ceil_div(HW, 256), B), dim3(256), 0, s, (const T*)src, dst, C, Cs, HW, npix));
  BX_CHECK_LAUNCH("bx_nhwc_to_nchw");
  return BX_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_relu(const T* __restrict__ x, T* __restrict__ y, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n8; i += stride) {
    float v[8];
    ld8(x, i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    st8(y, i * 8, v);
  }
}
extern "C" int bx_relu(const void* x, void* y, size_t n, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(x && y && n % 8 == 0, "bx_relu: n must be a multiple of 8");
  const size_t n8 = n / 8;
  const int grid = (int)(n8 / 256 + 1 > 2048 ? 2048 : n8 / 256 + 1);
  BX_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((k_relu<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n8));
  BX_CHECK_LAUNCH("bx_relu");
  return BX_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void k_axpby(const float* __restrict__ x, float* __restrict__ y, size_t n, float alpha, float beta) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = beta == 0.f ? alpha * x[i] : alpha * x[i] + beta * y[i];
}
__global__ void k_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = a[i] * b[i];
}
__global__ void k_scale_dev(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ o, size_t n) {
  const float k = s[0];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = x[i] * k;
}
__global__ void k_abs(const float* __restrict__ x, float* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = fabsf(x[i]);
}
static inline int ew_grid(size_t n);
extern "C" int bx_abs(const float* x, float* out, size_t n, bxStream stream) {
  BX_REQUIRE(x && out, "bx_abs: null pointer");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_abs, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
  BX_CHECK_LAUNCH("bx_abs");
  return BX_OK;
}
extern "C" int bx_scale_dev(const float* x, const float* scalar, float* out, size_t n, bxStream stream) {
  BX_REQUIRE(x && scalar && out, "bx_scale_dev: null pointer");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_scale_dev, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, out, n);
  BX_CHECK_LAUNCH("bx_scale_dev");
  return BX_OK;
}
static inline int ew_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g ? g : 1)); }
extern "C" int bx_axpby(const float* x, float* y, size_t n, float alpha, float beta, bxStream stream) {
  BX_REQUIRE(x && y, "bx_axpby: null pointer");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, alpha, beta);
  BX_CHECK_LAUNCH("bx_axpby");
  return BX_OK;
}
extern "C" int bx_mul(const float* a, const float* b, float* out, size_t n, bxStream stream) {
  BX_REQUIRE(a && b && out, "bx_mul: null pointer");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_mul, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  BX_CHECK_LAUNCH("bx_mul");
  return BX_OK;
}

// Integrated-gradients helpers: all K interpolants / all K weighted gradient slices of one pass in ONE launch each (the
// per-interpolant bx_axpby calls were 300 launches of ~3.6 us per attribution, 11 % of its time)
//   interpolate: out[k][i] = (1 - a_k) * base[i] + a_k * x[i]        accumulate: acc[i] += sum_k w_k * g[k][i]  (k ascending)
__global__ void k_ig_interpolate(const float* __restrict__ x, const float* __restrict__ base, const float* __restrict__ alphas,
                                 float* __restrict__ out, size_t n, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i], bv = base[i];
  for (int k = 0; k < K; ++k) {
    const float a = alphas[k];
    out[(size_t)k * n + i] = a * xv + (1.0f - a) * bv;
  }
}
__global__ void k_ig_accumulate(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ acc, size_t n, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = acc[i];
  int k = 0;
  for (; k + 4 <= K; k += 4) {                       // four loads in flight, added in k order
    const float g0 = g[(size_t)k * n + i], g1 = g[(size_t)(k + 1) * n + i], g2 = g[(size_t)(k + 2) * n + i], g3 = g[(size_t)(k + 3) * n + i];
    s = w[k] * g0 + s; s = w[k + 1] * g1 + s; s = w[k + 2] * g2 + s; s = w[k + 3] * g3 + s;
  }
  for (; k < K; ++k) s = w[k] * g[(size_t)k * n + i] + s;
  acc[i] = s;
}
extern "C" int bx_ig_interpolate(const float* x, const float* base, const float* alphas_device, float* out, size_t n, int K, bxStream stream) {
  BX_REQUIRE(x && base && alphas_device && out && K > 0, "bx_ig_interpolate: bad arguments");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_ig_interpolate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, base, alphas_device, out, n, K);
  BX_CHECK_LAUNCH("bx_ig_interpolate");
  return BX_OK;
}
extern "C" int bx_ig_accumulate(const float* grads, const float* weights_device, float* acc, size_t n, int K, bxStream stream) {
  BX_REQUIRE(grads && weights_device && acc && K > 0, "bx_ig_accumulate: bad arguments");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_ig_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grads, weights_device, acc, n, K);
  BX_CHECK_LAUNCH("bx_ig_accumulate");
  return BX_OK;
}

// ---------------------------------------------------------------------------------------------
// AdamW (decoupled weight decay), torch.optim.AdamW semantics:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__device__ __forceinline__ void adamw_one(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd,
                                          float gscale, float step_size, float inv_sqrt_bc2) {
  gi *= gscale;
  pi *= (1.f - lr * wd);
  mi = b1 * mi + (1.f - b1) * gi;
  vi = b2 * vi + (1.f - b2) * gi * gi;
  pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
}
// one float4 per thread and one trip per thread (a capped grid-stride loop made every trip a serial HBM round trip);
// the arenas are 16-byte aligned, the n % 4 tail goes to the first threads of block 0
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        size_t n, float lr, float b1, float b2, float eps, float wd, float gscale, const float* __restrict__ step) {
  const float t = step[0];
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const size_t n4 = n / 4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    adamw_one(pv.x, gv.x, mv.x, vv.x, lr, b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2);
    adamw_one(pv.y, gv.y, mv.y, vv.y, lr, b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2);
    adamw_one(pv.z, gv.z, mv.z, vv.z, lr, b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2);
    adamw_one(pv.w, gv.w, mv.w, vv.w, lr, b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2);
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t j = n4 * 4 + threadIdx.x;
    float pj = p[j], mj = m[j], vj = v[j];
    adamw_one(pj, g[j], mj, vj, lr, b1, b2, eps, wd, gscale, step_size, inv_sqrt_bc2);
    p[j] = pj; m[j] = mj; v[j] = vj;
  }
}
__global__ void k_step_inc(float* t) { t[0] += 1.f; }
extern "C" int bx_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float grad_scale, float* step_count, bxStream stream) {
  BX_REQUIRE(p && g && m && v && step_count, "bx_adamw_step: null pointer");
  if (n == 0) return BX_OK;
  hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step_count);
  BX_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "bx_adamw_step: arenas must be 16-byte aligned");
  const size_t nblk = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(k_adamw, dim3((unsigned)(nblk ? nblk : 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, grad_scale, step_count);
  BX_CHECK_LAUNCH("bx_adamw_step");
  return BX_OK;
}

// The same update with (i) every hyper-parameter read from a device buffer -- a captured hipGraph then follows a learning-rate
// schedule without being re-captured -- and (ii) the DDP loop's manual L2 penalty (training_distributed.py:52-53,
// total_loss = loss + lambda * sum p^2) folded in: its gradient 2*lambda*p joins g before the moments are updated (identical on
// every rank, so adding it after the all-reduce equals adding it before) and each workgroup leaves sum p^2 of its slice (of the
// parameters BEFORE the update, i.e. of the forward pass's weights) in sumsq_partials[blockIdx.x] for a fixed-order total.
// hyper[8] = {lr, beta1, beta2, eps, weight_decay, grad_scale, l2_lambda, unused}.
// The step count is advanced IN this launch (round 2: no k_step_inc launch in front): step[0] holds the number of steps taken so
// far, every workgroup reads it, uses t = step[0] + 1 and then draws a ticket; the workgroup holding the last ticket knows that
// every other workgroup has read step[0] and stores t.  Tickets are relaxed atomics issued right after the read (their round trip
// hides under the arena loads) and carry no data -- nothing but step[0] depends on them, so no release/acquire.  They are drawn
// in TWO LEVELS (groups of 64 workgroups, then one ticket per group): measured on MI355X, 1 978 workgroups drawing from ONE
// address made this 12 us kernel 37 us long -- device-scope atomics on one address retire at ~12.5 ns each.
// Counter words (u32, zero between launches) live behind the count, 64 bytes apart: word 16 = top level, word 16 * (2 + g) = group g.
#define BX_ADAMW_GROUP 64
__global__ __launch_bounds__(256) void k_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const float* __restrict__ hyper,
                                                   float* step, float* __restrict__ sumsq_partials) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gscale = hyper[5], l2 = 2.f * hyper[6];
  // ONE lane of the workgroup reads the count and draws the ticket; the other waves take the count from LDS behind a barrier.
  // (Round 2 let every wave load step[0] itself while only wave 0's load was ordered before the workgroup's ticket: a late wave
  // could have read the count the last ticket holder had already advanced and applied the next step's bias correction.)
  __shared__ float s_tprev;
  unsigned ticket = 0;
  unsigned* const cnt_top = reinterpret_cast<unsigned*>(step) + 16;
  unsigned* const cnt_grp = cnt_top + 16 * (1 + (blockIdx.x / BX_ADAMW_GROUP));
  if (threadIdx.x == 0) {
    float tp = *reinterpret_cast<volatile float*>(step);
    unsigned one = 1u;
    asm volatile("" : "+v"(one), "+v"(tp));              // the count has ARRIVED in a register before the ticket is requested
    s_tprev = tp;
    ticket = __hip_atomic_fetch_add(cnt_grp, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const size_t n4 = n / 4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the arena loads do not depend on the count: they are in flight while lane 0's read of it travels
  float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), mv = pv, vv = pv, gv = pv;
  if (i < n4) {
    pv = reinterpret_cast<float4*>(p)[i]; mv = reinterpret_cast<float4*>(m)[i]; vv = reinterpret_cast<float4*>(v)[i];
    gv = reinterpret_cast<const float4*>(g)[i];
  }
  __syncthreads();
  const float t = s_tprev + 1.f;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  float ss = 0.f;
  if (i < n4) {
    ss = pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
    adamw_one(pv.x, gv.x * gscale + l2 * pv.x, mv.x, vv.x, lr, b1, b2, eps, wd, 1.f, step_size, inv_sqrt_bc2);
    adamw_one(pv.y, gv.y * gscale + l2 * pv.y, mv.y, vv.y, lr, b1, b2, eps, wd, 1.f, step_size, inv_sqrt_bc2);
    adamw_one(pv.z, gv.z * gscale + l2 * pv.z, mv.z, vv.z, lr, b1, b2, eps, wd, 1.f, step_size, inv_sqrt_bc2);
    adamw_one(pv.w, gv.w * gscale + l2 * pv.w, mv.w, vv.w, lr, b1, b2, eps, wd, 1.f, step_size, inv_sqrt_bc2);
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t j = n4 * 4 + threadIdx.x;
    float pj = p[j], mj = m[j], vj = v[j];
    ss += pj * pj;
    adamw_one(pj, g[j] * gscale + l2 * pj, mj, vj, lr, b1, b2, eps, wd, 1.f, step_size, inv_sqrt_bc2);
    p[j] = pj; m[j] = mj; v[j] = vj;
  }
  if (sumsq_partials) {                                      // wave-uniform branch
    __shared__ float part[4];
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) sumsq_partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
  }
  if (threadIdx.x == 0) {
    const unsigned grp = blockIdx.x / BX_ADAMW_GROUP, ngrp = (gridDim.x + BX_ADAMW_GROUP - 1) / BX_ADAMW_GROUP;
    const unsigned gsize = grp + 1 < ngrp ? BX_ADAMW_GROUP : gridDim.x - grp * BX_ADAMW_GROUP;
    if (ticket == gsize - 1) {                               // last of its group: the group has read step[0]
      __hip_atomic_store(cnt_grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(cnt_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1) {      // last group: everybody has
        __hip_atomic_store(cnt_top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        step[0] = t;
      }
    }
  }
}
// fixed-order sum of the per-workgroup partials, scaled by hyper[6] (= lambda): out[0] = lambda * sum p^2
__global__ __launch_bounds__(1024) void k_l2_finalize(const float* __restrict__ part, size_t n, const float* __restrict__ hyper,
                                                       float* __restrict__ out) {
  __shared__ double wsum[16];
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += (double)part[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    out[0] = (float)(t * (double)hyper[6]);
  }
}
// eight fp32 values from the HOST into a device buffer as kernel arguments: they are copied at launch time, so the caller's
// buffer may be reused at once (a pinned staging buffer + asynchronous copy would have to outlive the copy -- under hipGraph replay
// the host runs many steps ahead of the GPU) and the store is ordered on `stream` like any kernel
struct BxF32x8 { float v[8]; };
__global__ void k_store_f32x8(float* dst, BxF32x8 a) { if (threadIdx.x < 8) dst[threadIdx.x] = a.v[threadIdx.x]; }
extern "C" int bx_store_f32x8(float* dst_device, const float* values_host, bxStream stream) {
  BX_REQUIRE(dst_device && values_host, "bx_store_f32x8: null pointer");
  BxF32x8 a;
  for (int i = 0; i < 8; ++i) a.v[i] = values_host[i];
  hipLaunchKernelGGL(k_store_f32x8, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_device, a);
  BX_CHECK_LAUNCH("bx_store_f32x8");
  return BX_OK;
}
// two 64-bit words (device addresses) into a device slot pair: a captured graph's kernels read their input pointers from such slots
// (bxEegDesc.x_slot, bx_conv3x3_pack_layout_ex), so a replay can be pointed at the caller's batch without copying it
__global__ void k_store_u64x2(uint64_t* dst, uint64_t a, uint64_t b) { if (threadIdx.x < 2) dst[threadIdx.x] = threadIdx.x ? b : a; }
extern "C" int bx_store_u64x2(uint64_t* dst_device, uint64_t a, uint64_t b, bxStream stream) {
  BX_REQUIRE(dst_device && ((uintptr_t)dst_device & 7) == 0, "bx_store_u64x2: dst must be an 8-byte aligned device pointer");
  hipLaunchKernelGGL(k_store_u64x2, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_device, a, b);
  BX_CHECK_LAUNCH("bx_store_u64x2");
  return BX_OK;
}
extern "C" size_t bx_adamw_partials(size_t n) { return (n / 4 + 255) / 256 + 1; }
extern "C" size_t bx_adamw_step_words(size_t n) {
  const size_t nblk = (n / 4 + 255) / 256, grid = nblk ? nblk : 1;
  return 16 * (2 + (grid + BX_ADAMW_GROUP - 1) / BX_ADAMW_GROUP);
}
extern "C" int bx_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float* step_count,
                                 float* sumsq_partials, float* l2_value, bxStream stream) {
  BX_REQUIRE(p && g && m && v && hyper && step_count, "bx_adamw_step_dev: null pointer");
  BX_REQUIRE((sumsq_partials == nullptr) == (l2_value == nullptr), "bx_adamw_step_dev: sumsq_partials and l2_value go together");
  if (n == 0) return BX_OK;
  BX_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "bx_adamw_step_dev: arenas must be 16-byte aligned");
  BX_REQUIRE(((uintptr_t)step_count & 3) == 0, "bx_adamw_step_dev: step_count is a buffer of bx_adamw_step_words(n) 32-bit words (see brainxai.h)");
  const size_t nblk = (n / 4 + 255) / 256;
  const unsigned grid = (unsigned)(nblk ? nblk : 1);
  hipLaunchKernelGGL(k_adamw_dev, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper, step_count, sumsq_partials);
  if (sumsq_partials)
    hipLaunchKernelGGL(k_l2_finalize, dim3(1), dim3(1024), 0, (hipStream_t)stream, sumsq_partials, (size_t)grid, hyper, l2_value);
  BX_CHECK_LAUNCH("bx_adamw_step_dev");
  return BX_OK;
}

__global__ void k_sumsq(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  // single workgroup, fixed order -> deterministic
  __shared__ float part[16];
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    out[0] = t;
  }
}
extern "C" int bx_sumsq(const float* x, size_t n, float* out, bxStream stream) {
  BX_REQUIRE(x && out, "bx_sumsq: null pointer");
  hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
  BX_CHECK_LAUNCH("bx_sumsq");
  return BX_OK;
}

// LIME's batched inference (reference XAI_Multimodality.py:1567-1574, predict_fn): uint8 images [N,H,W,C] (what LIME hands over
// after astype(np.uint8)) -> torchvision ToTensor (x / 255, channels first) -> the model's internal channels-last layout in one
// pass; and softmax over the returned scores.
template <typename T>
__global__ void k_u8_to_nhwc(const unsigned char* __restrict__ src, T* __restrict__ dst, long long npix, int C, int Cp, float scale) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  for (int c0 = 0; c0 < Cp; c0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c0 + j) < C ? (float)src[p * C + c0 + j] * scale : 0.f;
    st8(dst, (size_t)p * Cp + c0, v);
  }
}
extern "C" int bx_u8_to_nhwc(const unsigned char* src, void* dst, int N, int H, int W, int C, int Cp, float scale, int dtype, bxStream stream) {
  BX_DTYPE_OK(dtype);
  BX_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && Cp % 8 == 0 && C <= Cp, "bx_u8_to_nhwc: bad arguments");
  const long long npix = (long long)N * H * W;
  BX_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((k_u8_to_nhwc<T>), dim3(bx_ceil_div(npix, 256)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, npix,
                                                 C, Cp, scale));
  BX_CHECK_LAUNCH("bx_u8_to_nhwc");
  return BX_OK;
}
__global__ void k_softmax_rows(const float* __restrict__ x, float* __restrict__ y, int rows, int N) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * N;
  float m = -INFINITY, se = 0.f;
  for (int n = 0; n < N; ++n) m = fmaxf(m, xr[n]);
  for (int n = 0; n < N; ++n) se += expf(xr[n] - m);
  for (int n = 0; n < N; ++n) y[(size_t)r * N + n] = expf(xr[n] - m) / se;
}
extern "C" int bx_softmax_rows(const float* x, float* y, int rows, int N, bxStream stream) {
  BX_REQUIRE(x && y && rows > 0 && N > 0, "bx_softmax_rows: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows, dim3(bx_ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, y, rows, N);
  BX_CHECK_LAUNCH("bx_softmax_rows");
  return BX_OK;
}

// gradient seeds for attribution: seed[r][:] = onehot(class of sample r % B), the class being class_mode (>= 0) or the arg-max of
// logp[r % B] (first maximum, like torch.argmax) -- replaces zeros + argmax + scatter_ (+ repeat) framework launches
__global__ void k_class_seed(const float* __restrict__ logp, float* __restrict__ seed, int R, int B, int N, int class_mode) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  int c = class_mode;
  if (c < 0) {
    const float* row = logp + (size_t)(r % B) * N;
    c = 0;
    for (int n = 1; n < N; ++n) if (row[n] > row[c]) c = n;
  }
  for (int n = 0; n < N; ++n) seed[(size_t)r * N + n] = n == c ? 1.f : 0.f;
}
extern "C" int bx_class_seed(const float* logp, float* seed, int rows, int B, int N, int class_mode, bxStream stream) {
  BX_REQUIRE(seed && rows > 0 && B > 0 && N > 0 && class_mode < N && (class_mode >= 0 || logp), "bx_class_seed: bad arguments");
  hipLaunchKernelGGL(k_class_seed, dim3(bx_ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, logp, seed, rows, B, N, class_mode);
  BX_CHECK_LAUNCH("bx_class_seed");
  return BX_OK;
}

__global__ void k_seed_next(uint64_t* state, uint64_t* out) { const uint64_t v = state[0] + 1; state[0] = v; out[0] = v; }
__global__ void k_seed_next2(uint64_t* sa, uint64_t* oa, uint64_t* sb, uint64_t* ob) {
  if (threadIdx.x == 0) { const uint64_t v = sa[0] + 1; sa[0] = v; oa[0] = v; }
  else { const uint64_t v = sb[0] + 1; sb[0] = v; ob[0] = v; }
}
extern "C" int bx_seed_next2(uint64_t* state_a, uint64_t* out_a, uint64_t* state_b, uint64_t* out_b, bxStream stream) {
  BX_REQUIRE(state_a && out_a && state_b && out_b && state_a != state_b, "bx_seed_next2: bad arguments");
  hipLaunchKernelGGL(k_seed_next2, dim3(1), dim3(2), 0, (hipStream_t)stream, state_a, out_a, state_b, out_b);
  BX_CHECK_LAUNCH("bx_seed_next2");
  return BX_OK;
}
extern "C" int bx_seed_next(uint64_t* state, uint64_t* out, bxStream stream) {
  BX_REQUIRE(state && out, "bx_seed_next: null pointer");
  hipLaunchKernelGGL(k_seed_next, dim3(1), dim3(1), 0, (hipStream_t)stream, state, out);
  BX_CHECK_LAUNCH("bx_seed_next");
  return BX_OK;
}
