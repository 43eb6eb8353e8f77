// Device-side helpers shared by all kernels of libctunet_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctunet_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LRELU_SLOPE 0.01f
#define NORM_EPS 1e-5f

void ctu_set_error(const char* fmt, ...);
int ctu_check_launch(const char* what);
int ctu_option_generic_gemm();  // test hook, see ctu_set_option
int ctu_option_route();         // A/B routing bits, see ctu_set_option "route"
#define CTU_ROUTE_NT_NO_STREAM 1    /* short-K layers back on the general NT kernel */
#define CTU_ROUTE_NT_NO_BK128 2     /* no 128-deep stages for the 64x64 trunk tiles */
#define CTU_ROUTE_NT_NO_KG2 4       /* no two-k-group 8-wave trunk tiles */
#define CTU_ROUTE_HALO_KSPLIT_OLD 8 /* previous channel-split rule of the small-volume 3x3x3 convs */
#define CTU_ROUTE_HALO_TPS9 16      /* 9-tap weight stages in the halo kernels with one or two n tiles */
#define CTU_ROUTE_HALO_WGRAD_BURST 64  /* weight-gradient halo kernel: next brick's operand DMA in one burst behind the barrier (previous rule) */
#define CTU_ROUTE_HALO_WGRAD_ATOMICS 128 /* weight-gradient halo kernel: atomics into the panel even when a workspace for partial panels is given */
#define CTU_ROUTE_IN_REDUCE_LDS32 256  /* InstanceNorm backward reduce: the 32-KiB LDS reduction for every channel count (previous rule) */
#define CTU_ROUTE_HALO_THIN 512 /* halo forward / data-gradient kernels: one resident workgroup fewer per CU (padding dynamic LDS), leaving registers and wave slots to HBM-bound kernels of other streams */
#define CTU_ROUTE_IN_ROWS64 1024 /* InstanceNorm reductions: at least 64 rows per workgroup (previous rule) instead of 16 */
#define CTU_ROUTE_HALO_NO_BATCH_PAIR 2048 /* halo forward / data-gradient kernels: 4 x 8 x 8 bricks of one batch item for every volume (previous rule) */
#define CTU_ROUTE_HALO_NSPLIT 4096 /* halo forward / data-gradient kernels with four n tiles: the n-split variant (wave-private weight rings, one barrier per half chunk) */
#define CTU_ROUTE_NT_NO_NARROW 16384 /* narrow-output layers (N = 32 / 64) back on the general NT kernel */
#define CTU_ROUTE_IN_GRID_8192 32768 /* InstanceNorm apply kernels: 8 192 workgroups in all (previous rule) instead of 1 024 */
#define CTU_ROUTE_NT_NARROW_STATS 65536 /* narrow-output layers WITH fused InstanceNorm sums on the streaming kernel too (off: see gemm_narrow.hip) */
#define CTU_ROUTE_NO_GATHER_GEMM 131072 /* in-grid strided taps (patch / transposed convolutions, 1x1x1 s2) on the generic kernels, not the LDS-DMA GEMMs */
#define CTU_ROUTE_HALO_GATHER_WAVE0 32 /* wave 0 gathers the halo alone (previous rule) instead of 7 + 3 x 4 pieces over the four waves */

#define CTU_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      ctu_set_error(__VA_ARGS__);   \
      return CTU_ERR_ARG;           \
    }                               \
  } while (0)

// dispatch a templated launcher on the activation dtype
#define CTU_DISPATCH(dtype, CALL_F32, CALL_BF16)                    \
  do {                                                              \
    if ((dtype) == CTU_F32) { CALL_F32; }                           \
    else if ((dtype) == CTU_BF16) { CALL_BF16; }                    \
    else { ctu_set_error("bad dtype %d", (int)(dtype)); return CTU_ERR_ARG; } \
  } while (0)

// ---- 8-element vector access: 16 B for bf16, 2 x 16 B for f32 -------------------------------------------
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
// reduction over aligned groups of `width` lanes (power of two <= 64)
__device__ __forceinline__ float group_sum(float x, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Division of a value below 2^31 by a launch constant: q = (n * ceil(2^(32 + s) / d)) >> (32 + s), s = ceil(log2 d): exact
// for n < 2^31, four instructions instead of the ~40 of a 32-bit division.
struct FastDiv { unsigned long long m; int s; int d; };
static inline FastDiv fast_div(int d) {
  FastDiv f;
  f.d = d;
  f.s = 0;
  while ((1ll << f.s) < d) ++f.s;
  f.m = (((unsigned long long)1 << (32 + f.s)) + (unsigned long long)d - 1) / (unsigned long long)d;
  return f;
}
__device__ __forceinline__ int fdiv(unsigned n, const FastDiv& f) { return (int)(((unsigned long long)n * f.m) >> (32 + f.s)); }

static inline unsigned grid_for(int64_t work_items, int block, int64_t cap = 8192) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}
