// MFMA fragment helpers shared by the implicit-GEMM kernels (gfx950).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// MFMA wrappers.  Fragment = 8 consecutive-k elements of one row (lane r = row, lane half h -> k = 8h..8h+7)
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Mma;

template <> struct Mma<bf16> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag load(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
  // k-strided gather (transposed operand): element j at p[j*stride]
  static __device__ __forceinline__ Frag gather(const bf16* p, int stride) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = p[j * stride];
    return f;
  }
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct Mma<float> {
  struct Frag { float v[8]; };
  static __device__ __forceinline__ Frag load(const float* p) {
    Frag f;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
    return f;
  }
  static __device__ __forceinline__ Frag gather(const float* p, int stride) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = p[j * stride];
    return f;
  }
  // 32x32x2: lane (r,h) supplies A[r][k=h], B[k=h][r]; step j pairs k = j (h=0) with k = 8+j (h=1) on both operands
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
  }
};

__device__ __forceinline__ bool gather_coord(int o, int t, int s, int p, int n_in, int mode, int& i) {
  if (mode == 0) {
    i = o * s - p + t;
    return (unsigned)i < (unsigned)n_in;
  }
  const int z = o + p - t;  // stride is 1 or 2 in mode 1 (checked on the host)
  if (z < 0) return false;
  i = (s == 1) ? z : (z >> 1);
  return ((s == 1) || !(z & 1)) && i < n_in;
}

// XCD-aware bijective remap of a 1-D block id: blocks dealt round-robin to 8 XCDs get contiguous tile ranges,
// so neighbouring tiles (which share halo voxels / weight panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, rr = nwg & 7, x = bid & 7;
  return (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (bid >> 3);
}

