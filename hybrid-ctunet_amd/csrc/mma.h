// MFMA fragment helpers shared by the implicit-GEMM kernels (gfx950).
#pragma once
// Measurement build only (tools/build_variant.sh): -DCTU_MFMA_YIELD=n puts `s_nop n-1` behind every bf16 MFMA of the halo kernels.
// A wave whose next instruction is an MFMA waiting for the matrix pipe keeps every other wave of its SIMD from issuing
// (profiles/r04_corun_mechanism.log); the nop completes long before the pipe is free again and leaves those cycles to the others.
#ifndef CTU_MFMA_YIELD
#define CTU_MFMA_YIELD 0
#endif
#if CTU_MFMA_YIELD
#define CTU_MFMA_YIELD_HERE asm volatile("s_nop %0" ::"n"(CTU_MFMA_YIELD - 1));
#else
#define CTU_MFMA_YIELD_HERE
#endif

#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// MFMA wrappers.  Fragment = 8 consecutive-k elements of one row (lane r = row, lane half h -> k = 8h..8h+7)
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Mma;

template <> struct Mma<bf16> {
  typedef bf16x8 Frag;
  static __device__ __forceinline__ Frag load(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
  // k-strided gather (transposed operand) from LDS: element j at p[j*stride], where p is THIS lane's pointer
  // &M[k0 + 8h][c0 + r].  Done with two ds_read_b64_tr_b16: per 16-lane group the instruction reads a 4(k) x 16(c) block
  // and hands lane i column i; lane 4q+pp of the group must supply the address of block row q, columns 4pp..4pp+3,
  // i.e. its own pointer moved by q rows and (3pp - 4q) columns.  Needs EXEC all ones, (c0, stride) multiples of 4.
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
  typedef __attribute__((address_space(3))) bf16x4_* lds_v4_ptr;
  static __device__ __forceinline__ Frag gather2(const bf16* p, int stride, int second_rows) {
    const int i = threadIdx.x & 15, q = i >> 2, pp = i & 3;
    const bf16* a0 = p + q * stride + 3 * pp - 4 * q;
    const bf16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_ptr)a0);
    const bf16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_ptr)(a0 + second_rows * stride));
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
    return f;
  }
  static __device__ __forceinline__ Frag gather(const bf16* p, int stride) { return gather2(p, stride, 4); }
  // Three k-shifted fragments from ONE fetch of twelve rows: f0 = rows 0..7, f1 = rows 1..8, f2 = rows 2..9 of this lane's column
  // (rows 10, 11 are read and dropped).  Three transposed reads instead of six; f1 costs four v_alignbit.  Split in two so that the
  // reads of the next k step can be in flight while the fragments of this one are cut and multiplied.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  static __device__ __forceinline__ void row3_fetch(const bf16* p, int stride, unsigned (&v)[5]) {
    const int i = threadIdx.x & 15, q = i >> 2, pp = i & 3;
    const bf16* a0 = p + q * stride + 3 * pp - 4 * q;
    const u32x2_ r0 = __builtin_bit_cast(u32x2_, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_ptr)a0));
    const u32x2_ r1 = __builtin_bit_cast(u32x2_, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_ptr)(a0 + 4 * stride)));
    const u32x2_ r2 = __builtin_bit_cast(u32x2_, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4_ptr)(a0 + 8 * stride)));
    v[0] = r0[0]; v[1] = r0[1]; v[2] = r1[0]; v[3] = r1[1]; v[4] = r2[0];
  }
  static __device__ __forceinline__ void row3_frags(const unsigned (&v)[5], Frag& f0, Frag& f1, Frag& f2) {
    const u32x4_ w0 = {v[0], v[1], v[2], v[3]};
    const u32x4_ w1 = {__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                       __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(v[4], v[3], 16)};
    const u32x4_ w2 = {v[1], v[2], v[3], v[4]};
    f0 = __builtin_bit_cast(Frag, w0);
    f1 = __builtin_bit_cast(Frag, w1);
    f2 = __builtin_bit_cast(Frag, w2);
  }
  // rows +0..3 and +8..11: the order in which an accumulator tile presents itself as the other MFMA operand
  static __device__ __forceinline__ Frag gather_perm(const bf16* p, int stride) { return gather2(p, stride, 8); }
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    CTU_MFMA_YIELD_HERE
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
  static __device__ __forceinline__ Frag gather_perm(const float* p, int stride) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = p[((j & 3) + 8 * (j >> 2)) * stride];
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

