// LDS-DMA (global_load_lds_dwordx4) helpers shared by the kernels that stage operands without registers (gfx950).
#pragma once
#include "common.h"

// cache-policy modifiers of the DMA loads (experiments: " nt", " sc1", ...): per translation unit, before this include
#ifndef CTU_DMA_MOD
#define CTU_DMA_MOD ""        // dma16 (gathers: halo voxels, GEMM tails)
#endif
#ifndef CTU_DMA_GROUP_MOD
#define CTU_DMA_GROUP_MOD ""  // grouped issues (weight stages, GEMM tiles)
#endif

static __device__ __attribute__((aligned(16))) uint32_t g_zero16[4];  // 16 zero bytes: DMA source of padding slots

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// 64 lanes x 16 B: lane l's 16 bytes at gsrc land at LDS byte address lds_wave_base + 16 l.  Written in assembly so
// that hipcc does not count it: with the builtin it drains vmcnt(0) in front of the next LDS read that might alias,
// which serialises exactly the overlap these kernels are built around.  Completion is OUR job: counted
// s_waitcnt vmcnt(N), then a barrier, then the ds_read.  M0 is compiler-reserved, so it is saved and restored.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" CTU_DMA_MOD "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm_then_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_then_barrier_n(int n) {  // n is wave-uniform
  switch (n) {
    case 1: wait_vm_then_barrier<1>(); break;
    case 2: wait_vm_then_barrier<2>(); break;
    case 3: wait_vm_then_barrier<3>(); break;
    case 4: wait_vm_then_barrier<4>(); break;
    case 5: wait_vm_then_barrier<5>(); break;
    case 6: wait_vm_then_barrier<6>(); break;
    case 7: wait_vm_then_barrier<7>(); break;
    case 8: wait_vm_then_barrier<8>(); break;
    case 12: wait_vm_then_barrier<12>(); break;
    case 16: wait_vm_then_barrier<16>(); break;
    case 24: wait_vm_then_barrier<24>(); break;
    default: wait_vm_then_barrier<0>(); break;  // (any other count: wait for everything - always safe)
  }
}


// Group of 1..3 DMA instructions sharing one wave-uniform 64-bit base (SGPR pair) with per-lane 32-bit byte offsets,
// LDS destinations dst0, dst0 + 4096, dst0 + 8192: one M0 save/restore and no 64-bit vector address arithmetic for the
// group (the per-instruction form above costs ~15 scalar/vector instructions per DMA - as many as the MFMAs it feeds in
// the short stages of the halo kernel).  The string opens with s_nop 4: the base SGPRs come fresh from
// v_readfirstlane, and a VMEM instruction reading an SGPR written by a VALU needs 5 wait states, which hipcc does not
// insert around inline assembly (without it: a memory fault that came and went with the schedule).
__device__ __forceinline__ void dma16_group(int n, const void* sbase, unsigned v0, unsigned v1, unsigned v2,
                                            unsigned char* lds_wave_base) {
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
  const uint64_t a = (uint64_t)(uintptr_t)sbase;
  // (readfirstlane returns int: without the unsigned cast a low word with bit 31 set sign-extends over the high word)
  const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned keep;
  if (n == 3)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "s"(dst), "s"(base) : "memory", "scc");
  else if (n == 2)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(dst), "s"(base) : "memory", "scc");
  else if (n == 1)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "s"(dst), "s"(base) : "memory");
}

// Compile-time-sized variant: N (1, 2 or 4) instructions, LDS destinations STRIDE bytes apart.
template <int N, int STRIDE>
__device__ __forceinline__ void dma16_groupN(const void* sbase, const unsigned (&v)[N], unsigned char* lds_wave_base) {
  static_assert(N == 1 || N == 2 || N == 4, "group size");
  const unsigned dst =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
  const uint64_t a = (uint64_t)(uintptr_t)sbase;
  const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned keep;
  if constexpr (N == 4)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %6" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(dst), "s"(base), "n"(STRIDE)
                 : "memory", "scc");
  else if constexpr (N == 2)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4" CTU_DMA_GROUP_MOD "\n\t"
                 "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[0]), "v"(v[1]), "s"(dst), "s"(base), "n"(STRIDE) : "memory", "scc");
  else
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" CTU_DMA_GROUP_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v[0]), "s"(dst), "s"(base) : "memory");
}
