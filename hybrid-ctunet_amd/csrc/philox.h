// Counter-based RNG for the dropout path: Philox4x32-10 (Salmon et al., SC'11).  A mask is a pure
// function of (seed, offset, element index), so backward regenerates it instead of storing it.
//
// One call yields 4 x 32 bits = eight 16-bit draws; draw u keeps its element iff u >= thr16, thr16 = round(p * 65536).
//   * flat tensors (ctu_dropout): element i -> counter (i >> 3, i >> 35, 0xD0D0D0D0, site), draw i & 7
//   * attention probabilities of pair P = group * heads + head: a call covers 2 queries x 4 keys,
//     counter (key >> 2, query >> 1, P, site), draw 4 (query & 1) + (key & 3)
// where draw j is bits [16 (j & 1), 16 (j & 1) + 16) of output word j >> 1.  oracle/dropout_oracle.py restates both.
#pragma once
#include <stdint.h>

struct DropCtx {
  uint32_t thr16;      // 0 .. 65536; 0 = no dropout
  float scale;         // 65536 / (65536 - thr16)
  uint32_t k0, k1;     // seed
  uint32_t site;       // which dropout call of the step (offset)
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t philox_draw16(const uint32_t (&o)[4], int j) { return (o[j >> 1] >> (16 * (j & 1))) & 0xffffu; }

// the four draws of (query, keys 4 kq .. 4 kq + 3) of attention pair `pair`: bit i of the result = keep key 4 kq + i
__device__ __forceinline__ uint32_t attn_keep4(const DropCtx& d, uint32_t pair, int query, int kq) {
  uint32_t o[4];
  philox4x32_10((uint32_t)kq, (uint32_t)(query >> 1), pair, d.site, d.k0, d.k1, o);
  const uint32_t w0 = o[2 * (query & 1)], w1 = o[2 * (query & 1) + 1];
  return ((w0 & 0xffffu) >= d.thr16 ? 1u : 0u) | ((w0 >> 16) >= d.thr16 ? 2u : 0u) | ((w1 & 0xffffu) >= d.thr16 ? 4u : 0u) |
         ((w1 >> 16) >= d.thr16 ? 8u : 0u);
}
// keep flag of one (query, key) pair (dK/dV kernel: the lane owns a key, the accumulator rows are queries)
__device__ __forceinline__ bool attn_keep1(const DropCtx& d, uint32_t pair, int query, int key) {
  return (attn_keep4(d, pair, query, key >> 2) >> (key & 3)) & 1u;
}
// keep flags of (queries 2 qh, 2 qh + 1; key): bit 0 / bit 1 - one call serves both (the dK/dV kernel walks queries in fours)
__device__ __forceinline__ uint32_t attn_keep_qpair(const DropCtx& d, uint32_t pair, int qh, int key) {
  uint32_t o[4];
  philox4x32_10((uint32_t)(key >> 2), (uint32_t)qh, pair, d.site, d.k0, d.k1, o);
  const int j = key & 3, sh = 16 * (j & 1);
  const uint32_t u0 = (o[j >> 1] >> sh) & 0xffffu, u1 = (o[2 + (j >> 1)] >> sh) & 0xffffu;
  return (u0 >= d.thr16 ? 1u : 0u) | (u1 >= d.thr16 ? 2u : 0u);
}
