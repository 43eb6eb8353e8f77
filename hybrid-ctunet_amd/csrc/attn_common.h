// Geometry helpers shared by the attention translation units (VALU reference kernels and MFMA kernels).
#pragma once
#include "common.h"
#include "philox.h"

struct AttnCtx {
  ctu_attn_geom g;
  int ntok, nd, nh, nw, groups, relm, reloff, dim, ldq;
  DropCtx drop;  // dropout of the attention probabilities (thr16 = 0: none)
};
int ctu_make_drop_ctx(float p, uint64_t seed, uint64_t offset, DropCtx* d);  // dropout.hip

// row of token i of attention group grp in the natural channels-last row order (see ctu_attn_geom)
__device__ __forceinline__ int64_t attn_row(const AttnCtx& c, int grp, int i) {
  if (c.g.part == 0) return (int64_t)grp * c.ntok + i;
  int t = grp;
  const int wz = t % c.nw; t /= c.nw;
  const int wy = t % c.nh; t /= c.nh;
  const int wx = t % c.nd;
  const int b = t / c.nd;
  const int win = c.g.win;
  const int i3 = i % win, i2 = (i / win) % win, i1 = i / (win * win);
  int d, h, w;
  if (c.g.part == 1) { d = wx * win + i1; h = wy * win + i2; w = wz * win + i3; }
  else { d = i1 * c.nd + wx; h = i2 * c.nh + wy; w = i3 * c.nw + wz; }
  return (((int64_t)b * c.g.D + d) * c.g.H + h) * c.g.W + w;
}
// idx(i,j) = relcode(i) - relcode(j) + reloff  (hybrid_CTUNet.py:472-477)
__device__ __forceinline__ int relcode(const AttnCtx& c, int i) {
  const int win = c.g.win;
  const int i3 = i % win, i2 = (i / win) % win, i1 = i / (win * win);
  return (i1 * c.relm + i2) * c.relm + i3;
}

// MFMA kernels (attention_mfma.hip).  Return CTU_OK, or -1 when the shape does not fit their LDS budget (the caller
// then uses the VALU kernels).
int attn_mfma_fwd(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse, const AttnCtx& c,
                  hipStream_t s);
int attn_mfma_bwd(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out, const void* dout,
                  const float* lse, void* dqkv, float* dbias, const AttnCtx& c, hipStream_t s);
