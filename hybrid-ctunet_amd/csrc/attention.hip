// Multi-head self-attention cores: ViT MHSA (networks/vit.py:66-78) and MaxViT-style block/grid window attention
// with a learned relative-position bias (networks/hybrid_CTUNet.py:481-511,559-567).
//
// v1 design (correctness-first, VALU + LDS): one workgroup = 32 queries (fwd, bwd_q) or 32 keys (bwd_kv) of one
// (group, head); K/V (or Q/dO) are streamed through LDS in chunks of 128 rows as fp32 with an odd row stride so both
// "lane = row" dot products and "lane = channel" accumulations are bank-conflict-free; softmax is online
// (flash-style) with wavefront shuffle reductions; backward recomputes P from the saved log-sum-exp.
// Tokens of a window are addressed arithmetically in the natural channels-last row order, so the reference's
// Rearrange layers cost nothing.  These are < 1.5 % of the model's FLOPs; an MFMA version is listed in DESIGN.md.
#include "attn_common.h"

#include <string.h>

// test hook: "attn_valu" = 1 forces the VALU kernels below even where the MFMA kernels (attention_mfma.hip) apply
static int g_attn_force_valu = 0;
static int g_generic_gemm = 0;  // "generic_gemm" = 1: plain GEMMs stay on the generic implicit-GEMM kernels (igemm.hip)
int ctu_option_generic_gemm() { return g_generic_gemm; }
static int g_nt_debug = 0;  // measurement hook of gemm_nt_dma (see gemm_dma.h)
int ctu_option_nt_debug() { return g_nt_debug; }
static int g_route = 0;  // A/B routing bits (see ctu_set_option "route" in ctunet_hip.h); read from memory, never getenv
int ctu_option_route() { return g_route; }
extern "C" int ctu_set_option(const char* name, int value) {
  if (name && !strcmp(name, "attn_valu")) { g_attn_force_valu = value; return CTU_OK; }
  if (name && !strcmp(name, "generic_gemm")) { g_generic_gemm = value; return CTU_OK; }
  if (name && !strcmp(name, "nt_debug")) { g_nt_debug = value; return CTU_OK; }
  if (name && !strcmp(name, "route")) { g_route = value; return CTU_OK; }
  ctu_set_error("unknown option %s", name ? name : "(null)");
  return CTU_ERR_ARG;
}

#define ATT_KC 128   // rows per LDS chunk
#define ATT_QPW 8    // queries (keys) per wave
#define NEG_BIG (-1.0e30f)

// load a chunk of rows [r0, r0+128) of one (group, head) column block at `col0` into LDS (fp32, stride DH+1)
template <typename T, int DH>
__device__ __forceinline__ void load_chunk(const T* __restrict__ src, int ld, int col0, const AttnCtx& c, int grp, int r0,
                                           float* __restrict__ dst) {
  constexpr int VPRW = DH / 8;
  for (int v = threadIdx.x; v < ATT_KC * VPRW; v += 256) {
    const int key = v / VPRW, part = v % VPRW;
    const int j = r0 + key;
    float x[8];
    if (j < c.ntok) load8(src + attn_row(c, grp, j) * ld + col0 + part * 8, x);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[key * (DH + 1) + part * 8 + e] = x[e];
  }
}

// acc(lane = channel) += sum_j w[j] * M[j][channel] over the 128-row chunk.  DH=64: lane = d.  DH=32: d = lane&31 and
// the two lane halves take even / odd rows (combined by the caller with one shfl_xor 32).
template <int DH>
__device__ __forceinline__ float chunk_weighted_sum(const float* __restrict__ w, const float* __restrict__ M, int lane) {
  float acc = 0.f;
  if (DH == 64) {
#pragma unroll 4
    for (int j = 0; j < ATT_KC; j += 4) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(w + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(p[e], M[(j + e) * (DH + 1) + lane], acc);
    }
  } else {
    const int d = lane & 31, half = lane >> 5;
#pragma unroll 4
    for (int j = 0; j < ATT_KC; j += 4) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(w + j);
      acc = fmaf(p[half], M[(j + half) * (DH + 1) + d], acc);
      acc = fmaf(p[2 + half], M[(j + 2 + half) * (DH + 1) + d], acc);
    }
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------------------
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                       T* __restrict__ out, float* __restrict__ lse, const AttnCtx c) {
  constexpr int LDK = DH + 1;
  __shared__ __attribute__((aligned(16))) float Ks[ATT_KC * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[ATT_KC * LDK];
  __shared__ __attribute__((aligned(16))) float Ps[4][ATT_KC];
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qbase = blockIdx.y * 32 + wave * ATT_QPW;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  float m[ATT_QPW], l[ATT_QPW], o[ATT_QPW];
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) { m[t] = NEG_BIG; l[t] = 0.f; o[t] = 0.f; }

  for (int kc0 = 0; kc0 < c.ntok; kc0 += ATT_KC) {
    __syncthreads();
    load_chunk<T, DH>(qkv, c.ldq, colk, c, grp, kc0, Ks);
    load_chunk<T, DH>(qkv, c.ldq, colv, c, grp, kc0, Vs);
    __syncthreads();
    const int j0 = kc0 + lane, j1 = kc0 + lane + 64;
    const int rc0 = bias_table ? relcode(c, j0 < c.ntok ? j0 : 0) : 0;
    const int rc1 = bias_table ? relcode(c, j1 < c.ntok ? j1 : 0) : 0;
#pragma unroll
    for (int t = 0; t < ATT_QPW; ++t) {
      const int qi = qbase + t;
      if (qi >= c.ntok) continue;  // wave-uniform
      const T* qrow = qkv + attn_row(c, grp, qi) * c.ldq + colq;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int d8 = 0; d8 < DH / 8; ++d8) {
        float q[8];
        load8(qrow + d8 * 8, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s0 = fmaf(q[e], Ks[lane * LDK + d8 * 8 + e], s0);
          s1 = fmaf(q[e], Ks[(lane + 64) * LDK + d8 * 8 + e], s1);
        }
      }
      s0 *= c.g.scale; s1 *= c.g.scale;
      if (bias_table) {
        const int rq = relcode(c, qi) + c.reloff;
        s0 += bias_table[(rq - rc0) * heads + head];
        s1 += bias_table[(rq - rc1) * heads + head];
      }
      if (j0 >= c.ntok) s0 = NEG_BIG;
      if (j1 >= c.ntok) s1 = NEG_BIG;
      const float mnew = fmaxf(m[t], wave_max(fmaxf(s0, s1)));
      const float alpha = __expf(m[t] - mnew);
      const float p0 = __expf(s0 - mnew), p1 = __expf(s1 - mnew);
      l[t] = l[t] * alpha + wave_sum(p0 + p1);
      m[t] = mnew;
      Ps[wave][lane] = p0;
      Ps[wave][lane + 64] = p1;
      __builtin_amdgcn_wave_barrier();
      o[t] = o[t] * alpha + chunk_weighted_sum<DH>(Ps[wave], Vs, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) {
    const int qi = qbase + t;
    if (qi >= c.ntok) continue;
    float ov = o[t];
    if (DH == 32) ov += __shfl_xor(ov, 32, 64);
    ov /= l[t];
    if (lane < DH) out[attn_row(c, grp, qi) * c.dim + head * DH + lane] = (T)ov;
    if (lane == 0) lse[((size_t)grp * heads + head) * c.ntok + qi] = m[t] + __logf(l[t]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, query side: dq and (optionally) the relative-position-bias gradient
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                         const T* __restrict__ out, const T* __restrict__ dout,
                                                         const float* __restrict__ lse, T* __restrict__ dqkv,
                                                         float* __restrict__ dbias, const AttnCtx c) {
  constexpr int LDK = DH + 1;
  __shared__ __attribute__((aligned(16))) float Ks[ATT_KC * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[ATT_KC * LDK];
  __shared__ __attribute__((aligned(16))) float Ds[4][ATT_KC];
  extern __shared__ float tbl[];  // [relm^3] bias-gradient accumulator of this head (only when dbias != NULL)
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qbase = blockIdx.y * 32 + wave * ATT_QPW;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  const int ntbl = c.relm * c.relm * c.relm;
  if (dbias)
    for (int i = threadIdx.x; i < ntbl; i += 256) tbl[i] = 0.f;

  float dq[ATT_QPW], delta[ATT_QPW], lq[ATT_QPW];
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) {
    dq[t] = 0.f; delta[t] = 0.f; lq[t] = 0.f;
    const int qi = qbase + t;
    if (qi < c.ntok) {
      const int64_t row = attn_row(c, grp, qi);
      float part = 0.f;
      if (lane < DH) part = (float)dout[row * c.dim + head * DH + lane] * (float)out[row * c.dim + head * DH + lane];
      delta[t] = wave_sum(part);
      lq[t] = lse[((size_t)grp * heads + head) * c.ntok + qi];
    }
  }

  for (int kc0 = 0; kc0 < c.ntok; kc0 += ATT_KC) {
    __syncthreads();
    load_chunk<T, DH>(qkv, c.ldq, colk, c, grp, kc0, Ks);
    load_chunk<T, DH>(qkv, c.ldq, colv, c, grp, kc0, Vs);
    __syncthreads();
    const int j0 = kc0 + lane, j1 = kc0 + lane + 64;
    const int rc0 = bias_table ? relcode(c, j0 < c.ntok ? j0 : 0) : 0;
    const int rc1 = bias_table ? relcode(c, j1 < c.ntok ? j1 : 0) : 0;
#pragma unroll
    for (int t = 0; t < ATT_QPW; ++t) {
      const int qi = qbase + t;
      if (qi >= c.ntok) continue;
      const int64_t row = attn_row(c, grp, qi);
      const T* qrow = qkv + row * c.ldq + colq;
      const T* grow = dout + row * c.dim + head * DH;
      float s0 = 0.f, s1 = 0.f, dp0 = 0.f, dp1 = 0.f;
#pragma unroll
      for (int d8 = 0; d8 < DH / 8; ++d8) {
        float q[8], go[8];
        load8(qrow + d8 * 8, q);
        load8(grow + d8 * 8, go);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = d8 * 8 + e;
          s0 = fmaf(q[e], Ks[lane * LDK + d], s0);
          s1 = fmaf(q[e], Ks[(lane + 64) * LDK + d], s1);
          dp0 = fmaf(go[e], Vs[lane * LDK + d], dp0);
          dp1 = fmaf(go[e], Vs[(lane + 64) * LDK + d], dp1);
        }
      }
      s0 *= c.g.scale; s1 *= c.g.scale;
      int i0 = 0, i1 = 0;
      if (bias_table) {
        const int rq = relcode(c, qi) + c.reloff;
        i0 = rq - rc0; i1 = rq - rc1;
        s0 += bias_table[i0 * heads + head];
        s1 += bias_table[i1 * heads + head];
      }
      const float ds0 = (j0 < c.ntok) ? __expf(s0 - lq[t]) * (dp0 - delta[t]) : 0.f;
      const float ds1 = (j1 < c.ntok) ? __expf(s1 - lq[t]) * (dp1 - delta[t]) : 0.f;
      if (dbias) {
        if (j0 < c.ntok) atomicAdd(&tbl[i0], ds0);
        if (j1 < c.ntok) atomicAdd(&tbl[i1], ds1);
      }
      Ds[wave][lane] = ds0;
      Ds[wave][lane + 64] = ds1;
      __builtin_amdgcn_wave_barrier();
      dq[t] += chunk_weighted_sum<DH>(Ds[wave], Ks, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) {
    const int qi = qbase + t;
    if (qi >= c.ntok) continue;
    float v = dq[t];
    if (DH == 32) v += __shfl_xor(v, 32, 64);
    if (lane < DH) dqkv[attn_row(c, grp, qi) * c.ldq + colq + lane] = (T)(v * c.g.scale);
  }
  if (dbias) {
    __syncthreads();
    for (int i = threadIdx.x; i < ntbl; i += 256) {
      const float v = tbl[i];
      if (v != 0.f) atomicAdd(&dbias[(size_t)i * heads + head], v);
    }
  }
}

// backward, key/value side: dk, dv
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                          const T* __restrict__ out, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, T* __restrict__ dqkv,
                                                          const AttnCtx c) {
  constexpr int LDK = DH + 1;
  constexpr int VPRW = DH / 8;
  __shared__ __attribute__((aligned(16))) float Qs[ATT_KC * LDK];
  __shared__ __attribute__((aligned(16))) float Gs[ATT_KC * LDK];  // dO
  __shared__ __attribute__((aligned(16))) float Ps[4][ATT_KC];
  __shared__ __attribute__((aligned(16))) float Ds[4][ATT_KC];
  __shared__ float Ls[ATT_KC], Dl[ATT_KC];                          // lse_i, delta_i of the chunk's queries
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kbase = blockIdx.y * 32 + wave * ATT_QPW;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  float dk[ATT_QPW], dv[ATT_QPW];
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) { dk[t] = 0.f; dv[t] = 0.f; }

  for (int qc0 = 0; qc0 < c.ntok; qc0 += ATT_KC) {
    __syncthreads();
    load_chunk<T, DH>(qkv, c.ldq, colq, c, grp, qc0, Qs);
    // dO chunk + delta_i = sum_d dO*O (reduced over the DH/8 consecutive lanes that share a query)
    for (int v = threadIdx.x; v < ATT_KC * VPRW; v += 256) {
      const int key = v / VPRW, part = v % VPRW;
      const int i = qc0 + key;
      float go[8], oo[8];
      float pd = 0.f;
      if (i < c.ntok) {
        const int64_t row = attn_row(c, grp, i);
        load8(dout + row * c.dim + head * DH + part * 8, go);
        load8(out + row * c.dim + head * DH + part * 8, oo);
#pragma unroll
        for (int e = 0; e < 8; ++e) pd = fmaf(go[e], oo[e], pd);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) go[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) Gs[key * LDK + part * 8 + e] = go[e];
      pd = group_sum(pd, VPRW);
      if (part == 0) {
        Dl[key] = pd;
        Ls[key] = (i < c.ntok) ? lse[((size_t)grp * heads + head) * c.ntok + i] : 0.f;
      }
    }
    __syncthreads();
    const int i0 = qc0 + lane, i1 = qc0 + lane + 64;
    const int rq0 = bias_table ? relcode(c, i0 < c.ntok ? i0 : 0) + c.reloff : 0;
    const int rq1 = bias_table ? relcode(c, i1 < c.ntok ? i1 : 0) + c.reloff : 0;
#pragma unroll
    for (int t = 0; t < ATT_QPW; ++t) {
      const int kj = kbase + t;
      if (kj >= c.ntok) continue;
      const int64_t row = attn_row(c, grp, kj);
      const T* krow = qkv + row * c.ldq + colk;
      const T* vrow = qkv + row * c.ldq + colv;
      float s0 = 0.f, s1 = 0.f, dp0 = 0.f, dp1 = 0.f;
#pragma unroll
      for (int d8 = 0; d8 < DH / 8; ++d8) {
        float kk[8], vv[8];
        load8(krow + d8 * 8, kk);
        load8(vrow + d8 * 8, vv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = d8 * 8 + e;
          s0 = fmaf(kk[e], Qs[lane * LDK + d], s0);
          s1 = fmaf(kk[e], Qs[(lane + 64) * LDK + d], s1);
          dp0 = fmaf(vv[e], Gs[lane * LDK + d], dp0);
          dp1 = fmaf(vv[e], Gs[(lane + 64) * LDK + d], dp1);
        }
      }
      s0 *= c.g.scale; s1 *= c.g.scale;
      if (bias_table) {
        const int rk = relcode(c, kj);
        s0 += bias_table[(rq0 - rk) * heads + head];
        s1 += bias_table[(rq1 - rk) * heads + head];
      }
      const float p0 = (i0 < c.ntok) ? __expf(s0 - Ls[lane]) : 0.f;
      const float p1 = (i1 < c.ntok) ? __expf(s1 - Ls[lane + 64]) : 0.f;
      Ps[wave][lane] = p0;
      Ps[wave][lane + 64] = p1;
      Ds[wave][lane] = p0 * (dp0 - Dl[lane]);
      Ds[wave][lane + 64] = p1 * (dp1 - Dl[lane + 64]);
      __builtin_amdgcn_wave_barrier();
      dv[t] += chunk_weighted_sum<DH>(Ps[wave], Gs, lane);
      dk[t] += chunk_weighted_sum<DH>(Ds[wave], Qs, lane);
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int t = 0; t < ATT_QPW; ++t) {
    const int kj = kbase + t;
    if (kj >= c.ntok) continue;
    float a = dk[t], b = dv[t];
    if (DH == 32) { a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64); }
    if (lane < DH) {
      const int64_t row = attn_row(c, grp, kj);
      dqkv[row * c.ldq + colk + lane] = (T)(a * c.g.scale);
      dqkv[row * c.ldq + colv + lane] = (T)b;
    }
  }
}

static int make_ctx(const ctu_attn_geom* g, AttnCtx* c) {
  CTU_REQUIRE(g, "attn geom is null");
  CTU_REQUIRE(g->dh == 32 || g->dh == 64, "attention head dim must be 32 or 64 (%d)", g->dh);
  CTU_REQUIRE(g->heads > 0 && g->B > 0 && g->D > 0 && g->H > 0 && g->W > 0, "bad attention dims");
  c->g = *g;
  c->dim = g->heads * g->dh;
  c->ldq = 3 * c->dim;
  if (g->part == 0) {
    c->ntok = g->D * g->H * g->W;
    c->groups = g->B;
    c->nd = c->nh = c->nw = 1;
    c->relm = 1; c->reloff = 0;
  } else {
    CTU_REQUIRE(g->part == 1 || g->part == 2, "bad partition %d", g->part);
    CTU_REQUIRE(g->win > 0 && g->D % g->win == 0 && g->H % g->win == 0 && g->W % g->win == 0,
                "volume %dx%dx%d not divisible by window %d", g->D, g->H, g->W, g->win);
    c->ntok = g->win * g->win * g->win;
    c->nd = g->D / g->win; c->nh = g->H / g->win; c->nw = g->W / g->win;
    c->groups = g->B * c->nd * c->nh * c->nw;
    c->relm = 2 * g->win - 1;
    c->reloff = (g->win - 1) * (c->relm * c->relm + c->relm + 1);
  }
  CTU_REQUIRE((int64_t)c->groups * g->heads < (1ll << 31), "too many attention groups");
  c->drop.thr16 = 0; c->drop.scale = 1.f; c->drop.k0 = c->drop.k1 = c->drop.site = 0;
  return CTU_OK;
}

static int set_drop(AttnCtx* c, float p, uint64_t seed, uint64_t offset) {
  CTU_REQUIRE(ctu_make_drop_ctx(p, seed, offset, &c->drop) == 0, "attention dropout: p must be in [0, 1)");
  return CTU_OK;
}
static int attn_fwd_impl(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse,
                         const ctu_attn_geom* g, float p, uint64_t seed, uint64_t offset, ctu_stream_t stream) {
  AttnCtx c;
  if (int rc = make_ctx(g, &c)) return rc;
  if (int rc = set_drop(&c, p, seed, offset)) return rc;
  CTU_REQUIRE(qkv && out && lse, "null pointer");
  CTU_REQUIRE(!bias_table || g->part != 0, "bias table needs a window partition");
  dim3 grid(c.groups * g->heads, (c.ntok + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  if (!g_attn_force_valu) {
    const int rc = attn_mfma_fwd(dtype, qkv, bias_table, out, lse, c, s);
    if (rc >= 0) return rc;
  }
  CTU_REQUIRE(c.drop.thr16 == 0, "attention dropout runs on the MFMA kernels only (bf16, or fp32 with <= 224 tokens)");
#define ATT_FWD(T, DH) \
  hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), grid, dim3(256), 0, s, (const T*)qkv, bias_table, (T*)out, lse, c)
  if (g->dh == 32) { CTU_DISPATCH(dtype, ATT_FWD(float, 32), ATT_FWD(bf16, 32)); }
  else { CTU_DISPATCH(dtype, ATT_FWD(float, 64), ATT_FWD(bf16, 64)); }
#undef ATT_FWD
  return ctu_check_launch("attn_fwd");
}

extern "C" int ctu_attn_fwd(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse,
                            const ctu_attn_geom* g, ctu_stream_t stream) {
  return attn_fwd_impl(dtype, qkv, bias_table, out, lse, g, 0.f, 0, 0, stream);
}
extern "C" int ctu_attn_fwd_dropout(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse,
                                    const ctu_attn_geom* g, float p, uint64_t seed, uint64_t offset, ctu_stream_t stream) {
  return attn_fwd_impl(dtype, qkv, bias_table, out, lse, g, p, seed, offset, stream);
}

static int attn_bwd_impl(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out, const void* dout,
                         const float* lse, void* dqkv, float* dbias, const ctu_attn_geom* g, float p, uint64_t seed,
                         uint64_t offset, ctu_stream_t stream) {
  AttnCtx c;
  if (int rc = make_ctx(g, &c)) return rc;
  if (int rc = set_drop(&c, p, seed, offset)) return rc;
  CTU_REQUIRE(qkv && out && dout && lse && dqkv, "null pointer");
  CTU_REQUIRE((bias_table != nullptr) == (dbias != nullptr), "bias_table and dbias must be given together");
  CTU_REQUIRE(!bias_table || g->part != 0, "bias table needs a window partition");
  dim3 grid(c.groups * g->heads, (c.ntok + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  if (!g_attn_force_valu) {
    const int rc = attn_mfma_bwd(dtype, qkv, bias_table, out, dout, lse, dqkv, dbias, c, s);
    if (rc >= 0) return rc;
  }
  CTU_REQUIRE(c.drop.thr16 == 0, "attention dropout runs on the MFMA kernels only (bf16, or fp32 with <= 224 tokens)");
  const size_t tbl_bytes = dbias ? (size_t)c.relm * c.relm * c.relm * sizeof(float) : 0;
#define ATT_BWD(T, DH)                                                                                               \
  do {                                                                                                               \
    hipLaunchKernelGGL((attn_bwd_q_kernel<T, DH>), grid, dim3(256), tbl_bytes, s, (const T*)qkv, bias_table,         \
                       (const T*)out, (const T*)dout, lse, (T*)dqkv, dbias, c);                                      \
    hipLaunchKernelGGL((attn_bwd_kv_kernel<T, DH>), grid, dim3(256), 0, s, (const T*)qkv, bias_table, (const T*)out, \
                       (const T*)dout, lse, (T*)dqkv, c);                                                            \
  } while (0)
  if (g->dh == 32) { CTU_DISPATCH(dtype, ATT_BWD(float, 32), ATT_BWD(bf16, 32)); }
  else { CTU_DISPATCH(dtype, ATT_BWD(float, 64), ATT_BWD(bf16, 64)); }
#undef ATT_BWD
  return ctu_check_launch("attn_bwd");
}

extern "C" int ctu_attn_bwd(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out,
                            const void* dout, const float* lse, void* dqkv, float* dbias, const ctu_attn_geom* g,
                            ctu_stream_t stream) {
  return attn_bwd_impl(dtype, qkv, bias_table, out, dout, lse, dqkv, dbias, g, 0.f, 0, 0, stream);
}
extern "C" int ctu_attn_bwd_dropout(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out,
                                    const void* dout, const float* lse, void* dqkv, float* dbias, const ctu_attn_geom* g,
                                    float p, uint64_t seed, uint64_t offset, ctu_stream_t stream) {
  return attn_bwd_impl(dtype, qkv, bias_table, out, dout, lse, dqkv, dbias, g, p, seed, offset, stream);
}
