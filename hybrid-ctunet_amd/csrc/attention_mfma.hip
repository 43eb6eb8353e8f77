// MFMA attention for the ViT trunk (N = 432, dh = 64; networks/vit.py:66-78) and the 6^3 block/grid window attention
// with relative-position bias (N = 216, dh = 32; networks/hybrid_CTUNet.py:481-511).
//
// One workgroup = one (group, head); K and V (forward, dQ) or Q and dO (dK/dV) of that head sit in LDS in `T`.
// Forward / dQ:  a wave owns a 32-query tile and computes S^T = K Q^T with the KEY on the accumulator rows and the
//   QUERY on the lanes, so softmax statistics are lane-local (one xor-32 shuffle joins the two lane halves) and the
//   P^T accumulator registers feed the next MFMA (O^T = V^T P^T, dQ^T = K^T dS^T) directly as its B operand; the A operand
//   (V^T / K^T) is gathered from LDS in the matching permuted key order.
// dK/dV:  a wave owns a 32-key tile and uses the untransposed S = Q K^T (query on rows) for dV^T += dO^T P, dK^T += Q^T dS.
// Templated on T like the GEMM kernels: bf16 -> v_mfma_f32_32x32x16_bf16, f32 -> exact v_mfma_f32_32x32x2_f32.
#include "attn_common.h"
#include "mma.h"

#define MF_BIG 1.0e30f

template <typename T> struct FragOps;
template <> struct FragOps<bf16> {
  static __device__ __forceinline__ bf16x8 make(const float (&v)[8]) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)v[j];
    return f;
  }
  static __device__ __forceinline__ bf16x8 zero() { bf16x8 f; for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f; return f; }
  // element j = p[(j&3 + 8*(j>>2)) * stride]: the key order in which an accumulator tile presents itself as B operand
  static __device__ __forceinline__ bf16x8 gather_perm(const bf16* p, int stride) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = p[((j & 3) + 8 * (j >> 2)) * stride];
    return f;
  }
};
template <> struct FragOps<float> {
  typedef Mma<float>::Frag F;
  static __device__ __forceinline__ F make(const float (&v)[8]) {
    F f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = v[j];
    return f;
  }
  static __device__ __forceinline__ F zero() { F f; for (int j = 0; j < 8; ++j) f.v[j] = 0.f; return f; }
  static __device__ __forceinline__ F gather_perm(const float* p, int stride) {
    F f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = p[((j & 3) + 8 * (j >> 2)) * stride];
    return f;
  }
};

// rows [0, KT*32) of one (group, head) column block -> LDS [row][DH + EV] in T; rows >= ntok are zero
template <typename T, int DH, int KT>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int ld, int col0, const AttnCtx& c, int grp,
                                           T* __restrict__ dst) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int LD = DH + EV;
  constexpr int VPRW = DH / EV;
  for (int v = threadIdx.x; v < KT * 32 * VPRW; v += 256) {
    const int row = v / VPRW, part = v - row * VPRW;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (row < c.ntok) val = *reinterpret_cast<const u32x4*>(src + attn_row(c, grp, row) * ld + col0 + part * EV);
    *reinterpret_cast<u32x4*>(&dst[row * LD + part * EV]) = val;
  }
}

// B-operand fragments of a 32-row tile read straight from global: lane (r, h) takes row r, elements ks*16 + 8h .. +7
template <typename T, int DH>
__device__ __forceinline__ void load_row_frags(const T* __restrict__ src, int ld, int col0, const AttnCtx& c, int grp,
                                               int row, bool valid, int h, typename Mma<T>::Frag (&f)[DH / 16]) {
  if (valid) {
    const T* p = src + attn_row(c, grp, row) * ld + col0 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) f[ks] = Mma<T>::load(p + ks * 16);
  } else {
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) f[ks] = FragOps<T>::zero();
  }
}

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------------------
template <typename T, int DH, int KT, bool DROP>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                            T* __restrict__ out, float* __restrict__ lse, const AttnCtx c) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int LD = DH + EV;
  constexpr int NP = KT * 32;
  constexpr int CH = 7;  // key tiles per softmax chunk (7 x 16 accumulator registers)
  __shared__ __attribute__((aligned(16))) T Ks[NP * LD];
  __shared__ __attribute__((aligned(16))) T Vs[NP * LD];
  __shared__ float tbl[1331];
  __shared__ int krel[NP];
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  stage_rows<T, DH, KT>(qkv, c.ldq, colk, c, grp, Ks);
  stage_rows<T, DH, KT>(qkv, c.ldq, colv, c, grp, Vs);
  const bool has_bias = bias_table != nullptr;
  if (has_bias) {
    const int ntbl = c.relm * c.relm * c.relm;
    for (int i = threadIdx.x; i < ntbl; i += 256) tbl[i] = bias_table[(size_t)i * heads + head];
    for (int i = threadIdx.x; i < NP; i += 256) krel[i] = relcode(c, i < c.ntok ? i : 0);
  }
  __syncthreads();

  const int ktiles = (c.ntok + 31) / 32;
  for (int qt = wave + 4 * blockIdx.y; qt < ktiles; qt += 4 * gridDim.y) {  // blockIdx.y: split of the tile loop
    const int q = qt * 32 + r;
    const bool qok = q < c.ntok;
    typename Mma<T>::Frag fq[DH / 16];
    load_row_frags<T, DH>(qkv, c.ldq, colq, c, grp, q, qok, h, fq);
    const int relq = has_bias ? relcode(c, qok ? q : 0) + c.reloff : 0;
    float m = -MF_BIG, l = 0.f;
    f32x16 oacc[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;

    for (int kc = 0; kc < ktiles; kc += CH) {
      f32x16 s[CH];
#pragma unroll
      for (int t = 0; t < CH; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
        if (kc + t < ktiles) {
#pragma unroll
          for (int ks = 0; ks < DH / 16; ++ks)
            Mma<T>::mma(Mma<T>::load(&Ks[((kc + t) * 32 + r) * LD + ks * 16 + 8 * h]), fq[ks], s[t]);
        }
      }
      float mloc = -MF_BIG;
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = (kc + t) * 32 + acc_row(e, h);
          float v = s[t][e] * c.g.scale;
          if (has_bias) v += tbl[relq - krel[key < NP ? key : 0]];
          v = (key < c.ntok) ? v : -MF_BIG;
          s[t][e] = v;
          mloc = fmaxf(mloc, v);
        }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m, mloc);
      const float alpha = __expf(m - mnew);
      float lsum = 0.f;
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pv = __expf(s[t][e] - mnew);
          s[t][e] = pv;
          lsum += pv;
        }
      lsum += __shfl_xor(lsum, 32, 64);
      l = l * alpha + lsum;
      m = mnew;
      if (DROP) {  // the normaliser keeps every probability; dropped ones leave the P V product (1/(1-p) joins 1/l below)
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const uint32_t keep = attn_keep4(c.drop, blockIdx.x, q, (kc + t) * 8 + 2 * g4 + h);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!((keep >> j) & 1u)) s[t][4 * g4 + j] = 0.f;
          }
      }
#pragma unroll
      for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
      // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
      for (int t = 0; t < CH; ++t) {
        if (kc + t < ktiles) {
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = s[t][8 * sh + j];
            const typename Mma<T>::Frag fb = FragOps<T>::make(pv);
            const int kbase = (kc + t) * 32 + 16 * sh + 4 * h;
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt)
              Mma<T>::mma(Mma<T>::gather_perm(&Vs[kbase * LD + dt * 32 + r], LD), fb, oacc[dt]);
          }
        }
      }
    }
    if (qok) {
      const float inv = (DROP ? c.drop.scale : 1.0f) / l;
      T* orow = out + attn_row(c, grp, q) * c.dim + head * DH;
#pragma unroll
      for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) orow[dt * 32 + acc_row(e, h)] = (T)(oacc[dt][e] * inv);
      if (h == 0) lse[((size_t)grp * heads + head) * c.ntok + q] = m + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, query side: dQ (and the relative-position-bias gradient).  LDS: K, V.
template <typename T, int DH, int KT, bool DROP>
__global__ __launch_bounds__(256) void attn_mfma_bwd_q_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                              const T* __restrict__ out, const T* __restrict__ dout,
                                                              const float* __restrict__ lse, T* __restrict__ dqkv,
                                                              float* __restrict__ dbias, const AttnCtx c) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int LD = DH + EV;
  constexpr int NP = KT * 32;
  __shared__ __attribute__((aligned(16))) T Ks[NP * LD];
  __shared__ __attribute__((aligned(16))) T Vs[NP * LD];
  __shared__ float tbl[1331];
  // bias-gradient table, one private copy per wave, updated with PLAIN read-add-write: for a fixed key the map
  // query -> relative-position index is injective, so the 32 lanes of a half never collide and the two halves go one
  // after the other.  (ds_add_f32 retires about one lane per clock: 16 of them per lane and tile pair made this
  // kernel 4x slower than its key/value twin.)
  __shared__ float dtbl[4][1332];  // [1331] = dump slot of masked (out-of-range) pairs
  __shared__ int krel[NP];
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  stage_rows<T, DH, KT>(qkv, c.ldq, colk, c, grp, Ks);
  stage_rows<T, DH, KT>(qkv, c.ldq, colv, c, grp, Vs);
  const bool has_bias = bias_table != nullptr;
  const int ntbl = c.relm * c.relm * c.relm;
  if (has_bias) {
    for (int i = threadIdx.x; i < ntbl; i += 256) {
      tbl[i] = bias_table[(size_t)i * heads + head];
      dtbl[0][i] = 0.f; dtbl[1][i] = 0.f; dtbl[2][i] = 0.f; dtbl[3][i] = 0.f;
    }
    for (int i = threadIdx.x; i < NP; i += 256) krel[i] = relcode(c, i < c.ntok ? i : 0);
  }
  __syncthreads();

  const int ktiles = (c.ntok + 31) / 32;
  for (int qt = wave + 4 * blockIdx.y; qt < ktiles; qt += 4 * gridDim.y) {  // blockIdx.y: split of the tile loop
    const int q = qt * 32 + r;
    const bool qok = q < c.ntok;
    typename Mma<T>::Frag fq[DH / 16], fg[DH / 16], fo[DH / 16];
    load_row_frags<T, DH>(qkv, c.ldq, colq, c, grp, q, qok, h, fq);
    load_row_frags<T, DH>(dout, c.dim, head * DH, c, grp, q, qok, h, fg);
    load_row_frags<T, DH>(out, c.dim, head * DH, c, grp, q, qok, h, fo);
    // delta_q = sum_d dO*O: each lane half holds half of the row's elements
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
      const T* pg = reinterpret_cast<const T*>(&fg[ks]);
      const T* po = reinterpret_cast<const T*>(&fo[ks]);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta = fmaf((float)pg[j], (float)po[j], delta);
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lq = qok ? lse[((size_t)grp * heads + head) * c.ntok + q] : MF_BIG;
    const int relq = has_bias ? relcode(c, qok ? q : 0) + c.reloff : 0;
    f32x16 dq[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

    for (int kt = 0; kt < ktiles; ++kt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) {
        Mma<T>::mma(Mma<T>::load(&Ks[(kt * 32 + r) * LD + ks * 16 + 8 * h]), fq[ks], s);
        Mma<T>::mma(Mma<T>::load(&Vs[(kt * 32 + r) * LD + ks * 16 + 8 * h]), fg[ks], dp);
      }
      if (DROP) {  // dP = keep ? dP_dropped / (1 - p) : 0
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const uint32_t keep = attn_keep4(c.drop, blockIdx.x, q, kt * 8 + 2 * g4 + h);
#pragma unroll
          for (int j = 0; j < 4; ++j) dp[4 * g4 + j] = ((keep >> j) & 1u) ? dp[4 * g4 + j] * c.drop.scale : 0.f;
        }
      }
      float ds[16];
      int bis[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * 32 + acc_row(e, h);
        float v = s[e] * c.g.scale;
        int bi = 0;
        if (has_bias) { bi = relq - krel[key]; v += tbl[bi]; }
        const bool ok = qok && key < c.ntok;
        const float pv = ok ? __expf(v - lq) : 0.f;
        ds[e] = pv * (dp[e] - delta);  // 0 for masked pairs: adding it below is harmless
        bis[e] = ok ? bi : 1331;
      }
      if (has_bias) {
        float* mytbl = dtbl[wave];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (h == half) {
#pragma unroll
            for (int e = 0; e < 16; ++e) mytbl[bis[e]] += ds[e];
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = ds[8 * sh + j];
        const typename Mma<T>::Frag fb = FragOps<T>::make(pv);
        const int kbase = kt * 32 + 16 * sh + 4 * h;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt)
          Mma<T>::mma(Mma<T>::gather_perm(&Ks[kbase * LD + dt * 32 + r], LD), fb, dq[dt]);
      }
    }
    if (qok) {
      T* grow = dqkv + attn_row(c, grp, q) * c.ldq + colq;
#pragma unroll
      for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) grow[dt * 32 + acc_row(e, h)] = (T)(dq[dt][e] * c.g.scale);
    }
  }
  if (has_bias) {
    __syncthreads();
    for (int i = threadIdx.x; i < ntbl; i += 256) {
      const float v = (dtbl[0][i] + dtbl[1][i]) + (dtbl[2][i] + dtbl[3][i]);
      if (v != 0.f) atomicAdd(&dbias[(size_t)i * heads + head], v);
    }
  }
}

// backward, key/value side: dK, dV.  LDS: Q, dO, lse, delta.
template <typename T, int DH, int KT, bool DROP>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kv_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_table,
                                                               const T* __restrict__ out, const T* __restrict__ dout,
                                                               const float* __restrict__ lse, T* __restrict__ dqkv,
                                                               const AttnCtx c) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int LD = DH + EV;
  constexpr int NP = KT * 32;
  constexpr int VPRW = DH / EV;
  __shared__ __attribute__((aligned(16))) T Qs[NP * LD];
  __shared__ __attribute__((aligned(16))) T Gs[NP * LD];
  __shared__ float tbl[1331];
  __shared__ int qrel[NP];
  __shared__ float Ls[NP], Dl[NP];
  const int heads = c.g.heads;
  const int grp = blockIdx.x / heads, head = blockIdx.x % heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int colq = head * DH, colk = c.dim + head * DH, colv = 2 * c.dim + head * DH;
  stage_rows<T, DH, KT>(qkv, c.ldq, colq, c, grp, Qs);
  // dO rows + delta_i = sum_d dO*O (the DH/EV consecutive threads of a row reduce with shuffles)
  for (int v = threadIdx.x; v < NP * VPRW; v += 256) {
    const int row = v / VPRW, part = v - row * VPRW;
    u32x4 val = {0u, 0u, 0u, 0u};
    float pd = 0.f;
    if (row < c.ntok) {
      const int64_t grow = attn_row(c, grp, row);
      val = *reinterpret_cast<const u32x4*>(dout + grow * c.dim + head * DH + part * EV);
      const u32x4 ov = *reinterpret_cast<const u32x4*>(out + grow * c.dim + head * DH + part * EV);
      const T* pg = reinterpret_cast<const T*>(&val);
      const T* po = reinterpret_cast<const T*>(&ov);
#pragma unroll
      for (int j = 0; j < EV; ++j) pd = fmaf((float)pg[j], (float)po[j], pd);
    }
    *reinterpret_cast<u32x4*>(&Gs[row * LD + part * EV]) = val;
    pd = group_sum(pd, VPRW);
    if (part == 0) {
      Dl[row] = pd;
      Ls[row] = (row < c.ntok) ? lse[((size_t)grp * heads + head) * c.ntok + row] : MF_BIG;
    }
  }
  const bool has_bias = bias_table != nullptr;
  if (has_bias) {
    const int ntbl = c.relm * c.relm * c.relm;
    for (int i = threadIdx.x; i < ntbl; i += 256) tbl[i] = bias_table[(size_t)i * heads + head];
    for (int i = threadIdx.x; i < NP; i += 256) qrel[i] = relcode(c, i < c.ntok ? i : 0) + c.reloff;
  }
  __syncthreads();

  const int ktiles = (c.ntok + 31) / 32;
  for (int kt = wave + 4 * blockIdx.y; kt < ktiles; kt += 4 * gridDim.y) {  // blockIdx.y: split of the tile loop
    const int key = kt * 32 + r;
    const bool kok = key < c.ntok;
    typename Mma<T>::Frag fk[DH / 16], fv[DH / 16];
    load_row_frags<T, DH>(qkv, c.ldq, colk, c, grp, key, kok, h, fk);
    load_row_frags<T, DH>(qkv, c.ldq, colv, c, grp, key, kok, h, fv);
    const int relk = has_bias ? relcode(c, kok ? key : 0) : 0;
    f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }

    for (int qt = 0; qt < ktiles; ++qt) {
      f32x16 s, dp;  // rows = queries of tile qt, column = this lane's key
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < DH / 16; ++ks) {
        Mma<T>::mma(Mma<T>::load(&Qs[(qt * 32 + r) * LD + ks * 16 + 8 * h]), fk[ks], s);
        Mma<T>::mma(Mma<T>::load(&Gs[(qt * 32 + r) * LD + ks * 16 + 8 * h]), fv[ks], dp);
      }
      float pv[16], ds[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qi = qt * 32 + acc_row(e, h);
        float v = s[e] * c.g.scale;
        if (has_bias) v += tbl[qrel[qi] - relk];
        const float p = __expf(v - Ls[qi]);  // Ls = +BIG for padded queries -> 0
        pv[e] = p;
        ds[e] = p * (dp[e] - Dl[qi]);
      }
      if (DROP) {  // dV takes the dropped, rescaled probabilities; dS the rescaled gradient of the kept ones
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int q0 = qt * 32 + 8 * g4 + 4 * h + 2 * half;  // even: queries q0, q0 + 1 share one Philox call
            const uint32_t keep = attn_keep_qpair(c.drop, blockIdx.x, q0 >> 1, key);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int e = 4 * g4 + 2 * half + j;
              const bool k1 = (keep >> j) & 1u;
              const float p = pv[e];
              ds[e] = p * ((k1 ? dp[e] * c.drop.scale : 0.f) - Dl[q0 + j]);
              pv[e] = k1 ? p * c.drop.scale : 0.f;
            }
          }
      }
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = pv[8 * sh + j]; b[j] = ds[8 * sh + j]; }
        const typename Mma<T>::Frag fp = FragOps<T>::make(a), fd = FragOps<T>::make(b);
        const int qbase = qt * 32 + 16 * sh + 4 * h;
#pragma unroll
        for (int dt = 0; dt < DH / 32; ++dt) {
          Mma<T>::mma(Mma<T>::gather_perm(&Gs[qbase * LD + dt * 32 + r], LD), fp, dv[dt]);
          Mma<T>::mma(Mma<T>::gather_perm(&Qs[qbase * LD + dt * 32 + r], LD), fd, dk[dt]);
        }
      }
    }
    if (kok) {
      T* grow = dqkv + attn_row(c, grp, key) * c.ldq;
#pragma unroll
      for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          grow[colk + dt * 32 + acc_row(e, h)] = (T)(dk[dt][e] * c.g.scale);
          grow[colv + dt * 32 + acc_row(e, h)] = (T)dv[dt][e];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// grid: x = (group, head); y = split of the 32-row tile loop.  The ViT trunk has 2 x 12 (group, head) pairs with 14
// tiles each: without the split 24 workgroups serve 256 CUs.  Every workgroup stages the whole K/V (or Q/dO) of its
// pair; its waves then take tiles wave + 4 y, + 4 gridDim.y, ...; outputs are rows of disjoint tiles.
static dim3 attn_grid(const AttnCtx& c) {
  const int pairs = c.groups * c.g.heads;
  const int ktiles = (c.ntok + 31) / 32;
  int ys = 1;
  if (pairs < 256) {
    ys = (512 + pairs - 1) / pairs;
    const int maxy = (ktiles + 3) / 4;
    if (ys > maxy) ys = maxy;
    if (ys < 1) ys = 1;
  }
  return dim3(pairs, ys);
}

template <typename T, int DH, int KT>
static int launch_fwd(const void* qkv, const float* bias, void* out, float* lse, const AttnCtx& c, hipStream_t s) {
  if (c.drop.thr16)
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<T, DH, KT, true>), attn_grid(c), dim3(256), 0, s, (const T*)qkv, bias,
                       (T*)out, lse, c);
  else
    hipLaunchKernelGGL((attn_mfma_fwd_kernel<T, DH, KT, false>), attn_grid(c), dim3(256), 0, s, (const T*)qkv, bias,
                       (T*)out, lse, c);
  return ctu_check_launch("attn_mfma_fwd");
}
template <typename T, int DH, int KT>
static int launch_bwd(const void* qkv, const float* bias, const void* out, const void* dout, const float* lse, void* dqkv,
                      float* dbias, const AttnCtx& c, hipStream_t s) {
  const dim3 grid = attn_grid(c);
  if (c.drop.thr16) {
    hipLaunchKernelGGL((attn_mfma_bwd_q_kernel<T, DH, KT, true>), grid, dim3(256), 0, s, (const T*)qkv, bias, (const T*)out,
                       (const T*)dout, lse, (T*)dqkv, dbias, c);
    hipLaunchKernelGGL((attn_mfma_bwd_kv_kernel<T, DH, KT, true>), grid, dim3(256), 0, s, (const T*)qkv, bias, (const T*)out,
                       (const T*)dout, lse, (T*)dqkv, c);
  } else {
    hipLaunchKernelGGL((attn_mfma_bwd_q_kernel<T, DH, KT, false>), grid, dim3(256), 0, s, (const T*)qkv, bias, (const T*)out,
                       (const T*)dout, lse, (T*)dqkv, dbias, c);
    hipLaunchKernelGGL((attn_mfma_bwd_kv_kernel<T, DH, KT, false>), grid, dim3(256), 0, s, (const T*)qkv, bias, (const T*)out,
                       (const T*)dout, lse, (T*)dqkv, c);
  }
  return ctu_check_launch("attn_mfma_bwd");
}

// which instantiation fits: KT = 7 (<= 224 tokens) or 14 (<= 448); f32 with 448 tokens x dh 64 exceeds the 160 KiB LDS
static int pick_kt(ctu_dtype dtype, const AttnCtx& c) {
  if (c.relm * c.relm * c.relm > 1331) return -1;
  if (c.ntok <= 224) return 7;
  if (c.ntok <= 448 && dtype == CTU_BF16) return 14;
  return -1;
}

#define MF_DISPATCH(LAUNCH, ...)                                                              \
  do {                                                                                        \
    const int kt = pick_kt(dtype, c);                                                         \
    if (kt < 0) return -1;                                                                    \
    const int dh = c.g.dh;                                                                    \
    if (dtype == CTU_BF16) {                                                                  \
      if (kt == 7 && dh == 32) return LAUNCH<bf16, 32, 7>(__VA_ARGS__);                       \
      if (kt == 7 && dh == 64) return LAUNCH<bf16, 64, 7>(__VA_ARGS__);                       \
      if (kt == 14 && dh == 32) return LAUNCH<bf16, 32, 14>(__VA_ARGS__);                     \
      if (kt == 14 && dh == 64) return LAUNCH<bf16, 64, 14>(__VA_ARGS__);                     \
    } else if (dtype == CTU_F32) {                                                            \
      if (kt == 7 && dh == 32) return LAUNCH<float, 32, 7>(__VA_ARGS__);                      \
      if (kt == 7 && dh == 64) return LAUNCH<float, 64, 7>(__VA_ARGS__);                      \
    }                                                                                         \
    return -1;                                                                                \
  } while (0)

int attn_mfma_fwd(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse, const AttnCtx& c,
                  hipStream_t s) {
  MF_DISPATCH(launch_fwd, qkv, bias_table, out, lse, c, s);
}
int attn_mfma_bwd(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out, const void* dout,
                  const float* lse, void* dqkv, float* dbias, const AttnCtx& c, hipStream_t s) {
  MF_DISPATCH(launch_bwd, qkv, bias_table, out, dout, lse, dqkv, dbias, c, s);
}
