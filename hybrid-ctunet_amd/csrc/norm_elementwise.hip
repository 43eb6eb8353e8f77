// HBM-bound kernels: InstanceNorm3d(+residual)(+LeakyReLU), LayerNorm, GELU, adds, layout permutes, casts,
// and the binary cross-weight fusion core.  All access is 8 elements per lane (16 B bf16 / 2x16 B f32),
// channels-last, coalesced along C.  Reductions: wavefront shuffles -> LDS -> one fp32 atomic per block.
#include "common.h"

// =========================================================================================================
// InstanceNorm3d  (resnet.py:97-124,156-157,198; hybrid_CTUNet.py:84-104): x [B][S][C]
// =========================================================================================================
// stats[b][c] = (sum, sumsq).  grid (chunks, B); block 256 = (C/8 column groups) x (256/(C/8) row lanes)
template <typename T>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ x, double* __restrict__ stats,
                                                       const int64_t S, const int C, const int64_t rows_per_block) {
  __shared__ double red[256 * 16];
  const int ncg = C >> 3;                 // column groups (<= 256, checked on host)
  const int tid = threadIdx.x;
  const int cg = tid % ncg, rl = tid / ncg;
  const int rlanes = 256 / ncg;
  const int b = blockIdx.y;
  const int64_t s_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t s_end = min(S, s_begin + rows_per_block);
  // Shifted sums: accumulate (x - K), (x - K)^2 with K = the channel's first voxel.  Conv outputs fed by
  // post-activation tensors can have |mean| >> std; E[x^2] - E[x]^2 would then cancel catastrophically in fp32.
  // fp64 accumulators: this net amplifies rounding noise in the statistics ~1000x (DESIGN.md "Numerics"); the kernel
  // is HBM-bound, so the wider adds are free.
  double s1[8], s2[8];
  float shift[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.0; s2[e] = 0.0; shift[e] = 0.f; }
  if (rl < rlanes) {
    const T* base = x + ((size_t)b * S) * C + cg * 8;
    load8(base, shift);
    for (int64_t s = s_begin + rl; s < s_end; s += rlanes) {
      float v[8];
      load8(base + (size_t)s * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const double d = (double)(v[e] - shift[e]); s1[e] += d; s2[e] += d * d; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  // thread t < C*2 reduces one (channel, which) over the row lanes
  for (int o = tid; o < C * 2; o += 256) {
    const int c = o >> 1, which = o & 1;
    const int g = c >> 3, e = c & 7;
    double acc = 0.0;
    for (int r = 0; r < rlanes; ++r) acc += red[(r * ncg + g) * 16 + which * 8 + e];
    atomicAdd(&stats[((size_t)b * C + c) * 2 + which], acc);
  }
}

// (sum(x-K), sum(x-K)^2) in fp64 -> (mean, rstd) in fp32
// The accumulators are handed back zeroed, so one persistent workspace serves every call without a memset launch.
template <typename T>
__global__ __launch_bounds__(256) void in_finalize_kernel(const T* __restrict__ x, double* __restrict__ acc,
                                                          float* __restrict__ stats, const int B, const int64_t S,
                                                          const int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  const double inv_s = 1.0 / (double)S;
  const double k = (double)(float)x[((size_t)b * S) * C + c];
  const double m = acc[(size_t)i * 2] * inv_s;
  const double var = fmax(acc[(size_t)i * 2 + 1] * inv_s - m * m, 0.0);
  acc[(size_t)i * 2] = 0.0;
  acc[(size_t)i * 2 + 1] = 0.0;
  stats[(size_t)i * 2] = (float)(k + m);
  stats[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)NORM_EPS));
}

// same from UNSHIFTED sums (sum x, sum x^2) accumulated by a producer's epilogue (ctu_conv3_halo in_acc); bf16 path only
__global__ __launch_bounds__(256) void in_finalize_raw_kernel(double* __restrict__ acc, float* __restrict__ stats,
                                                              const int n, const double inv_s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double m = acc[(size_t)i * 2] * inv_s;
  const double var = fmax(acc[(size_t)i * 2 + 1] * inv_s - m * m, 0.0);
  acc[(size_t)i * 2] = 0.0;
  acc[(size_t)i * 2 + 1] = 0.0;
  stats[(size_t)i * 2] = (float)m;
  stats[(size_t)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)NORM_EPS));
}

__device__ __forceinline__ void in_mean_rstd(const float* stats, int b, int C, int c0, float /*inv_s*/, float (&mean)[8],
                                             float (&rstd)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = stats[((size_t)b * C + c0 + e) * 2];
    rstd[e] = stats[((size_t)b * C + c0 + e) * 2 + 1];
  }
}

// y = act((x - mean) * rstd + residual).  grid (gx, B) with gx * 256 a multiple of C/8: a thread keeps ONE 8-channel
// column group of ONE batch item, so mean / rstd are loaded once and the loop is a pure stream.
// raw (optional): UNSHIFTED fp64 sums (sum x, sum x^2) [B][C][2] left by the producer's epilogue: (mean, rstd) are then
// derived here - every thread for its own eight channels - and written to `stats` by the first workgroup of each batch item
// (the separate finalize launch disappears); `clear` (optional, another accumulator no launch still reads) is zeroed.
template <typename T>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* __restrict__ x, float* __restrict__ stats,
                                                       const T* __restrict__ res, T* __restrict__ y, const int64_t S,
                                                       const int C, const int act, const int64_t yb16,
                                                       uint8_t* __restrict__ mask, const double* __restrict__ raw = nullptr,
                                                       double* __restrict__ clear = nullptr, const int clear_n = 0) {
  const int ncg = C >> 3;
  const int b = blockIdx.y;
  const int64_t nvec = S * ncg;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (clear && b == 0)
    for (int64_t i = i0; i < clear_n; i += stride) clear[i] = 0.0;
  const int cg = (int)(i0 % ncg);
  float mean[8], rstd[8];
  if (raw) {
    // one fp64 division and square root per CHANNEL and workgroup (thread t: channels t, t + 256, ...), handed round through
    // LDS - every thread deriving its own eight channels cost 3 us per launch over 8 192 workgroups
    // (dynamic LDS, 8 bytes per channel: a kilobyte at 128 channels still fits beside a pair of halo-convolution workgroups)
    extern __shared__ float in_apply_lds[];
    float* sm = in_apply_lds;
    float* sr = in_apply_lds + C;
    const double inv_s = 1.0 / (double)S;
    for (int c = threadIdx.x; c < C; c += 256) {
      const size_t o = ((size_t)b * C + c) * 2;
      const double m = raw[o] * inv_s;
      const double var = fmax(raw[o + 1] * inv_s - m * m, 0.0);
      const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)NORM_EPS));
      sm[c] = mf;
      sr[c] = rf;
      if (blockIdx.x == 0) {
        stats[o] = mf;
        stats[o + 1] = rf;
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = sm[cg * 8 + e];
      rstd[e] = sr[cg * 8 + e];
    }
  } else {
    in_mean_rstd(stats, b, C, cg * 8, 0.f, mean, rstd);
  }
  const size_t base = (size_t)b * S * C;
  x += base;
  // yb16 > 0 (= voxels of the whole batch): y in CTU_LAYOUT_B16, element (m, c) at ((c >> 4) * yb16 + m) * 16 + (c & 15)
  T* yblk = y + ((size_t)(cg >> 1) * yb16 + (size_t)b * S) * 16 + (cg & 1) * 8;
  y += base;
  if (res) res += base;
  if (mask) mask += (size_t)b * nvec;   // one byte per 8-channel vector: bit e = (pre-activation value of channel e > 0)
  for (int64_t i = i0; i < nvec; i += stride) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[e]) * rstd[e];
    if (res) {
      float r[8];
      load8(res + i * 8, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (mask) {
      unsigned bits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) bits |= (v[e] > 0.f ? 1u : 0u) << e;
      mask[i] = (uint8_t)bits;
    }
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * LRELU_SLOPE;
    }
    if (yb16) store8(yblk + (i / ncg) * 16, v);
    else store8(y + i * 8, v);
  }
}

// y = act((x - mean) * rstd + (x2 - mean2) * rstd2): the last norm of a block whose shortcut is conv + norm (resnet.py:122-124 with
// downsample :196-199; hybrid_CTUNet.py:99-104) applied together with the shortcut's norm - the normalised shortcut is never
// written and read back (two passes over the block's output tensor and one launch less per such block).  raw / raw2: UNSHIFTED fp64
// sums from the producers' epilogues (NULL: stats / stats2 are inputs, e.g. after a ctu_in_stats pass); (mean, rstd) of both norms
// are written for the backward pass; clear / clear2 name the accumulators of the PREVIOUS main / shortcut norm, zeroed here.
template <typename T>
__global__ __launch_bounds__(256) void in_apply_dual_kernel(const T* __restrict__ x, float* __restrict__ stats, const double* __restrict__ raw,
                                                            const T* __restrict__ x2, float* __restrict__ stats2,
                                                            const double* __restrict__ raw2, T* __restrict__ y, const int64_t S,
                                                            const int C, const int act, uint8_t* __restrict__ mask,
                                                            double* __restrict__ clear, const int clear_n,
                                                            double* __restrict__ clear2, const int clear2_n) {
  extern __shared__ float in_dual_lds[];   // mean[C] rstd[C] mean2[C] rstd2[C]
  const int ncg = C >> 3;
  const int b = blockIdx.y;
  const int64_t nvec = S * ncg;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (b == 0) {
    if (clear) for (int64_t i = i0; i < clear_n; i += stride) clear[i] = 0.0;
    if (clear2) for (int64_t i = i0; i < clear2_n; i += stride) clear2[i] = 0.0;
  }
  const double inv_s = 1.0 / (double)S;
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const bool second = c >= C;
    const int cc = second ? c - C : c;
    const double* rw = second ? raw2 : raw;
    float* st = second ? stats2 : stats;
    const size_t o = ((size_t)b * C + cc) * 2;
    float mf, rf;
    if (rw) {
      const double m = rw[o] * inv_s;
      const double var = fmax(rw[o + 1] * inv_s - m * m, 0.0);
      mf = (float)m;
      rf = (float)(1.0 / sqrt(var + (double)NORM_EPS));
      if (blockIdx.x == 0) {
        st[o] = mf;
        st[o + 1] = rf;
      }
    } else {
      mf = st[o];
      rf = st[o + 1];
    }
    in_dual_lds[(second ? 2 * C : 0) + cc] = mf;
    in_dual_lds[(second ? 3 * C : C) + cc] = rf;
  }
  __syncthreads();
  const int cg = (int)(i0 % ncg);
  float mean[8], rstd[8], mean2[8], rstd2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = in_dual_lds[cg * 8 + e];
    rstd[e] = in_dual_lds[C + cg * 8 + e];
    mean2[e] = in_dual_lds[2 * C + cg * 8 + e];
    rstd2[e] = in_dual_lds[3 * C + cg * 8 + e];
  }
  const size_t base = (size_t)b * S * C;
  x += base;
  x2 += base;
  y += base;
  if (mask) mask += (size_t)b * nvec;
  for (int64_t i = i0; i < nvec; i += stride) {
    float v[8], w[8];
    load8(x + i * 8, v);
    load8(x2 + i * 8, w);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // the shortcut's normalised value rounded to the activation type first: what the two-launch form stored and read back
      // and the sum as ONE fma: what hipcc makes of in_apply_kernel's "(x - mean) * rstd" followed by "+= residual" - bit for bit
      // what the two-launch form computes from the stored shortcut (tools/dualdbg.py)
      const float rsd = (float)(T)__fmul_rn(w[e] - mean2[e], rstd2[e]);
      v[e] = fmaf(v[e] - mean[e], rstd[e], rsd);
    }
    if (mask) {
      unsigned bits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) bits |= (v[e] > 0.f ? 1u : 0u) << e;
      mask[i] = (uint8_t)bits;
    }
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * LRELU_SLOPE;
    }
    store8(y + i * 8, v);
  }
}

// sums[b][c] = (sum g, sum g*xhat), g = dy * act'(y)
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ y, const float* __restrict__ stats,
                                                            double* __restrict__ sums, const int64_t S, const int C,
                                                            const int act, const int64_t rows_per_block,
                                                            const uint8_t* __restrict__ mask, const int fold) {
  // LDS: [256][16] doubles (32 KiB) in general; [4 waves][C / 8][16] when C / 8 is a power of two <= 64 (the lanes of a wave
  // that share a channel group are folded by shuffles first): 4 KiB at 64 channels, 8 at 128 - small enough to sit beside a
  // pair of halo-convolution workgroups (150 of the CU's 160 KiB), which the 32-KiB form can not
  extern __shared__ double red[];
  const int ncg = C >> 3;
  const int tid = threadIdx.x;
  const bool folded = fold != 0;
  const int cg = tid % ncg, rl = tid / ncg;
  const int rlanes = 256 / ncg;
  const int b = blockIdx.y;
  const int64_t s_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t s_end = min(S, s_begin + rows_per_block);
  double s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
  if (rl < rlanes) {
    float mean[8], rstd[8];
    in_mean_rstd(stats, b, C, cg * 8, 1.0f / (float)S, mean, rstd);
    const size_t base = ((size_t)b * S) * C + cg * 8;
    for (int64_t s = s_begin + rl; s < s_end; s += rlanes) {
      float g[8], xv[8];
      const size_t off = base + (size_t)s * C;
      load8(dy + off, g);
      load8(x + off, xv);
      float xh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - mean[e]) * rstd[e];
      if (act) {
        if (mask) {  // residual case with the forward's sign bits: one byte instead of a 16-byte vector of y
          const unsigned bits = mask[((size_t)b * S + s) * ncg + cg];
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : g[e] * LRELU_SLOPE;
        } else if (y) {  // residual case: the activation saw xhat + residual, whose sign only y records
          float yv[8];
          load8(y + off, yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = yv[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
        } else {  // no residual: sign(y) == sign(xhat), no third input stream
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = xh[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += (double)g[e];
        s2[e] += (double)g[e] * (double)xh[e];
      }
    }
  }
  if (folded) {
    for (int d = ncg; d < 64; d <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], d, 64);
        s2[e] += __shfl_xor(s2[e], d, 64);
      }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < ncg) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(wave * ncg + lane) * 16 + e] = s1[e]; red[(wave * ncg + lane) * 16 + 8 + e] = s2[e]; }
    }
    __syncthreads();
    for (int o = tid; o < C * 2; o += 256) {
      const int c = o >> 1, which = o & 1;
      const int g = c >> 3, e = c & 7;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += red[(w * ncg + g) * 16 + which * 8 + e];
      atomicAdd(&sums[((size_t)b * C + c) * 2 + which], acc);
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  for (int o = tid; o < C * 2; o += 256) {
    const int c = o >> 1, which = o & 1;
    const int g = c >> 3, e = c & 7;
    double acc = 0.0;
    for (int r = 0; r < rlanes; ++r) acc += red[(r * ncg + g) * 16 + which * 8 + e];
    atomicAdd(&sums[((size_t)b * C + c) * 2 + which], acc);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)); same thread -> column-group mapping as in_apply_kernel.
// `clear` (optional) is another sums buffer that no launch still reads - the previous call's: it is zeroed here so the
// two buffers can alternate between calls without a memset launch.
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ stats,
                                                           const double* __restrict__ sums, T* __restrict__ dx,
                                                           T* __restrict__ dres, const int64_t S, const int C,
                                                           const int act, double* __restrict__ clear, const int clear_n,
                                                           const int64_t dxb16, const uint8_t* __restrict__ mask) {
  const int ncg = C >> 3;
  const int b = blockIdx.y;
  const int64_t nvec = S * ncg;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (clear && b == 0)
    for (int64_t i = i0; i < clear_n; i += stride) clear[i] = 0.0;
  const int cg = (int)(i0 % ncg);
  float mean[8], rstd[8], m1[8], m2[8];
  in_mean_rstd(stats, b, C, cg * 8, 0.f, mean, rstd);
  const double inv_s = 1.0 / (double)S;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    m1[e] = (float)(sums[((size_t)b * C + cg * 8 + e) * 2] * inv_s);
    m2[e] = (float)(sums[((size_t)b * C + cg * 8 + e) * 2 + 1] * inv_s);
  }
  const size_t base = (size_t)b * S * C;
  dy += base;
  x += base;
  T* dxblk = dx + ((size_t)(cg >> 1) * dxb16 + (size_t)b * S) * 16 + (cg & 1) * 8;  // CTU_LAYOUT_B16 destination (dxb16 > 0)
  dx += base;
  if (y) y += base;
  if (dres) dres += base;
  if (mask) mask += (size_t)b * nvec;
  for (int64_t i = i0; i < nvec; i += stride) {
    float g[8], xv[8], xh[8];
    load8(dy + i * 8, g);
    load8(x + i * 8, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - mean[e]) * rstd[e];
    if (act) {
      if (mask) {
        const unsigned bits = mask[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : g[e] * LRELU_SLOPE;
      } else if (y) {
        float yv[8];
        load8(y + i * 8, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = yv[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = xh[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
      }
    }
    if (dres) store8(dres + i * 8, g);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rstd[e] * (g[e] - m1[e] - xh[e] * m2[e]);
    if (dxb16) store8(dxblk + (i / ncg) * 16, o);
    else store8(dx + i * 8, o);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Reduce AND apply in one launch, for tensors that are launch-bound rather than byte-bound (<= 32 MB: 104 of the model's 124
// norms; their two launches cost 16 - 30 us each in the step however few bytes they move).  The grid is small enough to be
// resident as a whole (<= 256 workgroups of <= 128 registers: four fit a CU, so up to four such launches on different streams
// can wait at the same time without starving each other of the slots their last workgroups need).  A workgroup sums its rows,
// adds them into sums[b] (fp64 atomics, as the reduce kernel), arrives at a counter of its batch item, waits until the
// item's other workgroups have arrived, reads the sums back past the non-coherent L2 (agent-scope loads) and applies to the
// SAME rows - they are still in its L1 / its XCD's L2.  sync[2 b] counts arrivals, sync[2 b + 1] departures; the last
// workgroup to depart zeroes both, so one workspace serves every launch of a stream.  A wait that lasts 50 ms gives up
// (counted in g_sync_timeouts: a wrong result the tests see, never a hung GPU).
// ---------------------------------------------------------------------------------------------------------
__device__ unsigned g_sync_timeouts;

template <typename T>
__global__ __launch_bounds__(256, 4) void in_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ stats,
                                                           double* __restrict__ sums, T* __restrict__ dx, T* __restrict__ dres,
                                                           const int64_t S, const int C, const int act,
                                                           const int64_t rows_per_block, const uint8_t* __restrict__ mask,
                                                           const int fold, double* __restrict__ clear, const int clear_n,
                                                           const int64_t dxb16, unsigned* __restrict__ sync) {
  extern __shared__ double red[];
  const int ncg = C >> 3;
  const int tid = threadIdx.x;
  const int cg = tid % ncg, rl = tid / ncg;
  const int rlanes = 256 / ncg;
  const int b = blockIdx.y;
  const int64_t s_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t s_end = min(S, s_begin + rows_per_block);
  if (clear && b == 0)
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < clear_n; i += (int64_t)gridDim.x * 256) clear[i] = 0.0;
  const bool active = rl < rlanes;
  float mean[8], rstd[8];
  if (active) in_mean_rstd(stats, b, C, cg * 8, 0.f, mean, rstd);
  const size_t base = ((size_t)b * S) * C + cg * 8;
  // g = dy * act'(.), xhat of one (row, column group)
  auto fetch = [&](int64_t s, float (&g)[8], float (&xh)[8]) {
    const size_t off = base + (size_t)s * C;
    float xv[8];
    load8(dy + off, g);
    load8(x + off, xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - mean[e]) * rstd[e];
    if (act) {
      if (mask) {
        const unsigned bits = mask[((size_t)b * S + s) * ncg + cg];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : g[e] * LRELU_SLOPE;
      } else if (y) {
        float yv[8];
        load8(y + off, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = yv[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = xh[e] > 0.f ? g[e] : g[e] * LRELU_SLOPE;
      }
    }
  };
  double s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
  if (active) {
    for (int64_t s = s_begin + rl; s < s_end; s += 2 * rlanes) {   // two rows in flight per thread
      float g0[8], h0[8], g1[8], h1[8];
      const bool two = s + rlanes < s_end;
      fetch(s, g0, h0);
      if (two) fetch(s + rlanes, g1, h1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += (double)g0[e];
        s2[e] += (double)g0[e] * (double)h0[e];
      }
      if (two) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += (double)g1[e];
          s2[e] += (double)g1[e] * (double)h1[e];
        }
      }
    }
  }
  if (fold) {
    for (int d = ncg; d < 64; d <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], d, 64);
        s2[e] += __shfl_xor(s2[e], d, 64);
      }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < ncg) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(wave * ncg + lane) * 16 + e] = s1[e]; red[(wave * ncg + lane) * 16 + 8 + e] = s2[e]; }
    }
    __syncthreads();
    for (int o = tid; o < C * 2; o += 256) {
      const int c = o >> 1, which = o & 1;
      const int g = c >> 3, e = c & 7;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += red[(w * ncg + g) * 16 + which * 8 + e];
      const double old = atomicAdd(&sums[((size_t)b * C + c) * 2 + which], acc);   // returning form: back = performed at the memory side
      asm volatile("" ::"v"(old));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    for (int o = tid; o < C * 2; o += 256) {
      const int c = o >> 1, which = o & 1;
      const int g = c >> 3, e = c & 7;
      double acc = 0.0;
      for (int r = 0; r < rlanes; ++r) acc += red[(r * ncg + g) * 16 + which * 8 + e];
      const double old = atomicAdd(&sums[((size_t)b * C + c) * 2 + which], acc);
      asm volatile("" ::"v"(old));
    }
  }
  // ---- the item's workgroups meet.  No fences (an agent-scope release writes the XCD's whole L2 back, an acquire invalidates it:
  // 25 us per launch, measured): the sums are only ever touched by memory-side atomics and agent-scope (L2-bypassing) loads, the
  // returning adds above are back - i.e. performed - before the barrier, and the counter is a memory-side atomic polled past L2
  __syncthreads();
  if (tid == 0) {
    const unsigned n = gridDim.x;
    __hip_atomic_fetch_add(&sync[2 * b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(&sync[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) {   // 50 ms at 100 MHz
        atomicAdd(&g_sync_timeouts, 1u);
        break;
      }
    }
    const unsigned gone = __hip_atomic_fetch_add(&sync[2 * b + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == n - 1) {   // everybody is past the wait: hand the counters back zeroed
      __hip_atomic_store(&sync[2 * b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[2 * b + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!active) return;
  // ---- apply to the same rows
  float m1[8], m2[8];
  const double inv_s = 1.0 / (double)S;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const double* sp = &sums[((size_t)b * C + cg * 8 + e) * 2];
    m1[e] = (float)(__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * inv_s);
    m2[e] = (float)(__hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * inv_s);
  }
  T* dxblk = dx + ((size_t)(cg >> 1) * dxb16 + (size_t)b * S) * 16 + (cg & 1) * 8;  // CTU_LAYOUT_B16 destination (dxb16 > 0)
  auto put = [&](int64_t s, const float (&g)[8], const float (&xh)[8]) {
    const size_t off = base + (size_t)s * C;
    if (dres) store8(dres + off, g);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rstd[e] * (g[e] - m1[e] - xh[e] * m2[e]);
    if (dxb16) store8(dxblk + (size_t)s * 16, o);
    else store8(dx + off, o);
  };
  for (int64_t s = s_begin + rl; s < s_end; s += 2 * rlanes) {
    float g0[8], h0[8], g1[8], h1[8];
    const bool two = s + rlanes < s_end;
    fetch(s, g0, h0);
    if (two) fetch(s + rlanes, g1, h1);
    put(s, g0, h0);
    if (two) put(s + rlanes, g1, h1);
  }
}

// grid.x for the streaming kernels: ~8192 workgroups in total, a multiple of C/8 so a thread's column group is fixed
static unsigned in_stream_grid(int64_t S, int C, int B) {
  const int ncg = C / 8;
  // at least four vectors per thread: its per-channel constants (mean, rstd, the two gradient means) cost as much as a
  // vector's arithmetic (2 x 48 x 48 x 96 x 32 ch backward apply: 35 us with one vector per thread against 16 us of HBM time)
  int64_t g = (S * ncg + 1023) / 1024;
  // 1 024 workgroups in all (four per CU), each walking its column group down the rows: alone as fast as the 8 192 of rounds 1-3
  // (80.7 / 129.8 us against 81.9 / 132.0 at 64 ch @ 96^3), and in the overlapped step the longer-lived workgroups keep their CU slots
  // against the other stream's halo workgroups: 45.70 / 45.71 against 45.91 / 45.89 ms per step.  2 048: 45.75; 512: 46.0 (the
  // forward apply then loses a quarter of its bandwidth alone).  profiles/r04_experiment_in_grid.log
  const int total = (ctu_option_route() & CTU_ROUTE_IN_GRID_8192) ? 8192 : 1024;
  const int64_t cap = total / (B > 0 ? B : 1) > 0 ? total / (B > 0 ? B : 1) : 1;
  if (g > cap) g = cap;
  g = ((g + ncg - 1) / ncg) * ncg;
  return (unsigned)g;
}

static int check_in(const void* x, int B, int64_t S, int C) {
  CTU_REQUIRE(x, "null pointer");
  CTU_REQUIRE(B > 0 && S > 0, "bad B/S");
  CTU_REQUIRE(C >= 8 && C % 8 == 0 && C <= 2048, "InstanceNorm needs C %% 8 == 0 and C <= 2048 (C=%d)", C);
  return CTU_OK;
}
static int64_t in_rows_per_block(int64_t S, int B, int C = 2048) {
  // aim for ~2048 blocks in total, at least 64 rows each; fewer for few channels: every block ends with one fp64 atomic
  // per (channel, sum), and with 32 channels those land on eight cache lines - 1024 blocks per batch item made the
  // 32-channel reduce 4.7x slower than its HBM time (51.6 us against 11)
  // (measured at 2 x 48 x 48 x 96: 32 ch 51.6 -> 16.8 us with 512 blocks, 128 ch 72 -> 48 us with 1024; 64 ch @ 96^3 wants
  //  1024: 87 us against 103 with 2048 and 106 with 512)
  const int64_t total = C <= 32 ? 512 : (C <= 128 ? 1024 : 2048);
  int64_t chunks = total / (B > 0 ? B : 1);
  if (chunks < 1) chunks = 1;
  int64_t rows = (S + chunks - 1) / chunks;
  // small volumes with many channels (12 x 12 x 24 x 512, 6 x 6 x 12 x 1024): four row lanes per workgroup walk 64 rows one after
  // the other - the pass is as long as that chain (1024 ch @ 6 x 6 x 12: 21.0 us, 9.2 with 16 rows per workgroup).  Elsewhere more
  // workgroups only add atomics per address (256 ch @ 24 x 24 x 48: 23 -> 47 us); profiles/r03_kb_inorm_small_rows_floor.log
  const int64_t floor_rows = (S <= 4096 && C >= 256 && !(ctu_option_route() & CTU_ROUTE_IN_ROWS64)) ? 16 : 64;
  if (rows < floor_rows) rows = floor_rows;
  return rows;
}

extern "C" int ctu_in_stats(ctu_dtype dtype, const void* x, int32_t B, int64_t S, int32_t C, double* acc, float* stats,
                            ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(stats && acc, "null stats/acc");
  const int64_t rows = in_rows_per_block(S, B, C);
  dim3 grid((unsigned)((S + rows - 1) / rows), B);
  hipStream_t s = (hipStream_t)stream;
  const dim3 fgrid((B * C + 255) / 256);
  CTU_DISPATCH(dtype,
               {
                 hipLaunchKernelGGL(in_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, acc, S, C, rows);
                 hipLaunchKernelGGL(in_finalize_kernel<float>, fgrid, dim3(256), 0, s, (const float*)x, acc, stats, B, S, C);
               },
               {
                 hipLaunchKernelGGL(in_stats_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, acc, S, C, rows);
                 hipLaunchKernelGGL(in_finalize_kernel<bf16>, fgrid, dim3(256), 0, s, (const bf16*)x, acc, stats, B, S, C);
               });
  return ctu_check_launch("in_stats");
}

extern "C" int ctu_in_finalize(int32_t B, int64_t S, int32_t C, double* acc, float* stats, ctu_stream_t stream) {
  CTU_REQUIRE(B > 0 && S > 0 && C > 0 && acc && stats, "in_finalize: bad args");
  hipLaunchKernelGGL(in_finalize_raw_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc, stats,
                     B * C, 1.0 / (double)S);
  return ctu_check_launch("in_finalize");
}

static int in_apply_impl(ctu_dtype dtype, const void* x, const double* raw, float* stats, const void* residual, void* y,
                         int32_t B, int64_t S, int32_t C, int32_t act, int32_t y_layout, uint8_t* sign_mask,
                         double* clear_ws, int32_t clear_n, ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(stats && y, "null pointer");
  CTU_REQUIRE(y_layout == CTU_LAYOUT_NDHWC || (y_layout == CTU_LAYOUT_B16 && C % 16 == 0 && y != x), "in_apply: bad output layout");
  CTU_REQUIRE(clear_n >= 0 && (clear_n == 0 || clear_ws) && (clear_ws == nullptr || clear_ws != raw), "in_apply: bad clear workspace");
  const int64_t yb16 = y_layout == CTU_LAYOUT_B16 ? (int64_t)B * S : 0;
  const dim3 grid(in_stream_grid(S, C, B), B);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = raw ? (size_t)2 * C * sizeof(float) : 0;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(in_apply_kernel<float>, grid, dim3(256), lds, s, (const float*)x, stats,
                                  (const float*)residual, (float*)y, S, C, act, yb16, sign_mask, raw, clear_ws, clear_n),
               hipLaunchKernelGGL(in_apply_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)x, stats,
                                  (const bf16*)residual, (bf16*)y, S, C, act, yb16, sign_mask, raw, clear_ws, clear_n));
  return ctu_check_launch("in_apply");
}
extern "C" int ctu_in_apply(ctu_dtype dtype, const void* x, const float* stats, const void* residual, void* y,
                            int32_t B, int64_t S, int32_t C, int32_t act, int32_t y_layout, uint8_t* sign_mask,
                            ctu_stream_t stream) {
  return in_apply_impl(dtype, x, nullptr, const_cast<float*>(stats), residual, y, B, S, C, act, y_layout, sign_mask, nullptr, 0,
                       stream);
}
extern "C" int ctu_in_apply_acc(ctu_dtype dtype, const void* x, const double* raw_acc, float* stats, const void* residual,
                                void* y, int32_t B, int64_t S, int32_t C, int32_t act, int32_t y_layout,
                                uint8_t* sign_mask, double* clear_ws, int32_t clear_n, ctu_stream_t stream) {
  return in_apply_impl(dtype, x, raw_acc, stats, residual, y, B, S, C, act, y_layout, sign_mask, clear_ws, clear_n, stream);
}

extern "C" int ctu_in_apply_dual(ctu_dtype dtype, const void* x, const double* raw_acc, float* stats, const void* x2,
                                 const double* raw_acc2, float* stats2, void* y, int32_t B, int64_t S, int32_t C, int32_t act,
                                 uint8_t* sign_mask, double* clear_ws, int32_t clear_n, double* clear_ws2, int32_t clear_n2,
                                 ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(x2 && stats && stats2 && y, "in_apply_dual: null pointer");
  CTU_REQUIRE(clear_n >= 0 && (clear_n == 0 || clear_ws) && (clear_ws == nullptr || (clear_ws != raw_acc && clear_ws != raw_acc2)),
              "in_apply_dual: bad clear workspace");
  CTU_REQUIRE(clear_n2 >= 0 && (clear_n2 == 0 || clear_ws2) && (clear_ws2 == nullptr || (clear_ws2 != raw_acc && clear_ws2 != raw_acc2)),
              "in_apply_dual: bad second clear workspace");
  const dim3 grid(in_stream_grid(S, C, B), B);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)4 * C * sizeof(float);
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(in_apply_dual_kernel<float>, grid, dim3(256), lds, s, (const float*)x, stats, raw_acc, (const float*)x2,
                                  stats2, raw_acc2, (float*)y, S, C, act, sign_mask, clear_ws, clear_n, clear_ws2, clear_n2),
               hipLaunchKernelGGL(in_apply_dual_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)x, stats, raw_acc, (const bf16*)x2,
                                  stats2, raw_acc2, (bf16*)y, S, C, act, sign_mask, clear_ws, clear_n, clear_ws2, clear_n2));
  return ctu_check_launch("in_apply_dual");
}

extern "C" int ctu_in_bwd_reduce(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats,
                                 double* sums, int32_t B, int64_t S, int32_t C, int32_t act, const uint8_t* sign_mask,
                                 ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(dy && stats && sums, "null pointer");  // y == NULL and no mask: no residual was added (sign taken from xhat)
  const int64_t rows = in_rows_per_block(S, B, C);
  dim3 grid((unsigned)((S + rows - 1) / rows), B);
  hipStream_t s = (hipStream_t)stream;
  const int ncg = C / 8;
  const int fold = ncg <= 64 && (ncg & (ncg - 1)) == 0 && !(ctu_option_route() & CTU_ROUTE_IN_REDUCE_LDS32);
  const size_t lds = fold ? (size_t)4 * ncg * 16 * sizeof(double) : (size_t)256 * 16 * sizeof(double);
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(in_bwd_reduce_kernel<float>, grid, dim3(256), lds, s, (const float*)dy, (const float*)x,
                                  (const float*)y, stats, sums, S, C, act, rows, sign_mask, fold),
               hipLaunchKernelGGL(in_bwd_reduce_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)dy, (const bf16*)x,
                                  (const bf16*)y, stats, sums, S, C, act, rows, sign_mask, fold));
  return ctu_check_launch("in_bwd_reduce");
}

extern "C" int ctu_in_bwd_apply(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats,
                                const double* sums, void* dx, void* dres, int32_t B, int64_t S, int32_t C, int32_t act,
                                double* clear_ws, int32_t clear_n, int32_t dx_layout, const uint8_t* sign_mask,
                                ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(dy && stats && sums && dx, "null pointer");
  CTU_REQUIRE(dx_layout == CTU_LAYOUT_NDHWC || (dx_layout == CTU_LAYOUT_B16 && C % 16 == 0 && dx != dy && dx != x),
              "in_bwd_apply: bad output layout");
  const int64_t dxb16 = dx_layout == CTU_LAYOUT_B16 ? (int64_t)B * S : 0;
  CTU_REQUIRE(clear_n >= 0 && (clear_n == 0 || clear_ws) && clear_ws != sums, "bad clear workspace");
  const dim3 grid(in_stream_grid(S, C, B), B);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(in_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x,
                                  (const float*)y, stats, sums, (float*)dx, (float*)dres, S, C, act, clear_ws, clear_n, dxb16, sign_mask),
               hipLaunchKernelGGL(in_bwd_apply_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x,
                                  (const bf16*)y, stats, sums, (bf16*)dx, (bf16*)dres, S, C, act, clear_ws, clear_n, dxb16, sign_mask));
  return ctu_check_launch("in_bwd_apply");
}

extern "C" int ctu_in_bwd_fused(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats,
                                double* sums, void* dx, void* dres, int32_t B, int64_t S, int32_t C, int32_t act,
                                double* clear_ws, int32_t clear_n, int32_t dx_layout, const uint8_t* sign_mask,
                                uint32_t* sync_ws, ctu_stream_t stream) {
  if (int rc = check_in(x, B, S, C)) return rc;
  CTU_REQUIRE(dy && stats && sums && dx && sync_ws, "in_bwd_fused: null pointer");
  CTU_REQUIRE(B <= CTU_IN_FUSED_MAX_WG, "in_bwd_fused: at most %d batch items (one workgroup each at least)", CTU_IN_FUSED_MAX_WG);
  CTU_REQUIRE(dx_layout == CTU_LAYOUT_NDHWC || (dx_layout == CTU_LAYOUT_B16 && C % 16 == 0 && dx != dy && dx != x),
              "in_bwd_fused: bad output layout");
  const int64_t dxb16 = dx_layout == CTU_LAYOUT_B16 ? (int64_t)B * S : 0;
  CTU_REQUIRE(clear_n >= 0 && (clear_n == 0 || clear_ws) && clear_ws != sums, "in_bwd_fused: bad clear workspace");
  // the whole grid has to be resident at once: <= CTU_IN_FUSED_MAX_WG workgroups, at least 16 rows each
  int64_t chunks = CTU_IN_FUSED_MAX_WG / B;
  int64_t rows = (S + chunks - 1) / chunks;
  if (rows < 16) rows = 16;
  chunks = (S + rows - 1) / rows;
  dim3 grid((unsigned)chunks, B);
  hipStream_t s = (hipStream_t)stream;
  const int ncg = C / 8;
  const int fold = ncg <= 64 && (ncg & (ncg - 1)) == 0;
  const size_t lds = fold ? (size_t)4 * ncg * 16 * sizeof(double) : (size_t)256 * 16 * sizeof(double);
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(in_bwd_fused_kernel<float>, grid, dim3(256), lds, s, (const float*)dy, (const float*)x,
                                  (const float*)y, stats, sums, (float*)dx, (float*)dres, S, C, act, rows, sign_mask, fold, clear_ws,
                                  clear_n, dxb16, sync_ws),
               hipLaunchKernelGGL(in_bwd_fused_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)dy, (const bf16*)x,
                                  (const bf16*)y, stats, sums, (bf16*)dx, (bf16*)dres, S, C, act, rows, sign_mask, fold, clear_ws,
                                  clear_n, dxb16, sync_ws));
  return ctu_check_launch("in_bwd_fused");
}

// number of in-kernel waits that gave up since the library was loaded (synchronises the device; tests and bench.py read it)
extern "C" int ctu_sync_timeouts(void) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sync_timeouts), sizeof(v)) != hipSuccess) return -1;
  return (int)v;
}

// =========================================================================================================
// LayerNorm (vit.py:35,55,116,118; hybrid_CTUNet.py:456,518,630-631): a row is held by LPR lanes, up to 4 vectors
// of 8 per lane (dim <= LPR*32).  dim = LPR * VPL * 8.
// =========================================================================================================
#define LN_MAXV 4
template <typename T, int VPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean_rstd, const int64_t rows,
                                                            const int dim, const int lpr, const int vpl) {
  const int tid = threadIdx.x;
  const int rows_per_block = 256 / lpr;
  const int sub = tid % lpr, rloc = tid / lpr;
  const float inv_d = 1.0f / (float)dim;
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + rloc; row < rows; row += (int64_t)gridDim.x * rows_per_block) {
    float v[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
      if (k < vpl) {
        load8(x + (size_t)row * dim + (size_t)(sub + k * lpr) * 8, v[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[k][e];
      }
    const float mu = group_sum(s, lpr) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
      if (k < vpl) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mu; q = fmaf(d, d, q); }
      }
    const float rstd = rsqrtf(group_sum(q, lpr) * inv_d + NORM_EPS);
#pragma unroll
    for (int k = 0; k < VPL; ++k)
      if (k < vpl) {
        const int c0 = (sub + k * lpr) * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mu) * rstd * gamma[c0 + e] + beta[c0 + e];
        store8(y + (size_t)row * dim + c0, o);
      }
    if (sub == 0) { mean_rstd[row * 2] = mu; mean_rstd[row * 2 + 1] = rstd; }
  }
}

template <typename T, int VPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_rstd, T* __restrict__ dx,
                                                            float* __restrict__ ws, const int64_t rows, const int dim,
                                                            const int lpr, const int vpl, const T* __restrict__ dx_add) {
  const int tid = threadIdx.x;
  const int rows_per_block = 256 / lpr;
  const int sub = tid % lpr, rloc = tid / lpr;
  const float inv_d = 1.0f / (float)dim;
  float ag[VPL][8], ab[VPL][8];
#pragma unroll
  for (int k = 0; k < VPL; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + rloc; row < rows; row += (int64_t)gridDim.x * rows_per_block) {
    const float mu = mean_rstd[row * 2], rstd = mean_rstd[row * 2 + 1];
    float g[VPL][8], xh[VPL][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
      if (k < vpl) {
        const int c0 = (sub + k * lpr) * 8;
        float d[8], xv[8];
        load8(dy + (size_t)row * dim + c0, d);
        load8(x + (size_t)row * dim + c0, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[k][e] = (xv[e] - mu) * rstd;
          g[k][e] = d[e] * gamma[c0 + e];
          s1 += g[k][e];
          s2 = fmaf(g[k][e], xh[k][e], s2);
          ag[k][e] = fmaf(d[e], xh[k][e], ag[k][e]);
          ab[k][e] += d[e];
        }
      }
    const float m1 = group_sum(s1, lpr) * inv_d, m2 = group_sum(s2, lpr) * inv_d;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
      if (k < vpl) {
        const int c0 = (sub + k * lpr) * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (g[k][e] - m1 - xh[k][e] * m2);
        if (dx_add) {  // gradient that reached x through the residual branch around this LayerNorm (ops.GradStash)
          float a[8];
          load8(dx_add + (size_t)row * dim + c0, a);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += a[e];
        }
        store8(dx + (size_t)row * dim + c0, o);
      }
  }
  // column sums: reduce the block's row lanes through LDS, then one atomic per column per block
  __shared__ float red[256 * 8];
  for (int k = 0; k < VPL; ++k) {
    if (k >= vpl) break;
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid * 8 + e] = which ? ab[k][e] : ag[k][e];
      __syncthreads();
      if (rloc == 0) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int r = 0; r < rows_per_block; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += red[(r * lpr + sub) * 8 + e];
        // per-block partials (plain stores): ~1000 workgroups atomically adding into the same <= 512 addresses
        // serialise at the memory side (measured: 238 us per launch whatever the size); stage 2 sums the partials
        float* dst = ws + ((size_t)blockIdx.x * 2 + which) * dim + (sub + k * lpr) * 8;
        store8(dst, acc);
      }
    }
  }
}

static int ln_config(int dim, int* lpr, int* vpl) {
  CTU_REQUIRE(dim >= 8 && dim % 8 == 0, "LayerNorm dim must be a multiple of 8 (%d)", dim);
  const int nvec = dim / 8;
  int l = 64;
  while (l > 1 && (nvec % l != 0)) l >>= 1;
  CTU_REQUIRE(nvec / l <= LN_MAXV, "LayerNorm dim %d unsupported (needs <= %d vectors per lane)", dim, LN_MAXV);
  *lpr = l;
  *vpl = nvec / l;
  return CTU_OK;
}

extern "C" int ctu_layernorm_fwd(ctu_dtype dtype, const void* x, const float* gamma, const float* beta, void* y,
                                 float* mean_rstd, int64_t rows, int32_t dim, ctu_stream_t stream) {
  int lpr, vpl;
  if (int rc = ln_config(dim, &lpr, &vpl)) return rc;
  CTU_REQUIRE(x && gamma && beta && y && mean_rstd && rows > 0, "null pointer / bad rows");
  const unsigned grid = grid_for(rows, 256 / lpr, 4096);
  hipStream_t s = (hipStream_t)stream;
#define LN_FWD(T, V) \
  hipLaunchKernelGGL((layernorm_fwd_kernel<T, V>), dim3(grid), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)y, mean_rstd, rows, dim, lpr, vpl)
  switch (vpl) {  // vectors per lane as a template argument: register arrays sized exactly (occupancy)
    case 1: CTU_DISPATCH(dtype, LN_FWD(float, 1), LN_FWD(bf16, 1)); break;
    case 2: CTU_DISPATCH(dtype, LN_FWD(float, 2), LN_FWD(bf16, 2)); break;
    case 3: CTU_DISPATCH(dtype, LN_FWD(float, 3), LN_FWD(bf16, 3)); break;
    default: CTU_DISPATCH(dtype, LN_FWD(float, 4), LN_FWD(bf16, 4)); break;
  }
#undef LN_FWD
  return ctu_check_launch("layernorm_fwd");
}

// stage 2 of the dgamma/dbeta reduction: out[which][col] += sum over the nblk per-block partial rows of ws
__global__ __launch_bounds__(256) void layernorm_bwd_stage2_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, const int nblk, const int dim) {
  // workgroup = 16 consecutive (which, col) entries x 16 row groups; each thread sums nblk/16 partial rows
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;  // index into the [2][dim] row of a partial
  // blockIdx.y splits the partial rows: with dim = 32 .. 128 (the per-voxel LayerNorms, 1024 partial rows) four to
  // sixteen workgroups walking 64 rows each took 11.6 us per launch, 52 launches per step
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float acc = 0.f;
  if (i < 2 * dim)
    for (int b = b0 + rg; b < b1; b += 16) acc += ws[(size_t)b * 2 * dim + i];
  red[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && i < 2 * dim) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    const int which = i / dim, col = i - which * dim;
    float* dst = which ? dbeta : dgamma;
    if (gridDim.y > 1) atomicAdd(&dst[col], t);
    else dst[col] += t;
  }
}

extern "C" int ctu_layernorm_bwd_add(ctu_dtype dtype, const void* dy, const void* x, const float* gamma,
                                     const float* mean_rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                                     float* ws, int64_t rows, int32_t dim, ctu_stream_t stream);
extern "C" int ctu_layernorm_bwd(ctu_dtype dtype, const void* dy, const void* x, const float* gamma,
                                 const float* mean_rstd, void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows,
                                 int32_t dim, ctu_stream_t stream) {
  return ctu_layernorm_bwd_add(dtype, dy, x, gamma, mean_rstd, nullptr, dx, dgamma, dbeta, ws, rows, dim, stream);
}
extern "C" int ctu_layernorm_bwd_add(ctu_dtype dtype, const void* dy, const void* x, const float* gamma,
                                     const float* mean_rstd, const void* dx_add, void* dx, float* dgamma, float* dbeta,
                                     float* ws, int64_t rows, int32_t dim, ctu_stream_t stream) {
  int lpr, vpl;
  if (int rc = ln_config(dim, &lpr, &vpl)) return rc;
  CTU_REQUIRE(dy && x && gamma && mean_rstd && dx && dgamma && dbeta && ws && rows > 0, "null pointer / bad rows");
  const unsigned grid = grid_for(rows, 256 / lpr, CTU_LN_BWD_MAX_BLOCKS);
  hipStream_t s = (hipStream_t)stream;
#define LN_BWD(T, V)                                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, V>), dim3(grid), dim3(256), 0, s, (const T*)dy, (const T*)x, gamma, mean_rstd, \
                     (T*)dx, ws, rows, dim, lpr, vpl, (const T*)dx_add)
  switch (vpl) {
    case 1: CTU_DISPATCH(dtype, LN_BWD(float, 1), LN_BWD(bf16, 1)); break;
    case 2: CTU_DISPATCH(dtype, LN_BWD(float, 2), LN_BWD(bf16, 2)); break;
    case 3: CTU_DISPATCH(dtype, LN_BWD(float, 3), LN_BWD(bf16, 3)); break;
    default: CTU_DISPATCH(dtype, LN_BWD(float, 4), LN_BWD(bf16, 4)); break;
  }
#undef LN_BWD
  const unsigned ysplit = grid >= 256 ? 8 : (grid >= 64 ? 2 : 1);
  hipLaunchKernelGGL(layernorm_bwd_stage2_kernel, dim3((2 * dim + 15) / 16, ysplit), dim3(256), 0, s, ws, dgamma, dbeta,
                     (int)grid, dim);
  return ctu_check_launch("layernorm_bwd");
}

// =========================================================================================================
// elementwise: GELU, add, broadcast add, cast, column sums, permutes
// =========================================================================================================
template <typename T, int OP>  // 0 gelu fwd (a=x), 1 gelu bwd (a=dy, b=x), 2 add
__global__ __launch_bounds__(256) void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                                                 const int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float va[8], vb[8], o[8];
    load8(a + i * 8, va);
    if (OP != 0) load8(b + i * 8, vb);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (OP == 0) o[e] = gelu_erf(va[e]);
      else if (OP == 1) o[e] = va[e] * gelu_erf_grad(vb[e]);
      else o[e] = va[e] + vb[e];
    }
    store8(y + i * 8, o);
  }
}

template <typename T, int OP>
static int launch_ew(const void* a, const void* b, void* y, int64_t n, hipStream_t s, const char* name) {
  CTU_REQUIRE(a && y && (b || OP == 0) && n > 0 && n % 8 == 0, "%s: null pointer or n %% 8 != 0", name);
  hipLaunchKernelGGL((ew_kernel<T, OP>), dim3(grid_for(n / 8, 256)), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)y,
                     n / 8);
  return ctu_check_launch(name);
}
extern "C" int ctu_gelu_fwd(ctu_dtype dtype, const void* x, void* y, int64_t n, ctu_stream_t stream) {
  CTU_DISPATCH(dtype, return (launch_ew<float, 0>(x, nullptr, y, n, (hipStream_t)stream, "gelu_fwd")),
               return (launch_ew<bf16, 0>(x, nullptr, y, n, (hipStream_t)stream, "gelu_fwd")));
}
extern "C" int ctu_gelu_bwd(ctu_dtype dtype, const void* dy, const void* x, void* dx, int64_t n, ctu_stream_t stream) {
  CTU_DISPATCH(dtype, return (launch_ew<float, 1>(dy, x, dx, n, (hipStream_t)stream, "gelu_bwd")),
               return (launch_ew<bf16, 1>(dy, x, dx, n, (hipStream_t)stream, "gelu_bwd")));
}
extern "C" int ctu_add(ctu_dtype dtype, const void* a, const void* b, void* y, int64_t n, ctu_stream_t stream) {
  CTU_DISPATCH(dtype, return (launch_ew<float, 2>(a, b, y, n, (hipStream_t)stream, "add")),
               return (launch_ew<bf16, 2>(a, b, y, n, (hipStream_t)stream, "add")));
}

template <typename T>
__global__ __launch_bounds__(256) void add_bcast_kernel(const T* __restrict__ a, const float* __restrict__ bc,
                                                        T* __restrict__ y, const int64_t rows, const int cols,
                                                        const int64_t period) {
  const int ncg = cols >> 3;
  const int64_t nvec = rows * ncg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / ncg;
    const int cg = (int)(i - row * ncg);
    float v[8], w[8];
    load8(a + i * 8, v);
    load8(bc + (row % period) * cols + cg * 8, w);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += w[e];
    store8(y + i * 8, v);
  }
}
extern "C" int ctu_add_bcast(ctu_dtype dtype, const void* a, const float* bcast, void* y, int64_t rows, int32_t cols,
                             int64_t period, ctu_stream_t stream) {
  CTU_REQUIRE(a && bcast && y && rows > 0 && cols > 0 && cols % 8 == 0 && period > 0, "add_bcast: bad args");
  const unsigned grid = grid_for(rows * (cols / 8), 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(add_bcast_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, bcast, (float*)y,
                                  rows, cols, period),
               hipLaunchKernelGGL(add_bcast_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)a, bcast, (bf16*)y,
                                  rows, cols, period));
  return ctu_check_launch("add_bcast");
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, const int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dst[i] = (TD)(float)src[i];
}
extern "C" int ctu_cast(const void* src, ctu_dtype sd, void* dst, ctu_dtype dd, int64_t n, ctu_stream_t stream) {
  CTU_REQUIRE(src && dst && n > 0, "cast: bad args");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for(n, 256);
  if (sd == CTU_F32 && dd == CTU_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16*)dst, n);
  else if (sd == CTU_BF16 && dd == CTU_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (float*)dst, n);
  else if (sd == CTU_F32 && dd == CTU_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
  else if (sd == CTU_BF16 && dd == CTU_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n);
  else { ctu_set_error("cast: bad dtypes"); return CTU_ERR_ARG; }
  return ctu_check_launch("cast");
}

// dst[i0*d0+i1*d1+i2*d2] (=|+=) src[i0*s0+i1*s1+i2*s2]; thread index runs over (i0,i1,i2) with i2 fastest
template <typename TD, bool ACC, bool CLEAR = false>
__global__ __launch_bounds__(256) void permute3_kernel(float* __restrict__ src, TD* __restrict__ dst,
                                                       const int64_t n0, const int64_t n1, const int64_t n2,
                                                       const int64_t s0, const int64_t s1, const int64_t s2,
                                                       const int64_t d0, const int64_t d1, const int64_t d2) {
  const int64_t total = n0 * n1 * n2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t i2 = i % n2;
    const int64_t t = i / n2;
    const int64_t i1 = t % n1, i0 = t / n1;
    const float v = src[i0 * s0 + i1 * s1 + i2 * s2];
    if (CLEAR) src[i0 * s0 + i1 * s1 + i2 * s2] = 0.f;  // hand a scratch panel back zeroed (each element is read once)
    TD* p = dst + i0 * d0 + i1 * d1 + i2 * d2;
    if (ACC) *p = (TD)((float)*p + v);
    else *p = (TD)v;
  }
}
extern "C" int ctu_permute3(float* src, void* dst, ctu_dtype dd, int64_t n0, int64_t n1, int64_t n2, int64_t s0,
                            int64_t s1, int64_t s2, int64_t d0, int64_t d1, int64_t d2, int32_t accumulate,
                            ctu_stream_t stream) {
  CTU_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n2 > 0, "permute3: bad args");
  CTU_REQUIRE(!accumulate || dd == CTU_F32, "permute3: accumulate needs an fp32 destination");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for(n0 * n1 * n2, 256);
  if (dd == CTU_F32 && accumulate == 2)
    hipLaunchKernelGGL((permute3_kernel<float, true, true>), dim3(grid), dim3(256), 0, s, src, (float*)dst, n0, n1, n2, s0,
                       s1, s2, d0, d1, d2);
  else if (dd == CTU_F32 && accumulate)
    hipLaunchKernelGGL((permute3_kernel<float, true>), dim3(grid), dim3(256), 0, s, src, (float*)dst, n0, n1, n2, s0, s1,
                       s2, d0, d1, d2);
  else if (dd == CTU_F32)
    hipLaunchKernelGGL((permute3_kernel<float, false>), dim3(grid), dim3(256), 0, s, src, (float*)dst, n0, n1, n2, s0, s1,
                       s2, d0, d1, d2);
  else if (dd == CTU_BF16)
    hipLaunchKernelGGL((permute3_kernel<bf16, false>), dim3(grid), dim3(256), 0, s, src, (bf16*)dst, n0, n1, n2, s0, s1,
                       s2, d0, d1, d2);
  else { ctu_set_error("permute3: bad dtype"); return CTU_ERR_ARG; }
  return ctu_check_launch("permute3");
}

// out[n] += sum_m x[m][n]; block: 256 threads = (N/8 col groups, capped) x row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, const T* __restrict__ row_scale,
                                                     const int64_t M, const int N, const int ld,
                                                     float* __restrict__ out, const int64_t rows_per_block) {
  __shared__ float red[256 * 8];
  const int ncg = N >> 3;
  const int tid = threadIdx.x;
  const int gpb = ncg < 256 ? ncg : 256;  // column groups handled per block pass
  const int rlanes = 256 / gpb;
  const int cgl = tid % gpb, rl = tid / gpb;
  const int64_t m_begin = (int64_t)blockIdx.x * rows_per_block, m_end = min(M, m_begin + rows_per_block);
  for (int cg0 = 0; cg0 < ncg; cg0 += gpb) {
    const int cg = cg0 + cgl;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (rl < rlanes && cg < ncg) {
      int64_t m = m_begin + rl;
      // eight, then four rows in flight per thread: a narrow matrix runs few workgroups (see ctu_colsum), so latency must be hidden
      // inside the thread (64 columns x 1.77 M rows: 177 us with four rows in flight and 256 workgroups - 1.3 TB/s)
      for (; m + 7 * (int64_t)rlanes < m_end; m += 8 * (int64_t)rlanes) {
        float v[8][8];
        float sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) load8(x + (size_t)(m + u * (int64_t)rlanes) * ld + cg * 8, v[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) sc[u] = row_scale ? (float)row_scale[m + u * (int64_t)rlanes] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          acc[e] += ((sc[0] * v[0][e] + sc[1] * v[1][e]) + (sc[2] * v[2][e] + sc[3] * v[3][e])) +
                    ((sc[4] * v[4][e] + sc[5] * v[5][e]) + (sc[6] * v[6][e] + sc[7] * v[7][e]));
      }
      for (; m + 3 * (int64_t)rlanes < m_end; m += 4 * (int64_t)rlanes) {
        float v0[8], v1[8], v2[8], v3[8];
        load8(x + (size_t)m * ld + cg * 8, v0);
        load8(x + (size_t)(m + rlanes) * ld + cg * 8, v1);
        load8(x + (size_t)(m + 2 * rlanes) * ld + cg * 8, v2);
        load8(x + (size_t)(m + 3 * rlanes) * ld + cg * 8, v3);
        float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f;
        if (row_scale) {
          s0 = (float)row_scale[m]; s1 = (float)row_scale[m + rlanes];
          s2 = (float)row_scale[m + 2 * rlanes]; s3 = (float)row_scale[m + 3 * rlanes];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (s0 * v0[e] + s1 * v1[e]) + (s2 * v2[e] + s3 * v3[e]);
      }
      for (; m < m_end; m += rlanes) {
        float v[8];
        load8(x + (size_t)m * ld + cg * 8, v);
        const float sc = row_scale ? (float)row_scale[m] : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(sc, v[e], acc[e]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[e];
    __syncthreads();
    if (rl == 0 && cg < ncg) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = 0.f;
      for (int r = 0; r < rlanes; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += red[(r * gpb + cgl) * 8 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&out[cg * 8 + e], t[e]);
    }
  }
}
// few rows, many columns (the gradient of a parameter broadcast over the batch: position embedding, vit.py:118-126 - 2 rows of 331 776
// columns): one thread per 8-column group walks the rows and owns its outputs - no LDS, no atomics.  (Through colsum_kernel this was
// ONE workgroup looping over 41 472 column groups: 280 us for 1.3 MB.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, const T* __restrict__ row_scale, const int64_t M,
                                                          const int N, const int ld, float* __restrict__ out) {
  const int cg = blockIdx.x * 256 + threadIdx.x;
  if (cg >= (N >> 3)) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int64_t m = 0; m < M; ++m) {
    float v[8];
    load8(x + (size_t)m * ld + cg * 8, v);
    const float sc = row_scale ? (float)row_scale[m] : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(sc, v[e], acc[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) out[cg * 8 + e] += acc[e];
}

extern "C" int ctu_colsum(ctu_dtype dtype, const void* x, const void* row_scale, int64_t M, int32_t N, int32_t ld,
                          float* out, ctu_stream_t stream) {
  CTU_REQUIRE(x && out && M > 0 && N > 0 && N % 8 == 0 && ld >= N && ld % 8 == 0, "colsum: bad args");
  if (M <= 64 && N >= 2048) {
    const unsigned wgrid = (unsigned)((N / 8 + 255) / 256);
    hipStream_t ws = (hipStream_t)stream;
    CTU_DISPATCH(dtype,
                 hipLaunchKernelGGL(colsum_wide_kernel<float>, dim3(wgrid), dim3(256), 0, ws, (const float*)x, (const float*)row_scale, M, N,
                                    ld, out),
                 hipLaunchKernelGGL(colsum_wide_kernel<bf16>, dim3(wgrid), dim3(256), 0, ws, (const bf16*)x, (const bf16*)row_scale, M, N,
                                    ld, out));
    return ctu_check_launch("colsum");
  }
  // every workgroup ends with N atomics on the same N addresses: a narrow matrix takes fewer, longer workgroups
  // (1024 workgroups x 16 columns spent 200 us of a 265 us pass queueing on 16 addresses)
  const int64_t blocks = N <= 64 ? 256 : 1024;
  int64_t rows = (M + blocks - 1) / blocks;
  if (rows < 64) rows = 64;
  const unsigned grid = (unsigned)((M + rows - 1) / rows);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x,
                                  (const float*)row_scale, M, N, ld, out, rows),
               hipLaunchKernelGGL(colsum_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x,
                                  (const bf16*)row_scale, M, N, ld, out, rows));
  return ctu_check_launch("colsum");
}

// out[m][n] = x[m] * w[n]: the 1x1x1 convolution of a one-channel volume (ResBlock.conv3 shortcut of vit_encoder0,
// hybrid_CTUNet.py:75-83) - a pure store stream.  grid.x * 256 is a multiple of N / 8: a thread keeps its 8 columns.
template <typename T>
__global__ __launch_bounds__(256) void outer_rows_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         T* __restrict__ out, const int64_t M, const int N) {
  const int ncg = N >> 3;
  const int64_t nvec = M * ncg;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(i0 % ncg);
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) wv[e] = w[cg * 8 + e];
  for (int64_t i = i0; i < nvec; i += (int64_t)gridDim.x * 256) {
    const float xv = (float)x[i / ncg];
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = xv * wv[e];
    store8(out + i * 8, o);
  }
}
extern "C" int ctu_outer_rows(ctu_dtype dtype, const void* x, const float* w, void* out, int64_t M, int32_t N,
                              ctu_stream_t stream) {
  CTU_REQUIRE(x && w && out && M > 0 && N > 0 && N % 8 == 0, "outer_rows: bad args");
  const int ncg = N / 8;
  int64_t g = (M * ncg + 255) / 256;
  if (g > 8192) g = 8192;
  g = ((g + ncg - 1) / ncg) * ncg;
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(outer_rows_kernel<float>, dim3((unsigned)g), dim3(256), 0, s, (const float*)x, w,
                                  (float*)out, M, N),
               hipLaunchKernelGGL(outer_rows_kernel<bf16>, dim3((unsigned)g), dim3(256), 0, s, (const bf16*)x, w,
                                  (bf16*)out, M, N));
  return ctu_check_launch("outer_rows");
}

// patchify: x [B][H][W][F] (c = 1) -> tokens [B][(h w f)][(p1 p2 pf)]   (vit.py:115)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ x, T* __restrict__ tok, const int B,
                                                       const int H, const int W, const int F, const int p1, const int p2,
                                                       const int p3) {
  const int nh = H / p1, nw = W / p2, nf = F / p3;
  const int64_t total = (int64_t)B * H * W * F;
  const int pd = p1 * p2 * p3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i indexes the OUTPUT: (((b*nh + h)*nw + w)*nf + f)*pd + ((i1*p2 + i2)*p3 + i3)
    const int e = (int)(i % pd);
    int64_t t = i / pd;
    const int f = (int)(t % nf); t /= nf;
    const int w = (int)(t % nw); t /= nw;
    const int h = (int)(t % nh);
    const int b = (int)(t / nh);
    const int i3 = e % p3, i2 = (e / p3) % p2, i1 = e / (p3 * p2);
    tok[i] = x[(((size_t)b * H + h * p1 + i1) * W + w * p2 + i2) * F + f * p3 + i3];
  }
}
extern "C" int ctu_patchify(ctu_dtype dtype, const void* x, void* tokens, int32_t B, int32_t H, int32_t W, int32_t F,
                            int32_t p1, int32_t p2, int32_t p3, ctu_stream_t stream) {
  CTU_REQUIRE(x && tokens && B > 0 && H % p1 == 0 && W % p2 == 0 && F % p3 == 0, "patchify: bad args");
  const unsigned grid = grid_for((int64_t)B * H * W * F, 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)tokens, B,
                                  H, W, F, p1, p2, p3),
               hipLaunchKernelGGL(patchify_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)tokens, B, H,
                                  W, F, p1, p2, p3));
  return ctu_check_launch("patchify");
}

// pixel shuffle (hybrid_CTUNet.py:420-428): y[b][d*p1+i1][h*p2+i2][w*p3+i3][cc] = x[b][d][h][w][((cc*p1+i1)*p2+i2)*p3+i3]
// thread per output element group of 8 consecutive cc (requires c % 8 == 0): source elements are strided by p1*p2*p3.
template <typename T, bool INV>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const T* __restrict__ x, T* __restrict__ y, const int B,
                                                            const int D, const int H, const int W, const int c,
                                                            const int p1, const int p2, const int p3) {
  const int P = p1 * p2 * p3;
  const int ncg = c >> 3;
  const int64_t total = (int64_t)B * D * p1 * H * p2 * W * p3 * ncg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    int64_t t = i / ncg;
    const int ow = (int)(t % (W * p3)); t /= (W * p3);
    const int oh = (int)(t % (H * p2)); t /= (H * p2);
    const int od = (int)(t % (D * p1));
    const int b = (int)(t / (D * p1));
    const int w = ow / p3, i3 = ow % p3, h = oh / p2, i2 = oh % p2, d = od / p1, i1 = od % p1;
    const size_t big = ((((size_t)b * D + d) * H + h) * W + w) * (size_t)(c * P) + (size_t)(cg * 8) * P +
                       (size_t)((i1 * p2 + i2) * p3 + i3);
    const size_t small = (size_t)i * 8;
    if (!INV) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)x[big + (size_t)e * P];
      store8(y + small, v);
    } else {  // x is the shuffled-layout tensor (gradient), y the original layout
      float v[8];
      load8(x + small, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[big + (size_t)e * P] = (T)v[e];
    }
  }
}
// The same through LDS: a workgroup owns VOX consecutive input voxels.  Forward: their rows [c * P] arrive coalesced (16 B per
// lane), each output vector gathers its 8 channels cc .. cc + 7 of one sub-position from LDS (stride P elements) and leaves
// as a 16-byte store - every byte of the big tensor is read once (the per-output-element kernel above touches each 32-byte
// sector of a row from P / 2 different waves: 1.2 TB/s at 442 k voxels x 128 channels).  Inverse: the other way round.
template <bool INV>
__global__ __launch_bounds__(256) void pixel_shuffle_lds_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const int B,
                                                                const int D, const int H, const int W, const int c,
                                                                const int p1, const int p2, const int p3, const int VOX) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds[];
  bf16* tile = reinterpret_cast<bf16*>(ps_lds);   // [VOX][c * P]
  const int P = p1 * p2 * p3;
  const int CB = c * P;
  const int ncg = c >> 3;
  const int64_t nvox = (int64_t)B * D * H * W;
  const int64_t v0 = (int64_t)blockIdx.x * VOX;
  const int nv = (int)min((int64_t)VOX, nvox - v0);
  const bf16* big_in = x + v0 * CB;   // (forward: the big tensor is the input)
  bf16* big_out = y + v0 * CB;        // (inverse: the big tensor is the output)
  const int vec_rows = CB >> 3;       // 16-byte vectors per big row
  if (!INV) {
    for (int i = threadIdx.x; i < nv * vec_rows; i += 256)
      *reinterpret_cast<bf16x8*>(tile + (size_t)i * 8) = *reinterpret_cast<const bf16x8*>(big_in + (size_t)i * 8);
    __syncthreads();
  }
  // small-layout vectors of this workgroup: (voxel, sub-position, channel group)
  const int per_vox = P * ncg;
  for (int i = threadIdx.x; i < nv * per_vox; i += 256) {
    const int cg = i % ncg;
    const int sub = (i / ncg) % P;
    const int lv = i / per_vox;
    const int64_t v = v0 + lv;
    const int w = (int)(v % W);
    const int h = (int)((v / W) % H);
    const int d = (int)((v / ((int64_t)W * H)) % D);
    const int b = (int)(v / ((int64_t)W * H * D));
    const int i3 = sub % p3, i2 = (sub / p3) % p2, i1 = sub / (p3 * p2);
    const size_t small = (((((size_t)b * D * p1 + d * p1 + i1) * (H * p2) + h * p2 + i2) * (size_t)(W * p3) + w * p3 + i3) * c) +
                         (size_t)cg * 8;
    bf16* t = tile + (size_t)lv * CB + (size_t)(cg * 8) * P + sub;
    if (!INV) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = t[e * P];
      *reinterpret_cast<bf16x8*>(y + small) = o;
    } else {
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(x + small);
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e * P] = o[e];
    }
  }
  if (INV) {
    __syncthreads();
    for (int i = threadIdx.x; i < nv * vec_rows; i += 256)
      *reinterpret_cast<bf16x8*>(big_out + (size_t)i * 8) = *reinterpret_cast<const bf16x8*>(tile + (size_t)i * 8);
  }
}

extern "C" int ctu_pixel_shuffle(ctu_dtype dtype, const void* x, void* y, int32_t B, int32_t D, int32_t H, int32_t W,
                                 int32_t c, int32_t p1, int32_t p2, int32_t p3, int32_t inverse, ctu_stream_t stream) {
  CTU_REQUIRE(x && y && B > 0 && D > 0 && H > 0 && W > 0 && c > 0 && c % 8 == 0 && p1 > 0 && p2 > 0 && p3 > 0,
              "pixel_shuffle: bad args (c must be a multiple of 8)");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CTU_BF16) {
    const int CB = c * p1 * p2 * p3;
    int VOX = 16384 / (CB * 2);   // 16 KiB of LDS per workgroup
    if (VOX > 64) VOX = 64;
    if (VOX >= 1) {
      const int64_t nvox = (int64_t)B * D * H * W;
      const dim3 g((unsigned)((nvox + VOX - 1) / VOX));
      const size_t lds = (size_t)VOX * CB * 2;
      if (inverse)
        hipLaunchKernelGGL(pixel_shuffle_lds_kernel<true>, g, dim3(256), lds, s, (const bf16*)x, (bf16*)y, B, D, H, W, c, p1, p2, p3, VOX);
      else
        hipLaunchKernelGGL(pixel_shuffle_lds_kernel<false>, g, dim3(256), lds, s, (const bf16*)x, (bf16*)y, B, D, H, W, c, p1, p2, p3, VOX);
      return ctu_check_launch("pixel_shuffle");
    }
  }
  const unsigned grid = grid_for((int64_t)B * D * H * W * p1 * p2 * p3 * (c / 8), 256);
#define PS_LAUNCH(T, INV) \
  hipLaunchKernelGGL((pixel_shuffle_kernel<T, INV>), dim3(grid), dim3(256), 0, s, (const T*)x, (T*)y, B, D, H, W, c, p1, p2, p3)
  if (inverse) { CTU_DISPATCH(dtype, PS_LAUNCH(float, true), PS_LAUNCH(bf16, true)); }
  else { CTU_DISPATCH(dtype, PS_LAUNCH(float, false), PS_LAUNCH(bf16, false)); }
#undef PS_LAUNCH
  return ctu_check_launch("pixel_shuffle");
}

// =========================================================================================================
// Stride-2 helpers of the data gradients of the ResNet stage transitions (resnet.py:98 conv2 with stride 2, resnet.py:166-176 the
// strided 1x1x1 downsample).  The generic implicit GEMM computed them per INPUT voxel with 7 of 8 taps masked (25 - 47 TFLOP/s):
//   * 3x3x3, stride 2, padding 1:  dX = conv3x3x3_stride1(U, flipped W) with U = dY zero-upsampled by 2 - eight times the MACs,
//     but on the halo kernel (ctu_conv3_halo) they cost a fifth of the time; ctu_upsample2_zeros writes U.
//   * 1x1x1, stride 2:  dX is dY W at the even voxels and zero elsewhere: a plain GEMM over the OUTPUT rows, then
//     ctu_add_strided2 adds the compact result into the gradient it joins (the block input's) at the even voxels.
// =========================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void upsample2_zeros_kernel(const T* __restrict__ x, T* __restrict__ y, const int B, const int D,
                                                              const int H, const int W, const int C) {
  const int ncg = C >> 3;
  const int64_t total = (int64_t)B * 8 * D * H * W * ncg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i / ncg;
    const int cg = (int)(i - t * ncg);
    const int w = (int)(t % (2 * W)); t /= 2 * W;
    const int h = (int)(t % (2 * H)); t /= 2 * H;
    const int d = (int)(t % (2 * D));
    const int b = (int)(t / (2 * D));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (!((d | h | w) & 1)) load8(x + ((((size_t)b * D + (d >> 1)) * H + (h >> 1)) * W + (w >> 1)) * C + cg * 8, v);
    store8(y + i * 8, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_strided2_kernel(T* __restrict__ y, const T* __restrict__ x, const int B, const int D,
                                                           const int H, const int W, const int C) {
  const int ncg = C >> 3;
  const int64_t total = (int64_t)B * D * H * W * ncg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i / ncg;
    const int cg = (int)(i - t * ncg);
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    T* dst = y + ((((size_t)b * 2 * D + 2 * d) * 2 * H + 2 * h) * 2 * W + 2 * w) * C + cg * 8;
    float a[8], v[8];
    load8(dst, a);
    load8(x + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += v[e];
    store8(dst, a);
  }
}

extern "C" int ctu_upsample2_zeros(ctu_dtype dtype, const void* x, void* y, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                   ctu_stream_t stream) {
  CTU_REQUIRE(x && y && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "upsample2_zeros: bad args (C %% 8)");
  const unsigned grid = grid_for((int64_t)B * 8 * D * H * W * (C / 8), 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(upsample2_zeros_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, B, D, H, W, C),
               hipLaunchKernelGGL(upsample2_zeros_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, B, D, H, W, C));
  return ctu_check_launch("upsample2_zeros");
}

extern "C" int ctu_add_strided2(ctu_dtype dtype, void* y, const void* x, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                ctu_stream_t stream) {
  CTU_REQUIRE(x && y && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "add_strided2: bad args (C %% 8)");
  const unsigned grid = grid_for((int64_t)B * D * H * W * (C / 8), 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(add_strided2_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)y, (const float*)x, B, D, H, W, C),
               hipLaunchKernelGGL(add_strided2_kernel<bf16>, dim3(grid), dim3(256), 0, s, (bf16*)y, (const bf16*)x, B, D, H, W, C));
  return ctu_check_launch("add_strided2");
}

// =========================================================================================================
// binary cross-weight fusion core (hybrid_CTUNet.py:651-665).  4 lanes per (token, head of 32 channels).
// =========================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pwa_fwd_kernel(const T* __restrict__ qkv1, const T* __restrict__ qkv2,
                                                      T* __restrict__ out, const int64_t rows, const int C,
                                                      const float scale) {
  const int ncg = C >> 3;
  const int64_t nvec = rows * ncg;  // multiple of 4 (C % 32 == 0)
  const int64_t nvec_pad = (nvec + 255) / 256 * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_pad; i += (int64_t)gridDim.x * 256) {
    const bool ok = i < nvec;
    const int64_t row = ok ? i / ncg : 0;
    const int c0 = ok ? (int)(i - row * ncg) * 8 : 0;
    const T* r1 = qkv1 + (size_t)row * 3 * C;
    const T* r2 = qkv2 + (size_t)row * 3 * C;
    float q1[8], k1[8], v1[8], q2[8], k2[8], v2[8];
    load8(r1 + c0, q1); load8(r1 + C + c0, k1); load8(r1 + 2 * C + c0, v1);
    load8(r2 + c0, q2); load8(r2 + C + c0, k2); load8(r2 + 2 * C + c0, v2);
    float z = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) z += q2[e] * k1[e] - q1[e] * k2[e];
    z = group_sum(z, 4) * scale;          // d1 - d2 over the head's 32 channels
    const float a1 = 1.0f / (1.0f + __expf(-z));
    if (ok) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = a1 * v1[e] + (1.0f - a1) * v2[e];
      store8(out + (size_t)row * C + c0, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pwa_bwd_kernel(const T* __restrict__ qkv1, const T* __restrict__ qkv2,
                                                      const T* __restrict__ dout, T* __restrict__ dqkv1,
                                                      T* __restrict__ dqkv2, const int64_t rows, const int C,
                                                      const float scale) {
  const int ncg = C >> 3;
  const int64_t nvec = rows * ncg;
  const int64_t nvec_pad = (nvec + 255) / 256 * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_pad; i += (int64_t)gridDim.x * 256) {
    const bool ok = i < nvec;
    const int64_t row = ok ? i / ncg : 0;
    const int c0 = ok ? (int)(i - row * ncg) * 8 : 0;
    const T* r1 = qkv1 + (size_t)row * 3 * C;
    const T* r2 = qkv2 + (size_t)row * 3 * C;
    float q1[8], k1[8], v1[8], q2[8], k2[8], v2[8], go[8];
    load8(r1 + c0, q1); load8(r1 + C + c0, k1); load8(r1 + 2 * C + c0, v1);
    load8(r2 + c0, q2); load8(r2 + C + c0, k2); load8(r2 + 2 * C + c0, v2);
    load8(dout + (size_t)row * C + c0, go);
    float z = 0.f, da = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      z += q2[e] * k1[e] - q1[e] * k2[e];
      da += go[e] * (v1[e] - v2[e]);
    }
    z = group_sum(z, 4) * scale;
    da = group_sum(da, 4);
    const float a1 = 1.0f / (1.0f + __expf(-z));
    const float dz = da * a1 * (1.0f - a1) * scale;
    if (ok) {
      float o[8];
      T* g1 = dqkv1 + (size_t)row * 3 * C;
      T* g2 = dqkv2 + (size_t)row * 3 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = -dz * k2[e];
      store8(g1 + c0, o);               // dq1
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = dz * q2[e];
      store8(g1 + C + c0, o);           // dk1
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = a1 * go[e];
      store8(g1 + 2 * C + c0, o);       // dv1
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = dz * k1[e];
      store8(g2 + c0, o);               // dq2
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = -dz * q1[e];
      store8(g2 + C + c0, o);           // dk2
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (1.0f - a1) * go[e];
      store8(g2 + 2 * C + c0, o);       // dv2
    }
  }
}

extern "C" int ctu_pwa_fwd(ctu_dtype dtype, const void* qkv1, const void* qkv2, void* out, int64_t rows, int32_t C,
                           float scale, ctu_stream_t stream) {
  CTU_REQUIRE(qkv1 && qkv2 && out && rows > 0 && C > 0 && C % 32 == 0, "pwa_fwd: bad args (C %% 32)");
  const unsigned grid = grid_for(rows * (C / 8), 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(pwa_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)qkv1,
                                  (const float*)qkv2, (float*)out, rows, C, scale),
               hipLaunchKernelGGL(pwa_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)qkv1, (const bf16*)qkv2,
                                  (bf16*)out, rows, C, scale));
  return ctu_check_launch("pwa_fwd");
}
extern "C" int ctu_pwa_bwd(ctu_dtype dtype, const void* qkv1, const void* qkv2, const void* dout, void* dqkv1,
                           void* dqkv2, int64_t rows, int32_t C, float scale, ctu_stream_t stream) {
  CTU_REQUIRE(qkv1 && qkv2 && dout && dqkv1 && dqkv2 && rows > 0 && C > 0 && C % 32 == 0, "pwa_bwd: bad args");
  const unsigned grid = grid_for(rows * (C / 8), 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(pwa_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)qkv1,
                                  (const float*)qkv2, (const float*)dout, (float*)dqkv1, (float*)dqkv2, rows, C, scale),
               hipLaunchKernelGGL(pwa_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)qkv1, (const bf16*)qkv2,
                                  (const bf16*)dout, (bf16*)dqkv1, (bf16*)dqkv2, rows, C, scale));
  return ctu_check_launch("pwa_bwd");
}

// =========================================================================================================
// Patch matrix of a one-channel volume: P[m][k] = x[b][od*sd - pd + td][oh*sh - ph + th][ow*sw - pw + tw] for tap
// k = (td*kh + th)*kw + tw < taps, zero for padding voxels and for taps <= k < kpad (kpad a multiple of 64).
// The Cin = 1 convolutions (vit_encoder0.conv1 3x3x3, hybrid_CTUNet.py:57-65; ResNet stem 7x7x7 s(2,2,1),
// resnet.py:150-155) then run as plain LDS-DMA GEMMs: forward out = P W^T (57 MB + the patch matrix instead of a
// VALU-bound direct convolution at a tenth of the VALU peak), weight gradient dW = dY^T P.  bf16; the image is a few
// MB and stays in L2, the pass is bound by writing P (16 B per thread).
// =========================================================================================================
// One thread writes a whole row of P (kpad / 8 vectors): the row's output coordinate is decomposed once, the taps advance by
// increment with carry, the image reads (L2-resident) of neighbouring rows overlap.  (One thread per 16-byte vector paid nine
// integer divisions for eight 2-byte reads: 164 us for the 340 MB of the stem's patch matrix, 3x its store time.)
// (The row index is decomposed with FastDiv, common.h: the kernel spent more on its three 32-bit divisions per 16-byte store
// than on the store - 144 us per launch for 113 / 340 MB.)
__global__ __launch_bounds__(256) void im2col_cin1_kernel(const bf16* __restrict__ x, bf16* __restrict__ P, const ctu_geom g,
                                                          const int taps, const int kpad, const int64_t M, const FastDiv dkg,
                                                          const FastDiv dwo, const FastDiv dho, const FastDiv ddo) {
  const int kg = kpad >> 3;  // 8-tap groups per row
  // lanes of a wave take CONSECUTIVE vectors of P (coalesced 16-byte stores): vector i = (row, group); a thread keeps one group
  // index and walks down the rows, so only the row index is decomposed per iteration (three divisions)
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;   // a multiple of kg (host): the thread's group never changes
  const int grp = (int)(i0 % kg);
  const int k0 = grp * 8;
  int tw0 = k0 % g.kw;
  const int tq = k0 / g.kw;
  int th0 = tq % g.kh, td0 = tq / g.kh;
  const int64_t total = M * kg;
  for (int64_t i = i0; i < total; i += stride) {
    const int64_t m = i < (1ll << 31) ? (int64_t)fdiv((unsigned)i, dkg) : i / kg;
    int t = (int)m;
    int q = fdiv((unsigned)t, dwo);
    const int ow = t - q * g.Wo; t = q;
    q = fdiv((unsigned)t, dho);
    const int oh = t - q * g.Ho; t = q;
    q = fdiv((unsigned)t, ddo);
    const int od = t - q * g.Do;
    const int b = q;
    bf16x8 v;
    int tw = tw0, th = th0, td = td0;
    const int bd = od * g.sd - g.pd, bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;
    const bf16* xb = x + (size_t)b * g.Di * g.Hi * g.Wi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16 val = (bf16)0.f;
      const int id = bd + td, ih = bh + th, iw = bw + tw;
      if (k0 + j < taps && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi)
        val = xb[((size_t)id * g.Hi + ih) * g.Wi + iw];
      v[j] = val;
      if (++tw == g.kw) { tw = 0; if (++th == g.kh) { th = 0; ++td; } }
    }
    *reinterpret_cast<bf16x8*>(P + m * kpad + k0) = v;
  }
}

extern "C" int ctu_im2col_cin1(const void* x, void* P, const ctu_geom* g, int32_t kpad, ctu_stream_t stream) {
  CTU_REQUIRE(x && P && g, "im2col_cin1: null pointer");
  const int taps = g->kd * g->kh * g->kw;
  CTU_REQUIRE(kpad >= taps && kpad % 8 == 0, "im2col_cin1: kpad must be a multiple of 8 and >= taps");
  CTU_REQUIRE(g->B > 0 && g->Do > 0 && g->Ho > 0 && g->Wo > 0 && g->sd > 0 && g->sh > 0 && g->sw > 0, "im2col_cin1: bad geom");
  const int64_t M = (int64_t)g->B * g->Do * g->Ho * g->Wo;
  CTU_REQUIRE(M < (1ll << 31), "im2col_cin1: too many rows");
  const int64_t total = M * (kpad / 8);
  const int kg = kpad / 8;
  int64_t grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  grid = (grid + kg - 1) / kg * kg;   // grid x 256 a multiple of kg: a thread keeps its tap group
  hipLaunchKernelGGL(im2col_cin1_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x, (bf16*)P, *g, taps, kpad, M, fast_div(kg), fast_div(g->Wo), fast_div(g->Ho), fast_div(g->Do));
  return ctu_check_launch("im2col_cin1");
}
