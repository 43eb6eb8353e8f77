// Host-side interface of the LDS-DMA plain GEMM kernels (gemm_dma.hip), called from the ctu_igemm_* entry points.
#pragma once
#include "common.h"

// Gathered operand of a convolution whose taps never leave the grid (no padding, (Do - 1) s + k <= Di per axis: the
// kernel == stride patch convolutions and transposed-convolution gradients, the 1x1x1 stride-2 shortcuts).  Row m of the
// row space [B][Do][Ho][Wo] and tap t read source row base(m) + tapoff[t] of the gathered tensor [B*Di*Hi*Wi][C]:
// base(m) = b sB + od sD + oh sH + ow sW, separable from the tap - the DMA kernels add a per-lane row base to the
// wave-uniform tap address, nothing else changes.  taps <= 8.
struct GatherGeom {
  int on;
  int C;                  // channels = row length of the gathered tensor
  int taps;
  int sB, sD, sH, sW;     // source-row steps: Di*Hi*Wi, sd*Hi*Wi, sh*Wi, sw
  int tapoff[8];          // (td*Hi + th)*Wi + tw
  FastDiv dWo, dHo, dDo;  // decode of m
  FastDiv dKpt;           // NT kernel: 64-deep k steps per tap (filled by the launcher)
};
__device__ __forceinline__ int gather_base(const GatherGeom& g, int m) {
  int q = fdiv((unsigned)m, g.dWo);
  const int ow = m - q * g.dWo.d; m = q;
  q = fdiv((unsigned)m, g.dHo);
  const int oh = m - q * g.dHo.d; m = q;
  q = fdiv((unsigned)m, g.dDo);
  const int od = m - q * g.dDo.d;
  return q * g.sB + od * g.sD + oh * g.sH + ow * g.sW;
}
// tapoff[t] for a wave-uniform t without indexing the kernel argument dynamically (that copies the table to scratch memory
// and puts a scratch load on every stage issue)
__device__ __forceinline__ int gather_tapoff(const GatherGeom& g, int t) {
  int off = g.tapoff[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) off = t == i ? g.tapoff[i] : off;
  return off;
}
// fills `gg` when `g` is such a geometry (mode 0) with C1 % 64 == 0 channels and no second source
bool gather_geom_from(const ctu_geom* g, GatherGeom& gg);

struct GemmNtArgs {
  const bf16* a1;  // [M][C1]; with ga.on the gathered tensor [B*Di*Hi*Wi][C], K = taps * C and w = [taps][N][C]
  const bf16* a2;  // [M][C2] (k >= C1), or NULL
  const bf16* w;   // [N][K], K = C1 + C2; or, with w_kn, [K][N] (reduction-major)
  int w_kn;
  void* out;
  ctu_epilogue ep;
  int M, N, K, C1, C2;
  int splitk;      // in: requested split; out: the split used
  float* ws;       // fp32 [M][N], zeroed (split-K only)
  double* in_acc;  // optional [M / in_rows][N][2] += (sum, sum of squares) of the output columns per batch item
  int in_rows;     // rows per batch item (a multiple of 128: no tile straddles two items)
  int tiles_m, tiles_n, ksteps, ks_per_split, nwork;  // filled by the launcher
  int debug;       // measurement hook (ctu_set_option "nt_debug"): 1 = no output stores, 2 = no operand DMA
  GatherGeom ga;   // ga.on: A rows are gathered (see GatherGeom)
};

// returns 0 on launch, -1 if the shape is out of range (caller falls back to the generic kernel)
int launch_gemm_nt_dma(GemmNtArgs& p, hipStream_t stream);

struct GemmTnArgs {
  const bf16* p;    // [M][ldp], N columns used
  const bf16* q1;   // [M][C1]
  const bf16* q2;   // [M][C2] or NULL
  float* dw;        // [N][C] fp32, accumulated
  float* bias_grad; // [N] fp32 += column sums of P, or NULL
  int ldp, M, N, C, C1, C2;
  float* part;      // filled by the launcher: partial panels [splits][N][C] for the two-stage reduction, or NULL
  int tiles_n, tiles_c, splits, rows_per_split;
  GatherGeom ga;    // ga.on: Q rows are gathered, C = taps * ga.C columns (column t*ga.C + c = tap t, channel c) and dw = [taps][N][ga.C]
};

// returns 0 on launch (a.part / a.splits tell the caller whether a reduction pass over `ws` must follow), -1 if out of range
int launch_gemm_tn_dma(GemmTnArgs& a, float* ws, int64_t ws_floats, hipStream_t stream);
