// Host-side interface of the LDS-DMA plain GEMM kernels (gemm_dma.hip), called from the ctu_igemm_* entry points.
#pragma once
#include "common.h"

struct GemmNtArgs {
  const bf16* a1;  // [M][C1]
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
};

// returns 0 on launch (a.part / a.splits tell the caller whether a reduction pass over `ws` must follow), -1 if out of range
int launch_gemm_tn_dma(GemmTnArgs& a, float* ws, int64_t ws_floats, hipStream_t stream);
