/*
 * ctunet_hip.h — C ABI of libctunet_hip.so: the MI355X (gfx950) kernels under the Hybrid-CTUNet hot path.
 *
 * The reference (shouwangzhe134/Hybrid-CTUNet) has NO FFI: its boundary is the Python nn.Module API
 * (networks/hybrid_CTUNet.py:694-1036).  Every FLOP there is dispatched through ATen leaf ops; this header
 * declares the native entry points that replace those leaf-op families, one group per row of SURVEY.md
 * section 2.2 (K1..K15).  Each declaration cites the reference call sites it stands in for.  The host-side
 * mirror (the Python files under hybrid-ctunet_amd/networks) binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers unless noted.
 *   - activations are channels-last ("NDHWC"): a volume tensor is a row-major [rows = B*D*H*W][C] matrix.
 *   - dtype selects the activation/packed-weight element type: CTU_F32 (parity mode, f32-input MFMA) or
 *     CTU_BF16 (bf16 operands, fp32 accumulate).  Statistics, losses, master weights and weight
 *     gradients are always fp32.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing and keeps no
 *     global state; workspaces are passed in.  Graph-capture safe.
 *   - return value: CTU_OK or an error code; ctu_last_error() gives a thread-local message.
 */
#ifndef CTUNET_HIP_H
#define CTUNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTU_OK 0
#define CTU_ERR_ARG 1     /* unsupported shape / null pointer / misaligned channel count */
#define CTU_ERR_LAUNCH 2  /* hipGetLastError() after launch was not hipSuccess */

typedef enum { CTU_F32 = 0, CTU_BF16 = 1 } ctu_dtype;
/* Memory layout of a volume tensor [B][D][H][W][C].  CTU_LAYOUT_NDHWC: channels-last rows, element (voxel m, channel c)
 * at m*C + c - the layout of every tensor unless stated.  CTU_LAYOUT_B16: 16-channel blocks outermost,
 * [C/16][B*D*H*W][16], element at ((c >> 4) * voxels + m) * 16 + (c & 15).  It exists for ONE dataflow: the input of a
 * 3x3x3 convolution that is the output of an InstanceNorm (+LeakyReLU) (resnet.py:106-113, hybrid_CTUNet.py:93-99) - and,
 * backwards, the gradient an InstanceNorm backward hands to the convolution in front of it.  The halo convolution
 * gathers its input 16 channels at a time: from channels-last rows that is 32 bytes out of every voxel's row (one cache
 * line touched per voxel and pass), from B16 a halo row of 10 voxels is one 320-byte run.  Produced only by
 * ctu_in_apply / ctu_in_bwd_apply, consumed only by ctu_conv3_halo / ctu_conv3_halo_wgrad (bf16). */
typedef enum { CTU_LAYOUT_NDHWC = 0, CTU_LAYOUT_B16 = 1 } ctu_layout;
typedef void* ctu_stream_t; /* hipStream_t */

int ctu_abi_version(void);
const char* ctu_last_error(void);
/* Test hooks (process-wide): "attn_valu" = 1 makes ctu_attn_fwd/_bwd use the VALU reference kernels even where the MFMA
 * kernels apply; "generic_gemm" = 1 keeps plain bf16 GEMMs on the generic implicit-GEMM kernels instead of the LDS-DMA
 * GEMM kernels - so both implementations can be checked against the oracle in one process.  "route" = bit set of A/B
 * routing switches for measurements (1: short-K layers on the general NT kernel instead of gemm_nt_stream, 2: no 128-deep
 * stages, 4: no two-k-group trunk tiles, 8: previous channel-split rule of the small 3x3x3 convs, 16: 9-tap weight stages
 * in the halo kernels, 32: wave 0 gathers the halo alone, 64: weight-gradient operand DMA in one burst, 128: weight-gradient
 * atomics even with a workspace, 256: 32-KiB LDS reduction in every InstanceNorm backward reduce, 512: one resident halo
 * workgroup fewer per CU, 1024: at least 64 rows per workgroup in every InstanceNorm reduction, 2048: no batch-pair bricks
 * in the halo forward / data-gradient kernels - each bit restores a previous rule); "nt_debug" = bits
 * that switch a kernel's memory traffic off for timing.  Options are plain process-wide ints read per launch - no
 * launch path calls getenv. */
int ctu_set_option(const char* name, int32_t value);

/* Geometry of an implicit GEMM over a channels-last volume.
 * Row space  : M = B*Do*Ho*Wo rows (one per voxel of the Do x Ho x Wo grid).
 * Gather grid: B*Di*Hi*Wi voxels with C1 (+C2, a second tensor concatenated along channels) channels.
 * For row (b,od,oh,ow) and tap (td,th,tw) the gathered voxel is
 *   mode 0:  i = o*s - p + t                      (forward convolution, nn.Conv3d)
 *   mode 1:  i = (o + p - t) / s  if divisible    (data gradient of a strided convolution / transposed gather)
 * and contributes zero when out of range.  A plain GEMM is kd=kh=kw=1, s=1, p=0, Di=Do=M, others 1. */
typedef struct ctu_geom {
  int32_t B, Di, Hi, Wi;
  int32_t Do, Ho, Wo;
  int32_t C1, C2;
  int32_t N;
  int32_t kd, kh, kw;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw;
  int32_t mode;
} ctu_geom;

/* Epilogue of ctu_igemm_nt. */
typedef struct ctu_epilogue {
  const float* bias;    /* [N] fp32 or NULL                                  */
  const void* residual; /* [M][ldc] same dtype as out, added after act, or NULL; with a split output (n_split > 0) it is
                         * added to the columns that go to `out` only */
  int32_t act;          /* 0 none, 1 exact-erf GELU, 2 GELU backward: out = (a.w) * GELU'(residual) - `residual` holds the
                         * pre-activation the forward pass saved (pre_out) and is NOT added; plain bf16 LDS-DMA GEMM, no
                         * bias, no split-K: the data gradient of the Linear behind a GELU (vit.py:37-39,
                         * hybrid_CTUNet.py:519-522) leaves the GEMM already multiplied, no elementwise pass in between */
  int32_t ldc;          /* leading dimension (elements) of out                */
  void* out2;           /* second destination for columns >= n_split, or NULL  */
  int32_t n_split;      /* 0 = unused; multiple of 8                           */
  int32_t ldc2;
  /* transposed-convolution scatter (kernel == stride, nn.ConvTranspose3d at hybrid_CTUNet.py:177-185):
   * column n = tap*n_per_tap + co ; row m = (b,d,h,w) of the sc_D x sc_H x sc_W input grid is written to
   * output voxel (b, d*kd+td, h*kh+th, w*kw+tw), channel co.  scatter = 0 disables. */
  int32_t scatter;
  int32_t n_per_tap;
  int32_t sc_D, sc_H, sc_W, sc_kd, sc_kh, sc_kw;
  /* split-K for GEMMs with few output tiles and a long reduction (the ViT trunk: 864 tokens, K up to 3072):
   * splitk > 1 with a ZEROED fp32 workspace [M][N] makes `splitk` workgroups per tile sum partial tiles into the
   * workspace; a second kernel applies bias/act/residual, writes `out` and hands the workspace back ZEROED (one
   * persistent workspace serves every call on a stream without a memset).  Plain epilogue only. */
  int32_t splitk;
  /* W layout of a plain GEMM (1 tap, stride 1): 0 = [N][K] (row per output column), 1 = [K][N] (reduction-major):
   * the data gradient of a Linear / 1x1x1 conv then reads the forward weight as stored instead of a transposed copy.
   * Only with dtype bf16, K % 64 == 0 and N % 8 == 0 (the LDS-DMA GEMM); anything else is CTU_ERR_ARG. */
  int32_t w_kn;
  float* splitk_ws;
  /* fused InstanceNorm statistics (plain bf16 LDS-DMA GEMM without split-K only, else CTU_ERR_ARG): in_acc fp64
   * [M / in_rows][N][2] += (sum, sum of squares) of the fp32 results per batch item of in_rows rows (in_rows % 128 == 0,
   * M % in_rows == 0) and output column; finish with ctu_in_finalize. */
  double* in_acc;
  int32_t in_rows;
  int32_t reserved_;
  /* act == 1 only, same restrictions as in_acc: also store the PRE-activation (x + bias) to pre_out [M][ldc] - what
   * the GELU backward needs - so training needs no separate activation pass over the hidden tensor. */
  void* pre_out;
} ctu_epilogue;

/* K1/K3/K5/K2/K4 forward and data-gradient:  out[m][n] = sum_tap sum_c A[gather(m,tap)][c] * W[tap][n][c]
 * (+bias, GELU, +residual).  W is the packed [taps][N][C1+C2] panel in `dtype` (see ctu_permute3).
 * Replaces nn.Conv3d 3x3x3/1x1x1/strided (resnet.py:96-100,150-155,197; hybrid_CTUNet.py:57-83), nn.Linear
 * (vit.py:36,39,59,62,117; hybrid_CTUNet.py:402,457,465,519,522,632-633,641,679), nn.ConvTranspose3d
 * (hybrid_CTUNet.py:177-185,232-240,286-294) and their input gradients.  C1,C2,N multiples of 8. */
int ctu_igemm_nt(ctu_dtype dtype, const void* a1, const void* a2, const void* w, void* out,
                 const ctu_geom* g, const ctu_epilogue* ep, ctu_stream_t stream);

/* Weight gradient:  dw[tap][n][c] += sum_m P[m][n] * Q[gather(m,tap)][c]   (fp32 atomics into a zeroed panel).
 * P has g->N columns (leading dim ldp) over the row space, Q1/Q2 are the gathered tensors (C1/C2 channels).
 * bias_grad (optional, fp32 [N], zeroed): += column sums of P, i.e. the bias gradient of the same layer, computed
 * from the P vectors the kernel stages anyway (no second pass over dY).
 * ws (optional, fp32, ws_floats entries, contents irrelevant): scratch for a two-stage reduction, used when the panel
 * is small (taps*N*C <= 2^18) and the row space is split >= 8 ways - partial panels are stored there and summed by a
 * second pass instead of hundreds of atomics landing on the same addresses. */
int ctu_igemm_tn(ctu_dtype dtype, const void* p, int32_t ldp, const void* q1, const void* q2, float* dw,
                 float* bias_grad, const ctu_geom* g, float* ws, int64_t ws_floats, ctu_stream_t stream);

/* Patch matrix of a one-channel bf16 volume x [B][Di][Hi][Wi]: P[m][k] (bf16, [M][kpad], kpad % 8 == 0, zero for k >= taps and
 * for padding voxels), m over the Do x Ho x Wo output grid of `g`, k = (td*kh + th)*kw + tw.  With it the Cin == 1
 * convolutions run as plain GEMMs (ctu_igemm_nt / ctu_igemm_tn on P). */
int ctu_im2col_cin1(const void* x, void* P, const ctu_geom* g, int32_t kpad, ctu_stream_t stream);

/* Cin == 1 convolutions (vit_encoder0.conv1 1->64 3x3x3, hybrid_CTUNet.py:57-65; ResNet stem 7x7x7 s(2,2,1),
 * resnet.py:150-155).  x: [B][Di][Hi][Wi] ; w: fp32 [taps][N] ; out: [M][N].  kernel 1x1x1 (ResBlock.conv3 shortcut), 3x3x3 or 7x7x7. */
int ctu_conv_cin1_fwd(ctu_dtype dtype, const void* x, const float* w, void* out, const ctu_geom* g,
                      ctu_stream_t stream);
int ctu_conv_cin1_wgrad(ctu_dtype dtype, const void* x, const void* dy, float* dw, const ctu_geom* g,
                        ctu_stream_t stream);

/* K1 fast path: 3x3x3 / stride 1 / padding 1 convolution with an LDS-resident halo brick (4x8x8 output voxels per
 * workgroup, 27 taps served from one staged 6x10x10 halo per 32-channel chunk; weights streamed in MFMA-fragment
 * order).  Forward of ResBlock.conv1/conv2 (hybrid_CTUNet.py:57-74) and Bottleneck.conv2 (resnet.py:98) when the
 * stride is 1, and - with a flipped/transposed panel - their input gradient.  x1/x2: [B][D][H][W][C1|C2] (C multiples
 * of 32); wfrag: panel from ctu_pack_frag; out: [rows][ldc]; columns >= n_split go to out2 (ldc2) when n_split > 0.
 * in_acc (optional, fp64 [B][N][2], bf16 / n_split == 0 only): += (sum y, sum y^2) of the fp32 accumulators per batch
 * item and output channel - the InstanceNorm statistics of the layer that follows (resnet.py:97-99), so that no
 * separate pass re-reads the output; turn them into (mean, rstd) with ctu_in_finalize.
 * residual (optional, [rows][ldc] bf16, no in_acc): added to the part of the result that goes to `out` - used by the data
 * gradient to fold in a gradient that reached the same tensor through another branch (identity shortcut of a ResBlock);
 * residual2 (optional, [rows][ldc2], n_split > 0): the same for the part that goes to `out2` (a block whose shortcut
 * convolution reads the same channel-concatenated pair of tensors as its first convolution).
 * ws (optional fp32 scratch, ws_floats entries, contents irrelevant): volumes of a few bricks (the 12x12x24 and
 * 6x6x12 stages) split their input channels over workgroups, keep fp32 partial outputs there and sum them in a
 * second pass (bf16, n_split == 0).
 * x1_layout: ctu_layout of x1 (x2 is always channels-last). */
int ctu_conv3_halo(ctu_dtype dtype, const void* x1, const void* x2, const void* wfrag, void* out, void* out2,
                   int32_t B, int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N, int32_t n_split,
                   int32_t ldc, int32_t ldc2, double* in_acc, const void* residual, const void* residual2, float* ws,
                   int64_t ws_floats, int32_t x1_layout, ctu_stream_t stream);
/* Weight gradient of the same convolution with the halo staged once per brick:
 * dw[27][N][C1+C2] += sum_v dy[v][n] * x[v + tap - 1][c]  (added into the fp32 panel). dy: [B][D][H][W][N].
 * ws (optional fp32 scratch, ws_floats entries, contents irrelevant): with room for one partial panel per brick split
 * (256 workgroups x 54 tiles: 14.2 M floats cover every shape) the splits store plain partial panels and a second kernel
 * adds them into dw; without it every workgroup adds its tiles with fp32 atomics. */
int ctu_conv3_halo_wgrad(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw, int32_t B,
                         int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N, int32_t x1_layout,
                         int32_t dy_layout, float* ws, int64_t ws_floats, ctu_stream_t stream);
/* The same with the result ADDED straight into the parameter's own layout dw_param[N][C1+C2][27] (nn.Conv3d weight,
 * resnet.py:35-50): the per-split partial panels are summed and transposed by one reduce kernel - no [27][N][K] panel, no
 * permute pass.  bf16 LDS-DMA kernel only; ws must hold splits x 27 x N x (C1+C2) floats (<= 256 x 54 x 1024 for every
 * shape), else CTU_ERR_ARG. */
int ctu_conv3_halo_wgrad_param(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw_param, int32_t B,
                               int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N, int32_t x1_layout,
                               int32_t dy_layout, float* ws, int64_t ws_floats, ctu_stream_t stream);
/* Pack fp32 weights W(n, c, tap) = src[n*sn + c*sc + tap*st] into MFMA-fragment order
 * dst[K/32][taps][2][ceil(N/32)][64 lanes][8] (zero padded), optionally with the tap order reversed (flip = 1). */
int ctu_pack_frag(const float* src, void* dst, ctu_dtype dst_dtype, int32_t N, int32_t K, int32_t taps, int64_t sn,
                  int64_t sc, int64_t st, int32_t flip, ctu_stream_t stream);
/* The same for many panels in one launch: `jobs_dev` is a DEVICE array of `count` jobs (fields as the arguments of
 * ctu_pack_frag; ntn = ceil(N / 32), total = (K / 32) * taps * 2 * ntn * 512 elements of dst).  Job j owns the workgroups
 * [block0_j, block0_{j+1}) of a flat grid of `total_blocks` (block0 ascending, block0_0 = 0). */
typedef struct ctu_pack_job {
  const float* src;
  void* dst;
  int64_t sn, sc, st, total, block0;
  int32_t N, K, taps, flip, ntn, dst_dtype;
} ctu_pack_job;
int ctu_pack_frag_batched(const ctu_pack_job* jobs_dev, int32_t count, int64_t total_blocks, ctu_stream_t stream);

/* Strided 3-index permute + cast: dst[i0*d0 + i1*d1 + i2*d2] = (dst_dtype) src[i0*s0 + i1*s1 + i2*s2].
 * src is fp32 (master weights / packed fp32 gradients).  Used to pack weights into [taps][N][K] panels and
 * to unpack panel gradients back into the nn.Module's parameter layout (accumulate=1 adds into dst, fp32;
 * accumulate=2 additionally writes zeros back to every src element it read: a persistent scratch panel is handed
 * back clean, no memset launch). */
int ctu_permute3(float* src, void* dst, ctu_dtype dst_dtype, int64_t n0, int64_t n1, int64_t n2,
                 int64_t s0, int64_t s1, int64_t s2, int64_t d0, int64_t d1, int64_t d2, int32_t accumulate,
                 ctu_stream_t stream);
/* column sums (bias gradients): out[n] += sum_m s[m] * x[m][n], x is [M][ld]; s = row_scale [M] (same dtype) or 1 when
 * NULL.  With s = the one-channel image this is the weight gradient of a 1x1x1 Cin = 1 convolution. */
int ctu_colsum(ctu_dtype dtype, const void* x, const void* row_scale, int64_t M, int32_t N, int32_t ld, float* out,
               ctu_stream_t stream);
/* out[m][n] = x[m] * w[n] (w fp32): forward of a 1x1x1 convolution of a one-channel volume. */
int ctu_outer_rows(ctu_dtype dtype, const void* x, const float* w, void* out, int64_t M, int32_t N, ctu_stream_t stream);

/* K6/K7 InstanceNorm3d (eps 1e-5, no affine) fused with residual add and LeakyReLU(0.01)
 * (resnet.py:97-124,156-157,198; hybrid_CTUNet.py:84-104).  x: [B][S][C]; acc_ws: fp64 [B][C][2] workspace, zero on
 * entry and handed back ZEROED (one persistent workspace serves every call on a stream without a memset); stats:
 * fp32 [B][C][2] receives (mean, rstd).  Sums are shifted by the channel's first voxel and
 * accumulated in fp64 (no E[x^2]-E[x]^2 cancellation; the deep IN stack amplifies statistic noise ~1000x).  y = act((x-mean)*rstd + residual). */
int ctu_in_stats(ctu_dtype dtype, const void* x, int32_t B, int64_t S, int32_t C, double* acc_ws, float* stats,
                 ctu_stream_t stream);
/* (mean, rstd) from UNSHIFTED fp64 sums (sum x, sum x^2) accumulated by a producer (ctu_conv3_halo in_acc); acc is
 * handed back zeroed. */
int ctu_in_finalize(int32_t B, int64_t S, int32_t C, double* acc, float* stats, ctu_stream_t stream);
/* y_layout: ctu_layout of y (CTU_LAYOUT_B16 when the one consumer of y is ctu_conv3_halo; y must not alias x then). */
/* sign_mask (optional, B*S*C/8 bytes): bit e of byte (voxel, C/8 group) = pre-activation value of channel 8 group + e > 0.
 * With a residual the sign of the activation's argument is recorded nowhere else but in y; the backward kernels then read
 * this byte instead of a 16-byte vector of y (twice). */
int ctu_in_apply(ctu_dtype dtype, const void* x, const float* stats, const void* residual, void* y, int32_t B,
                 int64_t S, int32_t C, int32_t act, int32_t y_layout, uint8_t* sign_mask, ctu_stream_t stream);
/* ctu_in_apply with the finalize step folded in (launch lists: one launch per norm instead of two).  raw_acc (optional):
 * UNSHIFTED fp64 sums (sum x, sum x^2) [B][C][2] from a producer's epilogue (ctu_conv3_halo / ctu_igemm_nt in_acc); (mean, rstd)
 * are derived from them inside the kernel and WRITTEN to stats (the backward pass reads them there); with raw_acc == NULL
 * stats is an input as in ctu_in_apply.  raw_acc is left as it is: clear_ws[0..clear_n) (optional, != raw_acc) names another
 * accumulator that no launch still reads - the previous norm's - and is zeroed here, so two accumulators alternate
 * without a memset or finalize launch. */
int ctu_in_apply_acc(ctu_dtype dtype, const void* x, const double* raw_acc, float* stats, const void* residual, void* y,
                     int32_t B, int64_t S, int32_t C, int32_t act, int32_t y_layout, uint8_t* sign_mask, double* clear_ws,
                     int32_t clear_n, ctu_stream_t stream);
/* The last norm of a block whose shortcut is conv + norm, applied TOGETHER with the shortcut's norm (resnet.py:122-124 with the
 * downsample of :196-199; hybrid_CTUNet.py:99-104): y = act((x - mean) * rstd + (x2 - mean2) * rstd2).  The normalised shortcut is
 * never materialised.  raw_acc / raw_acc2 as in ctu_in_apply_acc (NULL: stats / stats2 are inputs); both (mean, rstd) tables are
 * written; clear_ws / clear_ws2: the accumulators of the previous main / shortcut norm, zeroed here (may be NULL). */
int ctu_in_apply_dual(ctu_dtype dtype, const void* x, const double* raw_acc, float* stats, const void* x2, const double* raw_acc2,
                      float* stats2, void* y, int32_t B, int64_t S, int32_t C, int32_t act, uint8_t* sign_mask, double* clear_ws,
                      int32_t clear_n, double* clear_ws2, int32_t clear_n2, ctu_stream_t stream);
/* backward: g = dy * act'(y) (y may be NULL when no residual was added: then sign(y) == sign(xhat) and the third
 * input stream is skipped; with sign_mask from ctu_in_apply, y is not read either); sums[b][c] = (sum g, sum g*xhat), fp64, zero on entry;
 * dx = rstd*(g - s1/S - xhat*s2/S); dres = g when dres != NULL.  ctu_in_bwd_apply also zeroes clear_ws[0..clear_n)
 * (optional, must differ from sums): pass the sums buffer of the PREVIOUS call on the stream so two buffers can
 * alternate without a memset launch.  dx_layout: ctu_layout of dx (CTU_LAYOUT_B16 when x is the output of a
 * ctu_conv3_halo convolution, whose data- and weight-gradient kernels are the only readers of dx). */
int ctu_in_bwd_reduce(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats,
                      double* sums, int32_t B, int64_t S, int32_t C, int32_t act, const uint8_t* sign_mask,
                      ctu_stream_t stream);
int ctu_in_bwd_apply(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats,
                     const double* sums, void* dx, void* dres, int32_t B, int64_t S, int32_t C, int32_t act,
                     double* clear_ws, int32_t clear_n, int32_t dx_layout, const uint8_t* sign_mask, ctu_stream_t stream);

/* ctu_in_bwd_reduce + ctu_in_bwd_apply in ONE launch, for tensors whose two launches cost more than their bytes (the callers
 * use it up to 32 MB; resnet.py:106-126 and hybrid_CTUNet.py:93-105 backward).  Same arguments, same arithmetic and the same
 * alternating-sums protocol; the grid is at most CTU_IN_FUSED_MAX_WG workgroups, all resident, which meet per batch item at a
 * counter between the two phases.  sync_ws: 2 * B uint32 words, zero before the first launch and handed back zeroed (one
 * per stream).  A wait longer than 50 ms gives up and is counted (ctu_sync_timeouts). */
#define CTU_IN_FUSED_MAX_WG 256
int ctu_in_bwd_fused(ctu_dtype dtype, const void* dy, const void* x, const void* y, const float* stats, double* sums,
                     void* dx, void* dres, int32_t B, int64_t S, int32_t C, int32_t act, double* clear_ws, int32_t clear_n,
                     int32_t dx_layout, const uint8_t* sign_mask, uint32_t* sync_ws, ctu_stream_t stream);
/* in-kernel waits that gave up since the library was loaded (0 in a healthy run); synchronises the device */
int ctu_sync_timeouts(void);

/* K8 LayerNorm (eps 1e-5, affine) (vit.py:35,55,116,118; hybrid_CTUNet.py:456,518,630-631).
 * x,y: [rows][dim]; mean_rstd: fp32 [rows][2]; dgamma/dbeta: fp32 [dim], accumulated (atomics). */
int ctu_layernorm_fwd(ctu_dtype dtype, const void* x, const float* gamma, const float* beta, void* y,
                      float* mean_rstd, int64_t rows, int32_t dim, ctu_stream_t stream);
/* ws: fp32 workspace of CTU_LN_BWD_MAX_BLOCKS * 2 * dim floats (per-workgroup partial column sums; a second-stage
 * kernel adds them into dgamma/dbeta - thousands of workgroups adding atomically into the same few hundred addresses
 * serialise at the memory side). */
#define CTU_LN_BWD_MAX_BLOCKS 1024
int ctu_layernorm_bwd(ctu_dtype dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd,
                      void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int32_t dim, ctu_stream_t stream);
/* dx = LayerNorm backward + dx_add (same shape / dtype as dx, may be NULL): the gradient that reached x through the
 * residual branch around the norm (x = f(LN(x)) + x, vit.py:93-96, hybrid_CTUNet.py:434-440) joins here instead of in an
 * elementwise pass of autograd. */
int ctu_layernorm_bwd_add(ctu_dtype dtype, const void* dy, const void* x, const float* gamma, const float* mean_rstd,
                          const void* dx_add, void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int32_t dim,
                          ctu_stream_t stream);

/* K12 GELU (exact erf) and plain adds (vit.py:37; hybrid_CTUNet.py:520; Residual :434-440). n multiple of 8. */
int ctu_gelu_fwd(ctu_dtype dtype, const void* x, void* y, int64_t n, ctu_stream_t stream);
int ctu_gelu_bwd(ctu_dtype dtype, const void* dy, const void* x, void* dx, int64_t n, ctu_stream_t stream);
int ctu_add(ctu_dtype dtype, const void* a, const void* b, void* y, int64_t n, ctu_stream_t stream);
/* y[r][c] = a[r][c] + bcast[r % period][c]  (vit.py:133 pos_embedding); fp32 bcast */
int ctu_add_bcast(ctu_dtype dtype, const void* a, const float* bcast, void* y, int64_t rows, int32_t cols,
                  int64_t period, ctu_stream_t stream);

/* K9/K10 multi-head self-attention core on a fused qkv matrix [rows][3*heads*dh] (q|k|v, heads inside each).
 * Token partition (which rows form one attention group of ntok tokens):
 *   part 0: contiguous: group g = rows [g*ntok, (g+1)*ntok)                      (vit.py:66-78)
 *   part 1: block windows of win^3 voxels of a [B][D][H][W] grid  '(h h1)'       (hybrid_CTUNet.py:559)
 *   part 2: grid  windows (dilated, stride D/win)                 '(h1 h)'       (hybrid_CTUNet.py:564)
 * s = scale*(q.k) + bias_table[relidx(i,j)][head] (bias_table NULL for ViT; [ (2win-1)^3 ][heads] fp32,
 * hybrid_CTUNet.py:470-500); softmax over keys; out = P.v written to out[rows][heads*dh].
 * lse: fp32 [groups*heads][ntok] (log-sum-exp, saved for backward). */
typedef struct ctu_attn_geom {
  int32_t part;        /* 0,1,2 */
  int32_t B, D, H, W;  /* volume grid (part 1,2); for part 0: B groups, D*H*W = ntok */
  int32_t win;         /* window edge (6) for part 1,2 */
  int32_t heads, dh;   /* dh in {32, 64} */
  float scale;
} ctu_attn_geom;
int ctu_attn_fwd(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse,
                 const ctu_attn_geom* g, ctu_stream_t stream);
/* backward: dqkv [rows][3*heads*dh]; dbias fp32 accumulated (atomics) or NULL. */
int ctu_attn_bwd(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out, const void* dout,
                 const float* lse, void* dqkv, float* dbias, const ctu_attn_geom* g, ctu_stream_t stream);

/* Fused FeedForward forward of the 128-wide token stages (hybrid_CTUNet.py:513-526 under Residual :434-440):
 * y = x + W2 gelu(W1 LayerNorm(x) + b1) + b2 in one kernel; the normalised rows and the hidden activations stay in registers
 * between the two products.  x, y: [M][128] bf16; w1: [Hd][128] bf16 row-major; w2_frag: W2 ([128][Hd] bf16) re-ordered by
 * ctu_ff_pack_w2 (once per optimizer step); pre, u: [M][Hd] bf16 - the pre-activation and gelu(pre) the backward pass reads
 * (GELU', operand of W2's weight gradient), both NULL for inference (nothing but y and the statistics is written); mean_rstd:
 * fp32 [M][2] as ctu_layernorm_fwd writes it.  M % 256 == 0, D == 128, Hd % 64 == 0. */
int ctu_ff_pack_w2(const void* w2, void* w2_frag, int32_t D, int32_t Hd, ctu_stream_t stream);
int ctu_ff_fwd(ctu_dtype dtype, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
               const void* w2_frag, const float* b2, void* y, void* pre, void* u, float* mean_rstd, int64_t M, int32_t D,
               int32_t Hd, ctu_stream_t stream);

/* pixelweight_attention.forward for C == 128 in one kernel (hybrid_CTUNet.py:645-669: norm1 / norm2, to_qkv1 / to_qkv2, the
 * cross-weight core below, to_out[0]): out = Wo . mix(Wq1 LN1(x1), Wq2 LN2(x2)).  x1, x2, out: [M][128] bf16.
 * ctu_pwa_pack: wq1, wq2 = to_qkv1/2.weight [384][128] bf16 row-major (rows q | k | v), wo = to_out[0].weight [128][128] bf16
 * -> packed, 4 * 56 * 512 bf16 (224 KiB): the MFMA fragments of the three matrices in the order the kernel streams them, per
 * head (once per optimizer step).
 * ctu_pwa_block_fwd: qkv1, qkv2: [M][384] bf16 - the projections the backward pass reads (both or neither; NULL: inference,
 * nothing saved); mean_rstd1/2: fp32 [M][2] as ctu_layernorm_fwd writes them.  M % 128 == 0. */
int ctu_pwa_pack(const void* wq1, const void* wq2, const void* wo, void* packed, int32_t C, ctu_stream_t stream);
int ctu_pwa_block_fwd(ctu_dtype dtype, const void* x1, const void* x2, const float* g1, const float* b1, const float* g2,
                      const float* b2, const void* w_packed, void* out, void* qkv1, void* qkv2, float* mean_rstd1,
                      float* mean_rstd2, int64_t M, int32_t C, float scale, ctu_stream_t stream);

/* K11 binary cross-weight fusion core (hybrid_CTUNet.py:651-665): per token, per head of 32 channels:
 * a1 = sigmoid(scale*(<q2,k1> - <q1,k2>)); out = a1*v1 + (1-a1)*v2.  qkv1,qkv2: [rows][3*C]; out: [rows][C]. */
int ctu_pwa_fwd(ctu_dtype dtype, const void* qkv1, const void* qkv2, void* out, int64_t rows, int32_t C,
                float scale, ctu_stream_t stream);
int ctu_pwa_bwd(ctu_dtype dtype, const void* qkv1, const void* qkv2, const void* dout, void* dqkv1, void* dqkv2,
                int64_t rows, int32_t C, float scale, ctu_stream_t stream);

/* Stride-2 helpers of the stage-transition data gradients (resnet.py:98 conv2 with stride 2; resnet.py:166-176 downsample).
 * upsample2_zeros: x [B][D][H][W][C] -> y [B][2D][2H][2W][C], y[b][2d][2h][2w] = x[b][d][h][w], zero elsewhere: the data gradient
 *   of a 3x3x3 stride-2 padding-1 convolution is ctu_conv3_halo(y, flipped weights) at the input's size.
 * add_strided2: y[b][2d][2h][2w][:] += x[b][d][h][w][:] for y [B][2D][2H][2W][C]: adds the (compact) data gradient of a 1x1x1
 *   stride-2 convolution - a plain GEMM over its output rows - into the gradient of the tensor it read.  C % 8 == 0. */
int ctu_upsample2_zeros(ctu_dtype dtype, const void* x, void* y, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                        ctu_stream_t stream);
int ctu_add_strided2(ctu_dtype dtype, void* y, const void* x, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                     ctu_stream_t stream);

/* K13 layout ops.  patchify: x [B][H][W][F] -> tokens [B][(H/p1)(W/p2)(F/p3)][p1*p2*p3]   (vit.py:115, c=1).
 * pixel shuffle: x [B][D][H][W][c*p1*p2*p3] -> y [B][D*p1][H*p2][W*p3][c]  (hybrid_CTUNet.py:420-428);
 * inverse = 1 applies the inverse map (its gradient). */
int ctu_patchify(ctu_dtype dtype, const void* x, void* tokens, int32_t B, int32_t H, int32_t W, int32_t F,
                 int32_t p1, int32_t p2, int32_t p3, ctu_stream_t stream);
int ctu_pixel_shuffle(ctu_dtype dtype, const void* x, void* y, int32_t B, int32_t D, int32_t H, int32_t W,
                      int32_t c, int32_t p1, int32_t p2, int32_t p3, int32_t inverse, ctu_stream_t stream);

/* K14 DiceCE (MONAI DiceCELoss(to_onehot_y, softmax, squared_pred, smooth_nr=0, smooth_dr), main_CTUNet.py:156-158)
 * with the nearest-neighbour deep-supervision target map of trainer_CTUNet.py:93-94 fused in.
 * logits: [B][D][H][W][ldl] (first n_cls columns valid); labels: fp32 class ids [B][LD][LH][LW] (full res);
 * idx_d/h/w: int32 source index per logit coordinate (length D,H,W).  acc: fp32 [B][n_cls][3] + [1]
 * (sum p*y, sum p^2, sum y, then CE sum at acc[B*n_cls*3]), zeroed by the caller.
 * loss_out[0] += weight * (dice + ce)  (device scalar, fp32). */
int ctu_dicece_fwd(ctu_dtype dtype, const void* logits, int32_t ldl, const float* labels, const int32_t* idx_d,
                   const int32_t* idx_h, const int32_t* idx_w, int32_t B, int32_t D, int32_t H, int32_t W,
                   int32_t LD, int32_t LH, int32_t LW, int32_t n_cls, float* acc, ctu_stream_t stream);
int ctu_dicece_finalize(const float* acc, int32_t B, int32_t n_cls, int64_t S, float smooth_nr, float smooth_dr,
                        float weight, float* loss_out, ctu_stream_t stream);
/* dlogits[..][ldl] = weight * gscale[0] * d(dice+ce)/dlogits (pad columns written as 0). gscale: device fp32 or NULL (=1). */
int ctu_dicece_bwd(ctu_dtype dtype, const void* logits, int32_t ldl, const float* labels, const int32_t* idx_d,
                   const int32_t* idx_h, const int32_t* idx_w, int32_t B, int32_t D, int32_t H, int32_t W,
                   int32_t LD, int32_t LH, int32_t LW, int32_t n_cls, const float* acc, float smooth_nr,
                   float smooth_dr, float weight, const float* gscale, void* dlogits, ctu_stream_t stream);

/* K15 fused AdamW over one flat fp32 buffer (torch.optim.AdamW semantics, main_CTUNet.py:192-193).
 * skip: up to 16 [begin,end) element ranges left untouched (parameters whose .grad is None this step,
 * trainer_CTUNet.py:88-89 + torch's "skip if grad is None").  mirror_bf16 (optional, bf16 [n]): receives the updated
 * parameters rounded to bf16 - the GEMM kernels then read layer weights from it and no per-step cast pass exists.
 * hyper_dev (optional, device, 4 x 32 bit: lr, step count as int32, 1/(1-beta1^step), 1/sqrt(1-beta2^step)): when given, lr
 * and the bias corrections are read from it instead of the arguments and `step` is ignored; ctu_adamw_tick advances the
 * step count and refreshes the corrections on the device - a captured HIP graph of the training step then replays
 * correctly, and the host changes the learning rate by rewriting hyper_dev[0]. */
int ctu_adamw(float* p, const float* g, float* m, float* v, void* mirror_bf16, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int32_t step, const int64_t* skip_host, int32_t n_skip,
              const float* hyper_dev, ctu_stream_t stream);
int ctu_adamw_tick(float* hyper_dev, float beta1, float beta2, ctu_stream_t stream);
/* ---- inference-side callers of forward ----------------------------------------------------
 * Sliding-window accumulation (trainer_CTUNet.py:538-548, trainer_CUNet.py:386-392): for one window whose prediction
 * element (c,d,h,w) is logits[c*sc + d*sd + h*sh + w*sw] (dtype fp32 or bf16, any strides - the models return
 * channels-last views), out[b][c][d0+d][h0+h][w0+w] += importance[d][h][w] * logits and, when count != NULL,
 * count[b][d0+d][h0+h][w0+w] += importance (one weight map serves every class).  out fp32 [B][C][D][H][W]. */
int ctu_sw_accumulate(ctu_dtype dtype, const void* logits, int64_t sc, int64_t sd, int64_t sh, int64_t sw,
                      const float* importance, float* out, float* count, int32_t C, int32_t rd, int32_t rh, int32_t rw,
                      int32_t b, int32_t d0, int32_t h0, int32_t w0, int32_t D, int32_t H, int32_t W, ctu_stream_t stream);
/* out[b][c][s] /= count[b][s]  (trainer_CTUNet.py:547-548) */
int ctu_sw_normalize(float* out, const float* count, int32_t B, int32_t C, int64_t S, ctu_stream_t stream);
/* Hybrid complementation (test_CTUNet_final.py:545-551): labels of softmax(p1), softmax(p2) and of their average, p1/p2
 * fp32 [C][S], C <= 32; labels1 / labels2 may be NULL.  First maximum wins (torch.argmax). */
int ctu_hybrid_argmax(const float* p1, const float* p2, int32_t C, int64_t S, int64_t* labels1, int64_t* labels2,
                      int64_t* labels_hybrid, ctu_stream_t stream);

/* Dropout (reference: nn.Dropout at networks/vit.py:38,40,57,63,74 and
 * networks/hybrid_CTUNet.py:459-467,521-523).  Masks are a pure function of (seed, offset, element index) through
 * Philox4x32-10 with 16-bit draws (keep iff draw >= round(p * 65536); survivors scaled by 65536 / (65536 - thr)), so the
 * backward pass calls the same entry point on the gradient with the forward's (p, seed, offset) and nothing is stored.
 * `offset` identifies the dropout call (low 32 bits used).  y = dropout(x) + residual (residual may be NULL: the
 * reference's `x = attn(x) + x` with the dropout inside attn).  x, residual, y: n elements of `dtype`, 16-byte aligned;
 * y may alias x. */
int ctu_dropout(ctu_dtype dtype, const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed,
                uint64_t offset, ctu_stream_t stream);
/* ctu_attn_fwd / ctu_attn_bwd with dropout of the attention probabilities (after the softmax, before P.v; the reference's
 * self.dropout(attn) / nn.Sequential(Softmax, Dropout)).  MFMA kernels only: bf16, or fp32 with <= 224 tokens. */
int ctu_attn_fwd_dropout(ctu_dtype dtype, const void* qkv, const float* bias_table, void* out, float* lse,
                         const ctu_attn_geom* g, float p, uint64_t seed, uint64_t offset, ctu_stream_t stream);
int ctu_attn_bwd_dropout(ctu_dtype dtype, const void* qkv, const float* bias_table, const void* out, const void* dout,
                         const float* lse, void* dqkv, float* dbias, const ctu_attn_geom* g, float p, uint64_t seed,
                         uint64_t offset, ctu_stream_t stream);
/* Verification hook: the keep flags the attention kernels apply, keep[pairs = groups*heads][ntok][ntok], one byte each. */
int ctu_attn_dropout_mask(uint8_t* keep, int32_t pairs, int32_t ntok, float p, uint64_t seed, uint64_t offset,
                          ctu_stream_t stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI ---------------------------------------------
 * Replaces DistributedDataParallel's bucket all-reduce (main_CTUNet.py:116-118 init_process_group("nccl"), :187-189
 * DDP(model, find_unused_parameters=True)).  One communicator per process (= per GPU), created once:
 *   rank 0:      ctu_comm_unique_id(rccl_path, id)           id: 128 host bytes, sent to the other ranks by the caller
 *   every rank:  ctu_comm_init(rccl_path, rank, world, id, &handle)   on the calling thread's current HIP device
 * rccl_path names the RCCL shared object the process already uses (PyTorch ships its own librccl.so: pass that file so
 * that one RCCL runtime serves both; a pure C++ host passes /opt/rocm/lib/librccl.so.1).  It is opened with dlopen - the
 * library itself does not link RCCL.
 * ctu_allreduce_bucket: buf[n] fp32 (device, 16-byte aligned) <- mean over the ranks, asynchronous on `stream` (the
 * caller's side stream, fenced against the compute stream by events).  payload CTU_F32: one ncclAllReduce(ncclAvg).
 * payload CTU_BF16: cast -> all-to-all (direct reduce-scatter: one message per peer, all 7 links at once) -> fp32 sum of
 * the received chunks, x 1/world -> all-gather of the bf16 means -> expand; every rank ends with identical bf16-rounded
 * means.  scratch: device memory of ctu_allreduce_scratch_bytes(world, n) bytes (unused for CTU_F32, may be NULL). */
int ctu_comm_unique_id(const char* rccl_path, void* id128_host);
int ctu_comm_init(const char* rccl_path, int32_t rank, int32_t world, const void* id128_host, void** handle);
int ctu_comm_destroy(void* handle);
int64_t ctu_allreduce_scratch_bytes(int32_t world, int64_t n);
int ctu_allreduce_bucket(void* handle, float* buf, int64_t n, int32_t payload, void* scratch, int64_t scratch_bytes,
                         ctu_stream_t stream);

/* ---- launch lists (host path) -------------------------------------------------------------------------------------------
 * The reference dispatches one ATen op per Python call (e.g. the nine leaf ops of Bottleneck.forward, networks/resnet.py:
 * 106-126, or the ~20 of its backward); the per-op crossing of the language boundary, not the device, then sets the pace of
 * the small-volume stages.  A plan is a recorded list of calls of THIS library's entry points - the argument blocks of one
 * module's forward or backward - created once and replayed by one call per (module, pass):
 *   words   : commands [opcode, stream index, n, n argument words]...; argument words follow the entry point's prototype
 *             (one 8-byte word per scalar / pointer, floats as their bit pattern, ctu_geom / ctu_epilogue / ctu_attn_geom
 *             inline); opcodes are the indices of csrc/plan_dispatch.inc (generated from this header), 1000 = record event
 *             w[0] on the stream, 1001 = make the stream wait for event w[0].
 *   patches : triples (word position, slot, byte offset): before a replay word[position] = slots[slot] + offset - the
 *             device pointers (and run-time integers) of this call's tensors.
 * ctu_plan_run calls the entry points in order on streams[stream index]; it stops at the first failing command and
 * reports its index in ctu_last_error().  A plan is not re-entrant (one replay at a time); different plans are independent. */
int ctu_plan_create(const uint64_t* words, int64_t nwords, const uint64_t* patches, int64_t npatches, int32_t nevents,
                    int32_t nslots, void** handle);
int ctu_plan_run(void* handle, const uint64_t* slots, int32_t nslots, void* const* streams, int32_t nstreams);
int ctu_plan_destroy(void* handle);

/* Verification hook: one local stage of the CTU_BF16 exchange as rank-independent kernels (0 cast, 1 reduce, 2 expand) on
 * the scratch layout of ctu_allreduce_bucket - send [world][chunk] | recv [world][chunk] | mean [chunk], bf16, chunk =
 * ceil(n / world) rounded up to 8 - so that a test can play the two collectives with copies on one GPU. */
int ctu_allreduce_bucket_stage(int32_t stage, int32_t world, float* buf, int64_t n, void* scratch, int64_t scratch_bytes,
                               ctu_stream_t stream);

/* fp32 <-> dtype casts and fills */
int ctu_cast(const void* src, ctu_dtype src_dtype, void* dst, ctu_dtype dst_dtype, int64_t n, ctu_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CTUNET_HIP_H */
