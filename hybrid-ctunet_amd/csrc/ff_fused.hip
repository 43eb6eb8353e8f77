// Fused FeedForward forward for the wide token stages (hybrid_CTUNet.py:513-526 under Residual :434-440; vit.py:31-44):
//     y = x + W2 gelu(W1 LayerNorm(x) + b1) + b2,   x [M][128] bf16, hidden width Hd (a multiple of 64)
// in ONE kernel per call instead of LayerNorm + two GEMMs: the normalised rows never leave registers, the hidden activations
// go from the accumulators of the first GEMM straight into the second as its operand, and of the hidden tensors only what the
// backward pass reads (`pre` for GELU', `u` as the operand of W2's weight gradient) is written - once, as whole 128-byte row
// segments.  Per 442 368-row call: 113 MB read + 1.13 GB written instead of 1.25 GB read + 1.13 GB written over three launches.
//
// Shape of the computation.  A workgroup (8 waves) walks 256-row tiles; wave w owns rows 32 w .. 32 w + 31 of the tile for the
// whole FeedForward.  Both products are taken TRANSPOSED (the weight fragment is the MFMA's A operand, the activations its B
// operand), so a lane always holds values of ONE row:
//   * x: lane (r, hf) loads the 64 channels {16 s + 8 hf + i} of row r (eight 16-byte loads) - exactly the B-operand fragments
//     of the eight k steps; LayerNorm is lane-local sums plus one exchange with lane r + 32.
//   * pre^T tile [32 hidden][32 rows] = W1 fragment x h fragments; the accumulator of lane (r, h) holds hidden units
//     {4 h + 8 q + j}: bias, GELU, and registers 8 a .. 8 a + 7 of the tile ARE the B fragment of k step (tile, a) of the
//     second product - with the hidden index order 4 hf + 16 a + 8 (i >> 2) + (i & 3), which the packed W2 panel mirrors
//     (ctu_ff_pack_w2), so no shuffle and no LDS round trip lies between the two GEMMs.
//   * y^T tiles [32 features][32 rows] accumulate over the hidden chunks.
// Weights stream L2 -> LDS by LDS-DMA in chunks of 64 hidden units (16 W1 fragments gathered from the row-major mirror + 16
// packed W2 fragments = 32 KiB), double buffered, one workgroup barrier per chunk (32 MFMAs per wave); 256-row tiles halve that
// stream against 128-row tiles (0.44 GB per 442 368-row call).  Results leave through a wave-private LDS tile as 16-byte
// lanes of whole row segments; the stores are written in assembly so that the counted vmcnt in front of each barrier knows
// exactly how many vector-memory operations are younger than the awaited weight stage (dma.h).
#include "dma.h"
#include "mma.h"

namespace {

constexpr int FF_D = 128;        // model width this kernel is built for
constexpr int FF_ROWS = 256;     // rows per workgroup tile (8 waves x 32)
constexpr int FF_CH = 64;        // hidden units per weight stage
constexpr int FF_STAGE = 32 * 1024;
constexpr int FF_STG_WAVE = 9216;  // per-wave staging: [32][72] bf16 x 2 (pre, u) or [32][68] fp32 (two y tiles)

struct FfArgs {
  const bf16* x;
  const float* gamma;
  const float* beta;
  const bf16* w1;   // [Hd][128] bf16 (the optimizer's mirror)
  const float* b1;
  const bf16* w2f;  // packed by ctu_ff_pack_w2
  const float* b2;
  bf16* y;
  bf16* pre;
  bf16* u;
  float* mr;
  int64_t M;
  int Hd, nch, ntiles;
};

// (s_nop behind the store: a store of more than 8 bytes reads its data registers over two cycles, and a VALU write to them in the
// next cycle corrupts what some lanes store - hipcc pads real store instructions for this hazard but cannot see into inline
// assembly; found as 64-bit ADDRESSES of the following store in every fourth 8-byte piece of `pre`)
__device__ __forceinline__ void store16_asm(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// GELU (erf form, nn.GELU's default) without branches: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// bf16 rounding of the result); the ocml erff costs ~38 vector instructions per value and a divergent branch - 1 240 vector
// instructions per 32 MFMAs in this kernel, far more than the 32 MFMAs can hide.
// For v < 0 the small factor 1 - erf(|v| / sqrt 2) is used directly (no cancellation in the tail).
__device__ __forceinline__ float gelu_fast(float v) {
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // (v_rcp_f32: 1 ulp; __frcp_rn is a full IEEE division, 6 instructions)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float pe = poly * t * __expf(-ax * ax);   // = 1 - erf(ax)
  const float half = 0.5f * v * pe;
  return v < 0.f ? half : v - half;
}

template <bool SAVE>
__global__ __launch_bounds__(512, 1) void ff_fwd_kernel(const FfArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* wst = smem;                                  // 2 weight stages
  unsigned char* stg_all = smem + 2 * FF_STAGE;               // 8 x FF_STG_WAVE
  float* cst = reinterpret_cast<float*>(stg_all + 8 * FF_STG_WAVE);   // gamma[128] beta[128] b2[128] b1[Hd]
  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < FF_D; i += 512) {
    cst[i] = p.gamma[i];
    cst[FF_D + i] = p.beta[i];
    cst[2 * FF_D + i] = p.b2[i];
  }
  for (int i = tid; i < p.Hd; i += 512) cst[3 * FF_D + i] = p.b1[i];
  const float* gam = cst;
  const float* bet = cst + FF_D;
  const float* b2s = cst + 2 * FF_D;
  const float* b1s = cst + 3 * FF_D;
  unsigned char* stg = stg_all + wave * FF_STG_WAVE;

  // weight stage of hidden chunk c into buffer `buf`: this wave's four of the 32 one-KiB pieces
  auto issue_stage = [&](int c, int buf) {
    unsigned char* dst = wst + buf * FF_STAGE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pc = wave * 4 + k;
      const bf16* g;
      if (pc < 16) {   // W1 fragment (tile pc >> 3, k step pc & 7): lane (m, hf) = 8 channels of hidden unit 64 c + 32 tile + m
        g = p.w1 + (size_t)(FF_CH * c + 32 * (pc >> 3) + r) * FF_D + 16 * (pc & 7) + 8 * hf;
      } else {         // packed W2 fragment
        g = p.w2f + ((size_t)c * 16 + (pc - 16)) * 512 + lane * 8;
      }
      dma16(g, dst + pc * 1024);
    }
  };

  int seq = 0;   // weight stages consumed so far: stage `seq` lives in buffer seq & 1
  issue_stage(0, 0);
  __syncthreads();   // constants in LDS
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t row = (int64_t)tile * FF_ROWS + wave * 32 + r;
    // ---- x rows -> LayerNorm -> B fragments of the first product
    bf16x8 hfrag[8];
    {
      float xs[64];
      const bf16* xr = p.x + (size_t)row * FF_D + 8 * hf;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + 16 * s);
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[8 * s + i] = (float)v[i];
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) sum += xs[i];
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / FF_D);
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i) { const float d = xs[i] - mean; var += d * d; }
      var += __shfl_xor(var, 32, 64);
      const float rstd = rsqrtf(var * (1.0f / FF_D) + NORM_EPS);
      if (hf == 0) {
        p.mr[(size_t)row * 2] = mean;
        p.mr[(size_t)row * 2 + 1] = rstd;
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int c0 = 16 * s + 8 * hf;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + c0), g1 = *reinterpret_cast<const f32x4*>(gam + c0 + 4);
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(bet + c0), e1 = *reinterpret_cast<const f32x4*>(bet + c0 + 4);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[i] = (bf16)((xs[8 * s + i] - mean) * rstd * g0[i] + e0[i]);
          o[4 + i] = (bf16)((xs[8 * s + 4 + i] - mean) * rstd * g1[i] + e1[i]);
        }
        hfrag[s] = o;
      }
    }
    f32x16 accY[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) accY[n][e] = 0.f;

    for (int c = 0; c < p.nch; ++c, ++seq) {
      // stage `seq` was issued one chunk ago (or in the prologue); younger than it at this point: the 8 stores of the
      // previous chunk (and, behind a tile's last chunk, the 8 stores of y) - never fewer, so vmcnt(8) covers the stage
      if (seq == 0) wait_vm_then_barrier<0>();
      else wait_vm_then_barrier<SAVE ? 8 : 0>();   // (nothing saved - inference: no chunk stores, only the 8 stores of y per tile)
      {
        const int cn = c + 1 < p.nch ? c + 1 : 0;   // (past the last tile: one stage nobody reads - keeps the counts static)
        issue_stage(cn, (seq + 1) & 1);
      }
      const unsigned char* wb = wst + (seq & 1) * FF_STAGE + lane * 16;
      bf16x8 ufrag[2][2];
      bf16* spre = reinterpret_cast<bf16*>(stg);
      bf16* su = reinterpret_cast<bf16*>(stg + 4608);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + (t * 8 + s) * 1024);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hfrag[s], acc, 0, 0, 0);
        }
        // lane (r, hf): hidden units 64 c + 32 t + 4 hf + 8 q + j of row r
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + FF_CH * c + 32 * t + 4 * hf + 8 * q);
          float pv[4], uv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            pv[j] = acc[4 * q + j] + bb[j];
#ifdef FF_NO_GELU
            uv[j] = pv[j];
#else
            uv[j] = gelu_fast(pv[j]);
#endif
            acc[4 * q + j] = uv[j];
          }
          if (SAVE) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            bf16x4 pb, ub;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pb[j] = (bf16)pv[j]; ub[j] = (bf16)uv[j]; }
            *reinterpret_cast<bf16x4*>(spre + r * 72 + 32 * t + 4 * hf + 8 * q) = pb;
            *reinterpret_cast<bf16x4*>(su + r * 72 + 32 * t + 4 * hf + 8 * q) = ub;
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          bf16x8 f;
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = (bf16)acc[8 * a + i];
          ufrag[t][a] = f;
        }
      }
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const bf16x8 w = *reinterpret_cast<const bf16x8*>(wb + (16 + (n * 2 + t) * 2 + a) * 1024);
            accY[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, ufrag[t][a], accY[n], 0, 0, 0);
          }
      // pre / u of this chunk leave as 128-byte row segments: lane -> (row 8 i + (lane >> 3), 16-byte group lane & 7)
      if (SAVE) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t row0 = (int64_t)tile * FF_ROWS + wave * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 8 * i + (lane >> 3), cg = lane & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(spre + rr * 72 + cg * 8);
          store16_asm(p.pre + (size_t)(row0 + rr) * p.Hd + FF_CH * c + cg * 8, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 8 * i + (lane >> 3), cg = lane & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(su + rr * 72 + cg * 8);
          store16_asm(p.u + (size_t)(row0 + rr) * p.Hd + FF_CH * c + cg * 8, v);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // ---- y = x + (W2 u) + b2: two feature tiles at a time through the staging tile in fp32 (one rounding, as the GEMM epilogue)
    {
      float* sy = reinterpret_cast<float*>(stg);
      const int64_t row0 = (int64_t)tile * FF_ROWS + wave * 32;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = accY[2 * hp + n2][4 * q + j];
            *reinterpret_cast<f32x4*>(sy + r * 68 + 32 * n2 + 4 * hf + 8 * q) = v;
          }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 8 * i + (lane >> 3), cg = lane & 7;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sy + rr * 68 + cg * 8), v1 = *reinterpret_cast<const f32x4*>(sy + rr * 68 + cg * 8 + 4);
          const int d0 = 64 * hp + cg * 8;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(b2s + d0), c1 = *reinterpret_cast<const f32x4*>(b2s + d0 + 4);
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(p.x + (size_t)(row0 + rr) * FF_D + d0);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = (bf16)(v0[j] + c0[j] + (float)xv[j]);
            o[4 + j] = (bf16)(v1[j] + c1[j] + (float)xv[4 + j]);
          }
          store16_asm(p.y + (size_t)(row0 + rr) * FF_D + d0, __builtin_bit_cast(u32x4, o));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage nobody reads has landed before the workgroup's LDS is released
}

// W2 [D = 128][Hd] row-major bf16 -> MFMA A fragments in the hidden order the first product's accumulators present:
// out[((c * 4 + n) * 2 + t) * 2 + a][lane = (m, hf)][i] = W2[32 n + m][64 c + 32 t + 4 hf + 16 a + 8 (i >> 2) + (i & 3)]
__global__ __launch_bounds__(256) void ff_pack_w2_kernel(const bf16* __restrict__ w2, bf16* __restrict__ out, const int Hd) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // one (fragment, lane) per thread
  const int total = (Hd / FF_CH) * 16 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, frag = idx >> 6;
  const int a = frag & 1, t = (frag >> 1) & 1, n = (frag >> 2) & 3, c = frag >> 4;
  const int m = lane & 31, hf = lane >> 5;
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = w2[(size_t)(32 * n + m) * Hd + FF_CH * c + 32 * t + 4 * hf + 16 * a + 8 * (i >> 2) + (i & 3)];
  *reinterpret_cast<bf16x8*>(out + (size_t)idx * 8) = o;
}

}  // namespace

extern "C" int ctu_ff_pack_w2(const void* w2, void* w2_frag, int32_t D, int32_t Hd, ctu_stream_t stream) {
  CTU_REQUIRE(w2 && w2_frag, "ff_pack_w2: null pointer");
  CTU_REQUIRE(D == FF_D && Hd > 0 && Hd % FF_CH == 0, "ff_pack_w2: D = 128 and Hd %% 64 == 0 (D=%d Hd=%d)", D, Hd);
  const int total = (Hd / FF_CH) * 16 * 64;
  hipLaunchKernelGGL(ff_pack_w2_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)w2, (bf16*)w2_frag, Hd);
  return ctu_check_launch("ff_pack_w2");
}

extern "C" int ctu_ff_fwd(ctu_dtype dtype, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                          const void* w2_frag, const float* b2, void* y, void* pre, void* u, float* mean_rstd, int64_t M,
                          int32_t D, int32_t Hd, ctu_stream_t stream) {
  CTU_REQUIRE(dtype == CTU_BF16, "ff_fwd: bf16 only");
  CTU_REQUIRE(x && gamma && beta && w1 && b1 && w2_frag && b2 && y && mean_rstd, "ff_fwd: null pointer");
  CTU_REQUIRE((pre == nullptr) == (u == nullptr), "ff_fwd: pre and u are saved together or not at all");
  CTU_REQUIRE(D == FF_D && Hd >= 2 * FF_CH && Hd % FF_CH == 0 && Hd <= 4096, "ff_fwd: D = 128, Hd a multiple of 64 in [128, 4096] (D=%d Hd=%d)", D, Hd);
  CTU_REQUIRE(M > 0 && M % FF_ROWS == 0 && M * (int64_t)Hd < (1ll << 40), "ff_fwd: M must be a multiple of 256 (M=%lld)", (long long)M);
  FfArgs p;
  p.x = (const bf16*)x; p.gamma = gamma; p.beta = beta; p.w1 = (const bf16*)w1; p.b1 = b1; p.w2f = (const bf16*)w2_frag; p.b2 = b2;
  p.y = (bf16*)y; p.pre = (bf16*)pre; p.u = (bf16*)u; p.mr = mean_rstd; p.M = M; p.Hd = Hd; p.nch = Hd / FF_CH;
  p.ntiles = (int)(M / FF_ROWS);
  const size_t lds = 2 * FF_STAGE + 8 * FF_STG_WAVE + (3 * FF_D + Hd) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ff_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(ff_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      ctu_set_error("ff_fwd: cannot raise the dynamic LDS limit");
      return CTU_ERR_ARG;
    }
    attr_set = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  if (pre) hipLaunchKernelGGL(ff_fwd_kernel<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ff_fwd_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  return ctu_check_launch("ff_fwd");
}
