// Fused forward of pixelweight_attention for the 128-channel stages (hybrid_CTUNet.py:622-669):
//     out = Wo . cross_weight(Wq1 LayerNorm1(x1), Wq2 LayerNorm2(x2)),      x1, x2, out [M][128] bf16
// in ONE kernel per call instead of two LayerNorms, two [M][128] x [128][384] GEMMs, the cross-weight core and the output
// GEMM: the normalised rows stay in registers as the B operand of all six projections, each head's q / k / v tiles leave the
// accumulators only as the saved copies the backward pass reads (qkv1 / qkv2, optional), and the mixed head goes straight
// from registers into the output projection.  Per 442 368-row call: 226 MB read + 113 MB (+ 680 MB saved projections)
// written, instead of 1.9 GB read + 1.25 GB written over six launches.
//
// Same construction as ff_fused.hip: 8 waves walk 256-row tiles, wave w owns rows 32 w .. 32 w + 31; every product is taken
// transposed (weight fragment = MFMA A operand, activations = B operand), so a lane holds values of ONE row:
//   * lane (r, hf) of a q / k / v tile [32 head channels][32 rows] holds channels {4 hf + 8 q + j} of row r: the two dot
//     products <q2, k1>, <q1, k2> of a head are 16 lane-local products plus one exchange with lane r + 32, the sigmoid
//     and the mix a1 v1 + (1 - a1) v2 are lane-local;
//   * registers 8 a .. 8 a + 7 of the mixed tile are the B fragment of k step (head, a) of the output projection, in the
//     channel order 4 hf + 16 a + 8 (i >> 2) + (i & 3), which the packed Wo fragments mirror.
// Weights stream L2 -> LDS per HEAD: 24 fragments of Wq1 (q, k, v rows of the head x 8 k steps), 24 of Wq2, 8 of Wo = 56 KiB,
// contiguous in the packed panel ctu_pwa_pack writes once per optimizer step; double buffered, one workgroup barrier per
// head (56 MFMAs per wave).
#include "dma.h"
#include "mma.h"

namespace {

constexpr int PW_C = 128;
constexpr int PW_ROWS = 256;
constexpr int PW_STAGE = 56 * 1024;
constexpr int PW_STG_WAVE = 4608;   // [32][72] bf16: 64 output features of the wave's rows

struct PwaArgs {
  const bf16* x1;
  const bf16* x2;
  const float* g1;
  const float* b1;
  const float* g2;
  const float* b2;
  const bf16* wpk;   // the three weight matrices in stage order (ctu_pwa_pack): [head][56 fragments][64 lanes][8]
  bf16* out;
  bf16* qkv1;        // [M][384] or NULL
  bf16* qkv2;
  float* mr1;        // [M][2] (mean, rstd) as ctu_layernorm_fwd writes it
  float* mr2;
  int64_t M;
  int ntiles;
  float scale;
};

__device__ __forceinline__ void store16_asm(void* p, const u32x4& v) {   // (s_nop: see ff_fused.hip)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void store8_asm(void* p, const u32x2_t& v) {
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// LDS reads placed by hand: the data registers are written some hundred cycles after the instruction; lds_wait2 is the counted
// wait (LDS operations return in order) and ties the registers to it, so no use of them can be scheduled in front of the wait
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4& d, const unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// rows of x -> LayerNorm -> the eight B fragments (lane (r, hf): channels 16 s + 8 hf + i of row r)
__device__ __forceinline__ void ln_rows(const u32x4 (&raw)[8], const float* gam, const float* bet, float* mr, const int hf, bf16x8 (&frag)[8]) {
  float xs[64];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[8 * s + i] = (float)__builtin_bit_cast(bf16x8, raw[s])[i];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) sum += xs[i];
  sum += __shfl_xor(sum, 32, 64);
  const float mean = sum * (1.0f / PW_C);
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) { const float d = xs[i] - mean; var += d * d; }
  var += __shfl_xor(var, 32, 64);
  const float rstd = rsqrtf(var * (1.0f / PW_C) + NORM_EPS);
  if (hf == 0) { mr[0] = mean; mr[1] = rstd; }
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
    frag[s] = o;
  }
}

#ifdef PW_STAMPS   // measurement build (tools/build_variant.sh): shader-clock stamps of the first two tiles of every wave into qkv1
#define PW_STAMP()                                                                                                   \
  do {                                                                                                               \
    if (nst < 64 && lane == 0) reinterpret_cast<long long*>(p.qkv1)[((size_t)blockIdx.x * 8 + wave) * 64 + nst] = __builtin_amdgcn_s_memtime(); \
    ++nst;                                                                                                           \
  } while (0)
#else
#define PW_STAMP() do {} while (0)
#endif

template <bool SAVE>
__global__ __launch_bounds__(512, 1) void pwa_block_fwd_kernel(const PwaArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* wst = smem;                                           // 2 weight stages
  unsigned char* stg_all = smem + 2 * PW_STAGE;                        // 8 x PW_STG_WAVE
  float* cst = reinterpret_cast<float*>(stg_all + 8 * PW_STG_WAVE);    // g1 b1 g2 b2
  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < PW_C; i += 512) {
    cst[i] = p.g1[i];
    cst[PW_C + i] = p.b1[i];
    cst[2 * PW_C + i] = p.g2[i];
    cst[3 * PW_C + i] = p.b2[i];
  }
  unsigned char* stg = stg_all + wave * PW_STG_WAVE;

  // weight stage of head hd: this wave's seven of the 56 one-KiB pieces, each one contiguous KiB of the packed panel (a
  // fragment gathered from the row-major matrix touches 32 cache lines for 32 bytes each: the stage issue alone then took as
  // long as the head's MFMAs)
  auto issue_stage = [&](int hd, int buf) {
    unsigned char* dst = wst + buf * PW_STAGE + wave * 7 * 1024;
    const bf16* g = p.wpk + ((size_t)hd * 56 + wave * 7) * 512 + lane * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) dma16(g + k * 512, dst + k * 1024);
  };
  constexpr int YOUNGER = SAVE ? 24 : 0;   // vector-memory operations issued behind a stage's DMA before it is awaited (the 24
                                           // stores of saved projections): never fewer

  int seq = 0;
  [[maybe_unused]] int nst = 0;
  issue_stage(0, 0);
  __syncthreads();   // constants in LDS
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int64_t row = (int64_t)tile * PW_ROWS + wave * 32 + r;
    PW_STAMP();
    bf16x8 h1[8], h2[8];
    {
      // The rows arrive by COALESCED loads (instruction j: lane l = 16 bytes of row 4 j + (l >> 4), chunk l & 15 - eight whole
      // cache lines per instruction) and are turned into the row-per-lane layout through the wave's staging tile, sixteen
      // rows at a time.  Loading the row-per-lane layout directly (lane (r, hf): 16 bytes of row r) touches 32 cache lines
      // per instruction for 32 bytes each; the texture path works per line, and 45 % of the kernel went into those loads.
      u32x4 c1[8], c2[8];
      const size_t ld_off = ((size_t)tile * PW_ROWS + wave * 32 + (lane >> 4)) * PW_C + (lane & 15) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) c1[j] = *reinterpret_cast<const u32x4*>(p.x1 + ld_off + (size_t)j * 4 * PW_C);
#pragma unroll
      for (int j = 0; j < 8; ++j) c2[j] = *reinterpret_cast<const u32x4*>(p.x2 + ld_off + (size_t)j * 4 * PW_C);
      unsigned char* wr = stg + (lane >> 4) * 272 + (lane & 15) * 16;      // rows padded to 272 bytes: conflict-free both ways
      const unsigned char* rd = stg + (r & 15) * 272 + hf * 16;
      // (the pointer is laundered per tile: gamma and beta are tile-invariant, and hoisted out of the tile loop their 256
      // values per lane were spilled to scratch and re-read every tile)
      const float* cl = cst;
      asm volatile("" : "+v"(cl));
      auto transpose = [&](const u32x4 (&c)[8], u32x4 (&raw)[8]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<u32x4*>(wr + jj * 4 * 272) = c[4 * hh + jj];
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const bool mine = (r >> 4) == hh;   // (every lane reads - rows r and r + 16 share a slot - and keeps its own half)
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(rd + s2 * 32);
#pragma unroll
            for (int d = 0; d < 4; ++d) raw[s2][d] = (hh == 0 || mine) ? v[d] : raw[s2][d];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
      };
      {
        u32x4 raw[8];
        transpose(c1, raw);
        PW_STAMP();
        ln_rows(raw, cl, cl + PW_C, p.mr1 + (size_t)row * 2, hf, h1);
        PW_STAMP();
      }
      {
        u32x4 raw[8];
        transpose(c2, raw);
        PW_STAMP();
        ln_rows(raw, cl + 2 * PW_C, cl + 3 * PW_C, p.mr2 + (size_t)row * 2, hf, h2);
      }
    }
    f32x16 accO[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) accO[n][e] = 0.f;
    PW_STAMP();

#ifdef PW_NO_HEADS   // measurement build: loads, LayerNorms and stores only
    if (tile < 0)
#endif
    for (int hd = 0; hd < 4; ++hd, ++seq) {
      if (seq == 0) wait_vm_then_barrier<0>();
      else wait_vm_then_barrier<YOUNGER>();
      PW_STAMP();
      issue_stage((hd + 1) & 3, (seq + 1) & 1);   // (past the last tile: one stage nobody reads - keeps the counts static)
      // The head's 56 weight fragments are read from LDS by a hand-placed stream (ds_read_b128 in assembly, four steps = eight
      // fragments ahead of the MFMAs that use them, counted lgkmcnt waits): every MFMA here needs one 1-KiB fragment, four SIMDs
      // at one MFMA per 32 cycles ask for the whole 128 B/clk of the LDS, and the compiler's schedule (read, wait, MFMA) exposed
      // the LDS latency 28 times per head with only two waves per SIMD to cover it.
      // Steps g = 0..7: k1 | q2, 8..15: q1 | k2, 16..23: v1 | v2 (part A of input 1 and part B of input 2, k step g & 7);
      // g = 24..27: the two Wo fragments of output tile g - 24.
      const unsigned wba = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(wst + (seq & 1) * PW_STAGE + lane * 16);
      u32x4 ring[8];
#define PW_PA(g) (((g) >> 3) == 0 ? 1 : ((g) >> 3) == 1 ? 0 : 2)
#define PW_PB(g) (((g) >> 3) == 0 ? 0 : ((g) >> 3) == 1 ? 1 : 2)
#define PW_OFFA(g) ((g) < 24 ? (PW_PA(g) * 8 + ((g) & 7)) * 1024 : (48 + 2 * ((g) - 24)) * 1024)
#define PW_OFFB(g) ((g) < 24 ? ((3 + PW_PB(g)) * 8 + ((g) & 7)) * 1024 : (49 + 2 * ((g) - 24)) * 1024)
#define PW_ISSUE(g)                                           \
  do {                                                        \
    if ((g) < 28) {                                           \
      lds_read16<PW_OFFA((g) < 28 ? (g) : 0)>(ring[2 * ((g) & 3)], wba);     \
      lds_read16<PW_OFFB((g) < 28 ? (g) : 0)>(ring[2 * ((g) & 3) + 1], wba); \
    }                                                         \
  } while (0)
#define PW_WAIT(g) lds_wait2<2 * ((g) + 3 > 27 ? 27 - (g) : 3)>(ring[2 * ((g) & 3)], ring[2 * ((g) & 3) + 1])
#define PW_STEP(g)                                                                                                              \
  do {                                                                                                                          \
    PW_WAIT(g);                                                                                                                 \
    acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[2 * ((g) & 3)]), h1[(g) & 7], acca, 0, 0, 0);     \
    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[2 * ((g) & 3) + 1]), h2[(g) & 7], accb, 0, 0, 0); \
    PW_ISSUE((g) + 4);                                                                                                          \
  } while (0)
#define PW_PAIR(g0) \
  do { PW_STEP(g0); PW_STEP(g0 + 1); PW_STEP(g0 + 2); PW_STEP(g0 + 3); PW_STEP(g0 + 4); PW_STEP(g0 + 5); PW_STEP(g0 + 6); PW_STEP(g0 + 7); } while (0)
      PW_ISSUE(0); PW_ISSUE(1); PW_ISSUE(2); PW_ISSUE(3);
      f32x16 acca, accb;
      // rounds a pair's tiles to bf16 (as the stored projections are), stores them if asked, hands back the rounded values
      auto finish_pair = [&](int pa, int pb, float (&ta)[16], float (&tb)[16]) {
        bf16* da = SAVE ? p.qkv1 + (size_t)row * (3 * PW_C) + pa * PW_C + 32 * hd + 4 * hf : nullptr;
        bf16* db = SAVE ? p.qkv2 + (size_t)row * (3 * PW_C) + pb * PW_C + 32 * hd + 4 * hf : nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4 ba, bb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ba[j] = (bf16)acca[4 * q + j]; ta[4 * q + j] = (float)ba[j];
            bb[j] = (bf16)accb[4 * q + j]; tb[4 * q + j] = (float)bb[j];
          }
          if (SAVE) {
            store8_asm(da + 8 * q, __builtin_bit_cast(u32x2_t, ba));
            store8_asm(db + 8 * q, __builtin_bit_cast(u32x2_t, bb));
          }
        }
      };
      float z;
      {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acca[e] = 0.f; accb[e] = 0.f; }
        PW_PAIR(0);
        float ka[16], qb[16];
        finish_pair(1, 0, ka, qb);   // k1, q2
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s = fmaf(qb[e], ka[e], s);
        z = s;
      }
      PW_STAMP();
      {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acca[e] = 0.f; accb[e] = 0.f; }
        PW_PAIR(8);
        float qa[16], kb[16];
        finish_pair(0, 1, qa, kb);   // q1, k2
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s = fmaf(qa[e], kb[e], s);
        z -= s;
      }
      PW_STAMP();
      z += __shfl_xor(z, 32, 64);
      const float a1 = __builtin_amdgcn_rcpf(1.0f + __expf(-z * p.scale));
      bf16x8 ofrag[2];
      {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acca[e] = 0.f; accb[e] = 0.f; }
        PW_PAIR(16);
        float va[16], vb[16];
        finish_pair(2, 2, va, vb);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          bf16x8 f;
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = (bf16)(a1 * va[8 * a + i] + (1.0f - a1) * vb[8 * a + i]);
          ofrag[a] = f;
        }
      }
      PW_STAMP();
#define PW_OSTEP(g)                                                                                                                      \
  do {                                                                                                                                   \
    PW_WAIT(g);                                                                                                                          \
    accO[(g) - 24] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[2 * ((g) & 3)]), ofrag[0], accO[(g) - 24], 0, 0, 0);     \
    accO[(g) - 24] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[2 * ((g) & 3) + 1]), ofrag[1], accO[(g) - 24], 0, 0, 0); \
  } while (0)
      PW_OSTEP(24); PW_OSTEP(25); PW_OSTEP(26); PW_OSTEP(27);
      PW_STAMP();
#undef PW_OSTEP
#undef PW_PAIR
#undef PW_STEP
#undef PW_WAIT
#undef PW_ISSUE
#undef PW_OFFA
#undef PW_OFFB
#undef PW_PA
#undef PW_PB
    }
    // ---- out rows: 64 features at a time through the wave's staging tile, stored as whole 128-byte row segments
    {
      bf16* so = reinterpret_cast<bf16*>(stg);
      const int64_t row0 = (int64_t)tile * PW_ROWS + wave * 32;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 b;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = (bf16)accO[2 * hp + n2][4 * q + j];
            *reinterpret_cast<bf16x4*>(so + r * 72 + 32 * n2 + 4 * hf + 8 * q) = b;
          }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 8 * i + (lane >> 3), cg = lane & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(so + rr * 72 + cg * 8);
          store16_asm(p.out + (size_t)(row0 + rr) * PW_C + 64 * hp + cg * 8, v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    PW_STAMP();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage nobody reads has landed before the workgroup's LDS is released
}

// packed[head][piece][lane = (m, hf)][i], piece < 24: Wq1[(piece >> 3) * 128 + 32 head + m][16 (piece & 7) + 8 hf + i];
// 24 <= piece < 48: the same of Wq2; piece = 48 + 2 n + a: Wo[32 n + m][32 head + 4 hf + 16 a + 8 (i >> 2) + (i & 3)]
__global__ __launch_bounds__(256) void pwa_pack_kernel(const bf16* __restrict__ wq1, const bf16* __restrict__ wq2,
                                                       const bf16* __restrict__ wo, bf16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // one (head, piece, lane) per thread
  if (idx >= 4 * 56 * 64) return;
  const int lane = idx & 63, pc = (idx >> 6) % 56, hd = (idx >> 6) / 56;
  const int m = lane & 31, hf = lane >> 5;
  bf16x8 o;
  if (pc < 48) {
    const bf16* w = pc < 24 ? wq1 : wq2;
    const int q = pc < 24 ? pc : pc - 24;
    o = *reinterpret_cast<const bf16x8*>(w + (size_t)((q >> 3) * PW_C + 32 * hd + m) * PW_C + 16 * (q & 7) + 8 * hf);
  } else {
    const int n = (pc - 48) >> 1, a = (pc - 48) & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = wo[(size_t)(32 * n + m) * PW_C + 32 * hd + 4 * hf + 16 * a + 8 * (i >> 2) + (i & 3)];
  }
  *reinterpret_cast<bf16x8*>(out + (size_t)idx * 8) = o;
}

}  // namespace

extern "C" int ctu_pwa_pack(const void* wq1, const void* wq2, const void* wo, void* packed, int32_t C, ctu_stream_t stream) {
  CTU_REQUIRE(wq1 && wq2 && wo && packed, "pwa_pack: null pointer");
  CTU_REQUIRE(C == PW_C, "pwa_pack: C = 128 (C=%d)", C);
  hipLaunchKernelGGL(pwa_pack_kernel, dim3(4 * 56 * 64 / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)wq1, (const bf16*)wq2,
                     (const bf16*)wo, (bf16*)packed);
  return ctu_check_launch("pwa_pack");
}

extern "C" int ctu_pwa_block_fwd(ctu_dtype dtype, const void* x1, const void* x2, const float* g1, const float* b1, const float* g2,
                                 const float* b2, const void* w_packed, void* out, void* qkv1, void* qkv2, float* mean_rstd1,
                                 float* mean_rstd2, int64_t M, int32_t C, float scale, ctu_stream_t stream) {
  CTU_REQUIRE(dtype == CTU_BF16, "pwa_block_fwd: bf16 only");
  CTU_REQUIRE(x1 && x2 && g1 && b1 && g2 && b2 && w_packed && out && mean_rstd1 && mean_rstd2, "pwa_block_fwd: null pointer");
  CTU_REQUIRE((qkv1 == nullptr) == (qkv2 == nullptr), "pwa_block_fwd: qkv1 and qkv2 are saved together or not at all");
  CTU_REQUIRE(C == PW_C, "pwa_block_fwd: C = 128 (C=%d)", C);
  CTU_REQUIRE(M > 0 && M % PW_ROWS == 0 && M < (1ll << 31), "pwa_block_fwd: M must be a multiple of 256 (M=%lld)", (long long)M);
  PwaArgs p;
  p.x1 = (const bf16*)x1; p.x2 = (const bf16*)x2; p.g1 = g1; p.b1 = b1; p.g2 = g2; p.b2 = b2;
  p.wpk = (const bf16*)w_packed; p.out = (bf16*)out;
  p.qkv1 = (bf16*)qkv1; p.qkv2 = (bf16*)qkv2; p.mr1 = mean_rstd1; p.mr2 = mean_rstd2; p.M = M; p.scale = scale;
  p.ntiles = (int)(M / PW_ROWS);
  const size_t lds = 2 * PW_STAGE + 8 * PW_STG_WAVE + 4 * PW_C * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pwa_block_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pwa_block_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      ctu_set_error("pwa_block_fwd: cannot raise the dynamic LDS limit");
      return CTU_ERR_ARG;
    }
    attr_set = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
#ifdef PW_STAMPS
  hipLaunchKernelGGL(pwa_block_fwd_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  return ctu_check_launch("pwa_block_fwd");
#endif
  if (qkv1) hipLaunchKernelGGL(pwa_block_fwd_kernel<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pwa_block_fwd_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
  return ctu_check_launch("pwa_block_fwd");
}
