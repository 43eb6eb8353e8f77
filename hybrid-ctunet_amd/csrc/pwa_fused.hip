// Fused pixelweight_attention for the 128-channel stages (hybrid_CTUNet.py:622-669):
//     out = Wo . cross_weight(Wq1 LayerNorm1(x1), Wq2 LayerNorm2(x2)),      x1, x2, out [M][128] bf16
// in ONE kernel per call instead of two LayerNorms, two [M][128] x [128][384] GEMMs, the cross-weight core and the output
// GEMM: the normalised rows stay in registers as the B operand of all six projections, each head's q / k / v tiles leave the
// accumulators only as the saved copies a backward pass reads (qkv1 / qkv2, optional), and the mixed head goes straight
// from registers into the output projection.  Per 442 368-row call: 226 MB read + 113 MB (+ 680 MB saved projections)
// written, instead of 1.9 GB read + 1.25 GB written over six launches.
//
// Construction (as ff_fused.hip): a wave owns 32 rows for the whole block; every product is taken transposed (weight fragment =
// MFMA A operand, activations = B operand), so a lane holds values of ONE row:
//   * lane (r, hf) of a q / k / v tile [32 head channels][32 rows] holds channels {4 hf + 8 q + j} of row r: the two dot
//     products <q2, k1>, <q1, k2> of a head are 16 lane-local products plus one exchange with lane r + 32, the sigmoid
//     and the mix a1 v1 + (1 - a1) v2 are lane-local;
//   * registers 8 a .. 8 a + 7 of the mixed tile are the B fragment of k step (head, a) of the output projection, in the
//     channel order 4 hf + 16 a + 8 (i >> 2) + (i & 3), which the packed Wo fragments mirror.
// ONE workgroup of FOUR waves per CU walks 128-row tiles: one wave per SIMD owns the SIMD's whole register file (512 registers:
// the accumulators live in AGPRs), which is what this block needs - 64 registers of normalised rows, 64 of output accumulators,
// two projection pairs in flight, the LDS fragment ring AND the next tile's rows (64), requested a whole tile ahead.  With two
// waves per SIMD (256 registers each: an 8-wave workgroup per CU, or two 4-wave workgroups) the kernel spilled, could not hold
// the next tile's rows, and spent two thirds of a tile waiting - vector-memory operations of a wave return in order on ONE
// counter, so a load (or a spill reload) issued behind a tile's output stores waits for those stores' acknowledgement, 20 k
// cycles (profiles/r04_pwa_block_fwd.txt: 214 / 195 us per 442 k-row call in those forms, loads + LayerNorms + stores alone 104 us,
// the head loop 100 us, not overlapped).  Here every global access is issued at least one head ahead of its use or wait.
// Weights stream L2 -> LDS per HEAD: 56 fragments (k1 | q2 | q1 | k2 | v1 | v2, 8 k steps each, and 8 of Wo) = 56 KiB, contiguous
// in the panel ctu_pwa_pack writes once per weight update; double buffered, one workgroup barrier per head (56 MFMAs per wave).
// The fragments are read from LDS by a hand-placed stream (ds_read_b128 in assembly, four steps = eight fragments ahead of
// the MFMAs that use them, counted lgkmcnt): every MFMA here needs one 1-KiB fragment, and the compiler's order (read, wait,
// MFMA) exposes the LDS latency at every step - with one wave per SIMD nothing else would cover it.
#include <atomic>

#include "dma.h"
#include "mma.h"

namespace {

// Tile dispensers: a launch takes the next of 64 slots (two launches on two streams never share one; a slot comes round again
// 64 launches later, long after its last workgroup zeroed it).
__device__ unsigned g_pwa_ticket[64];
__device__ unsigned g_pwa_done[64];

constexpr int PW_C = 128;
constexpr int PW_ROWS = 128;          // rows per workgroup tile (4 waves x 32)
constexpr int PW_HEAD = 56;           // fragments (1 KiB each) of a head
constexpr int PW_STAGE = PW_HEAD * 1024;
constexpr int PW_STG_WAVE = 8192;     // [32 rows][256 B]: a wave's output rows on their way to whole-line stores

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
  unsigned* ticket;  // tile dispenser of this launch: tiles beyond the first (blockIdx.x) are taken with atomicAdd (see the kernel)
  unsigned* done;    // workgroups that are through: the last one hands both counters back zeroed
};

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
// (in assembly: the counted vmcnt waits know these stores; wave-uniform base + 32-bit byte offset per lane - 64-bit per-lane
// pointers kept across the tile are what the kernel spills first)
__device__ __forceinline__ void store8_asm(void* base, const unsigned off, const u32x2_t& v) {
  asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
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

// rows of x (lane (r, hf): channels 16 s + 8 hf + i of row r) -> LayerNorm -> the eight B fragments.  Three passes over the raw
// bf16 registers, converting on the fly: a 64-value fp32 copy per lane (twice, once the scheduler interleaves the two inputs)
// does not fit beside what the tile keeps live.
__device__ __forceinline__ void ln_rows(const bf16x8 (&raw)[8], const float* gam, const float* bet, float* mr, const int hf, bf16x8 (&frag)[8]) {
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += (float)raw[s][i];
  sum += __shfl_xor(sum, 32, 64);
  const float mean = sum * (1.0f / PW_C);
  float var = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = (float)raw[s][i] - mean; var = fmaf(d, d, var); }
  var += __shfl_xor(var, 32, 64);
  const float rstd = rsqrtf(var * (1.0f / PW_C) + NORM_EPS);
  if (hf == 0) { mr[0] = mean; mr[1] = rstd; }
  float mean3 = mean;
  asm volatile("" : "+v"(mean3));   // (else the 64 differences x - mean of the variance pass are kept for this one - and spilled)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int c0 = 16 * s + 8 * hf;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gam + c0), g1 = *reinterpret_cast<const f32x4*>(gam + c0 + 4);
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(bet + c0), e1 = *reinterpret_cast<const f32x4*>(bet + c0 + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = (bf16)(((float)raw[s][i] - mean3) * rstd * g0[i] + e0[i]);
      o[4 + i] = (bf16)(((float)raw[s][4 + i] - mean3) * rstd * g1[i] + e1[i]);
    }
    frag[s] = o;
  }
}

#ifdef PW_STAMPS   // measurement build (tools/build_variant.sh): shader-clock stamps of the first tiles of every wave into qkv1
#define PW_STAMP()                                                                                                   \
  do {                                                                                                               \
    if (nst < 64 && lane == 0) reinterpret_cast<long long*>(p.qkv1)[((size_t)blockIdx.x * 4 + wave) * 64 + nst] = __builtin_amdgcn_s_memtime(); \
    ++nst;                                                                                                           \
  } while (0)
#else
#define PW_STAMP() do {} while (0)
#endif

// Fragment stream of a head.  Step g reads two fragments into ring slot g & 3.  Steps 0..15 interleave the pairs k1 | q2 and
// q1 | k2 k step by k step (pair g & 1, k step g >> 1): FOUR independent accumulation chains - one wave per SIMD issues a
// dependent MFMA only every ~52 cycles, two chains alone leave the matrix pipe half idle; steps 16..23 are v1 | v2 (k step g & 7),
// steps 24..27 the two Wo fragments of output tile g - 24.  (Panel groups of pair p: 2 p and 2 p + 1.)  PW_WAIT(g) waits until step g's pair has landed with the steps up to min(g + 3, 27) in flight behind it.
#define PW_PAIR_OF(g) ((g) < 16 ? ((g) & 1) : 2)
#define PW_KSTEP_OF(g) ((g) < 16 ? ((g) >> 1) : ((g) & 7))
#define PW_OFFA(g) ((g) < 24 ? (16 * PW_PAIR_OF(g) + PW_KSTEP_OF(g)) * 1024 : (48 + 2 * ((g) - 24)) * 1024)
#define PW_OFFB(g) ((g) < 24 ? (16 * PW_PAIR_OF(g) + 8 + PW_KSTEP_OF(g)) * 1024 : (49 + 2 * ((g) - 24)) * 1024)
#define PW_ISSUE(g)                                                               \
  do {                                                                            \
    if ((g) < 28) {                                                               \
      lds_read16<PW_OFFA((g) < 28 ? (g) : 0)>(ring[2 * ((g) & 3)], wba);          \
      lds_read16<PW_OFFB((g) < 28 ? (g) : 0)>(ring[2 * ((g) & 3) + 1], wba);      \
    }                                                                             \
  } while (0)
#define PW_WAIT(g) lds_wait2<2 * ((g) + 3 > 27 ? 27 - (g) : 3)>(ring[2 * ((g) & 3)], ring[2 * ((g) & 3) + 1])
#define PW_FRAG(g, i) __builtin_bit_cast(bf16x8, ring[2 * ((g) & 3) + (i)])
#define PW_STEP(g, A, B)                                                                                  \
  do {                                                                                                    \
    PW_WAIT(g);                                                                                           \
    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 0), h1[PW_KSTEP_OF(g)], A, 0, 0, 0);           \
    B = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 1), h2[PW_KSTEP_OF(g)], B, 0, 0, 0);           \
    PW_ISSUE((g) + 4);                                                                                    \
  } while (0)
// (the first k step takes a literal zero as its C operand: zeroing 64 accumulator registers per head was a tenth of the head's
// instructions, and with one wave per SIMD every instruction is on the critical path)
#define PW_STEP_FIRST(g, A, B)                                                                            \
  do {                                                                                                    \
    PW_WAIT(g);                                                                                           \
    A = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 0), h1[PW_KSTEP_OF(g)], fzero, 0, 0, 0);       \
    B = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 1), h2[PW_KSTEP_OF(g)], fzero, 0, 0, 0);       \
    PW_ISSUE((g) + 4);                                                                                    \
  } while (0)
#define PW_PAIR(g0, A, B)                                                                                 \
  do {                                                                                                    \
    PW_STEP_FIRST(g0, A, B); PW_STEP(g0 + 1, A, B); PW_STEP(g0 + 2, A, B); PW_STEP(g0 + 3, A, B);          \
    PW_STEP(g0 + 4, A, B); PW_STEP(g0 + 5, A, B); PW_STEP(g0 + 6, A, B); PW_STEP(g0 + 7, A, B);            \
  } while (0)
#define PW_OSTEP(g)                                                                                                  \
  do {                                                                                                               \
    PW_WAIT(g);                                                                                                      \
    accO[(g) - 24] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 0), ofrag[0], accO[(g) - 24], 0, 0, 0);      \
    accO[(g) - 24] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PW_FRAG(g, 1), ofrag[1], accO[(g) - 24], 0, 0, 0);      \
  } while (0)

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void pwa_block_fwd_kernel(const PwaArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* wst = smem;                                                       // 2 weight stages
  unsigned char* stg_all = smem + 2 * PW_STAGE;                                    // 4 x PW_STG_WAVE
  float* cst = reinterpret_cast<float*>(stg_all + 4 * PW_STG_WAVE);                // g1 b1 g2 b2
  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < PW_C; i += 256) {
    cst[i] = p.g1[i];
    cst[PW_C + i] = p.b1[i];
    cst[2 * PW_C + i] = p.g2[i];
    cst[3 * PW_C + i] = p.b2[i];
  }
  unsigned char* so = stg_all + wave * PW_STG_WAVE;
  // weight stage of head hd into buffer `buf`: this wave's 14 of the 56 one-KiB pieces, contiguous in the panel (a fragment
  // gathered from the row-major matrix touches 32 cache lines for 32 bytes each: the stage issue alone then took as long as
  // the MFMAs it feeds)
  // (One assembly block per stage: source and destination are both contiguous, so a piece is an immediate offset - it moves the
  // global AND the LDS address - and every fourth piece a step of M0 and of the lane offset; the per-piece form cost ~15
  // instructions per DMA, 210 per head for a wave that has nobody to hide its instruction issue behind.)
  const unsigned dv0 = lane * 16, dv1 = dv0 + 4096, dv2 = dv0 + 8192, dv3 = dv0 + 12288;
  auto issue_stage = [&](int hd, int buf) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(wst + buf * PW_STAGE + wave * 14 * 1024));
    const uint64_t a = (uint64_t)(uintptr_t)(p.wpk + ((size_t)hd * PW_HEAD + wave * 14) * 512);
    const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
                          (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
    unsigned keep;
    asm volatile(
        "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6\n\tglobal_load_lds_dwordx4 %1, %6 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %6 offset:2048\n\tglobal_load_lds_dwordx4 %1, %6 offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %6\n\tglobal_load_lds_dwordx4 %2, %6 offset:1024\n\t"
        "global_load_lds_dwordx4 %2, %6 offset:2048\n\tglobal_load_lds_dwordx4 %2, %6 offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %6\n\tglobal_load_lds_dwordx4 %3, %6 offset:1024\n\t"
        "global_load_lds_dwordx4 %3, %6 offset:2048\n\tglobal_load_lds_dwordx4 %3, %6 offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %6\n\tglobal_load_lds_dwordx4 %4, %6 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(dv0), "v"(dv1), "v"(dv2), "v"(dv3), "s"(dst), "s"(base) : "memory", "scc");
  };
  // Vector-memory operations of a wave return in order on one counter.  A stage is awaited one head after its DMA was issued;
  // what is issued BEHIND the DMA in between (never fewer) may stay in flight: the 24 stores of a head's saved projections, and
  // in a tile's first head also the 8 stores of the previous tile's rows and the 16 loads of the next tile's rows.
  constexpr int YOUNGER = SAVE ? 24 : 0;

  // (addresses: wave-uniform base + 32-bit byte offset per lane; the host checks M * 768 < 2^32)
  bf16x8 raw1[8], raw2[8];
  auto request_rows = [&](int t) {
    const unsigned off = (unsigned)((t * PW_ROWS + wave * 32 + r) * (PW_C * 2) + 16 * hf);
    const unsigned char* b1 = reinterpret_cast<const unsigned char*>(p.x1);
    const unsigned char* b2 = reinterpret_cast<const unsigned char*>(p.x2);
#pragma unroll
    for (int s = 0; s < 8; ++s) raw1[s] = *reinterpret_cast<const bf16x8*>(b1 + (off + 32 * s));
#pragma unroll
    for (int s = 0; s < 8; ++s) raw2[s] = *reinterpret_cast<const bf16x8*>(b2 + (off + 32 * s));
  };
  // a tile's rows leave LDS (wave-private [32][256 B], 16-byte slots XOR-swizzled by the row) as whole cache lines
  auto store_rows = [&](int t) {
    unsigned char* ob = reinterpret_cast<unsigned char*>(p.out);
    const unsigned ooff = (unsigned)(t * PW_ROWS + wave * 32 + (lane >> 4)) * (PW_C * 2) + (lane & 15) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 4 * i + (lane >> 4);
      const u32x4 v = *reinterpret_cast<const u32x4*>(so + rr * 256 + (((lane & 15) ^ (rr & 15)) << 4));
      *reinterpret_cast<u32x4*>(ob + (ooff + (unsigned)(i * 4 * PW_C * 2))) = v;
    }
  };

  // Tiles are DISPENSED, not dealt: the first is blockIdx.x, every further one the next ticket.  The kernel wants a whole CU per
  // workgroup (146 KiB of LDS), and beside another stream's kernel part of the 256 workgroups start late - dealt tiles then made
  // the launch wait for the latecomers' whole share (inference: 360 us alone, 630 us beside the other branch).  Wave 0 draws the
  // ticket for the tile after next in head 2 (behind that head's DMA), parks it in LDS behind head 3's wait - which covers the
  // atomic's return on the in-order counter anyway - and every wave reads it behind the next tile's first barrier.
  unsigned* tick = reinterpret_cast<unsigned*>(cst + 4 * PW_C);   // [0]: the tile after the current one
  int seq = 0;
  [[maybe_unused]] int nst = 0;
  issue_stage(0, 0);
  if ((int)blockIdx.x < p.ntiles) request_rows(blockIdx.x);
  if (tid == 0) tick[0] = (unsigned)gridDim.x + atomicAdd(p.ticket, 1u);
  __syncthreads();   // constants in LDS (and, this once, the first stage, rows and ticket: everything has landed)
  int prev_tile = -1;
  int next_tile = (int)tick[0];
  unsigned drawn = 0;
  for (int tile = blockIdx.x; tile < p.ntiles; tile = next_tile) {
    const unsigned row = (unsigned)(tile * PW_ROWS + wave * 32 + r);
    const unsigned qoff = row * (3 * PW_C * 2) + 8 * hf;   // byte offset of this lane's first 4 channels in a [M][384] row
    PW_STAMP();
    bf16x8 h1[8], h2[8];
    {
      unsigned coff = 0;
      asm volatile("" : "+v"(coff));   // (gamma / beta are tile-invariant: hoisted out of the tile loop they are 256 values per lane;
                                       // an offset is laundered, not the pointer - a laundered pointer loses its address space)
      const float* cl = cst + coff;
      ln_rows(raw1, cl, cl + PW_C, reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(p.mr1) + row * 8u), hf, h1);
      ln_rows(raw2, cl + 2 * PW_C, cl + 3 * PW_C, reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(p.mr2) + row * 8u), hf, h2);
    }
    f32x16 accO[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) accO[n][e] = 0.f;
    PW_STAMP();

    for (int hd = 0; hd < 4; ++hd, ++seq) {
      if (seq == 0) wait_vm_then_barrier<0>();
      else if (hd == 0) wait_vm_then_barrier<YOUNGER + 2>();    // behind the DMA of this stage: the two stores of the LayerNorm statistics
      else if (hd == 1) wait_vm_then_barrier<YOUNGER + 24>();   // behind the DMA of this stage: 8 row stores + 16 row loads (below)
      else wait_vm_then_barrier<YOUNGER>();
      if (hd == 3 && tid == 0) tick[0] = drawn;   // (here, in front of the DMA: the wait above has already covered the atomic's return;
                                                  // every wave reads it behind the next tile's first barrier)
      issue_stage((hd + 1) & 3, (seq + 1) & 1);   // (past the last tile: one stage nobody reads - keeps the counts static)
      if (hd == 0) {
        // the previous tile's rows (first tile: this tile's staging area as it is - rewritten below by the real rows) and the
        // next tile's rows (past the last tile: the same rows again): behind this head's DMA, ahead of everything that waits
        if (prev_tile >= 0) next_tile = (int)tick[0];   // (drawn in the previous tile's head 2, parked behind its head 3 wait)
        store_rows(prev_tile >= 0 ? prev_tile : tile);
        request_rows(next_tile < p.ntiles ? next_tile : tile);
      } else if (hd == 2) {
        if (tid == 0) drawn = (unsigned)gridDim.x + atomicAdd(p.ticket, 1u);
      }
      const unsigned wba = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(wst + (seq & 1) * PW_STAGE + lane * 16);
      PW_STAMP();
      u32x4 ring[8];
      f32x16 acc0a, acc0b, acc1a, acc1b;
      const f32x16 fzero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // a finished pair of tiles, consumed four values at a time - either their dot product or the mixed head.  With SAVE the
      // values are rounded to bf16 first and stored: the forward pass then uses exactly what a backward pass will read; without,
      // the fp32 accumulators are used as they are (192 conversions per head less)
      auto pair_store = [&](int pa, int pb, int q, const bf16x4& ba, const bf16x4& bb) {
        if (SAVE) {
          unsigned char* q1 = reinterpret_cast<unsigned char*>(p.qkv1);
          unsigned char* q2 = reinterpret_cast<unsigned char*>(p.qkv2);
          *reinterpret_cast<bf16x4*>(q1 + (qoff + (unsigned)((pa * PW_C + 32 * hd + 8 * q) * 2))) = ba;
          *reinterpret_cast<bf16x4*>(q2 + (qoff + (unsigned)((pb * PW_C + 32 * hd + 8 * q) * 2))) = bb;
        }
      };
      auto pair_dot = [&](const f32x16& A, const f32x16& B, int pa, int pb) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (SAVE) {
            bf16x4 ba, bb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ba[j] = (bf16)A[4 * q + j];
              bb[j] = (bf16)B[4 * q + j];
              s = fmaf((float)ba[j], (float)bb[j], s);
            }
            pair_store(pa, pb, q, ba, bb);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(A[4 * q + j], B[4 * q + j], s);
          }
        }
        return s;
      };
      PW_ISSUE(0); PW_ISSUE(1); PW_ISSUE(2); PW_ISSUE(3);
      PW_STEP_FIRST(0, acc0a, acc0b); PW_STEP_FIRST(1, acc1a, acc1b);   // k1 | q2 and q1 | k2, k step by k step
      PW_STEP(2, acc0a, acc0b); PW_STEP(3, acc1a, acc1b); PW_STEP(4, acc0a, acc0b); PW_STEP(5, acc1a, acc1b);
      PW_STEP(6, acc0a, acc0b); PW_STEP(7, acc1a, acc1b); PW_STEP(8, acc0a, acc0b); PW_STEP(9, acc1a, acc1b);
      PW_STEP(10, acc0a, acc0b); PW_STEP(11, acc1a, acc1b); PW_STEP(12, acc0a, acc0b); PW_STEP(13, acc1a, acc1b);
      PW_STEP(14, acc0a, acc0b); PW_STEP(15, acc1a, acc1b);
      f32x16 acc2a, acc2b;
      PW_PAIR(16, acc2a, acc2b);                      // v1 | v2   (the dot products below are taken under these MFMAs)
      float z = pair_dot(acc0a, acc0b, 1, 0);         // <k1, q2>
      z -= pair_dot(acc1a, acc1b, 0, 1);              // <q1, k2>
      z += __shfl_xor(z, 32, 64);
      const float a1 = __builtin_amdgcn_rcpf(1.0f + __expf(-z * p.scale));
      bf16x8 ofrag[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (SAVE) {
          bf16x4 ba, bb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ba[j] = (bf16)acc2a[4 * q + j];
            bb[j] = (bf16)acc2b[4 * q + j];
            ofrag[q >> 1][4 * (q & 1) + j] = (bf16)(a1 * (float)ba[j] + (1.0f - a1) * (float)bb[j]);
          }
          pair_store(2, 2, q, ba, bb);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            ofrag[q >> 1][4 * (q & 1) + j] = (bf16)fmaf(a1, acc2a[4 * q + j] - acc2b[4 * q + j], acc2b[4 * q + j]);
        }
      }
      PW_OSTEP(24); PW_OSTEP(25); PW_OSTEP(26); PW_OSTEP(27);
      PW_STAMP();
    }
    // ---------------- out rows into the wave's staging area (stored in the next tile's first head, or after the loop)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 b;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (bf16)accO[n][4 * q + j];
        *reinterpret_cast<bf16x4*>(so + r * 256 + (((4 * n + q) ^ (r & 15)) << 4) + 8 * hf) = b;
      }
    __builtin_amdgcn_wave_barrier();
    prev_tile = tile;
    PW_STAMP();
  }
  if (prev_tile >= 0) store_rows(prev_tile);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage nobody reads has landed before the workgroup's LDS is released
  if (tid == 0 && atomicAdd(p.done, 1u) == gridDim.x - 1) {   // the last workgroup through hands the dispenser back
    *p.ticket = 0;
    *p.done = 0;
  }
}
#undef PW_OSTEP
#undef PW_PAIR
#undef PW_STEP

// packed[head][piece][lane = (m, hf)][i] in the order the kernel consumes a head (W = Wq1 for input 1, Wq2 for input 2; rows of
// Wq: q | k | v, 128 each): pieces 0-7 k1, 8-15 q2, 16-23 q1, 24-31 k2, 32-39 v1, 40-47 v2 - piece = 8 group + s holds
// W[part * 128 + 32 head + m][16 s + 8 hf + i] - and 48 + 2 n + a: Wo[32 n + m][32 head + 4 hf + 16 a + 8 (i >> 2) + (i & 3)]
__global__ __launch_bounds__(256) void pwa_pack_kernel(const bf16* __restrict__ wq1, const bf16* __restrict__ wq2,
                                                       const bf16* __restrict__ wo, bf16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // one (head, piece, lane) per thread
  if (idx >= 4 * PW_HEAD * 64) return;
  const int lane = idx & 63, pc = (idx >> 6) % PW_HEAD, hd = (idx >> 6) / PW_HEAD;
  const int m = lane & 31, hf = lane >> 5;
  bf16x8 o;
  if (pc < 48) {
    const int grp = pc >> 3, s = pc & 7;
    const bool second = grp == 1 || grp == 3 || grp == 5;          // q2, k2, v2
    const int part = grp == 0 || grp == 3 ? 1 : grp == 1 || grp == 2 ? 0 : 2;   // k1, q2, q1, k2, v1, v2
    const bf16* w = second ? wq2 : wq1;
    o = *reinterpret_cast<const bf16x8*>(w + (size_t)(part * PW_C + 32 * hd + m) * PW_C + 16 * s + 8 * hf);
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
  hipLaunchKernelGGL(pwa_pack_kernel, dim3(4 * PW_HEAD * 64 / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)wq1, (const bf16*)wq2,
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
  CTU_REQUIRE(M > 0 && M % PW_ROWS == 0 && M * 768 < (1ll << 32), "pwa_block_fwd: M must be a multiple of 128 below 2^32 / 768 (M=%lld)", (long long)M);
  PwaArgs p;
  p.x1 = (const bf16*)x1; p.x2 = (const bf16*)x2; p.g1 = g1; p.b1 = b1; p.g2 = g2; p.b2 = b2;
  p.wpk = (const bf16*)w_packed; p.out = (bf16*)out;
  p.qkv1 = (bf16*)qkv1; p.qkv2 = (bf16*)qkv2; p.mr1 = mean_rstd1; p.mr2 = mean_rstd2; p.M = M; p.scale = scale;
  p.ntiles = (int)(M / PW_ROWS);
  const size_t lds = 2 * PW_STAGE + 4 * PW_STG_WAVE + 4 * PW_C * sizeof(float) + 16;
  static std::atomic<unsigned> launches{0};
  const unsigned slot = launches.fetch_add(1) & 63;
  // (the counters' device addresses are looked up once per device: hipGetSymbolAddress per launch made inference passes erratic)
  static unsigned* d_ticket[16] = {};
  static unsigned* d_done[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    ctu_set_error("pwa_block_fwd: device ordinal out of range");
    return CTU_ERR_ARG;
  }
  if (!d_ticket[dev] || !d_done[dev]) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&d_ticket[dev]), HIP_SYMBOL(g_pwa_ticket)) != hipSuccess ||
        hipGetSymbolAddress(reinterpret_cast<void**>(&d_done[dev]), HIP_SYMBOL(g_pwa_done)) != hipSuccess) {
      ctu_set_error("pwa_block_fwd: cannot locate the tile dispensers");
      return CTU_ERR_ARG;
    }
  }
  p.ticket = d_ticket[dev] + slot;
  p.done = d_done[dev] + slot;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pwa_block_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(pwa_block_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      ctu_set_error("pwa_block_fwd: cannot raise the dynamic LDS limit");
      return CTU_ERR_ARG;
    }
    attr_set = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;   // one resident workgroup per CU
#ifdef PW_STAMPS
  hipLaunchKernelGGL(pwa_block_fwd_kernel<false>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  return ctu_check_launch("pwa_block_fwd");
#endif
  if (qkv1) hipLaunchKernelGGL(pwa_block_fwd_kernel<true>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(pwa_block_fwd_kernel<false>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  return ctu_check_launch("pwa_block_fwd");
}
