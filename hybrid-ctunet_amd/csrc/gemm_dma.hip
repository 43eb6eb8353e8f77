// Plain bf16 GEMMs (no gather: 1x1x1 stride-1 convolutions, nn.Linear, transposed-conv panels) with LDS-DMA staging.
//
//   gemm_nt_dma : out[m][n] = sum_k A[m][k] W[n][k]  (+ the ctu_epilogue: bias / GELU / residual / split / scatter)
//   gemm_tn_dma : dw[n][c] += sum_m P[m][n] Q[m][c]  (weight gradients of the same layers)
//
// The generic implicit-GEMM kernels (igemm.hip) stage 32-deep k slices through registers with one barrier per
// 8 MFMAs and a fresh workgroup per output tile; at this model's shapes (K = 32..3072, M = 864 or 10^5..10^6) they reach
// ~2.2 TB/s on the HBM-bound layers and ~90 TFLOP/s on the 864-token ViT trunk.  Here:
//   * operands go global -> LDS by DMA (no VGPR staging), 64-deep stages, two stages per workgroup and two
//     workgroups per CU (64 KiB in flight per CU);
//   * workgroups are persistent: the DMA of the NEXT tile's first stage is issued before the epilogue of the
//     current tile, so short-K tiles (K = 128 is two stages) do not expose a load latency per tile;
//   * 128-B LDS rows with the 16-B slot XOR-swizzled by (row >> 1) & 7 on the DMA SOURCE side: every 16-lane
//     ds_read_b128 group of a 32-row fragment covers all 16 slots of the 256-B bank row.
//
// Replaces (reference call sites): nn.Linear at vit.py:36,39,59,62,117 and hybrid_CTUNet.py:402-679; 1x1x1 nn.Conv3d at
// resnet.py:96,100 and hybrid_CTUNet.py:75-83; the ConvTranspose3d panels of get_conv_layer (resnet.py:17-50).
#include "mma.h"
#include "dma.h"
#include "gemm_dma.h"

#include <stdlib.h>

namespace {

struct WorkItem {
  int m0, n0, kb, ke;  // tile origin, k-step range [kb, ke)
};

// destination of the 8 consecutive columns n.. of row m under the ctu_epilogue contract (plain / split / scattered)
__device__ __forceinline__ size_t epilogue_offset(const ctu_epilogue& ep, bf16* out, int m, int n, bf16*& dst) {
  dst = out;
  if (ep.scatter) {
    const int tap = n / ep.n_per_tap, co = n - tap * ep.n_per_tap;
    const int tw = tap % ep.sc_kw;
    const int tq = tap / ep.sc_kw;
    const int th = tq % ep.sc_kh, td = tq / ep.sc_kh;
    int t = m;
    const int ww = t % ep.sc_W; t /= ep.sc_W;
    const int hh = t % ep.sc_H; t /= ep.sc_H;
    const int dd = t % ep.sc_D;
    const int bb = t / ep.sc_D;
    const size_t orow = (((size_t)bb * (ep.sc_D * ep.sc_kd) + dd * ep.sc_kd + td) * (ep.sc_H * ep.sc_kh) + hh * ep.sc_kh + th) *
                            (size_t)(ep.sc_W * ep.sc_kw) + ww * ep.sc_kw + tw;
    return orow * ep.ldc + co;
  }
  if (ep.n_split > 0 && n >= ep.n_split) {
    dst = reinterpret_cast<bf16*>(ep.out2);
    return (size_t)m * ep.ldc2 + (n - ep.n_split);
  }
  return (size_t)m * ep.ldc + n;
}

}  // namespace

// wait for the oldest of `a + 1` stages in flight (a in [0, MAXA], steady state a = MAXA): a compile-time vmcnt each
// EXTRA: vector-memory operations issued AFTER those stages that may stay in flight as well (vmcnt retires loads, LDS-DMA
// and stores together, in issue order: the epilogue stores of the previous tile are younger than the stage waited for)
template <int IPW, int MAXA, int EXTRA = 0> __device__ __forceinline__ void wait_stage(int a) {
  if constexpr (MAXA <= 0) {
    wait_vm_then_barrier<EXTRA>();
  } else {
    if (a >= MAXA) wait_vm_then_barrier<MAXA * IPW + EXTRA>();
    else wait_stage<IPW, MAXA - 1, EXTRA>(a);
  }
}

// R = ring depth (stages of 64 k): R - 1 stages are in flight while one is computed.  A step of a small tile is a
// handful of MFMAs, far less than one DMA latency, so small tiles (and launches with at most one workgroup per CU)
// take a deeper ring; the wait is a counted vmcnt that leaves the younger stages in flight (epilogue stores pending
// on the same counter only make it wait for more, never less: loads retire in order among themselves).
// BT = true ("NN"): W is stored reduction-major, W[k][n] with leading dimension N - the data gradient of a Linear /
// 1x1x1 conv reads the forward weight [N_fwd][K_fwd] as is, no transposed copy.  Its B tile is staged as 32-column
// panels [panel][64 k rows][32 n] (64-B rows, plain DMA) and read transposed, exactly like the TN kernel's operands.
// BK = 32 serves K % 64 == 32 (the 32-channel layers): 64-B LDS rows, four to a bank row, slot ^ ((row >> 2) & 3).
// KG = 2 (the 864-token trunk, 64 x 64 tiles, one workgroup per CU): EIGHT waves, two k groups of four.  Both groups work
// on every stage - group g takes the k steps [g BK/32, (g + 1) BK/32) of it - so each SIMD holds two waves whose
// ds_read -> MFMA chains interleave (with one wave per SIMD a 64 x 64 tile is a chain of exposed LDS latencies: 2.2 k
// cycles per 128-deep stage for 256 cycles of MFMA); at the end of a tile group 1 hands its accumulators to group 0
// through LDS.  No split-K atomics, no second pass.
template <int BM, int BN, int R, bool BT, int BK = 64, int KG = 1, bool GA = false>
__global__ __launch_bounds__(256 * KG, (KG == 2 || R * (BM + BN) * BK * 2 > 76 * 1024) ? 1 : 2) void gemm_nt_dma_kernel(const GemmNtArgs p) {
  static_assert(BK == 64 || BK == 128 || (BK == 32 && !BT), "stage depth");
  static_assert(!GA || (BK == 64 && !BT && KG == 1), "gathered A operand: 64-deep stages, row-major weight panels");
  constexpr int ROWB = BK * 2;         // bytes of a tile row in LDS
  constexpr int SPR = BK / 8;          // 16-B slots per row
  constexpr int RPI = 1024 / ROWB;     // tile rows per DMA wave-instruction
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int WM = BM / 2, WN = BN / 2;  // wave tile: WM rows x WN columns (waves 2 x 2)
  constexpr int MI = WM / 32, NJ = WN / 32;
  constexpr int NW = 4 * KG;              // waves
  constexpr int APW = BM / RPI / NW;  // DMA instructions (1 KiB = RPI rows each) per wave and stage: A ...
  constexpr int BPW = BN / RPI / NW;  // ... and B
  static_assert(APW >= 1 && BPW >= 1 && (KG == 1 || (BK / 16) % 2 == 0), "tile too small for the wave count");
  constexpr int EPI_LD = 32 + 4;
  constexpr int EPI_BYTES = 4 * 16 * EPI_LD * 4;
  constexpr int IPW = APW + BPW;   // DMA instructions per wave and stage
  static_assert((R - 2) * IPW + 8 <= 63, "vmcnt is a 6-bit count");
  constexpr int RED_BYTES = 4 * WN * 2 * 4;
  constexpr int KRED_BYTES = KG == 2 ? 4 * 64 * MI * NJ * 16 * 4 : 0;  // group 1's accumulators on their way to group 0
  __shared__ __attribute__((aligned(1024))) unsigned char smem[R * STAGE + EPI_BYTES + RED_BYTES + KRED_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 2 ? wave >> 2 : 0, w4 = wave & 3;  // k group, wave inside the group
  const int wm = w4 >> 1, wn = w4 & 1;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  // source-side bank swizzle of a tile row's 16-B slots (conflict-free ds_read_b128 of 32-row fragments)
  auto swz = [](int row) { return BK == 128 ? row & 15 : BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };

  // Work items = (m tile, n tile, k split), n fastest.  The grid is a multiple of cols = tiles_n * splitk (launcher), so a
  // persistent workgroup keeps its (n tile, k split) for life and only walks down the m tiles: no division per item (the
  // scalar div / mod sequences of a per-item decode were a tenth of a short-K tile's instruction stream).
  const int cols = p.tiles_n * p.splitk;
  const int my_col = vid % cols, tm_first = vid / cols, tm_step = (int)gridDim.x / cols;
  const int my_n0 = (my_col % p.tiles_n) * BN;
  const int my_kb = (my_col / p.tiles_n) * p.ks_per_split;
  const int my_ke = min(p.ksteps, my_kb + p.ks_per_split);
  auto item_at = [&](int tm) {
    WorkItem it;
    it.m0 = tm * BM;
    it.n0 = my_n0;
    it.kb = my_kb;
    it.ke = my_ke;
    return it;
  };

  // this lane's rows / source slots in the DMA instructions of its wave: instruction j covers tile rows 8 j' .. 8 j' + 7
  // (lane >> 3 picks the row, lane & 7 the LDS slot, which holds source slot (lane & 7) ^ ((row >> 1) & 7))
  int arow[APW], aslot[APW], brow[BPW], bslot[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    arow[j] = RPI * (APW * wave + j) + lane / SPR;
    aslot[j] = (lane % SPR) ^ swz(arow[j]);
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    if constexpr (BT) {  // instruction (panel, 16-row group): brow = k row inside the stage, bslot = first column
      const int ins = BPW * wave + j;  // BK / 16 instructions (16 k rows x 32 columns each) per 32-column panel
      brow[j] = (ins % (BK / 16)) * 16 + (lane >> 2);
      bslot[j] = (ins / (BK / 16)) * 32 + (lane & 3) * 8;
    } else {
      brow[j] = RPI * (BPW * wave + j) + lane / SPR;
      bslot[j] = (lane % SPR) ^ swz(brow[j]);
    }
  }

  // gathered A operand (GatherGeom): source row of each of this lane's tile rows, re-derived when the m tile changes
  int abase[APW], abase_m0 = -1;
  auto issue = [&](const WorkItem& it, int ks, int st) {
    if (p.debug & 2) return;
    if constexpr (GA) {
      const int C = p.ga.C;
      const int mrem = p.M - 1 - it.m0;
      if (it.m0 != abase_m0) {
#pragma unroll
        for (int j = 0; j < APW; ++j) abase[j] = gather_base(p.ga, it.m0 + min(arow[j], mrem));
        abase_m0 = it.m0;
      }
      const int t = fdiv((unsigned)ks, p.ga.dKpt);   // k steps [t kpt, (t + 1) kpt) belong to tap t
      const int kin = (ks - t * p.ga.dKpt.d) * BK;
      unsigned char* sa = smem + st * STAGE;
      unsigned va[APW], vb[BPW];
#pragma unroll
      for (int j = 0; j < APW; ++j) va[j] = ((unsigned)abase[j] * (unsigned)C + (unsigned)(aslot[j] * 8)) * 2u;
      dma16_groupN<APW, 1024>(p.a1 + (size_t)gather_tapoff(p.ga, t) * C + kin, va, sa + APW * wave * 1024);
      const int nrem = p.N - 1 - it.n0;
#pragma unroll
      for (int j = 0; j < BPW; ++j) vb[j] = (unsigned)((min(brow[j], nrem) * C + bslot[j] * 8) * 2);
      dma16_groupN<BPW, 1024>(p.w + ((size_t)t * p.N + it.n0) * C + kin, vb, sa + A_BYTES + BPW * wave * 1024);
      return;
    }
    const int k0 = ks * BK;
    const bool first = k0 < p.C1;
    const bf16* a = first ? p.a1 : p.a2;
    const int lda = first ? p.C1 : p.C2;
    const int ka = first ? k0 : k0 - p.C1;
    unsigned char* sa = smem + st * STAGE;
    // grouped issue (dma.h): wave-uniform 64-bit base of the tile's k slice + per-lane 32-bit byte offsets
    unsigned va[APW], vb[BPW];
    const int mrem = p.M - 1 - it.m0;  // tail rows repeat the last row; their outputs are never stored
#pragma unroll
    for (int j = 0; j < APW; ++j) va[j] = (unsigned)((min(arow[j], mrem) * lda + aslot[j] * 8) * 2);
    dma16_groupN<APW, 1024>(a + (size_t)it.m0 * lda + ka, va, sa + APW * wave * 1024);
    if constexpr (BT) {
#pragma unroll
      for (int j = 0; j < BPW; ++j)  // columns past N feed only outputs that are never stored
        vb[j] = (unsigned)((brow[j] * p.N + min(it.n0 + bslot[j], p.N - 8)) * 2);
      dma16_groupN<BPW, 1024>(p.w + (size_t)k0 * p.N, vb, sa + A_BYTES + BPW * wave * 1024);
    } else {
      const int nrem = p.N - 1 - it.n0;
#pragma unroll
      for (int j = 0; j < BPW; ++j) vb[j] = (unsigned)((min(brow[j], nrem) * p.K + bslot[j] * 8) * 2);
      dma16_groupN<BPW, 1024>(p.w + (size_t)it.n0 * p.K + k0, vb, sa + A_BYTES + BPW * wave * 1024);
    }
  };

  // fragment read offsets inside a stage: row * ROWB + ((slot ^ swz(row)) << 4), slot = 2 kk + h
  int aoff[MI], boff[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * WM + i * 32 + r;
    aoff[i] = row * ROWB + ((h ^ swz(row)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * WN + j * 32 + r;
    if constexpr (BT) boff[j] = A_BYTES + (wn * NJ + j) * (BK * 64) + (h * 8 * 32 + r) * 2;  // panel, k row 8 h, column r
    else boff[j] = A_BYTES + row * ROWB + ((h ^ swz(row)) << 4);
  }

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bf16* out = reinterpret_cast<bf16*>(p.out);

  if (tm_first >= p.tiles_m) return;
  // running InstanceNorm sums of this workgroup (in_acc): lane (r, h) owns column r of each of its n tiles
  float rs1[NJ], rs2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { rs1[j] = 0.f; rs2[j] = 0.f; }
  int stat_b = 0, stat_n0 = 0;
  bool stat_live = false;
  auto stat_flush = [&]() {
    // lane halves by shuffle, the two waves that share the columns through LDS, one fp64 atomic pair per column.
    // The host guarantees that a tile never straddles two batch items and that all its rows exist.
    float* red = reinterpret_cast<float*>(smem + R * STAGE + EPI_BYTES);  // [4 waves][WN][2]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float s1 = rs1[j] + __shfl_xor(rs1[j], 32, 64);
      const float s2 = rs2[j] + __shfl_xor(rs2[j], 32, 64);
      if (h == 0) {
        red[(w4 * WN + j * 32 + r) * 2] = s1;
        red[(w4 * WN + j * 32 + r) * 2 + 1] = s2;
      }
      rs1[j] = 0.f; rs2[j] = 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (vmcnt untouched: the next stages stay in flight)
    if (tid < BN) {
      const int wq = tid / WN, col = tid - wq * WN;  // waves wq (wm = 0) and wq + 2 (wm = 1) hold these columns
      const int n = stat_n0 + tid;
      if (n < p.N) {
        const float t1 = red[(wq * WN + col) * 2] + red[((wq + 2) * WN + col) * 2];
        const float t2 = red[(wq * WN + col) * 2 + 1] + red[((wq + 2) * WN + col) * 2 + 1];
        atomicAdd(&p.in_acc[((size_t)stat_b * p.N + n) * 2], (double)t1);
        atomicAdd(&p.in_acc[((size_t)stat_b * p.N + n) * 2 + 1], (double)t2);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // red may be rewritten by the next flush
  };
  // two cursors over the flat sequence of (work item, k step) stages of this workgroup: issue runs R - 1 ahead of compute
  int tm = tm_first, itm = tm_first;
  WorkItem cur = item_at(tm), icur = cur;
  int ks = cur.kb, iks = cur.kb;
  bool idone = false;
  auto issue_next = [&](int slot) {
    issue(icur, iks, slot);
    if (++iks == icur.ke) {
      itm += tm_step;
      if (itm < p.tiles_m) { icur.m0 = itm * BM; iks = icur.kb; }
      else idone = true;
    }
  };
  int ahead = 0, st = 0, ist = 0;  // stages issued but not yet computed; compute slot; next issue slot
  bool after_epi = false;
  constexpr int EPI_STORES = (BM / 64) * (32 / (512 / (BN / 2)));  // MI * NR store instructions per wave and whole tile
#pragma unroll 1
  for (int q = 0; q < R - 1 && !idone; ++q) { issue_next(ist); ist = ist + 1 == R ? 0 : ist + 1; ++ahead; }
  while (true) {
    // stage st has landed in every wave (the ahead - 1 younger ones may still be in flight), and every wave is done
    // reading the slot computed in the previous iteration - the one refilled next
    // right after a whole tile's epilogue its stores need not drain: the stage needed now was issued before them
    // (a write-heavy short-K GEMM otherwise alternates between "stores acknowledged" and "next stage landed")
    if (after_epi) wait_stage<IPW, R - 2, EPI_STORES>(ahead - 1);
    else wait_stage<IPW, R - 2>(ahead - 1);
    after_epi = false;
    if (!idone) { issue_next(ist); ist = ist + 1 == R ? 0 : ist + 1; ++ahead; }
    const bool last = ks + 1 == cur.ke;
    --ahead;

    const unsigned char* sa = smem + st * STAGE;
#pragma unroll
    for (int k2 = 0; k2 < BK / 16 / KG; ++k2) {
      const int kk = kg * (BK / 16 / KG) + k2;
      bf16x8 fa[MI], fb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + (aoff[i] ^ (kk << 5)));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (BT) fb[j] = Mma<bf16>::gather(reinterpret_cast<const bf16*>(sa + boff[j] + kk * 16 * 64), 32);
        else fb[j] = *reinterpret_cast<const bf16x8*>(sa + (boff[j] ^ (kk << 5)));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }

    if (last) {
      if constexpr (KG == 2) {
        // lane-contiguous fp32 dump (conflict-free), one barrier; group 1 then goes straight to the next tile, and cannot
        // overwrite the buffer before group 0 has read it: its next dump sits behind the loop-top barriers of a whole tile
        float* kred = reinterpret_cast<float*>(smem + R * STAGE + EPI_BYTES + RED_BYTES) + w4 * (64 * MI * NJ * 16);
        if (kg == 1) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int e = 0; e < 16; ++e) kred[((i * NJ + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kg == 0) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[i][j][e] += kred[((i * NJ + j) * 16 + e) * 64 + lane];
        }
      }
      if (KG == 1 && p.in_acc) {
        // InstanceNorm statistics of the output (the 1x1x1 convs of the ResNet bottlenecks feed an InstanceNorm): column
        // sums of the fp32 accumulators, kept in registers ACROSS the tiles of this persistent workgroup while they
        // belong to the same (batch item, n tile) - with gridDim a multiple of tiles_n that is all of them but one
        // switch per batch item - and flushed as one fp64 atomic pair per column (stat_flush).  Per address that is
        // ~one atomic per workgroup instead of one per tile (3456 same-address atomics made a 29 us GEMM 46 us).
        const int sb = cur.m0 / p.in_rows;
        if (stat_live && (sb != stat_b || cur.n0 != stat_n0)) stat_flush();
        stat_b = sb; stat_n0 = cur.n0; stat_live = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e]; rs1[j] += v; rs2[j] += v * v; }
      }
      if (kg == 0) {  // (KG == 2: group 1 has handed its sums over and goes on to the next tile)
      // epilogue through a wave-private fp32 patch of PR rows x the wave's WN columns (outside the DMA ring: the next
      // tile is already in flight): every store instruction then writes whole rows of the wave's column range
      // (128 B for WN = 64) instead of 64-B halves of a line.  The residual vectors of all MI NR patches are requested
      // FIRST: loaded patch by patch they are dependent global round trips (most of a short trunk GEMM's life).
      constexpr int PR = 512 / WN;    // patch rows: 16 (WN = 32) or 8 (WN = 64)
      constexpr int PLD = WN + 4;
      constexpr int NR = 32 / PR;     // patches per 32-row MFMA tile
      constexpr int RPR = 16 / NR;    // accumulator registers per patch
      constexpr int VPRW = WN / 8;    // 8-column vectors per patch row
      static_assert(4 * PR * PLD * 4 <= EPI_BYTES, "patch must fit the epilogue region");
      const int erow = lane / VPRW, ecv = lane % VPRW;
      const bool want_res = p.ep.residual != nullptr && p.splitk <= 1;
      bf16x8 resv[MI * NR];
      if (want_res) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int t = 0; t < NR; ++t) {
            const int m = cur.m0 + wm * WM + i * 32 + t * PR + erow, n = cur.n0 + wn * WN + ecv * 8;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (m < p.M && n < p.N) {
              bf16* dst;
              const size_t off = epilogue_offset(p.ep, out, m, n, dst);
              if (dst == out)   // (split output: the residual belongs to the `out` part)
                v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.ep.residual) + off);
            }
            resv[i * NR + t] = v;
          }
      }
      // this lane's 8 bias values: its columns are the same in every patch of the tile - one load, not one per patch
      float bv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      if (p.ep.bias && p.splitk <= 1) {
        const int n = cur.n0 + wn * WN + ecv * 8;
        if (n < p.N) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[e] = p.ep.bias[n + e];
        }
      }
      float* patch = reinterpret_cast<float*>(smem + R * STAGE) + w4 * (EPI_BYTES / 16);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int t = 0; t < NR; ++t) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int ee = 0; ee < RPR; ++ee)
              patch[((ee & 3) + 8 * (ee >> 2) + 4 * h) * PLD + j * 32 + r] = acc[i][j][RPR * t + ee];
          __builtin_amdgcn_wave_barrier();
          const int mb = cur.m0 + wm * WM + i * 32 + t * PR;
          const int nb = cur.n0 + wn * WN;
          if (p.splitk > 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int idx = lane + 64 * q;
              const int row = idx / WN, col = idx % WN;
              if (mb + row < p.M && nb + col < p.N)
                atomicAdd(p.ws + (size_t)(mb + row) * p.N + nb + col, patch[row * PLD + col]);
            }
          } else {
            const int m = mb + erow, n = nb + ecv * 8;
            if (m < p.M && n < p.N) {
              float x[8];
              load8(&patch[erow * PLD + ecv * 8], x);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] += bv[e];
              if (p.ep.pre_out) store8(reinterpret_cast<bf16*>(p.ep.pre_out) + (size_t)m * p.ep.ldc + n, x);
              if (p.ep.act == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = gelu_erf(x[e]);
              }
              if (want_res) {
                const bf16x8 rv = resv[i * NR + t];
                if (p.ep.act == 2) {   // GELU backward: `residual` holds the saved pre-activation, the product is dY.W
#pragma unroll
                  for (int e = 0; e < 8; ++e) x[e] *= gelu_erf_grad((float)rv[e]);
                } else {
#pragma unroll
                  for (int e = 0; e < 8; ++e) x[e] += (float)rv[e];
                }
              }
              bf16* dst;
              const size_t off = epilogue_offset(p.ep, out, m, n, dst);
              if (!(p.debug & 1)) store8(dst + off, x);
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      // (whole tiles only: a tail tile may skip a store, and counting one that was never issued would end the wait early;
      //  pre_out doubles the stores - counting fewer than issued only waits longer)
      after_epi = kg == 0 && p.splitk <= 1 && cur.m0 + BM <= p.M && cur.n0 + BN <= p.N && !(p.debug & 1);
      tm += tm_step;
      if (tm >= p.tiles_m) break;
      cur.m0 = tm * BM;
      ks = cur.kb;
    } else {
      ++ks;
    }
    st = st + 1 == R ? 0 : st + 1;
  }
  if (stat_live) stat_flush();
}


// ---------------------------------------------------------------------------------------------------------
// gemm_nt_stream : the short-K, long-M layers (1x1x1 convs of the ResNet bottlenecks and their data gradients:
// M = 10^5..10^6 rows, K = 32 / 64 / 128, N a multiple of 128) - pure streams of A in and out out, where the general
// kernel above spends ~10 k cycles per 128 x 128 tile on its own instruction stream and LDS round trips (measured with
// neither loads nor stores issued) against ~1 k cycles of MFMA.  Here:
//   * a workgroup keeps ONE 128-column tile of W for life, as MFMA fragments in registers (K / 16 x 2 per wave), and walks
//     down the m tiles: a stage is a whole 128 x K tile of A (one barrier and one DMA group per tile, no W traffic);
//   * the product is taken transposed (W fragment as the MFMA A operand): lane (r, h) owns output ROW r of a 32-row tile
//     and, in registers 4 g + c, columns 8 g + 4 h + c.  One v_permlane32_swap per register pair trades the upper lane
//     half of group 2 q against the lower half of group 2 q + 1, leaving 8 consecutive columns (16 q + 8 h ..) per lane:
//     16-byte bf16 stores straight from registers - no LDS patch, no write -> read round trips in the epilogue;
//   * the epilogue stores of tile t stay in flight across the wait for tile t + 1 (vmcnt retires in issue order and the
//     stage needed was issued before them);
//   * InstanceNorm sums (in_acc): per-lane partial column sums in registers across all tiles of a batch item, reduced over
//     the 32 row lanes once per flush.
// KS = K / 16.  Requires M % 128 == 0, N % 128 == 0, one source, plain epilogue (optional residual), no split K.
// ---------------------------------------------------------------------------------------------------------
// all-reduce (sum) inside each 16-lane DPP row: row_ror 8, 4, 2, 1
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

template <int KS, bool STATS>
__global__ __launch_bounds__(256, 2) void gemm_nt_stream_kernel(const GemmNtArgs p) {
  constexpr int K = KS * 16;
  constexpr int ROWB = K * 2;               // bytes of an A row
  constexpr int SPR = K / 8;                // 16-B slots per row
  constexpr int RPI = 1024 / ROWB;          // rows per DMA wave-instruction
  constexpr int STAGE = 128 * ROWB;
  constexpr int R = STAGE >= 32768 ? 2 : 4; // ring depth: 64 KiB (K = 128, 64), 32 KiB (K = 32)
  constexpr int APW = 128 / RPI / 4;        // DMA instructions per wave and stage: 8 / 4 / 2
  constexpr int GRP = APW >= 4 ? 4 : APW;   // ... issued in groups of
  constexpr int RBR = 256 / ROWB;           // rows per 256-B bank row
  constexpr int STORES = 8;                 // store instructions per wave and tile (2 x 2 x 2 vectors)
  constexpr int OUT_BYTES = 4 * 32 * 128;   // per wave 32 rows x 64 bf16 columns: the store staging (and the stat_flush scratch)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[R * STAGE + OUT_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int cols = p.N >> 7;
  const int n0 = (vid % cols) * 128;
  // a workgroup takes a CONTIGUOUS range of m tiles (consecutive rows stream from consecutive DRAM pages, and a range
  // rarely straddles two batch items: one stat_flush per workgroup); the `cols` workgroups of an index share their A
  // tiles through the XCD's L2 (consecutive vid)
  const int per_col = (int)gridDim.x / cols;
  const int chunk = ((p.M >> 7) + per_col - 1) / per_col;
  int tm = (vid / cols) * chunk;
  const int tiles_m = min(p.M >> 7, tm + chunk);
  constexpr int tm_step = 1;
  if (tm >= tiles_m) return;
  auto swz = [](int row) { return (row / RBR) & (SPR - 1); };

  // ---- this wave's W fragments: columns n0 + 64 wn + 32 j + r, k = 16 kk + 8 h .. + 7 ----
  bf16x8 wf[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if (p.w_kn) {  // W[k][n] (data gradient reads the forward weight as stored)
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[j][kk][e] = p.w[(size_t)(kk * 16 + 8 * h + e) * p.N + n];
      } else {
        wf[j][kk] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)n * K + kk * 16 + 8 * h);
      }
    }
  }

  // ---- DMA: per-lane source offsets inside a tile (constant for the kernel), lane-linear LDS image ----
  unsigned va[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = RPI * (APW * wave + j) + lane / SPR;
    va[j] = (unsigned)(row * ROWB + (((lane % SPR) ^ swz(row)) << 4));
  }
  auto issue = [&](int t, int slot) {
    const bf16* base = p.a1 + (size_t)t * 128 * K;
    unsigned char* dst = smem + slot * STAGE + APW * wave * 1024;
#pragma unroll
    for (int g = 0; g < APW / GRP; ++g) {
      unsigned v[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) v[j] = va[g * GRP + j];
      dma16_groupN<GRP, 1024>(base, v, dst + g * GRP * 1024);
    }
  };
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + r;
    aoff[i] = row * ROWB + ((h ^ swz(row)) << 4);  // k step kk: ^ (kk << 5)
  }

  f32x16 acc[2][2];
  float rs1[STATS ? 2 : 1][16], rs2[STATS ? 2 : 1][16];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { rs1[j][e] = 0.f; rs2[j][e] = 0.f; }
  }
  int stat_b = -1;
  auto stat_flush = [&]() {
    if constexpr (STATS) {
      float* red = reinterpret_cast<float*>(smem + R * STAGE);  // [4 waves][64 columns][2] (the idle store staging)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          // this lane's rows of column (e & 3) + 8 (e >> 2) + 4 h: sum the 32 row lanes - rotations inside the 16-lane
          // DPP rows (VALU, no LDS crossbar), then one exchange between the two rows of the half
          float s1 = row16_sum(rs1[j][e]), s2 = row16_sum(rs2[j][e]);
          s1 += __shfl_xor(s1, 16, 64);
          s2 += __shfl_xor(s2, 16, 64);
          if (r == 0) {
            const int col = j * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            red[(wave * 64 + col) * 2] = s1;
            red[(wave * 64 + col) * 2 + 1] = s2;
          }
          rs1[j][e] = 0.f; rs2[j][e] = 0.f;
        }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (vmcnt untouched: the ring stays in flight)
      if (tid < 128) {
        const int wq = tid >> 6, col = tid & 63;  // waves wq (wm = 0) and wq + 2 (wm = 1) hold these columns
        const float t1 = red[(wq * 64 + col) * 2] + red[((wq + 2) * 64 + col) * 2];
        const float t2 = red[(wq * 64 + col) * 2 + 1] + red[((wq + 2) * 64 + col) * 2 + 1];
        atomicAdd(&p.in_acc[((size_t)stat_b * p.N + n0 + tid) * 2], (double)t1);
        atomicAdd(&p.in_acc[((size_t)stat_b * p.N + n0 + tid) * 2 + 1], (double)t2);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // red may be rewritten by the next flush
    }
  };

  bf16* out = reinterpret_cast<bf16*>(p.out);
  // (the InstanceNorm-sum variant never carries a residual - stream_ok - and has no registers to spare for one)
  const bf16* res = STATS ? nullptr : reinterpret_cast<const bf16*>(p.ep.residual);
  const size_t ldc = (size_t)p.ep.ldc;
  const size_t lane_off = (size_t)(wm * 64 + r) * ldc + n0 + wn * 64 + 8 * h;  // + (128 tm + 32 i) ldc + 32 j + 16 q

  // ---- ring: tile t of this workgroup lives in slot (count % R); R - 1 tiles in flight ahead of the one computed ----
  int itm = tm, ahead = 0, slot = 0, islot = 0;
#pragma unroll 1
  for (int q = 0; q < R - 1 && itm < tiles_m; ++q) { issue(itm, islot); itm += tm_step; islot = islot + 1 == R ? 0 : islot + 1; ++ahead; }
  bool after_store = false;
#pragma unroll 1
  for (; tm < tiles_m; tm += tm_step) {
    if (after_store) wait_stage<APW, R - 2, STORES>(ahead - 1);
    else wait_stage<APW, R - 2>(ahead - 1);
    if (itm < tiles_m) { issue(itm, islot); itm += tm_step; islot = islot + 1 == R ? 0 : islot + 1; ++ahead; }
    --ahead;

    const unsigned char* sa = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      bf16x8 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + (aoff[i] ^ (kk << 5)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][kk], fa[i], acc[i][j], 0, 0, 0);
    }

    if constexpr (STATS) {
      const int sb = (tm * 128) / p.in_rows;
      if (sb != stat_b) {
        if (stat_b >= 0) stat_flush();
        stat_b = sb;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i) { const float v = acc[i][j][e]; rs1[j][e] += v; rs2[j][e] += v * v; }
    }

    // rows of this lane in the register layout (residual reads) and in the store layout (8 lanes per 128-B row segment)
    const size_t tile_off = (size_t)tm * 128 * ldc + lane_off;
    bf16x8 resv[2][2][2];
    if (res) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            resv[i][j][q] = *reinterpret_cast<const bf16x8*>(res + tile_off + (size_t)(32 * i) * ldc + 32 * j + 16 * q);
    }
    unsigned char* stg = smem + R * STAGE + wave * (32 * 128);
    const size_t st_off = (size_t)(tm * 128 + wm * 64 + (lane >> 3)) * ldc + n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float x[8];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * q + c]),
                                                              __float_as_uint(acc[i][j][8 * q + 4 + c]), false, false);
            x[c] = __uint_as_float(sw[0]);
            x[4 + c] = __uint_as_float(sw[1]);
          }
          if (res) {
            const bf16x8 rv = resv[i][j][q];
            if (p.ep.act == 2) {   // GELU backward: `residual` holds the saved pre-activation (ctu_epilogue.act)
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] *= gelu_erf_grad((float)rv[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) x[e] += (float)rv[e];
            }
          }
          // row r, 16-B slot 4 j + 2 q + h of the 128-B row, slot ^ (row & 7): conflict-free for the 8-lane write groups
          // (8 rows, one slot) and for the 16-lane read groups below (2 rows x 8 slots)
          store8(reinterpret_cast<bf16*>(stg + r * 128 + (((4 * j + 2 * q + h) ^ (r & 7)) << 4)), x);
        }
      __builtin_amdgcn_wave_barrier();
      bf16x8 ov[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = 8 * t + (lane >> 3);
        ov[t] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
      }
      __builtin_amdgcn_wave_barrier();
      if (!(p.debug & 1)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)  // each store: 8 whole 128-B row segments
          *reinterpret_cast<bf16x8*>(out + st_off + (size_t)(32 * i + 8 * t) * ldc) = ov[t];
      }
    }
    after_store = !(p.debug & 1);
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (STATS && stat_b >= 0) {
    // the flush scratch aliases wave 0's store staging tile: every wave must be through its last epilogue first (the
    // flushes inside the loop sit behind the loop-top barrier; without this one a fast wave's partial sums could land in
    // the tile wave 0 was still reading back - rare garbage / NaN rows in the last tile of a workgroup's range)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stat_flush();
  }
}

static bool stream_ok(const GemmNtArgs& p) {
  const ctu_epilogue& e = p.ep;
  return (p.K == 32 || p.K == 64 || p.K == 128) && p.a2 == nullptr && p.C1 == p.K && p.N % 128 == 0 && p.M % 128 == 0 &&
         p.M >= 128 * 256 && p.splitk <= 1 && !e.bias && (e.act == 0 || (e.act == 2 && e.residual)) && !e.pre_out &&
         !e.scatter && e.n_split <= 0 &&
         (!p.in_acc || (p.in_rows % 128 == 0 && !e.residual)) && !(ctu_option_route() & CTU_ROUTE_NT_NO_STREAM);
}
static void launch_stream(const GemmNtArgs& p, hipStream_t stream) {
  const int cols = p.N / 128;
  const int grid = cols >= 512 ? cols : 512 / cols * cols;
  const dim3 g(grid), b(256);
#define CTU_STREAM(KS)                                                                                   \
  do {                                                                                                   \
    if (p.in_acc) hipLaunchKernelGGL((gemm_nt_stream_kernel<KS, true>), g, b, 0, stream, p);             \
    else hipLaunchKernelGGL((gemm_nt_stream_kernel<KS, false>), g, b, 0, stream, p);                     \
  } while (0)
  if (p.K == 128) CTU_STREAM(8);
  else if (p.K == 64) CTU_STREAM(4);
  else CTU_STREAM(2);
#undef CTU_STREAM
}

bool gather_geom_from(const ctu_geom* g, GatherGeom& gg) {
  gg.on = 0;
  const int taps = g->kd * g->kh * g->kw;
  const int64_t src_rows = (int64_t)g->B * g->Di * g->Hi * g->Wi;
  if (g->mode != 0 || g->C2 != 0 || g->C1 % 64 != 0 || taps < 1 || taps > 8 || g->pd || g->ph || g->pw ||
      (g->Do - 1) * g->sd + g->kd > g->Di || (g->Ho - 1) * g->sh + g->kh > g->Hi || (g->Wo - 1) * g->sw + g->kw > g->Wi ||
      src_rows * g->C1 * 2 >= (1ll << 32) || (int64_t)g->B * g->Do * g->Ho * g->Wo >= (1ll << 31))
    return false;
  gg.on = 1;
  gg.C = g->C1;
  gg.taps = taps;
  gg.sB = g->Di * g->Hi * g->Wi;
  gg.sD = g->sd * g->Hi * g->Wi;
  gg.sH = g->sh * g->Wi;
  gg.sW = g->sw;
  for (int t = 0; t < 8; ++t) gg.tapoff[t] = 0;
  for (int td = 0, t = 0; td < g->kd; ++td)
    for (int th = 0; th < g->kh; ++th)
      for (int tw = 0; tw < g->kw; ++tw, ++t) gg.tapoff[t] = (td * g->Hi + th) * g->Wi + tw;
  gg.dWo = fast_div(g->Wo);
  gg.dHo = fast_div(g->Ho);
  gg.dDo = fast_div(g->Do);
  gg.dKpt = fast_div(g->C1 / 64);
  return true;
}

int ctu_option_nt_debug();
bool gemm_nt_narrow_ok(const GemmNtArgs& p);                       // gemm_narrow.hip
void launch_gemm_nt_narrow(const GemmNtArgs& p, hipStream_t stream);
int launch_gemm_nt_dma(GemmNtArgs& p, hipStream_t stream) {
  p.debug = ctu_option_nt_debug();
  if (p.ga.on && (p.w_kn || p.a2 || p.ga.C % 64 != 0 || p.K != p.ga.taps * p.ga.C)) return -1;
  if (!p.ga.on && stream_ok(p)) { launch_stream(p, stream); return 0; }
  if (!p.ga.on && gemm_nt_narrow_ok(p)) { launch_gemm_nt_narrow(p, stream); return 0; }
  // tile choice: the largest tile that still yields ~200 work items for the 512 resident workgroups; the 864-token
  // ViT trunk (M = 864) gets 64 x 64 tiles rather than a split K with its atomics and second pass
  const auto items = [&](int bm, int bn) { return (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const bool bk32 = p.K % 64 != 0;  // (K % 32 == 0 checked by the caller); 128-row tiles only
  int BM = 128, BN = p.N <= 64 ? 64 : 128;
  if (!bk32 && items(BM, BN) < 200 && p.splitk <= 1) {
    BN = 64;
    if (items(BM, BN) < 200) BM = 64;
  }
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  // 64 x 64 tiles (the 864-token ViT trunk: a handful of MFMAs per 64-deep stage, the stage hand-over dominates) take
  // 128-deep stages when K allows: half the barriers and waits per tile, 8 MFMAs per wave between them
  const bool bk128 = !bk32 && BM == 64 && BN == 64 && p.K % 128 == 0 && p.K >= 512 && (!p.a2 || p.C1 % 128 == 0) &&
                     !p.ga.on && !(ctu_option_route() & CTU_ROUTE_NT_NO_BK128);
  if (p.ga.on) p.ga.dKpt = fast_div(p.ga.C / 64);
  p.ksteps = p.K / (bk32 ? 32 : bk128 ? 128 : 64);
  if (p.splitk > p.ksteps) p.splitk = p.ksteps;
  if (p.splitk < 1) p.splitk = 1;
  p.ks_per_split = (p.ksteps + p.splitk - 1) / p.splitk;
  p.splitk = (p.ksteps + p.ks_per_split - 1) / p.ks_per_split;
  const int64_t nwork = (int64_t)p.tiles_m * p.tiles_n * p.splitk;
  if (nwork >= (1ll << 31)) return -1;
  p.nwork = (int)nwork;
  const bool one_per_cu = p.nwork <= 256;  // a single workgroup per CU may take the whole LDS for a deeper ring
  // two resident workgroups per CU, persistent over the work items; a multiple of the (n tile, k split) columns so that a
  // workgroup keeps its column (see the kernel)
  const int cols = p.tiles_n * p.splitk;
  const int grid = p.nwork < 512 ? p.nwork : (cols >= 512 ? cols : 512 / cols * cols);
  const dim3 g(grid), b(256);
  if (p.ga.on) {
    if (BM == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 4, false, 64, 1, true>), g, b, 0, stream, p);
    else if (BN == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 64, 2, false, 64, 1, true>), g, b, 0, stream, p);
    else if (one_per_cu) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 3, false, 64, 1, true>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 2, false, 64, 1, true>), g, b, 0, stream, p);
  } else if (bk32) {
    if (p.w_kn) return -1;
    if (BN == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 64, 3, false, 32>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 3, false, 32>), g, b, 0, stream, p);
  } else if (bk128 && !p.in_acc && p.splitk <= 1 && !(ctu_option_route() & CTU_ROUTE_NT_NO_KG2)) {
    const dim3 b8(512);  // eight waves: two k groups per tile (see the kernel)
    if (p.w_kn) hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 3, true, 128, 2>), g, b8, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 3, false, 128, 2>), g, b8, 0, stream, p);
  } else if (bk128) {
    if (p.w_kn) hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 3, true, 128>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 3, false, 128>), g, b, 0, stream, p);
  } else if (p.w_kn) {
    if (BM == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 4, true>), g, b, 0, stream, p);
    else if (BN == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 64, 2, true>), g, b, 0, stream, p);
    else if (one_per_cu) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 3, true>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 2, true>), g, b, 0, stream, p);
  } else {
    if (BM == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<64, 64, 4, false>), g, b, 0, stream, p);
    else if (BN == 64) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 64, 2, false>), g, b, 0, stream, p);
    else if (one_per_cu) hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 3, false>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((gemm_nt_dma_kernel<128, 128, 2, false>), g, b, 0, stream, p);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// gemm_tn_dma : dw[n][c] += sum_m P[m][n] Q[m][c].  The reduction index m is the ROW of both operands, so both are
// read "transposed" (ds_read_b64_tr_b16: 8 consecutive rows of one column per lane).  LDS image per stage of 64 rows:
// 32-column panels [panel][64 rows][32 cols] (64-B rows, plain lane-linear DMA): a transposed read touches
// 4 consecutive rows x 64 B = one whole 256-B bank row, conflict free without padding or swizzle.
// Workgroup = TN x TC tile of dw over one row split; 4 waves as 2 x 2; 2 stages; 2 workgroups per CU.
// Row tail (m >= M): P rows come from the zero page (Q rows are clamped - any finite value times zero).
// Column tails are clamped too: a column >= N only feeds output rows that are never written.
// bias_grad (column sums of P) costs one extra MFMA against a fragment of ones in the workgroups of c tile 0.
// ---------------------------------------------------------------------------------------------------------
template <int TN, int TC, int R>
__global__ __launch_bounds__(256, (R * (TN + TC) * 128 > 80 * 1024) ? 1 : 2) void gemm_tn_dma_kernel(const GemmTnArgs a) {
  constexpr int BKM = 64;
  constexpr int PANEL = BKM * 64;              // bytes of one 32-column panel of a stage
  constexpr int NP = TN / 32, NQ = TC / 32;    // panels
  constexpr int STAGE = (NP + NQ) * PANEL;
  constexpr int PI = NP, QI = NQ;              // DMA instructions per wave and stage (4 per panel, 4 waves)
  constexpr int AI = TN / 64, AJ = TC / 64;    // MFMA tiles per wave
  static_assert((R - 2) * (PI + QI) <= 8 || (R - 2) * (PI + QI) == 16, "counted vmcnt switch: 0..8, 12, 16, 24");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[R * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = a.tiles_n * a.tiles_c;
  const int tile = vid % tiles, split = vid / tiles;  // the tiles of one split sit together on one XCD
  const int n0 = (tile / a.tiles_c) * TN;
  const int c0 = (tile % a.tiles_c) * TC;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const bf16* zero = reinterpret_cast<const bf16*>(g_zero16);

  // this lane's source columns per DMA instruction: instruction q of the wave = (panel, 16-row group) pair
  // wave w owns instruction indices w + 4 q over the 4 NP (4 NQ) instructions of P (Q)
  int pcol[PI], qcol[QI];
  const bf16* qbase[QI];
  int qld[QI];
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int ins = wave + 4 * q, panel = ins >> 2;
    pcol[q] = min(n0 + panel * 32 + (lane & 3) * 8, a.N - 8);
  }
#pragma unroll
  for (int q = 0; q < QI; ++q) {
    const int ins = wave + 4 * q, panel = ins >> 2;
    const int c = min(c0 + panel * 32 + (lane & 3) * 8, a.C - 8);
    const bool first = c < a.C1;
    qbase[q] = first ? a.q1 : a.q2;
    qld[q] = first ? a.C1 : a.C2;
    qcol[q] = first ? c : c - a.C1;
  }
  // full stages take the grouped issue (dma.h): the per-lane byte offsets inside a stage never change, only the
  // wave-uniform base advances by 64 rows; the last (partial) stage and two-source Q operands keep the per-lane form
  unsigned vpo[PI], vqo[QI];
#pragma unroll
  for (int q = 0; q < PI; ++q) {
    const int ins = wave + 4 * q;
    vpo[q] = (unsigned)((((ins & 3) * 16 + (lane >> 2)) * a.ldp + pcol[q]) * 2);
  }
#pragma unroll
  for (int q = 0; q < QI; ++q) {
    const int ins = wave + 4 * q;
    vqo[q] = (unsigned)((((ins & 3) * 16 + (lane >> 2)) * a.C1 + qcol[q]) * 2);
  }
  // gathered Q (GatherGeom): a 32-column panel lies inside one tap (ga.C % 32 == 0), so tap and channel offset of each DMA
  // instruction are wave-uniform; every instruction of a wave covers the same 16 rows of the stage (ins & 3 == wave)
  const bool gathered = a.ga.on != 0;
  int qtap[QI];
  if (gathered) {
#pragma unroll
    for (int q = 0; q < QI; ++q) {
      const int cp = min(c0 + q * 32, a.C - 32);
      const int t = cp / a.ga.C;
      qtap[q] = gather_tapoff(a.ga, t);
      qcol[q] = cp - t * a.ga.C + (lane & 3) * 8;
    }
  }
  const bool groupable = a.C2 == 0;
  auto issue = [&](int mb, int st) {
    unsigned char* dst = smem + st * STAGE;
    if (gathered) {
      const int mrow = mb + wave * 16 + (lane >> 2);
      if (mb + BKM <= m_end) {
        dma16_groupN<PI, 4096>(a.p + (size_t)mb * a.ldp, vpo, dst + wave * 1024);
      } else {
#pragma unroll
        for (int q = 0; q < PI; ++q) dma16(mrow < m_end ? a.p + (size_t)mrow * a.ldp + pcol[q] : zero, dst + (wave + 4 * q) * 1024);
      }
      const size_t rb = (size_t)gather_base(a.ga, min(mrow, a.M - 1));
#pragma unroll
      for (int q = 0; q < QI; ++q)
        dma16(a.q1 + (rb + qtap[q]) * a.ga.C + qcol[q], dst + NP * PANEL + (wave + 4 * q) * 1024);
      return;
    }
    if (groupable && mb + BKM <= m_end) {
      dma16_groupN<PI, 4096>(a.p + (size_t)mb * a.ldp, vpo, dst + wave * 1024);
      dma16_groupN<QI, 4096>(a.q1 + (size_t)mb * a.C1, vqo, dst + NP * PANEL + wave * 1024);
      return;
    }
#pragma unroll
    for (int q = 0; q < PI; ++q) {
      const int ins = wave + 4 * q;
      const int m = mb + (ins & 3) * 16 + (lane >> 2);
      dma16(m < m_end ? a.p + (size_t)m * a.ldp + pcol[q] : zero, dst + ins * 1024);
    }
#pragma unroll
    for (int q = 0; q < QI; ++q) {
      const int ins = wave + 4 * q;
      const int m = min(mb + (ins & 3) * 16 + (lane >> 2), a.M - 1);
      dma16(qbase[q] + (size_t)m * qld[q] + qcol[q], dst + NP * PANEL + ins * 1024);
    }
  };

  f32x16 acc[AI][AJ], accb[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < AJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  const bool do_bias = a.bias_grad != nullptr && (tile % a.tiles_c) == 0;  // workgroup-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  {
    // R - 1 stages in flight ahead of the one computed; counted vmcnt leaves the younger ones pending
    const int nst = (m_end - m_begin + BKM - 1) / BKM;
    int issued = 0, st = 0, ist = 0;
#pragma unroll 1
    for (; issued < R - 1 && issued < nst; ++issued) { issue(m_begin + issued * BKM, ist); ist = ist + 1 == R ? 0 : ist + 1; }
    for (int sidx = 0; sidx < nst; ++sidx, st = st + 1 == R ? 0 : st + 1) {
      // stage st landed in every wave; every wave is done with the slot computed last iteration (refilled next)
      wait_vm_then_barrier_n((issued - sidx - 1) * (PI + QI));
      if (issued < nst) { issue(m_begin + issued * BKM, ist); ist = ist + 1 == R ? 0 : ist + 1; ++issued; }
      const bf16* sp = reinterpret_cast<const bf16*>(smem + st * STAGE);
      const bf16* sq = sp + NP * (PANEL / 2);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int krow = kk * 16 + h * 8;
        bf16x8 fa[AI], fb[AJ];
#pragma unroll
        for (int i = 0; i < AI; ++i) fa[i] = Mma<bf16>::gather(sp + (wn * AI + i) * (PANEL / 2) + krow * 32 + r, 32);
#pragma unroll
        for (int j = 0; j < AJ; ++j) fb[j] = Mma<bf16>::gather(sq + (wc * AJ + j) * (PANEL / 2) + krow * 32 + r, 32);
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int j = 0; j < AJ; ++j) Mma<bf16>::mma(fa[i], fb[j], acc[i][j]);
        if (do_bias && wc == 0) {
#pragma unroll
          for (int i = 0; i < AI; ++i) Mma<bf16>::mma(fa[i], ones, accb[i]);
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < AI; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + (wn * AI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int c = c0 + (wc * AJ + j) * 32 + r;
        if (c < a.C) {
          size_t o = (size_t)n * a.C + c;
          if (gathered) {   // dw = [taps][N][ga.C]
            const int t = (c0 + (wc * AJ + j) * 32) / a.ga.C;
            o = ((size_t)t * a.N + n) * a.ga.C + (c - t * a.ga.C);
          }
          // (a plain `dw[o] += v` for a sole writer is a dependent load-add-store per element: 64 serialized global
          // round trips per lane, 25 us of a 35 us trunk launch; the no-return atomic is fire-and-forget)
          if (a.part) a.part[(size_t)split * a.N * a.C + o] = acc[i][j][e];
          else atomicAdd(&a.dw[o], acc[i][j][e]);
        }
      }
      if (do_bias && wc == 0 && r == 0) atomicAdd(&a.bias_grad[n], accb[i][e]);
    }
  }
}

int launch_gemm_tn_dma(GemmTnArgs& a, float* ws, int64_t ws_floats, hipStream_t stream) {
  const auto ntiles = [&](int tn, int tc) { return (int64_t)((a.N + tn - 1) / tn) * ((a.C + tc - 1) / tc); };
  // 128 x 128 tiles when they alone give the chip enough work or the panel is small and M long (row splits fill the
  // chip); otherwise 64 x 64
  int TN = 128, TC = 128;
  if (a.N <= 64 || a.C <= 64 || (ntiles(128, 128) < 128 && a.M < 8192)) { TN = 64; TC = 64; }
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_c = (a.C + TC - 1) / TC;
  const int64_t tiles = (int64_t)a.tiles_n * a.tiles_c;
  // row splits: aim at ~512 workgroups, but every split costs a pass over the whole panel - bound that reduction
  // traffic (two-stage: splits x E x 8 B at ~4 TB/s) by half the MMA time it can save (~0.45 us per 64-row stage)
  const int64_t E = (int64_t)a.N * a.C;
  const int64_t stages = (a.M + 63) / 64;
  int splits = tiles >= 256 ? 1 : (int)((512 + tiles - 1) / tiles);
  if (E > (1 << 18)) {
    const double cap = (double)stages * 0.45 * 4.0e6 / (16.0 * (double)E);
    if (splits > (int)cap) splits = (int)cap;
  }
  const int max_splits = (a.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.part = nullptr;
  const int two_stage_min = E > (1 << 18) ? 2 : 8;  // small panels with a few splits: plain atomics are cheaper
  bool two_stage = ws && splits >= two_stage_min;
  if (two_stage && (int64_t)splits * E > ws_floats) {
    if (E > (1 << 18)) two_stage = false;  // large panel that does not fit: atomics
    else splits = (int)(ws_floats / E);
  }
  int rps = (a.M + splits - 1) / splits;
  rps = ((rps + 63) / 64) * 64;
  splits = (a.M + rps - 1) / rps;
  if (two_stage && splits >= two_stage_min) a.part = ws;
  a.rows_per_split = rps;
  a.splits = splits;
  const int64_t nwork = tiles * splits;
  if (nwork >= (1ll << 31)) return -1;
  const dim3 g((unsigned)nwork), b(256);
  if (TN == 64) hipLaunchKernelGGL((gemm_tn_dma_kernel<64, 64, 4>), g, b, 0, stream, a);
  else if (nwork <= 256) hipLaunchKernelGGL((gemm_tn_dma_kernel<128, 128, 3>), g, b, 0, stream, a);  // one per CU: deeper ring
  else hipLaunchKernelGGL((gemm_tn_dma_kernel<128, 128, 2>), g, b, 0, stream, a);
  return 0;
}
