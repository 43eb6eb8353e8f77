// 3x3x3, stride 1, padding 1 convolution (forward and data gradient) with an LDS-resident halo brick.
//
// The generic implicit GEMM (igemm.hip) re-gathers every input voxel once per tap through L2 and pays a workgroup
// barrier every 4 MFMAs per wave.  Here a workgroup owns a 4 x 8 x 8 brick of output voxels: for each 32-channel
// chunk it stages the 6 x 10 x 10 input halo in LDS ONCE and all 27 taps read their A fragments from it at a
// constant address offset; the B fragments (weights) are streamed global -> registers in MFMA-fragment order
// (1 KiB contiguous per wave-instruction, L2-resident panel), so there is no barrier inside a chunk at all.
//
// Replaces nn.Conv3d(k=3, s=1, p=1, bias=False) and its input gradient: ResBlock.conv1/conv2
// (networks/hybrid_CTUNet.py:57-74) and Bottleneck.conv2 (networks/resnet.py:98) - 80 % of the model's FLOPs
// (SURVEY.md section 2.2 row K1).  Input may be the channel concat of two tensors (torch.cat, hybrid_CTUNet.py:199,618);
// the data gradient then splits its output columns over two destinations.
#include "mma.h"
#include "dma.h"

#define HB_D 4
#define HB_H 8
#define HB_W 8
#define HALO_D (HB_D + 2)
#define HALO_H (HB_H + 2)
#define HALO_W (HB_W + 2)
#define HALO_VOX (HALO_D * HALO_H * HALO_W)  // 600

int ctu_option_nt_debug();  // attention.hip

struct HaloArgs {
  const void* x1;
  const void* x2;
  const void* wfrag;  // [K/32][27][2][Npad/32][64 lanes][8]  (ctu_pack_frag)
  void* out;
  void* out2;
  int B, D, H, W, C1, C2, N, n_split, ldc, ldc2;
  int nbd, nbh, nbw, ntn;  // brick grid, 32-wide n tiles in the whole panel
  double* in_acc;          // optional [B][N][2]: += (sum y, sum y^2) per batch item and channel (InstanceNorm statistics)
  const void* residual;    // optional [rows][ldc] (same dtype): added to the part of the result that goes to `out` (bf16 DMA kernel)
  const void* residual2;   // optional [rows][ldc2]: added to the part that goes to `out2` (n_split > 0)
  float* part;             // split over input channels (small volumes): fp32 partial outputs [split][rows][ntn * 32], else NULL
  int hc_per_split;        // 16-channel half chunks per split (blockIdx.z)
  // source layout of x1 (bf16 DMA kernel): element (voxel m, channel c) sits at m * vs1 + (c >> 4) * bs1 + (c & 15).
  // channels-last: vs1 = C1, bs1 = 16.  CTU_LAYOUT_B16 ("[C/16][voxels][16]", written by ctu_in_apply / ctu_in_bwd_apply for
  // exactly this consumer): vs1 = 16, bs1 = 16 * voxels - a halo row of 10 voxels is then ONE 320-byte run instead of ten
  // 32-byte pieces of ten different cache lines.
  int vs1;
  int64_t bs1;
  int debug;               // measurement hook (ctu_set_option "nt_debug"): 1 = no output stores, 4 = no weight DMA, 8 = no halo DMA
  void* part_stamps;       // nt_debug & 16: the workspace receives the STAMP build's cycle sums instead of split partials
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv3_halo_kernel(const HaloArgs p) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int CK = 32;              // channels per chunk
  constexpr int LDT = CK + EV;        // padded voxel row in LDS (80 B bf16 / 144 B f32)
  constexpr int VPV = CK / EV;        // 16-byte vectors per voxel per chunk
  constexpr int FRAG = 64 * 8;        // elements of one packed fragment tile (64 lanes x 8)
  constexpr int STAGE_LD = 32 + 4;
  constexpr size_t LDS_HALO = (size_t)HALO_VOX * LDT * sizeof(T);
  constexpr size_t LDS_EPI = 4 * 32 * (size_t)STAGE_LD * sizeof(float);
  constexpr size_t LDS_BYTES = LDS_HALO > LDS_EPI ? LDS_HALO : LDS_EPI;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  T* halo = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  // brick coordinates (blockIdx.x) and n block (blockIdx.y)
  int t = blockIdx.x;
  const int bw = t % p.nbw; t /= p.nbw;
  const int bh = t % p.nbh; t /= p.nbh;
  const int bd = t % p.nbd;
  const int b = t / p.nbd;
  const int d0 = bd * HB_D, h0 = bh * HB_H, w0 = bw * HB_W;
  const int nt0 = blockIdx.y * NT;  // first 32-wide n tile of this block
  const int K = p.C1 + p.C2;
  const int nchunks = K / CK;
  const T* x1 = reinterpret_cast<const T*>(p.x1);
  const T* x2 = reinterpret_cast<const T*>(p.x2);
  const T* wf = reinterpret_cast<const T*>(p.wfrag);

  // A-fragment base addresses (elements) of this lane for its two 32-voxel row tiles: wave = d slice,
  // local voxel v = 32 i + r -> (hh, ww) = (v >> 3, v & 7); tap (0,0,0) reads halo voxel (wave, hh, ww)
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = 32 * i + r;
    abase[i] = ((wave * HALO_H + (v >> 3)) * HALO_W + (v & 7)) * LDT + h * 8;
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fp32 parity mode: the MFMA chain of one tap (32 products, exact f32 FMAs) is added into a float64 accumulator and
  // restarted.  A single f32 chain over K = 27 x C (1728 .. 27648 terms) carries a rounding error of ~sqrt(K) ulp-sized
  // steps relative to the RUNNING sum - the dominant noise source of this network in fp32 (the InstanceNorm stack
  // amplifies it ~1000x); restarted chains keep it at the level of the final rounding.
  constexpr bool WIDE = sizeof(T) == 4;
  double acc64[WIDE ? 2 : 1][WIDE ? NT : 1][WIDE ? 16 : 1];
  if (WIDE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc64[i][j][e] = 0.0;
  }

  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * CK;
    const T* src = (c0 < p.C1) ? x1 : x2;
    const int cs = (c0 < p.C1) ? p.C1 : p.C2;
    const int cc = (c0 < p.C1) ? c0 : c0 - p.C1;
    __syncthreads();  // every wave is done reading the previous chunk's halo
    for (int v = tid; v < HALO_VOX * VPV; v += 256) {
      const int vox = v / VPV, part = v - vox * VPV;
      const int hw = vox % HALO_W;
      const int tq = vox / HALO_W;
      const int hh = tq % HALO_H, hd = tq / HALO_H;
      const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
        val = *reinterpret_cast<const u32x4*>(src + ((((size_t)b * p.D + gd) * p.H + gh) * p.W + gw) * cs + cc + part * EV);
      *reinterpret_cast<u32x4*>(&halo[vox * LDT + part * EV]) = val;
    }
    __syncthreads();

    // 54 k-steps (27 taps x 2 halves of the chunk); B fragments stream from the packed panel, prefetched one step
    // ahead in registers
    const T* wch = wf + ((size_t)ch * 54 * p.ntn + nt0) * FRAG + lane * 8;
    // two named register sets (static indexing: a runtime-indexed fragment array would go to scratch)
    typename Mma<T>::Frag fb0[NT], fb1[NT];
    auto load_b = [&](int ks, typename Mma<T>::Frag (&f)[NT]) {
#pragma unroll
      for (int j = 0; j < NT; ++j) f[j] = Mma<T>::load(wch + ((size_t)ks * p.ntn + j) * FRAG);
    };
    auto step = [&](int tap, int kk, const typename Mma<T>::Frag (&f)[NT]) {
      const int tw = tap % 3, tq = tap / 3;
      const int th = tq % 3, td = tq / 3;
      const int aoff = ((td * HALO_H + th) * HALO_W + tw) * LDT + kk * 16;
      typename Mma<T>::Frag fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = Mma<T>::load(&halo[abase[i] + aoff]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) Mma<T>::mma(fa[i], f[j], acc[i][j]);
    };
    load_b(0, fb0);
    for (int tap = 0; tap < 27; ++tap) {
      load_b(2 * tap + 1, fb1);
      step(tap, 0, fb0);
      if (tap + 1 < 27) load_b(2 * tap + 2, fb0);
      step(tap, 1, fb1);
      if (WIDE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              acc64[i][j][e] += (double)acc[i][j][e];
              acc[i][j][e] = 0.f;
            }
      }
    }
  }
  if (WIDE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = (float)acc64[i][j][e];
  }

  // ---- epilogue: per (row tile, n tile): accumulators -> wave-private LDS tile -> 8-wide vectors -> global ----
  __syncthreads();
  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * STAGE_LD;
  T* out = reinterpret_cast<T*>(p.out);
  T* out2 = reinterpret_cast<T*>(p.out2);
  const int gd = d0 + wave;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * h) * STAGE_LD + r] = acc[i][j][e];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int vv = lane + 64 * q;       // 128 vectors: 32 rows x 4 groups of 8 channels
        const int row = vv >> 2, cv = vv & 3;
        const int v = 32 * i + row;
        const int gh = h0 + (v >> 3), gw = w0 + (v & 7);
        const int n = (nt0 + j) * 32 + cv * 8;
        if (gd < p.D && gh < p.H && gw < p.W && n < p.N) {
          float xv[8];
          load8(&stage[row * STAGE_LD + cv * 8], xv);
          const size_t m = (((size_t)b * p.D + gd) * p.H + gh) * p.W + gw;
          if (p.n_split > 0 && n >= p.n_split) store8(out2 + m * p.ldc2 + (n - p.n_split), xv);
          else store8(out + m * p.ldc + n, xv);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 production variant: every operand reaches LDS by LDS-DMA (global_load_lds_dwordx4), nothing is staged
// through registers and no wave waits on a global load it issued itself in the same stage.
//   * halo: 16-channel half chunks (600 voxels x 32 B), double buffered; the next half chunk is in flight while
//     the 27 taps of the current one are computed.  The LDS image is lane-linear (DMA rule), so the bank swizzle
//     sits on the SOURCE address: 16-B slot of (voxel v, k-half hs) = 2 v + (hs ^ ((row(v) >> 1) & 1)), and the
//     32 rows of an M tile are dealt to voxels so that each 16-lane ds_read_b128 group covers two halo rows two
//     apart -> all 16 slots of the 256-B bank row, conflict free for every tap.
//   * weights: the 1 KiB MFMA-order fragments of 3 taps form a stage, fetched ONCE per workgroup into a 3-stage
//     ring (the register-streaming kernel above fetched them once per wave: 64 B/clk/CU of L1 traffic at full
//     MFMA rate).  One raw s_barrier per stage (6 NT MFMAs per wave); counted vmcnt keeps two stages plus the
//     halo prefetch in flight across it.
// ---------------------------------------------------------------------------------------------------------
// row of a 32-row M tile -> (h, w) inside the tile's 4 x 8 voxel patch (see the bank argument above)
__device__ __forceinline__ void halo_row_to_hw(int row, int& hh, int& ww) {
  const int q = row >> 2;
  hh = 2 * (q >> 2) + ((q ^ (q >> 1) ^ (q >> 2)) & 1);
  ww = 4 * ((q >> 1) & 1) + (row & 3);
}

// RD = depth of the weight ring: RD - 1 stages are in flight ahead of the one computed.  NT <= 2 takes RD = 2: with
// 51 KiB of LDS and ~110 VGPRs three workgroups fit a CU, and their interleaving hides more than a deeper ring does.
// TPS = taps per weight stage: 3 (one (td, th) row of taps; nine stages and barriers per half chunk) or 9 (one td plane; three
// stages per half chunk: a two-deep ring then prefetches 9 taps = 1 152 MFMA cycles ahead at NT = 2 instead of 384, for 75 KiB of
// LDS and two workgroups per CU instead of three - ctu_set_option("route", 16), measured in DESIGN.md section 8).
// STAMP (diagnostic build, ctu_set_option("nt_debug", 16); never launched by the product): every wave accumulates the shader
// cycles it spends (0) before its first stage barrier opens, (1) waiting at stage barriers (own DMA wait + s_barrier),
// (2) between barriers (fragment reads + MFMAs + DMA issue), (3) in the epilogue, and writes the four sums to
// stamps[workgroup][wave][4] - memory nothing else reads (MI355X_MICROARCH.md, in-kernel stamps).
// HSP = gather pieces per weight-loader wave (waves 1 - 3); wave 0 keeps the other 19 - 3 HSP.  HSP = 0: wave 0 gathers alone (the
// rule above).  HSP > 0 deals the burst over all four waves so that no wave reaches the next barriers later than the others by a
// whole gather; a loader wave issues its pieces AFTER the weight stage of the same barrier, and for the RD - 1 barriers that follow
// the pieces are younger than the awaited weights: those waits leave HSP more operations in flight (counted, not vmcnt(0)).
// BP ("batch pair", volumes whose height is 4 or 12 mod 8, e.g. 12 x 12 x 24): the brick is 4 x 4 x 8 voxels of TWO batch items
// (b = 2 pair + i for the wave's M tile i) instead of 4 x 8 x 8 of one - a 12-row volume then takes 3 bricks of 4 rows per pair of
// items where it took 2 x 2 bricks of 8 rows, a quarter of them padding (36 -> 27 bricks at 2 x 12 x 12 x 24).  The halo image is two
// 6 x 6 x 10 boxes (720 voxels, 23 DMA pieces; image row = 36 i + 6 hd + hh), everything else - weight stages, barriers, the M-tile
// row order - is unchanged.
template <int NT, int RD = (NT <= 2 ? 2 : 3), int TPS = 3, bool STAMP = false, int HSP = 4, bool BP = false>
__global__ __launch_bounds__(256, (RD == 2 && TPS == 3 && !BP) ? 3 : 2) void conv3_halo_dma_kernel(const HaloArgs p) {
  unsigned long long st_t0 = 0, st_wait = 0, st_work = 0, st_pro = 0, st_mark = 0;
  if (STAMP) st_t0 = st_mark = __builtin_amdgcn_s_memtime();
  constexpr int BH = BP ? 4 : HB_H;                       // brick rows
  constexpr int HH = BH + 2;                              // halo rows per d slice (and item)
  constexpr int HVOX = (BP ? 2 : 1) * HALO_D * HH * HALO_W;   // 600 | 720
  constexpr int HINS = (2 * HVOX + 63) / 64;          // DMA wave-instructions per halo half chunk: 19 (1216 slots >= 1200) | 23
  constexpr int HBUF = HINS * 1024;
  constexpr int SPC = 27 / TPS;       // stages per half chunk
  constexpr int SFR = TPS * NT;       // weight fragments (1 KiB each) per stage
  constexpr int SBYTES = SFR * 1024;
  constexpr int RING0 = 2 * HBUF;
  constexpr int STAGE_LD = 32 + 4;
  constexpr int LDS_MAIN = RING0 + RD * SBYTES;
  constexpr int LDS_EPI = 4 * 32 * STAGE_LD * 4;
  constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int bw = t % p.nbw; t /= p.nbw;
  const int bh = t % p.nbh; t /= p.nbh;
  const int bd = t % p.nbd;
  const int b = (BP ? 2 : 1) * (t / p.nbd);   // BP: the first of the two batch items (p.nbh counts 4-row bricks then)
  const int d0 = bd * HB_D, h0 = bh * BH, w0 = bw * HB_W;
  const int nt0 = blockIdx.y * NT;
  const int hc_b = blockIdx.z * p.hc_per_split;                      // this workgroup's range of 16-channel half chunks
  const int HC = min((p.C1 + p.C2) / 16, hc_b + p.hc_per_split);    // (exclusive end)
  const int U = (HC - hc_b) * SPC;                                   // stages
  const bf16* x1 = reinterpret_cast<const bf16*>(p.x1);
  const bf16* x2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* wf = reinterpret_cast<const bf16*>(p.wfrag);

  // ---- DMA roles.  vmcnt retires a wave's vector-memory operations IN ISSUE ORDER: if one wave issues a halo prefetch
  // (needed nine stages later) and then weight stages (needed two stages later), every wait for those weights also waits
  // for the halo gather - its HBM round trip lands on the critical path three stages after it was issued (measured: 22 %
  // of the kernel at 128 -> 128 @ 48 x 48 x 96, 37 % at 64 -> 64 @ 96^3).  So wave 0 issues ALL halo instructions and
  // nothing else - its one wait (vmcnt(0) at the first stage of a chunk) sees a prefetch that is nine stages old - and
  // waves 1 - 3 issue the weight stages.
  // per-lane halo sources of this wave's gather instructions (piece k fills slots 64 k ..; this wave owns pieces kbase + j)
  constexpr int HP0 = HINS - 3 * HSP;          // pieces of wave 0
  constexpr int HPM = HP0 > HSP ? HP0 : HSP;
  static_assert(HP0 >= 1, "wave 0 keeps at least one gather piece");
  const int kbase = wave == 0 ? 0 : HP0 + (wave - 1) * HSP;
  int hm[HPM];
  unsigned hpart = 0;
#pragma unroll
  for (int k = 0; k < HPM; ++k) {
    const int S = (kbase + k) * 64 + lane;
    int m = -1;
    if (S < 2 * HVOX) {
      const int vox = S >> 1, hs = S & 1;
      const int R = vox / HALO_W, hw = vox - R * HALO_W;   // image row (BP: 36 item + 6 hd + hh)
      const int item = BP ? R / (HALO_D * HH) : 0;
      const int Ri = R - item * (HALO_D * HH);
      const int hd = Ri / HH, hh = Ri - hd * HH;
      const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
      hpart |= (unsigned)(hs ^ ((R >> 1) & 1)) << k;
      if ((unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
        m = (((b + item) * p.D + gd) * p.H + gh) * p.W + gw;
    }
    hm[k] = m;
  }
  constexpr int BW = SFR / 3;  // weight DMA instructions per loader wave (1 - 3) and stage: NT (3-tap stages), 3 NT (9-tap)

  // this wave's gather pieces of half chunk hc (hm[] keeps static indices)
  auto issue_halo = [&](int hc) {
    if (p.debug & 8) return;
    const int c0 = hc * 16;
    const bool first = c0 < p.C1;
    const bf16* src = first ? x1 : x2;
    const int vs = first ? p.vs1 : p.C2;
    const int cc = first ? c0 : c0 - p.C1;
    src += first ? (size_t)(cc >> 4) * p.bs1 : (size_t)cc;
    unsigned char* dst = smem + (hc & 1) * HBUF;
    dst += kbase * 1024;
#pragma unroll
    for (int k = 0; k < HPM; ++k) {
      if (wave == 0 ? k < HP0 : k < HSP) {
        const bf16* g = hm[k] >= 0 ? src + (size_t)hm[k] * vs + ((hpart >> k) & 1) * 8
                                   : reinterpret_cast<const bf16*>(g_zero16);
        dma16(g, dst + k * 1024);
      }
    }
  };
  // weight stage = fragments f = wave, wave + 4, wave + 8 of this wave: per-lane byte offsets inside the stage are
  // fixed for the kernel, the stage's base address is wave-uniform -> one grouped DMA issue per stage (dma16_group)
  // loader wave w (1 - 3) takes fragments f = (w - 1) + 3 k, k < NT: LDS destinations 3 KiB apart
  unsigned bvoff[BW];
#pragma unroll
  for (int k = 0; k < BW; ++k) {
    const int f = (wave > 0 ? wave - 1 : 0) + 3 * k, tapi = f / NT, j = f - tapi * NT;
    bvoff[k] = (unsigned)(((tapi * 2 * p.ntn + j) * 512 + lane * 8) * 2);
  }
  auto issue_b = [&](int hc, int s, int slot) {  // waves 1 - 3 only
    if (p.debug & 4) return;
    const bf16* base = wf + ((size_t)(((hc >> 1) * 27 + TPS * s) * 2 + (hc & 1)) * p.ntn + nt0) * 512;
    unsigned char* dst = smem + RING0 + slot * SBYTES + (wave - 1) * 1024;
    if constexpr (BW == 1 || BW == 2 || BW == 4) {
      dma16_groupN<BW, 3072>(base, bvoff, dst);
    } else if constexpr (BW == 3) {
      const unsigned a[2] = {bvoff[0], bvoff[1]}, c[1] = {bvoff[2]};
      dma16_groupN<2, 3072>(base, a, dst);
      dma16_groupN<1, 3072>(base, c, dst + 2 * 3072);
    } else {
      static_assert(BW == 6, "weight instructions per loader wave");
      const unsigned a[4] = {bvoff[0], bvoff[1], bvoff[2], bvoff[3]}, c[2] = {bvoff[4], bvoff[5]};
      dma16_groupN<4, 3072>(base, a, dst);
      dma16_groupN<2, 3072>(base, c, dst + 4 * 3072);
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int vrow, vcol;
  halo_row_to_hw(r, vrow, vcol);

  if (wave == 0) {
    issue_halo(hc_b);
  } else {
    if (HSP > 0) issue_halo(hc_b);  // older than every weight stage: the first weight wait covers it
    issue_b(hc_b, 0, 0);
    if (RD == 3) issue_b(hc_b, 1, 1);
  }
  int hyoung = 0;  // loader waves: barriers to come at which the last gather is younger than the awaited weights
  int hc = hc_b, s = 0, rs = 0;               // stage being computed: half chunk, (td, th) index, ring slot
  int ihc = hc_b, is = RD - 1, irs = RD - 1;  // stage being fetched (RD - 1 ahead)
  for (int u = 0; u < U; ++u) {
    if (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (u > 0) st_work += t - st_mark;
      st_mark = t;
    }
    if (wave == 0) {
      // the halo of this chunk (issued nine stages ago, the only vector-memory traffic of this wave) has landed
      if (s == 0) wait_vm_then_barrier<0>();
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // the next half chunk's 19 gather pieces go out in ONE burst, into the buffer whose last readers passed this barrier
      // (dealt out three per stage, or one per tap behind the MFMAs, this wave reaches more barriers late and every wave
      // waits longer: +3.5 % / +10 % kernel time)
      if (s == 0 && hc + 1 < HC) issue_halo(hc + 1);
    } else {
      // own weight DMAs of this stage have landed; still in flight: the RD - 2 younger stages (and a younger gather)
      if (HSP > 0 && hyoung > 0) {
        wait_vm_then_barrier<(RD - 2) * BW + HSP>();
        --hyoung;
      } else {
        wait_vm_then_barrier<(RD - 2) * BW>();
      }
      issue_b(ihc, is, irs);  // past the end: refetches the last stage into a slot nobody reads (keeps the counts)
      if (HSP > 0 && s == 0 && hc + 1 < HC) {  // (behind the MFMAs of this stage instead: no better, r41 in profiles/)
        issue_halo(hc + 1);
        hyoung = (p.debug & 8) ? 0 : RD - 1;
      }
    }
    if (STAMP) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if (u == 0) st_pro = t - st_t0; else st_wait += t - st_mark;
      st_mark = t;
    }
    if (u + RD < U) {
      if (++is == SPC) { is = 0; ++ihc; }
    }
    irs = irs == RD - 1 ? 0 : irs + 1;

    const int td = TPS == 9 ? s : ((s * 11) >> 5);
    const unsigned char* hb = smem + (hc & 1) * HBUF;
    const unsigned char* rb = smem + RING0 + rs * SBYTES + lane * 16;
#pragma unroll
    for (int t3 = 0; t3 < TPS / 3; ++t3) {
      const int th = TPS == 9 ? t3 : s - 3 * td;
      int aoff[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int R = BP ? i * (HALO_D * HH) + (wave + td) * HH + vrow + th : (wave + td) * HALO_H + 4 * i + vrow + th;
        aoff[i] = ((R * HALO_W + vcol) * 2 + (h ^ ((R >> 1) & 1))) * 16;
      }
      // (measured: requesting the fragments of all three taps up front + s_setprio around the 6 NT MFMAs is 5-10 % SLOWER
      // than letting hipcc interleave reads and MFMAs tap by tap - the second wave of the SIMD hides the read latency)
#pragma unroll
      for (int tw = 0; tw < 3; ++tw) {
        bf16x8 fa[2], fb[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(hb + aoff[i] + tw * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(rb + ((t3 * 3 + tw) * NT + j) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            CTU_MFMA_YIELD_HERE
          }
      }
    }
    if (++s == SPC) { s = 0; ++hc; }
    rs = rs == RD - 1 ? 0 : rs + 1;
  }
  wait_vm_then_barrier<0>();  // the tail refetches have landed; LDS is free for the epilogue
  if (STAMP) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    st_work += t - st_mark;
    st_mark = t;
  }

  if (p.in_acc && !p.part) {
    // InstanceNorm statistics of this output, taken from the fp32 accumulators: the separate pass that re-reads the
    // tensor from HBM (in_stats_kernel) disappears.  Lane (r, h) sums its 2 x 16 rows of column r per n tile, the two
    // lane halves are combined by a shuffle, the four waves through LDS, then one fp64 atomic pair per channel and brick.
    float* red = reinterpret_cast<float*>(smem + 20 * 1024);  // [items][4 waves][NT * 32 columns][2], beyond the staging tiles
    constexpr int NI = BP ? 2 : 1;   // batch items per brick: tile i belongs to item i (BP) or both tiles to the one item
    const bool full = d0 + HB_D <= p.D && h0 + BH <= p.H && w0 + HB_W <= p.W;  // brick entirely inside the volume
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s1[NI], s2[NI];
#pragma unroll
      for (int it = 0; it < NI; ++it) { s1[it] = 0.f; s2[it] = 0.f; }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[i][j][e];
          if (!full) {
            int hh, ww;
            halo_row_to_hw((e & 3) + 8 * (e >> 2) + 4 * h, hh, ww);
            if (d0 + wave >= p.D || h0 + (BP ? 0 : 4 * i) + hh >= p.H || w0 + ww >= p.W) v = 0.f;
          }
          s1[BP ? i : 0] += v;
          s2[BP ? i : 0] += v * v;
        }
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const float a1 = s1[it] + __shfl_xor(s1[it], 32, 64), a2 = s2[it] + __shfl_xor(s2[it], 32, 64);
        if (h == 0) {
          red[((it * 4 + wave) * NT * 32 + j * 32 + r) * 2] = a1;
          red[((it * 4 + wave) * NT * 32 + j * 32 + r) * 2 + 1] = a2;
        }
      }
    }
    __syncthreads();
    if (tid < NT * 32) {
      const int n = nt0 * 32 + tid;
      if (n < p.N) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) {
            t1 += red[((it * 4 + wv) * NT * 32 + tid) * 2];
            t2 += red[((it * 4 + wv) * NT * 32 + tid) * 2 + 1];
          }
          atomicAdd(&p.in_acc[((size_t)(b + it) * p.N + n) * 2], (double)t1);
          atomicAdd(&p.in_acc[((size_t)(b + it) * p.N + n) * 2 + 1], (double)t2);
        }
      }
    }
    __syncthreads();  // red is inside the region the staging tiles of other waves do not touch, but keep phases apart
  }

  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * STAGE_LD;
  bf16* out = reinterpret_cast<bf16*>(p.out);
  bf16* out2 = reinterpret_cast<bf16*>(p.out2);
  const int gd = d0 + wave;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * h) * STAGE_LD + r] = acc[i][j][e];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int vv = lane + 64 * q;  // 128 vectors: 32 rows x 4 groups of 8 channels
        const int row = vv >> 2, cv = vv & 3;
        int hh, ww;
        halo_row_to_hw(row, hh, ww);
        const int gh = h0 + (BP ? 0 : 4 * i) + hh, gw = w0 + ww;
        const int n = (nt0 + j) * 32 + cv * 8;
        if (gd < p.D && gh < p.H && gw < p.W && n < p.N) {
          float xv[8];
          load8(&stage[row * STAGE_LD + cv * 8], xv);
          const size_t m = (((size_t)(b + (BP ? i : 0)) * p.D + gd) * p.H + gh) * p.W + gw;
          if (p.part) store8(p.part + ((size_t)blockIdx.z * p.B * p.D * p.H * p.W + m) * (p.ntn * 32) + n, xv);
          else if (p.n_split > 0 && n >= p.n_split) {
            if (p.residual2) {
              float rr[8];
              load8(reinterpret_cast<const bf16*>(p.residual2) + m * p.ldc2 + (n - p.n_split), rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] += rr[e];
            }
            store8(out2 + m * p.ldc2 + (n - p.n_split), xv);
          } else {
            if (p.residual) {
              float rr[8];
              load8(reinterpret_cast<const bf16*>(p.residual) + m * p.ldc + n, rr);
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] += rr[e];
            }
            if (!(p.debug & 1)) store8(out + m * p.ldc + n, xv);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  if (STAMP && lane == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.part_stamps) +
                              ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + wave) * 4;
    dst[0] = st_pro; dst[1] = st_wait; dst[2] = st_work; dst[3] = __builtin_amdgcn_s_memtime() - st_mark;
  }
}

// ---------------------------------------------------------------------------------------------------------
// N-split variant (round 4; four n tiles = 128 output channels per workgroup): wave w owns n tile w for ALL 256 voxels of the
// brick (8 M tiles) instead of a d slice for all n tiles.  What that buys: the weights a wave needs are its own - fetched by
// LDS-DMA into a WAVE-PRIVATE ring (PF fragments of 1 KiB, one per tap, PF - 1 taps ahead), waited for with a counted vmcnt and
// read back by the same wave, so no other wave is involved: the workgroup barrier per weight stage (every 6 NT MFMAs per
// wave; 26 - 32 % of the kernel waiting there) is gone, one barrier per 16-channel half chunk remains
// (216 MFMAs per wave) for the shared halo image.  What it costs: every wave reads the whole halo image (8 A fragments per tap
// instead of 2: 9 KiB of LDS reads per 8 MFMAs instead of 6), and 128 accumulator registers per lane.
// Same HaloArgs, same fragment-order panel, same halo image and swizzle, same epilogue forms as conv3_halo_dma_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int PF>
__global__ __launch_bounds__(256, 2) void conv3_halo_ns_kernel(const HaloArgs p) {
  constexpr int HVOX = HALO_D * HALO_H * HALO_W;      // 600
  constexpr int HINS = (2 * HVOX + 63) / 64;          // 19 DMA wave-instructions per halo half chunk
  constexpr int HBUF = HINS * 1024;
  constexpr int HPW = (HINS + 3) / 4;                 // gather pieces per wave: 5, 5, 5, 4
  constexpr int RING0 = 2 * HBUF;
  constexpr int STAGE_LD = 32 + 4;
  constexpr int LDS_MAIN = RING0 + 4 * PF * 1024;
  constexpr int LDS_EPI = 4 * 32 * STAGE_LD * 4;
  constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(27 % PF == 0 || PF > 27, "ring slots must repeat per half chunk");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int bw = t % p.nbw; t /= p.nbw;
  const int bh = t % p.nbh; t /= p.nbh;
  const int bd = t % p.nbd;
  const int b = t / p.nbd;
  const int d0 = bd * HB_D, h0 = bh * HB_H, w0 = bw * HB_W;
  const int nt = blockIdx.y * 4 + wave;                              // this wave's n tile
  const int hc_b = blockIdx.z * p.hc_per_split;
  const int HC = min((p.C1 + p.C2) / 16, hc_b + p.hc_per_split);    // (exclusive end)
  const bf16* x1 = reinterpret_cast<const bf16*>(p.x1);
  const bf16* x2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* wf = reinterpret_cast<const bf16*>(p.wfrag);

  // per-lane halo sources of this wave's gather pieces (piece k fills slots 64 k ..): as in conv3_halo_dma_kernel
  // (a wave whose share is one piece short repeats its last piece: every wave issues HPW instructions, static vmcnt counts)
  int hm[HPW], hdst[HPW];
  unsigned hpart = 0;
#pragma unroll
  for (int k = 0; k < HPW; ++k) {
    const int piece = min(wave * HPW + k, HINS - 1);
    const int S = piece * 64 + lane;
    int m = -1;
    if (S < 2 * HVOX) {
      const int vox = S >> 1, hs = S & 1;
      const int R = vox / HALO_W, hw = vox - R * HALO_W;
      const int hd = R / HALO_H, hh = R - hd * HALO_H;
      const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
      hpart |= (unsigned)(hs ^ ((R >> 1) & 1)) << k;
      if ((unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
        m = ((b * p.D + gd) * p.H + gh) * p.W + gw;
    }
    hm[k] = m;
    hdst[k] = piece * 1024;
  }
  auto issue_halo = [&](int hc) {
    const int c0 = hc * 16;
    const bool first = c0 < p.C1;
    const bf16* src = first ? x1 : x2;
    const int vs = first ? p.vs1 : p.C2;
    const int cc = first ? c0 : c0 - p.C1;
    src += first ? (size_t)(cc >> 4) * p.bs1 : (size_t)cc;
    unsigned char* dst = smem + (hc & 1) * HBUF;
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
      const bf16* g = (hm[k] >= 0 && !(p.debug & 8)) ? src + (size_t)hm[k] * vs + ((hpart >> k) & 1) * 8 : reinterpret_cast<const bf16*>(g_zero16);
      dma16(g, dst + hdst[k]);
    }
  };
  // weight fragment of (half chunk hc, tap) for this wave's n tile: 1 KiB contiguous in the fragment-order panel
  unsigned char* ring = smem + RING0 + wave * PF * 1024;
  auto issue_b = [&](int hc, int tap, int slot) {
    const bf16* g = wf + ((size_t)((((hc >> 1) * 27 + tap) * 2 + (hc & 1)) * p.ntn + nt)) * 512 + lane * 8;
    dma16((p.debug & 4) ? reinterpret_cast<const void*>(g_zero16) : reinterpret_cast<const void*>(g), ring + slot * 1024);
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  int vrow, vcol;
  halo_row_to_hw(r, vrow, vcol);
  int aoff[8];   // A-fragment byte offsets of the eight M tiles (d slice i >> 1, rows 4 (i & 1) ..) for the current (td, th)

  // prologue: first halo half chunk, then the first PF - 1 weight fragments
  issue_halo(hc_b);
#pragma unroll
  for (int q = 0; q < PF - 1; ++q) {
    const int tq = q % 27, hq = hc_b + q / 27;
    issue_b(hq < HC ? hq : HC - 1, tq, q % PF);
  }
  for (int hc = hc_b; hc < HC; ++hc) {
    // halo(hc): issued a whole half chunk ago (or in the prologue), older than the PF - 1 weight fragments in flight
    wait_vm_then_barrier<PF - 1>();
    // the buffer halo(hc + 1) goes to was read during hc - 1: every wave is past those reads (barrier).  Past the end: refetch
    // the current chunk's image into the idle buffer (keeps the counts static, nobody reads it)
    issue_halo(hc + 1 < HC ? hc + 1 : hc);
    const unsigned char* hb = smem + (hc & 1) * HBUF;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
      // in flight behind the fragment of this tap: PF - 2 younger fragments, and - until the fragments issued before this half
      // chunk's gather are used up - the HPW gather pieces
      if (tap < PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF - 2 + HPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF - 2) : "memory");
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ring + (tap % PF) * 1024 + lane * 16);
      // the eight fragment offsets of a (td, th) row of taps are computed where they are used (opaque lane terms: hipcc otherwise
      // hoists all 72 of a half chunk out of the loop and spills them)
      if (tw == 0) {
        int vr = vrow, vc = vcol, hh_ = h;
        asm volatile("" : "+v"(vr), "+v"(vc), "+v"(hh_));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int R = (i >> 1) * HALO_H + 4 * (i & 1) + vr + td * HALO_H + th;
          aoff[i] = ((R * HALO_W + vc) * 2 + (hh_ ^ ((R >> 1) & 1))) * 16;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(hb + aoff[i] + tw * 32);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
      }
      // refill the slot that the fragment PF - 1 taps ahead will use: the one read one tap ago (its ds_read has returned: the
      // MFMAs of that tap consumed it)
      {
        const int q = tap + PF - 1;
        const int tq = q % 27, hq = hc + q / 27;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_b(hq < HC ? hq : HC - 1, tq, q % PF);
      }
      __builtin_amdgcn_sched_barrier(0);   // (one tap's fragments live at a time: 36 registers, not 100)
    }
  }
  wait_vm_then_barrier<0>();  // tail refetches have landed; LDS is free for the epilogue

  const int n0 = nt * 32;
  if (p.in_acc && !p.part) {
    // InstanceNorm statistics from the fp32 accumulators: lane (r, h) holds rows of column n0 + r; one fp64 atomic pair per
    // channel and brick, no LDS (the wave owns its 32 channels for the whole brick)
    const bool full = d0 + HB_D <= p.D && h0 + HB_H <= p.H && w0 + HB_W <= p.W;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][e];
        if (!full) {
          int hh, ww;
          halo_row_to_hw((e & 3) + 8 * (e >> 2) + 4 * h, hh, ww);
          if (d0 + (i >> 1) >= p.D || h0 + 4 * (i & 1) + hh >= p.H || w0 + ww >= p.W) v = 0.f;
        }
        s1 += v;
        s2 += v * v;
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (h == 0 && n0 + r < p.N) {
      atomicAdd(&p.in_acc[((size_t)b * p.N + n0 + r) * 2], (double)s1);
      atomicAdd(&p.in_acc[((size_t)b * p.N + n0 + r) * 2 + 1], (double)s2);
    }
  }

  float* stage = reinterpret_cast<float*>(smem) + wave * 32 * STAGE_LD;
  bf16* out = reinterpret_cast<bf16*>(p.out);
  bf16* out2 = reinterpret_cast<bf16*>(p.out2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int gd = d0 + (i >> 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) stage[((e & 3) + 8 * (e >> 2) + 4 * h) * STAGE_LD + r] = acc[i][e];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int vv = lane + 64 * q;  // 128 vectors: 32 rows x 4 groups of 8 channels
      const int row = vv >> 2, cv = vv & 3;
      int hh, ww;
      halo_row_to_hw(row, hh, ww);
      const int gh = h0 + 4 * (i & 1) + hh, gw = w0 + ww;
      const int n = n0 + cv * 8;
      if (gd < p.D && gh < p.H && gw < p.W && n < p.N) {
        float xv[8];
        load8(&stage[row * STAGE_LD + cv * 8], xv);
        const size_t m = (((size_t)b * p.D + gd) * p.H + gh) * p.W + gw;
        if (p.part) store8(p.part + ((size_t)blockIdx.z * p.B * p.D * p.H * p.W + m) * (p.ntn * 32) + n, xv);
        else if (p.n_split > 0 && n >= p.n_split) {
          if (p.residual2) {
            float rr[8];
            load8(reinterpret_cast<const bf16*>(p.residual2) + m * p.ldc2 + (n - p.n_split), rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] += rr[e];
          }
          store8(out2 + m * p.ldc2 + (n - p.n_split), xv);
        } else {
          if (p.residual) {
            float rr[8];
            load8(reinterpret_cast<const bf16*>(p.residual) + m * p.ldc + n, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] += rr[e];
          }
          if (!(p.debug & 1)) store8(out + m * p.ldc + n, xv);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Second pass of a channel-split convolution: out[m][n] = bf16(sum_s part[s][m][n]) and, optionally, the InstanceNorm
// sums of the result.  grid (row chunks, B); block = (N / 8 column groups) x (256 / (N / 8) row lanes).
__global__ __launch_bounds__(256) void halo_split_finish_kernel(const float* __restrict__ part, bf16* __restrict__ out,
                                                                double* __restrict__ in_acc, const int nsplit,
                                                                const int64_t S, const int64_t rows_total, const int N,
                                                                const int npad, const int ldc, const int64_t rows_per_block) {
  __shared__ float red[256 * 16];
  const int ncg = N >> 3;
  const int tid = threadIdx.x;
  const int cg = tid % ncg, rl = tid / ncg;
  const int rlanes = 256 / ncg;
  const int b = blockIdx.y;
  const int64_t s_begin = (int64_t)blockIdx.x * rows_per_block;
  const int64_t s_end = min(S, s_begin + rows_per_block);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (rl < rlanes) {
    for (int64_t sidx = s_begin + rl; sidx < s_end; sidx += rlanes) {
      const int64_t m = (int64_t)b * S + sidx;
      float v[8];
      load8(part + m * npad + cg * 8, v);
      for (int sp = 1; sp < nsplit; ++sp) {
        float w[8];
        load8(part + ((int64_t)sp * rows_total + m) * npad + cg * 8, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += w[e];
      }
      store8(out + m * ldc + cg * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
    }
  }
  if (!in_acc) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  for (int o = tid; o < N * 2; o += 256) {
    const int c = o >> 1, which = o & 1;
    const int g = c >> 3, e = c & 7;
    double a = 0.0;
    for (int q = 0; q < rlanes; ++q) a += (double)red[(q * ncg + g) * 16 + which * 8 + e];
    atomicAdd(&in_acc[((size_t)b * N + c) * 2 + which], a);
  }
}

template <typename T> struct HaloDma {
  static bool launch(const HaloArgs&, float*, int64_t, hipStream_t) { return false; }
};
template <> struct HaloDma<bf16> {
  static bool launch(const HaloArgs& p, float* ws, int64_t ws_floats, hipStream_t s) {
    // 32-bit voxel indices and element offsets inside the kernel
    const int64_t vox = (int64_t)p.B * p.D * p.H * p.W;
    const int cmax = p.C1 > p.C2 ? p.C1 : p.C2;
    if (vox * cmax >= (1ll << 31)) return false;
    const int ntn = p.ntn;
    const int NT = ntn % 4 == 0 ? 4 : (ntn % 2 == 0 ? 2 : 1);
    // batch-pair bricks (see the kernel): fewer bricks when the height leaves an 8-row brick at most half full (12 rows: 3 bricks
    // of 4 for two items against 2 x 2 of 8); four n tiles only (the shapes that have it: 128 .. 512 channels at 12 x 12 x 24)
    const int nbh4 = (p.H + 3) / 4;
    const bool bp = NT == 4 && p.B % 2 == 0 && (p.B / 2) * nbh4 < p.B * p.nbh && !(ctu_option_route() & CTU_ROUTE_HALO_NO_BATCH_PAIR) &&
                    !(ctu_option_nt_debug() & 16);   // (the diagnostic STAMP build has no batch-pair variant)
    const int bricks = bp ? (p.B / 2) * p.nbd * nbh4 * p.nbw : p.B * p.nbd * p.nbh * p.nbw;
    // Small volumes (the 12x12x24 and 6x6x12 stages: 27 and 4 bricks) leave most CUs idle: split the input channels
    // over workgroups, keep fp32 partial outputs in the workspace and sum them in a second pass (no atomics).
    const int HCT = (p.C1 + p.C2) / 16;
    const int blocks = bricks * (ntn / NT);
    int ksplit = 1;
    const int64_t rows = (int64_t)p.B * p.D * p.H * p.W;
    if (ws && !p.residual && !p.residual2 && blocks < 256 && rows <= 16384 && HCT >= 4 && p.n_split == 0 && p.N % 8 == 0 && p.N <= 2048 && p.ldc % 8 == 0) {
      // fill the 512 resident slots (2 workgroups per CU) ONCE: rounding up (540 workgroups for 512 -> 512 @ 12 x 12 x 24)
      // costs a second round for a handful of stragglers; down to one half chunk (9 stages) per workgroup if need be
      // (256 -> 256 @ 6 x 6 x 12: 72.6 -> 28.1 us)
      const int slots = (ctu_option_route() & CTU_ROUTE_HALO_KSPLIT_OLD) ? 0 : 512;
      ksplit = slots ? slots / blocks : (512 + blocks - 1) / blocks;
      const int kmax = slots ? HCT : HCT / 2;
      if (ksplit > kmax) ksplit = kmax;
      const int64_t cap = ws_floats / (rows * ntn * 32);
      if (ksplit > cap) ksplit = (int)cap;
      if (ksplit < 2) ksplit = 1;
    }
    HaloArgs q = p;
    if (bp) q.nbh = nbh4;
    q.debug = ctu_option_nt_debug();
    q.hc_per_split = (HCT + ksplit - 1) / ksplit;
    ksplit = (HCT + q.hc_per_split - 1) / q.hc_per_split;
    q.part = ksplit > 1 ? ws : nullptr;
    q.part_stamps = nullptr;
    const dim3 grid(bricks, ntn / NT, ksplit);
    if ((q.debug & 16) && ksplit == 1 && ws && ws_floats >= (int64_t)grid.x * grid.y * 32 && NT >= 2) {   // diagnostic build
      q.part_stamps = ws;
      if (NT == 4) hipLaunchKernelGGL((conv3_halo_dma_kernel<4, 3, 3, true, 4>), grid, dim3(256), 0, s, q);
      else hipLaunchKernelGGL((conv3_halo_dma_kernel<2, 2, 3, true, 4>), grid, dim3(256), 0, s, q);
      return true;
    }
    const bool long_stages = (ctu_option_route() & CTU_ROUTE_HALO_TPS9) != 0;
    const bool alone = (ctu_option_route() & CTU_ROUTE_HALO_GATHER_WAVE0) != 0;  // previous rule: wave 0 gathers the halo alone
    // CTU_ROUTE_HALO_THIN: a few KiB of unused dynamic LDS push the third (NT <= 2) / second (NT = 4) resident workgroup
    // off the CU - its registers, wave slots and LDS are then free for kernels of other streams
    const size_t thin = (ctu_option_route() & CTU_ROUTE_HALO_THIN) ? (NT == 4 ? 12 * 1024 : 4 * 1024) : 0;
    if (NT == 4 && !bp && (ctu_option_route() & CTU_ROUTE_HALO_NSPLIT)) hipLaunchKernelGGL(conv3_halo_ns_kernel<9>, grid, dim3(256), 0, s, q);
    else if (bp) hipLaunchKernelGGL((conv3_halo_dma_kernel<4, 2, 3, false, 4, true>), grid, dim3(256), 0, s, q);   // (two-stage ring: 70 KiB, two workgroups per CU)
    else if (NT == 4 && alone) hipLaunchKernelGGL((conv3_halo_dma_kernel<4, 3, 3, false, 0>), grid, dim3(256), thin, s, q);
    else if (NT == 4) hipLaunchKernelGGL(conv3_halo_dma_kernel<4>, grid, dim3(256), thin, s, q);
    else if (NT == 2 && long_stages) hipLaunchKernelGGL((conv3_halo_dma_kernel<2, 2, 9, false, 0>), grid, dim3(256), thin, s, q);
    else if (NT == 2 && alone) hipLaunchKernelGGL((conv3_halo_dma_kernel<2, 2, 3, false, 0>), grid, dim3(256), thin, s, q);
    else if (NT == 2) hipLaunchKernelGGL(conv3_halo_dma_kernel<2>, grid, dim3(256), thin, s, q);
    else if (long_stages) hipLaunchKernelGGL((conv3_halo_dma_kernel<1, 2, 9, false, 0>), grid, dim3(256), thin, s, q);
    else if (alone) hipLaunchKernelGGL((conv3_halo_dma_kernel<1, 2, 3, false, 0>), grid, dim3(256), thin, s, q);
    else hipLaunchKernelGGL(conv3_halo_dma_kernel<1>, grid, dim3(256), thin, s, q);
    if (ksplit > 1) {
      const int64_t S = (int64_t)p.D * p.H * p.W;
      int64_t chunks = 1024 / p.B;
      if (chunks < 1) chunks = 1;
      int64_t rpb = (S + chunks - 1) / chunks;
      if (rpb < 16) rpb = 16;
      hipLaunchKernelGGL(halo_split_finish_kernel, dim3((unsigned)((S + rpb - 1) / rpb), p.B), dim3(256), 0, s, ws,
                         reinterpret_cast<bf16*>(p.out), p.in_acc, ksplit, S, rows, p.N, ntn * 32, p.ldc, rpb);
    }
    return true;
  }
};

template <typename T>
static int launch_halo(const HaloArgs& p, float* ws, int64_t ws_floats, hipStream_t s) {
  if (HaloDma<T>::launch(p, ws, ws_floats, s)) return ctu_check_launch("conv3_halo");
  const int ntn = p.ntn;
  const int bricks = p.B * p.nbd * p.nbh * p.nbw;
  if (ntn >= 4 && ntn % 4 == 0 && sizeof(T) == 2)  // (fp32 mode keeps float64 accumulators: two n tiles per wave at most)
    hipLaunchKernelGGL((conv3_halo_kernel<T, 4>), dim3(bricks, ntn / 4), dim3(256), 0, s, p);
  else if (ntn % 2 == 0)
    hipLaunchKernelGGL((conv3_halo_kernel<T, 2>), dim3(bricks, ntn / 2), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv3_halo_kernel<T, 1>), dim3(bricks, ntn), dim3(256), 0, s, p);
  return ctu_check_launch("conv3_halo");
}

extern "C" int ctu_conv3_halo(ctu_dtype dtype, const void* x1, const void* x2, const void* wfrag, void* out, void* out2,
                              int32_t B, int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N,
                              int32_t n_split, int32_t ldc, int32_t ldc2, double* in_acc, const void* residual,
                              const void* residual2, float* ws, int64_t ws_floats, int32_t x1_layout, ctu_stream_t stream) {
  CTU_REQUIRE(x1 && wfrag && out, "conv3_halo: null pointer");
  CTU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3_halo: bad dims");
  CTU_REQUIRE(C1 > 0 && C1 % 32 == 0 && C2 >= 0 && C2 % 32 == 0 && (C2 == 0 || x2), "conv3_halo: C1, C2 must be multiples of 32");
  CTU_REQUIRE(N > 0 && N % 8 == 0 && ldc % 8 == 0 && ldc > 0, "conv3_halo: N, ldc must be multiples of 8");
  CTU_REQUIRE(n_split % 32 == 0 && (n_split == 0 || (out2 && ldc2 > 0 && ldc2 % 8 == 0)), "conv3_halo: bad split");
  HaloArgs p;
  p.x1 = x1; p.x2 = x2; p.wfrag = wfrag; p.out = out; p.out2 = out2;
  p.B = B; p.D = D; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.N = N; p.n_split = n_split; p.ldc = ldc; p.ldc2 = ldc2;
  p.in_acc = in_acc;
  p.residual = residual;
  p.residual2 = residual2;
  CTU_REQUIRE((!residual && !residual2) || (dtype == CTU_BF16 && !in_acc &&
                                            (int64_t)B * D * H * W * (C1 > C2 ? C1 : C2) < (1ll << 31)),
              "conv3_halo: a residual input needs the bf16 LDS-DMA kernel without statistics");
  CTU_REQUIRE(!residual2 || (n_split > 0 && out2), "conv3_halo: residual2 belongs to the out2 part of a split output");
  p.part = nullptr;
  p.hc_per_split = (C1 + C2) / 16;
  CTU_REQUIRE(x1_layout == CTU_LAYOUT_NDHWC || (x1_layout == CTU_LAYOUT_B16 && dtype == CTU_BF16 &&
                                               (int64_t)B * D * H * W * (C1 > C2 ? C1 : C2) < (1ll << 31)),
              "conv3_halo: the blocked input layout needs the bf16 LDS-DMA kernel");
  p.vs1 = x1_layout == CTU_LAYOUT_B16 ? 16 : C1;
  p.bs1 = x1_layout == CTU_LAYOUT_B16 ? (int64_t)16 * B * D * H * W : 16;
  CTU_REQUIRE(ws_floats >= 0 && (ws_floats == 0 || ws), "conv3_halo: bad workspace");
  CTU_REQUIRE(!in_acc || (dtype == CTU_BF16 && n_split == 0 &&
                          (int64_t)B * D * H * W * (C1 > C2 ? C1 : C2) < (1ll << 31)),
              "conv3_halo: fused InstanceNorm statistics need the bf16 LDS-DMA kernel (no split, < 2^31 elements)");
  p.nbd = (D + HB_D - 1) / HB_D; p.nbh = (H + HB_H - 1) / HB_H; p.nbw = (W + HB_W - 1) / HB_W;
  p.ntn = (N + 31) / 32;
  CTU_REQUIRE((int64_t)B * p.nbd * p.nbh * p.nbw < (1ll << 31), "conv3_halo: too many bricks");
  CTU_DISPATCH(dtype, return launch_halo<float>(p, ws, ws_floats, (hipStream_t)stream),
               return launch_halo<bf16>(p, ws, ws_floats, (hipStream_t)stream));
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution:  dw[tap][n][c] += sum_v dY[v][n] * X[v + tap - 1][c].
// A workgroup owns one (32 output channels) x (32 input channels) tile for ALL 27 taps and walks a range of bricks
// with the accumulators resident: wave w holds taps w, w+4, ... (7 or 6 tiles of 32x32).  Per brick the X halo chunk
// and the dY tile are staged in LDS once; the K dimension of the MFMA is the voxel index, so both operands are
// read "transposed" (8 voxels of one channel per lane).  Finally each wave adds its tiles into the fp32 panel.
// ---------------------------------------------------------------------------------------------------------
struct HaloWgArgs {
  const void* dy;
  const void* x1;
  const void* x2;
  float* dw;  // [27][N][K]
  int B, D, H, W, C1, C2, N;
  int nbd, nbh, nbw, nbricks, bricks_per_block, tiles_c, tiles_n;
  int xvs, yvs;      // voxel strides of x1 and dy (bf16 DMA kernel), see HaloArgs::vs1
  int64_t xbs, ybs;  // 16-channel-block strides
  float* part;  // [splits][27][N][K] partial panels (workspace) or null: atomics straight into dw
  void* stamps; // STAMP build only
  int debug;  // measurement hook (ctu_set_option "nt_debug"): 4 = no operand DMA, 1 = no epilogue
};

template <typename T>
__global__ __launch_bounds__(256) void conv3_halo_wgrad_kernel(const HaloWgArgs p) {
  constexpr int EV = 16 / sizeof(T);
  constexpr int CK = 32;
  constexpr int LDT = CK + EV;   // halo voxel row (elements)
  constexpr int LDY = 32 + EV;   // dY tile row
  constexpr int VPV = CK / EV;
  constexpr int BRICK = HB_D * HB_H * HB_W;  // 256
  __shared__ __attribute__((aligned(16))) T halo[HALO_VOX * LDT];
  __shared__ __attribute__((aligned(16))) T dyt[BRICK * LDY];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int n0 = (blockIdx.x / p.tiles_c) * 32;
  const int c0 = (blockIdx.x % p.tiles_c) * CK;
  const int K = p.C1 + p.C2;
  const T* src = (c0 < p.C1) ? reinterpret_cast<const T*>(p.x1) : reinterpret_cast<const T*>(p.x2);
  const int cs = (c0 < p.C1) ? p.C1 : p.C2;
  const int cc = (c0 < p.C1) ? c0 : c0 - p.C1;
  const T* dy = reinterpret_cast<const T*>(p.dy);

  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // fp32 parity mode: per-brick f32 chains (256 voxels) summed in float64 (see conv3_halo_kernel)
  constexpr bool WIDE = sizeof(T) == 4;
  double acc64[WIDE ? 7 : 1][WIDE ? 16 : 1];
  if (WIDE) {
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc64[i][e] = 0.0;
  }

  const int brick_begin = blockIdx.y * p.bricks_per_block;
  const int brick_end = min(p.nbricks, brick_begin + p.bricks_per_block);
  for (int brick = brick_begin; brick < brick_end; ++brick) {
    int t = brick;
    const int bw = t % p.nbw; t /= p.nbw;
    const int bh = t % p.nbh; t /= p.nbh;
    const int bd = t % p.nbd;
    const int b = t / p.nbd;
    const int d0 = bd * HB_D, h0 = bh * HB_H, w0 = bw * HB_W;
    __syncthreads();
    for (int v = tid; v < HALO_VOX * VPV; v += 256) {
      const int vox = v / VPV, part = v - vox * VPV;
      const int hw = vox % HALO_W;
      const int tq = vox / HALO_W;
      const int hh = tq % HALO_H, hd = tq / HALO_H;
      const int gd = d0 + hd - 1, gh = h0 + hh - 1, gw = w0 + hw - 1;
      u32x4 val = {0u, 0u, 0u, 0u};
      if ((unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W)
        val = *reinterpret_cast<const u32x4*>(src + ((((size_t)b * p.D + gd) * p.H + gh) * p.W + gw) * cs + cc + part * EV);
      *reinterpret_cast<u32x4*>(&halo[vox * LDT + part * EV]) = val;
    }
    for (int v = tid; v < BRICK * VPV; v += 256) {
      const int vox = v / VPV, part = v - vox * VPV;
      const int gd = d0 + (vox >> 6), gh = h0 + ((vox >> 3) & 7), gw = w0 + (vox & 7);
      const int n = n0 + part * EV;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (gd < p.D && gh < p.H && gw < p.W && n < p.N)
        val = *reinterpret_cast<const u32x4*>(dy + ((((size_t)b * p.D + gd) * p.H + gh) * p.W + gw) * p.N + n);
      *reinterpret_cast<u32x4*>(&dyt[vox * LDY + part * EV]) = val;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 16; ++s) {
      const int d = s >> 2, hh = 2 * (s & 3) + h;  // this lane half's row of 8 voxels along w
      const typename Mma<T>::Frag fa = Mma<T>::gather(&dyt[(d * 64 + hh * 8) * LDY + r], LDY);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int tap = wave + 4 * i;
        if (tap < 27) {  // wave-uniform
          const int tw = tap % 3, tq = tap / 3;
          const int th = tq % 3, td = tq / 3;
          const typename Mma<T>::Frag fb =
              Mma<T>::gather(&halo[(((d + td) * HALO_H + hh + th) * HALO_W + tw) * LDT + r], LDT);
          Mma<T>::mma(fa, fb, acc[i]);
        }
      }
    }
    if (WIDE) {
#pragma unroll
      for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc64[i][e] += (double)acc[i][e];
          acc[i][e] = 0.f;
        }
    }
  }
  if (WIDE) {
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = (float)acc64[i][e];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wave + 4 * i;
    if (tap < 27) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int c = c0 + r;
        if (n < p.N && c < K) atomicAdd(&p.dw[((size_t)tap * p.N + n) * K + c], acc[i][e]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 production variant of the weight gradient: LDS-DMA staging, double buffered over bricks.
// One workgroup of 8 waves per CU owns (32 NTN output channels) x (32 input channels) for all 27 taps; wave w holds
// n tile w & 1 and taps (w >> 1) + 4 i (NTN = 2), or taps w + 8 i (NTN = 1).  While the MFMAs of brick k run, the halo
// chunk and dY tile of brick k+1 are in flight into the other buffer: one barrier per brick, no register staging.
// Both LDS images are plain [voxel][32 channels] (64-B rows, lane-linear as LDS-DMA requires): a transposed
// ds_read_b64_tr_b16 touches 4 consecutive voxels x 64 B = one whole 256-B bank row, conflict free without padding.
// Workgroups that walk the same brick range (all tiles of one split) are placed on one XCD so that the halo / dY
// bytes they share are served by that XCD's L2.
// ---------------------------------------------------------------------------------------------------------
// STAMP (diagnostic build, ctu_set_option("nt_debug", 16); never launched by the product): every wave sums the clock ticks it
// spends (0) until its first operands have landed, (1) waiting for its own DMA at later brick barriers, (2) at the barriers
// themselves, (3) in the k loops, (4) in the epilogue, and writes them to stamps[workgroup][wave][8].
template <int NTN, bool BURST = false, bool STAMP = false>
__global__ __launch_bounds__(512, 1) void conv3_halo_wgrad_dma_kernel(const HaloWgArgs p) {
  unsigned long long st_v[5] = {0, 0, 0, 0, 0}, st_mark = 0, st_c0 = 0, st_r0 = 0;
  if (STAMP) {
    st_c0 = st_mark = __builtin_amdgcn_s_memtime();
    st_r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz: (memtime ticks) / (memrealtime ticks) x 0.1 = shader clock in GHz
  }
  constexpr int HINS = 38;              // DMA wave-instructions per halo chunk: 600 voxels x 4 slots = 2400 <= 2432
  constexpr int HBYTES = HINS * 1024;
  constexpr int YBYTES = 256 * 64;      // dY tile of one 32-wide n tile
  constexpr int BUF = HBYTES + NTN * YBYTES;
  constexpr int ROWS = NTN == 2 ? 2 : 1;  // whole (td, th) rows of three taps per wave
  constexpr int NI = 3 * ROWS + 1;        // accumulator tiles per wave: the rows' taps + one tap of row (2, 2)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = NTN == 2 ? (wave & 1) : 0;
  const int tg = NTN == 2 ? (wave >> 1) : wave;  // tap group: rows ROWS tg .. and, for tg < 3, tap (2, 2, tg)
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.tiles_n * p.tiles_c;
  const int tile = vid % tiles, split = vid / tiles;
  const int n0 = (tile / p.tiles_c) * 32 * NTN;
  const int c0 = (tile % p.tiles_c) * 32;
  const int K = p.C1 + p.C2;
  const bool first = c0 < p.C1;
  const bf16* src = first ? reinterpret_cast<const bf16*>(p.x1) : reinterpret_cast<const bf16*>(p.x2);
  const int cs = first ? p.xvs : p.C2;                       // voxel stride of the halo operand
  const int64_t cbs = first ? p.xbs : 16;                    // its 16-channel-block stride
  const int cc = first ? c0 : c0 - p.C1;
  const bf16* dy = reinterpret_cast<const bf16*>(p.dy);
  const bf16* zero = reinterpret_cast<const bf16*>(g_zero16);

  // brick-independent part of this lane's DMA sources: halo voxel (hd, hh, hw) and 8-channel part per instruction
  int hv[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int S = (wave + 8 * k) * 64 + lane;
    const int vox = S >> 2;
    const int R = vox / HALO_W, hw = vox - R * HALO_W;
    const int hd = R / HALO_H, hh = R - hd * HALO_H;
    hv[k] = S < 4 * HALO_VOX ? (hd | (hh << 4) | (hw << 8) | ((S & 3) << 12)) : -1;
  }

  // operand DMA of one brick = NP pieces per wave (5 halo + 2 NTN dY instructions); piece q is a compile-time index
  constexpr int NP = 5 + 2 * NTN;
  struct BrickAt { int b, d0, h0, w0; };
  auto locate = [&](int brick) {
    BrickAt a;
    int t = brick;
    const int bw = t % p.nbw; t /= p.nbw;
    const int bh = t % p.nbh; t /= p.nbh;
    const int bd = t % p.nbd;
    a.b = t / p.nbd;
    a.d0 = bd * HB_D; a.h0 = bh * HB_H; a.w0 = bw * HB_W;
    return a;
  };
  auto issue_piece = [&](const BrickAt& a, int buf, int q) {
    if (p.debug & 4) return;
    unsigned char* dst = smem + buf * BUF;
    if (q < 5) {
      const int k = q;
      const int i = wave + 8 * k;
      if (i < HINS) {
        const int gd = a.d0 + (hv[k] & 15) - 1, gh = a.h0 + ((hv[k] >> 4) & 15) - 1, gw = a.w0 + ((hv[k] >> 8) & 15) - 1;
        const bool ok = hv[k] >= 0 && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H &&
                        (unsigned)gw < (unsigned)p.W;
        const int m = ((a.b * p.D + gd) * p.H + gh) * p.W + gw;
        const int ch = cc + ((hv[k] >> 12) & 3) * 8;  // first of this lane's 8 channels
        dma16(ok ? src + (size_t)m * cs + (size_t)(ch >> 4) * cbs + (ch & 15) : zero, dst + i * 1024);
      }
    } else {
      const int k = q - 5;
      const int j = wave + 8 * k;  // < 16 NTN
      const int S = (j & 15) * 64 + lane;
      const int vox = S >> 2;
      const int gd = a.d0 + (vox >> 6), gh = a.h0 + ((vox >> 3) & 7), gw = a.w0 + (vox & 7);
      const int n = n0 + (j >> 4) * 32 + (S & 3) * 8;
      const bool ok = gd < p.D && gh < p.H && gw < p.W && n < p.N;
      const int m = ((a.b * p.D + gd) * p.H + gh) * p.W + gw;
      dma16(ok ? dy + (size_t)m * p.yvs + (size_t)(n >> 4) * p.ybs + (n & 15) : zero, dst + HBYTES + j * 1024);
    }
  };

  f32x16 acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // Taps: a wave owns whole (td, th) rows.  The three taps of a row read the same ten halo voxels along w, shifted by one: they
  // are fetched ONCE (three transposed reads: voxels 0-3, 4-7, 8-11 of this lane's channel) and the fragments of tw = 0, 1, 2 are
  // cut from those registers (tw = 1 with four v_alignbit) - 10 LDS reads per k step and wave instead of 16.  With one read pair
  // per tap the k loop was bound by LDS bandwidth (8 fragments per 7 MFMAs: 146 B/clk/CU asked at full matrix rate, 128 there).
  // Row (2, 2) is dealt out tap by tap to groups 0 - 2; the group without one repeats a tap into an accumulator never written.
  int rowoff[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    const int R = ROWS * tg + j, td = R / 3, th = R - 3 * td;
    rowoff[j] = ((td * HALO_H + th) * HALO_W) * 32 + r;
  }
  const int singleoff = ((2 * HALO_H + 2) * HALO_W + (tg < 3 ? tg : 2)) * 32 + r;
  auto tap_of = [&](int i) { return i < 3 * ROWS ? (ROWS * tg + i / 3) * 3 + i % 3 : (tg < 3 ? 24 + tg : -1); };

  const int brick_begin = split * p.bricks_per_block;
  const int brick_end = min(p.nbricks, brick_begin + p.bricks_per_block);
  if (brick_begin < brick_end) {
    const BrickAt a = locate(brick_begin);
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(a, 0, q);
  }
  int buf = 0;
  for (int brick = brick_begin; brick < brick_end; ++brick, buf ^= 1) {
    // this brick's DMAs (issued one compute phase ago) have landed in every wave, and every wave is done reading
    // the other buffer
    if (STAMP) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      if (brick > brick_begin) st_v[3] += t - st_mark;
      st_mark = t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      t = __builtin_amdgcn_s_memtime();
      st_v[brick > brick_begin ? 1 : 0] += t - st_mark;
      st_mark = t;
      asm volatile("s_barrier" ::: "memory");
      t = __builtin_amdgcn_s_memtime();
      st_v[2] += t - st_mark;
      st_mark = t;
    } else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // the next brick's pieces go out one per k step, between the MFMAs (in one burst behind the barrier, all eight waves spend
    // ~10 % of the brick on address arithmetic with the matrix pipe idle)
    const bool more = brick + 1 < brick_end;
    const BrickAt an = locate(more ? brick + 1 : brick);
    const bf16* halo = reinterpret_cast<const bf16*>(smem + buf * BUF);
    const bf16* dyt = reinterpret_cast<const bf16*>(smem + buf * BUF + HBYTES + nt * YBYTES);
    // k loop, software pipelined by hand: the LDS reads of step s + 1 are issued before the MFMAs of step s (hipcc alone puts each
    // read right in front of its use - a wave then runs ~55 % of the time and the SIMD's second wave has to fill the rest)
    bf16x8 fa[2], fs[2];
    unsigned rw[2][ROWS][5];
    auto fetch = [&](int s, int slot) {
      const int d = s >> 2, hh = 2 * (s & 3) + h;  // this lane half's row of 8 voxels along w
      const int vb = ((d * HALO_H + hh) * HALO_W) * 32;
      fa[slot] = Mma<bf16>::gather(&dyt[(d * 64 + hh * 8) * 32 + r], 32);
#pragma unroll
      for (int j = 0; j < ROWS; ++j) Mma<bf16>::row3_fetch(&halo[rowoff[j] + vb], 32, rw[slot][j]);
      fs[slot] = Mma<bf16>::gather(&halo[singleoff + vb], 32);
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int cur = s & 1;
      if (s + 1 < 16) fetch(s + 1, cur ^ 1);
      if (BURST ? s == 0 : s < NP) {
        if (more) {
          if (BURST) {
#pragma unroll
            for (int q = 0; q < NP; ++q) issue_piece(an, buf ^ 1, q);
          } else {
            issue_piece(an, buf ^ 1, s);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        bf16x8 f0, f1, f2;
        Mma<bf16>::row3_frags(rw[cur][j], f0, f1, f2);
        Mma<bf16>::mma(fa[cur], f0, acc[3 * j]);
        Mma<bf16>::mma(fa[cur], f1, acc[3 * j + 1]);
        Mma<bf16>::mma(fa[cur], f2, acc[3 * j + 2]);
      }
      Mma<bf16>::mma(fa[cur], fs[cur], acc[3 * ROWS]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAMP) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    st_v[3] += t - st_mark;
    st_mark = t;
  }
  if (p.debug & 1) return;
  if (p.part) {
    // 16 one-dword stores per tile and lane are bound by the store instruction rate (6.9 K wave instructions per CU: ~60 us for
    // 64 -> 64 @ 96^3); through a per-wave LDS tile each lane stores four float4 instead, 128-B rows contiguous over 8 lanes
    __syncthreads();  // all waves are done reading the operand buffers
    constexpr int SLD = 36;
    float* st = reinterpret_cast<float*>(smem) + wave * 32 * SLD;
    float* dst = p.part + (size_t)split * 27 * p.N * K;
    const int srow = lane >> 3, scol = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int tap = tap_of(i);
      if (tap >= 0) {  // wave-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) st[((e & 3) + 8 * (e >> 2) + 4 * h) * SLD + r] = acc[i][e];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 8 * q + srow;
          const int n = n0 + nt * 32 + row, c = c0 + scol;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&st[row * SLD + scol]);
          // (streaming "nt" and write-through "sc0 sc1" stores: same kernel time, r60 / r61 in profiles/)
          if (n < p.N && c < K) *reinterpret_cast<f32x4*>(&dst[((size_t)tap * p.N + n) * K + c]) = v;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if (STAMP && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long* d = reinterpret_cast<unsigned long long*>(p.stamps) + ((size_t)blockIdx.x * 8 + wave) * 8;
      const unsigned long long tc = __builtin_amdgcn_s_memtime();
      st_v[4] = tc - st_mark;
#pragma unroll
      for (int e = 0; e < 5; ++e) d[e] = st_v[e];
      d[5] = tc - st_c0;
      const unsigned long long tr = __builtin_amdgcn_s_memrealtime();
      d[6] = tr - st_r0;
      d[7] = st_r0;  // start on the 100 MHz clock all CUs share: dispatch stagger

    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int tap = tap_of(i);
    if (tap >= 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int c = c0 + r;
        if (n < p.N && c < K) atomicAdd(&p.dw[((size_t)tap * p.N + n) * K + c], acc[i][e]);
      }
    }
  }
}

// dw[i] += sum over splits of part[s][i].  Block = 64 element lanes (VEC floats each) x 4 split lanes: split lane g sums the
// partial panels g, g + 4, ..., the four sums meet in LDS and lane g = 0 adds the result into dw - no atomics (a grid that split the
// partial panels over workgroups and combined them with atomics took 18 - 27 us for 57 MB, this takes the 11 us of the bytes).
// VEC = 1 for the small panels keeps >= 400 workgroups in flight.
template <int VEC>
__global__ __launch_bounds__(256) void halo_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 const int64_t panel, const int splits) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ float red[3][64][VEC];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * VEC;
  vec_t a = 0.f;
  if (i < panel) {
    const float* src = part + i;
#pragma unroll 8
    for (int sp = g; sp < splits; sp += 4) a += *reinterpret_cast<const vec_t*>(src + (int64_t)sp * panel);
  }
  if (g > 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[g - 1][lane][e] = a[e];
  }
  __syncthreads();
  if (g == 0 && i < panel) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] += red[0][lane][e] + red[1][lane][e] + red[2][lane][e];
    vec_t d = *reinterpret_cast<const vec_t*>(dw + i);
    d += a;
    *reinterpret_cast<vec_t*>(dw + i) = d;
  }
}

// Sum of the per-split partial panels part[split][27][NK], ADDED into the parameter layout dw[NK][27].  A workgroup owns PAIRS
// consecutive (n, k) pairs for ALL 27 taps: 256 / PAIRS split lanes per pair read rows of PAIRS x 4 contiguous bytes per (split, tap)
// and keep 27 sums in registers; one LDS exchange; then the PAIRS x 27 results leave as ONE contiguous run of read-modify-writes.
// (One workgroup per (64 pairs, tap) - the first version - updated 4-byte elements 108 bytes apart from 27 different workgroups: PMC
// 36.6 MB written and 64 MB fetched per launch for panels of 0.1 - 28 MB, 4.6 GB per step, profiles/r03_pmc_hbm_traffic*.)
template <int PAIRS>
__global__ __launch_bounds__(256) void halo_wgrad_reduce_param_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                       const int64_t NK, const int splits) {
  constexpr int SL = 256 / PAIRS;
  __shared__ float red[SL][PAIRS * 27];
  const int pl = threadIdx.x % PAIRS, g = threadIdx.x / PAIRS;
  const int64_t nk0 = (int64_t)blockIdx.x * PAIRS;
  const int64_t nk = nk0 + pl;
  const int64_t panel = 27 * NK;
  float a[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) a[t] = 0.f;
  if (nk < NK) {
    for (int sp = g; sp < splits; sp += SL) {
      const float* src = part + (int64_t)sp * panel + nk;
#pragma unroll
      for (int t = 0; t < 27; ++t) a[t] += src[(int64_t)t * NK];
    }
  }
#pragma unroll
  for (int t = 0; t < 27; ++t) red[g][pl * 27 + t] = a[t];
  __syncthreads();
  const int64_t left = NK - nk0;
  const int total = (int)(left < PAIRS ? left : PAIRS) * 27;
  for (int o = threadIdx.x; o < total; o += 256) {
    float v = 0.f;
#pragma unroll
    for (int gg = 0; gg < SL; ++gg) v += red[gg][o];
    dw[nk0 * 27 + o] += v;
  }
}

static int halo_wgrad_impl(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw, int32_t B,
                           int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N,
                           int32_t x1_layout, int32_t dy_layout, float* ws, int64_t ws_floats,
                           ctu_stream_t stream, const bool param_layout);
extern "C" int ctu_conv3_halo_wgrad(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw, int32_t B,
                                    int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N,
                                    int32_t x1_layout, int32_t dy_layout, float* ws, int64_t ws_floats,
                                    ctu_stream_t stream) {
  return halo_wgrad_impl(dtype, dy, x1, x2, dw, B, D, H, W, C1, C2, N, x1_layout, dy_layout, ws, ws_floats, stream, false);
}
extern "C" int ctu_conv3_halo_wgrad_param(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw_param,
                                          int32_t B, int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N,
                                          int32_t x1_layout, int32_t dy_layout, float* ws, int64_t ws_floats,
                                          ctu_stream_t stream) {
  return halo_wgrad_impl(dtype, dy, x1, x2, dw_param, B, D, H, W, C1, C2, N, x1_layout, dy_layout, ws, ws_floats, stream, true);
}
static int halo_wgrad_impl(ctu_dtype dtype, const void* dy, const void* x1, const void* x2, float* dw, int32_t B,
                           int32_t D, int32_t H, int32_t W, int32_t C1, int32_t C2, int32_t N,
                           int32_t x1_layout, int32_t dy_layout, float* ws, int64_t ws_floats,
                           ctu_stream_t stream, const bool param_layout) {
  CTU_REQUIRE(dy && x1 && dw, "conv3_halo_wgrad: null pointer");
  CTU_REQUIRE(ws_floats >= 0 && (ws_floats == 0 || ws), "conv3_halo_wgrad: bad workspace");
  CTU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "conv3_halo_wgrad: bad dims");
  CTU_REQUIRE(C1 > 0 && C1 % 32 == 0 && C2 >= 0 && C2 % 32 == 0 && (C2 == 0 || x2), "conv3_halo_wgrad: C1, C2 %% 32");
  CTU_REQUIRE(N > 0 && N % 8 == 0, "conv3_halo_wgrad: N %% 8");
  HaloWgArgs p;
  p.debug = ctu_option_nt_debug();
  p.part = nullptr;
  p.stamps = nullptr;
  p.dy = dy; p.x1 = x1; p.x2 = x2; p.dw = dw;
  p.B = B; p.D = D; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.N = N;
  p.nbd = (D + HB_D - 1) / HB_D; p.nbh = (H + HB_H - 1) / HB_H; p.nbw = (W + HB_W - 1) / HB_W;
  const int64_t nbricks = (int64_t)B * p.nbd * p.nbh * p.nbw;
  CTU_REQUIRE(nbricks < (1ll << 31), "conv3_halo_wgrad: too many bricks");
  p.nbricks = (int)nbricks;
  p.tiles_c = (C1 + C2) / 32;
  hipStream_t s = (hipStream_t)stream;
  const int cmax = (C1 > C2 ? C1 : C2) > N ? (C1 > C2 ? C1 : C2) : N;
  const bool dma_ok = dtype == CTU_BF16 && nbricks * 256 * cmax < (1ll << 31);
  CTU_REQUIRE((x1_layout == CTU_LAYOUT_NDHWC && dy_layout == CTU_LAYOUT_NDHWC) ||
                  (dma_ok && x1_layout <= CTU_LAYOUT_B16 && dy_layout <= CTU_LAYOUT_B16 && x1_layout >= 0 && dy_layout >= 0),
              "conv3_halo_wgrad: blocked operand layouts need the bf16 LDS-DMA kernel");
  const int64_t vox = (int64_t)B * D * H * W;
  p.xvs = x1_layout == CTU_LAYOUT_B16 ? 16 : C1;
  p.xbs = x1_layout == CTU_LAYOUT_B16 ? 16 * vox : 16;
  p.yvs = dy_layout == CTU_LAYOUT_B16 ? 16 : N;
  p.ybs = dy_layout == CTU_LAYOUT_B16 ? 16 * vox : 16;
  if (dma_ok) {
    // LDS-DMA kernel: one resident workgroup per CU (256 CUs), 32-bit element offsets
    const int ntn = N > 32 ? 2 : 1;
    p.tiles_n = (N + 32 * ntn - 1) / (32 * ntn);
    const int tiles = p.tiles_n * p.tiles_c;
    int splits = (256 + tiles - 1) / tiles;
    if (splits > p.nbricks) splits = p.nbricks;
    const int64_t panel = (int64_t)27 * N * (C1 + C2);
    // the parameter-layout form has no atomics fallback: fewer, longer brick ranges when the partial panels of the preferred
    // split count do not fit the workspace (tile counts that are not powers of two: N = 512, K = 768 -> 192 tiles x 2 splits)
    if (param_layout && ws && (int64_t)splits * panel > ws_floats && panel <= ws_floats) splits = (int)(ws_floats / panel);
    p.bricks_per_block = (p.nbricks + splits - 1) / splits;
    splits = (p.nbricks + p.bricks_per_block - 1) / p.bricks_per_block;
    CTU_REQUIRE((int64_t)tiles * splits < (1ll << 31), "conv3_halo_wgrad: too many workgroups");
    const bool partials = param_layout ||
                          (splits > 1 && ws && (int64_t)splits * panel <= ws_floats && (C1 + C2) % 4 == 0 &&
                           !(ctu_option_route() & CTU_ROUTE_HALO_WGRAD_ATOMICS));
    CTU_REQUIRE(!param_layout || (ws && (int64_t)splits * panel <= ws_floats),
                "conv3_halo_wgrad_param: workspace of %lld floats needed", (long long)splits * panel);
    if (partials) p.part = ws;
    if ((p.debug & 16) && partials && ntn == 2 && ws_floats >= (int64_t)splits * panel + (int64_t)tiles * splits * 8 * 8 * 2) {  // diagnostic build
      p.stamps = ws + (int64_t)splits * panel;
      hipLaunchKernelGGL((conv3_halo_wgrad_dma_kernel<2, false, true>), dim3(tiles * splits), dim3(512), 0, s, p);
      return ctu_check_launch("conv3_halo_wgrad");
    }
    const bool burst = (ctu_option_route() & CTU_ROUTE_HALO_WGRAD_BURST) != 0;
    if (ntn == 2 && burst) hipLaunchKernelGGL((conv3_halo_wgrad_dma_kernel<2, true>), dim3(tiles * splits), dim3(512), 0, s, p);
    else if (ntn == 2) hipLaunchKernelGGL(conv3_halo_wgrad_dma_kernel<2>, dim3(tiles * splits), dim3(512), 0, s, p);
    else if (burst) hipLaunchKernelGGL((conv3_halo_wgrad_dma_kernel<1, true>), dim3(tiles * splits), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv3_halo_wgrad_dma_kernel<1>, dim3(tiles * splits), dim3(512), 0, s, p);
    if (partials && param_layout) {
      const int64_t NK = (int64_t)N * (C1 + C2);
      if (NK >= 32 * 512) hipLaunchKernelGGL(halo_wgrad_reduce_param_kernel<32>, dim3((unsigned)((NK + 31) / 32)), dim3(256), 0, s, ws, dw, NK, splits);
      else hipLaunchKernelGGL(halo_wgrad_reduce_param_kernel<16>, dim3((unsigned)((NK + 15) / 16)), dim3(256), 0, s, ws, dw, NK, splits);
    } else if (partials) {
      if (panel >= 4 * 64 * 1024)
        hipLaunchKernelGGL(halo_wgrad_reduce_kernel<4>, dim3((unsigned)((panel / 4 + 63) / 64)), dim3(256), 0, s, ws, dw, panel, splits);
      else
        hipLaunchKernelGGL(halo_wgrad_reduce_kernel<1>, dim3((unsigned)((panel + 63) / 64)), dim3(256), 0, s, ws, dw, panel, splits);
    }
    return ctu_check_launch("conv3_halo_wgrad");
  }
  CTU_REQUIRE(!param_layout, "conv3_halo_wgrad_param: bf16 LDS-DMA kernel only");
  p.tiles_n = (N + 31) / 32;
  const int tiles = p.tiles_n * p.tiles_c;
  int splits = (512 + tiles - 1) / tiles;  // ~2 resident workgroups per CU
  if (splits > p.nbricks) splits = p.nbricks;
  if (splits < 1) splits = 1;
  p.bricks_per_block = (p.nbricks + splits - 1) / splits;
  splits = (p.nbricks + p.bricks_per_block - 1) / p.bricks_per_block;
  CTU_REQUIRE(splits <= 65535, "conv3_halo_wgrad: too many splits");
  dim3 grid(tiles, splits);
  CTU_DISPATCH(dtype, hipLaunchKernelGGL(conv3_halo_wgrad_kernel<float>, grid, dim3(256), 0, s, p),
               hipLaunchKernelGGL(conv3_halo_wgrad_kernel<bf16>, grid, dim3(256), 0, s, p));
  return ctu_check_launch("conv3_halo_wgrad");
}

// ---------------------------------------------------------------------------------------------------------
// Fragment-order weight packing.  dst[((chunk*taps + tap)*2 + kk)*ntn + nt][lane][j] = W(n, c, tap_src) with
// n = nt*32 + (lane & 31), c = chunk*32 + kk*16 + 8*(lane >> 5) + j, tap_src = flip ? taps-1-tap : tap,
// W(n, c, t) = src[n*sn + c*sc + t*st]; zero for n >= N or c >= K.
// ---------------------------------------------------------------------------------------------------------
template <typename TD>
__global__ __launch_bounds__(256) void pack_frag_kernel(const float* __restrict__ src, TD* __restrict__ dst, const int N,
                                                        const int K, const int taps, const int64_t sn, const int64_t sc,
                                                        const int64_t st, const int flip, const int ntn,
                                                        const int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    int64_t q = i >> 9;
    const int nt = (int)(q % ntn); q /= ntn;
    const int kk = (int)(q & 1); q >>= 1;
    const int tap = (int)(q % taps);
    const int chunk = (int)(q / taps);
    const int n = nt * 32 + (lane & 31);
    const int c = chunk * 32 + kk * 16 + 8 * (lane >> 5) + j;
    const int ts = flip ? taps - 1 - tap : tap;
    float v = 0.f;
    if (n < N && c < K) v = src[(int64_t)n * sn + (int64_t)c * sc + (int64_t)ts * st];
    dst[i] = (TD)v;
  }
}

extern "C" int ctu_pack_frag(const float* src, void* dst, ctu_dtype dst_dtype, int32_t N, int32_t K, int32_t taps,
                             int64_t sn, int64_t sc, int64_t st, int32_t flip, ctu_stream_t stream) {
  CTU_REQUIRE(src && dst && N > 0 && K > 0 && K % 32 == 0 && taps > 0, "pack_frag: bad args (K %% 32)");
  const int ntn = (N + 31) / 32;
  const int64_t total = (int64_t)(K / 32) * taps * 2 * ntn * 512;
  const unsigned grid = grid_for(total, 256);
  hipStream_t s = (hipStream_t)stream;
  if (dst_dtype == CTU_F32)
    hipLaunchKernelGGL(pack_frag_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, N, K, taps, sn, sc, st, flip,
                       ntn, total);
  else if (dst_dtype == CTU_BF16)
    hipLaunchKernelGGL(pack_frag_kernel<bf16>, dim3(grid), dim3(256), 0, s, src, (bf16*)dst, N, K, taps, sn, sc, st, flip,
                       ntn, total);
  else { ctu_set_error("pack_frag: bad dtype"); return CTU_ERR_ARG; }
  return ctu_check_launch("pack_frag");
}

// Many panels in ONE launch (94 single launches of ~7 us each per training step otherwise - pure launch latency): job table
// in device memory, blockIdx.y = job.
__global__ __launch_bounds__(256) void pack_frag_batched_kernel(const ctu_pack_job* __restrict__ jobs, const int count) {
  // flat grid: job j owns blocks [block0_j, block0_{j+1}) (sized by its element count: the panels span 27 k .. 7 M elements)
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ctu_pack_job jb = jobs[lo];
  const float* __restrict__ src = jb.src;
  const int64_t nblk = (lo + 1 < count ? jobs[lo + 1].block0 : (int64_t)gridDim.x) - jb.block0;
  // 27-tap bf16 panels of nn.Conv3d weights (every job of a training step): a workgroup takes (32 n x 32 c) tiles with all 27
  // taps.  The source is read as 32 runs of 864 contiguous floats - the (c, t) run of one output channel for the forward
  // panel (sc == 27), the (n', t) run of one reduction channel for the data-gradient panel (sn == 27) - i.e. every byte once
  // and coalesced; the element-per-thread loop below reads 4 bytes 108 bytes apart and fetches every line ~16 times (345 us
  // per step for 220 MB of weights).  The transposition to fragment order goes through LDS (rows padded to an odd dword count).
  if (jb.dst_dtype == CTU_BF16 && jb.taps == 27 && jb.st == 1 && (jb.sc == 27 || jb.sn == 27)) {
    constexpr int ROW = 32 * 27 + 2;
    __shared__ bf16 tile[32 * ROW];
    const bool fwd = jb.sc == 27;          // runs over (c, t) for fixed n; else over (n', t) for fixed c'
    const int chunks = jb.K >> 5;
    const int ntiles = jb.ntn * chunks;
    bf16* __restrict__ dstb = reinterpret_cast<bf16*>(jb.dst);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tl = (int)((int64_t)blockIdx.x - jb.block0); tl < ntiles; tl += (int)nblk) {
      const int nt = tl % jb.ntn, chunk = tl / jb.ntn;
      const int n0 = nt * 32, c0 = chunk * 32;
      __syncthreads();   // the previous tile's readers are done
      // 27 float4 per thread (the runs are 16-byte aligned: K % 32 == 0), nine independent loads in flight at a time (one
      // dependent 4-byte load per iteration made a tile cost ~100 us of pure latency)
#pragma unroll 1
      for (int k0 = 0; k0 < 27; k0 += 9) {
        f32x4 v[9];
        int run[9], q[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
          const int e = threadIdx.x + 256 * (k0 + u);
          run[u] = e / 216;
          q[u] = (e - run[u] * 216) * 4;
          const int n_l = fwd ? run[u] : 0;   // (the data-gradient runs mix n' inside: bounds per element below)
          const float* g = fwd ? src + (int64_t)(n0 + n_l) * jb.sn + (int64_t)c0 * 27 + q[u]
                               : src + (int64_t)(c0 + run[u]) * jb.sc + (int64_t)n0 * 27 + q[u];
          const bool ok = fwd ? n0 + n_l < jb.N : n0 + (q[u] + 3) / 27 < jb.N;   // whole vector inside the parameter
          v[u] = ok ? *reinterpret_cast<const f32x4*>(g) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (!ok && !fwd) {   // ragged end of a data-gradient run (N % 32 != 0): element by element
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (n0 + (q[u] + i) / 27 < jb.N) v[u][i] = g[i];
          }
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = q[u] + i;
            const int inner = r / 27, t = r - inner * 27;
            const int n_l = fwd ? run[u] : inner, c_l = fwd ? inner : run[u];
            tile[n_l * ROW + c_l * 27 + (jb.flip ? 26 - t : t)] = (bf16)v[u][i];
          }
        }
      }
      __syncthreads();
      for (int f = wave; f < 54; f += 4) {
        const int tap = f >> 1, kk = f & 1;
        const bf16* t0 = tile + (lane & 31) * ROW + (kk * 16 + 8 * (lane >> 5)) * 27 + tap;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = t0[j * 27];
        *reinterpret_cast<bf16x8*>(dstb + ((((int64_t)chunk * 27 + tap) * 2 + kk) * jb.ntn + nt) * 512 + lane * 8) = o;
      }
    }
    return;
  }
  for (int64_t i = ((int64_t)blockIdx.x - jb.block0) * 256 + threadIdx.x; i < jb.total; i += nblk * 256) {
    const int j = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    int64_t q = i >> 9;
    const int nt = (int)(q % jb.ntn); q /= jb.ntn;
    const int kk = (int)(q & 1); q >>= 1;
    const int tap = (int)(q % jb.taps);
    const int chunk = (int)(q / jb.taps);
    const int n = nt * 32 + (lane & 31);
    const int c = chunk * 32 + kk * 16 + 8 * (lane >> 5) + j;
    const int ts = jb.flip ? jb.taps - 1 - tap : tap;
    float v = 0.f;
    if (n < jb.N && c < jb.K) v = src[(int64_t)n * jb.sn + (int64_t)c * jb.sc + (int64_t)ts * jb.st];
    if (jb.dst_dtype == CTU_BF16) reinterpret_cast<bf16*>(jb.dst)[i] = (bf16)v;
    else reinterpret_cast<float*>(jb.dst)[i] = v;
  }
}

extern "C" int ctu_pack_frag_batched(const ctu_pack_job* jobs_dev, int32_t count, int64_t total_blocks, ctu_stream_t stream) {
  CTU_REQUIRE(jobs_dev && count > 0 && total_blocks >= count && total_blocks < (1ll << 31), "pack_frag_batched: bad args");
  hipLaunchKernelGGL(pack_frag_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, count);
  return ctu_check_launch("pack_frag_batched");
}
