// gemm_nt_narrow : out[M][N] = A[M][K] W^T for the long-M layers whose OUTPUT is narrow - N = 32 or 64: conv1 of the ResNet bottlenecks
// (resnet.py:96: 128 -> 32 @ 48 x 48 x 96, 256 -> 64 @ 24 x 24 x 48) and the data gradient of their conv3 (resnet.py:100).  The general
// LDS-DMA kernel moves these 141 MB in 57 us (2.5 TB/s): a 128 x 64 tile whose second half of columns does not exist, a ring of stages
// and an epilogue patch per tile, for 8 MFMAs of work per 32 rows.  Here the layer is what it is, a stream:
//   * every WAVE walks 32-row tiles of its own (no workgroup barrier in the loop); W (K x N <= 256 x 64) sits in its registers as MFMA
//     A fragments for the kernel's life - from either storage order (w_kn: the forward weight read reduction-major by a data gradient);
//   * lane (r, hf) loads the 16-byte pieces 16 s + 8 hf of row r - the B fragments of the transposed product, as in ff_fused.hip -
//     with the next tile's rows requested before the current tile's MFMAs;
//   * the result leaves through a wave-private LDS tile as 16-byte lanes of whole rows: 16 consecutive 64-byte rows = 1 KiB contiguous
//     per store instruction at N = 32;
//   * InstanceNorm sums (in_acc) as per-lane partials over all tiles of the batch item the workgroup is assigned to (grid.y), reduced
//     over the row lanes and the four waves once, then one fp64 atomic pair per column and workgroup.
#include "gemm_dma.h"
#include "mma.h"

namespace {

template <int KS, int NTN>
__global__ __launch_bounds__(256, 2) void gemm_nt_narrow_kernel(const GemmNtArgs p) {
  constexpr int N = 32 * NTN;
  constexpr int PITCH = N + 8;   // bf16 elements per staged row (16 bytes of padding: the 8-byte writes of 32 rows spread over the banks)
  __shared__ __attribute__((aligned(16))) bf16 stg_all[4][32 * PITCH];
  __shared__ float red[4][N][2];
  const int tid = threadIdx.x;
  const int lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = tid >> 6;
  bf16* stg = stg_all[wave];
  const int K = KS * 16;
  // ---- W fragments: A operand of the transposed product, lane (m = output column within the tile, hf): k = 16 s + 8 hf .. + 7
  bf16x8 wf[NTN][KS];
#pragma unroll
  for (int n = 0; n < NTN; ++n)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int col = 32 * n + r, k0 = 16 * s + 8 * hf;
      if (p.w_kn) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p.w[(size_t)(k0 + i) * N + col];
        wf[n][s] = v;
      } else {
        wf[n][s] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)col * K + k0);
      }
    }
  // tiles of this workgroup: batch item blockIdx.y when statistics are wanted (rows [item * in_rows, (item + 1) * in_rows)), else all rows
  const int64_t row_begin = p.in_acc ? (int64_t)blockIdx.y * p.in_rows : 0;
  const int ntiles = (int)((p.in_acc ? (int64_t)p.in_rows : (int64_t)p.M) >> 5);
  const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  bf16* out = reinterpret_cast<bf16*>(p.out);
  float s1[NTN][16], s2[NTN][16];
#pragma unroll
  for (int n = 0; n < NTN; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) { s1[n][e] = 0.f; s2[n][e] = 0.f; }

  bf16x8 xf[KS], xn[KS];
  auto request = [&](int t, bf16x8 (&dst)[KS]) {
    const bf16* src = p.a1 + (size_t)(row_begin + (int64_t)t * 32 + r) * K + 8 * hf;
#pragma unroll
    for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(src + 16 * s);
  };
  int t = wid;
  if (t < ntiles) request(t, xf);
  for (; t < ntiles; t += nw) {
    const int tn = t + nw;
    if (tn < ntiles) request(tn, xn);   // the next tile's rows are in flight under this tile's MFMAs and stores
    f32x16 acc[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][s], xf[s], acc[n], 0, 0, 0);
    }
    if (p.in_acc) {
#pragma unroll
      for (int n = 0; n < NTN; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float v = acc[n][e]; s1[n][e] += v; s2[n][e] += v * v; }
    }
    // lane (r, hf) holds columns 32 n + 4 hf + 8 q + j of row r: through the staging tile to whole rows
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
    for (int n = 0; n < NTN; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (bf16)acc[n][4 * q + j];
        *reinterpret_cast<bf16x4*>(stg + r * PITCH + 32 * n + 4 * hf + 8 * q) = v;
      }
    __builtin_amdgcn_wave_barrier();
    const size_t row0 = (size_t)(row_begin + (int64_t)t * 32);
    constexpr int VPR = N / 8;            // 16-byte vectors per row: 4 | 8
    constexpr int RPI = 64 / VPR;         // rows per store instruction: 16 | 8
    if (!(p.debug & 1)) {
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int rr = i * RPI + lane / VPR, cg = lane % VPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + rr * PITCH + cg * 8);
        *reinterpret_cast<u32x4*>(out + (row0 + rr) * p.ep.ldc + cg * 8) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = xn[s];
  }
  if (p.in_acc) {
    // column sums: over the 32 row lanes of each lane half (xor-shuffles), then over the four waves through LDS
#pragma unroll
    for (int n = 0; n < NTN; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float a = s1[n][e], b = s2[n][e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r == 0) {
          const int col = 32 * n + 4 * hf + 8 * (e >> 2) + (e & 3);
          red[wave][col][0] = a;
          red[wave][col][1] = b;
        }
      }
    __syncthreads();
    if (tid < N) {
      const float t1 = red[0][tid][0] + red[1][tid][0] + red[2][tid][0] + red[3][tid][0];
      const float t2 = red[0][tid][1] + red[1][tid][1] + red[2][tid][1] + red[3][tid][1];
      atomicAdd(&p.in_acc[((size_t)blockIdx.y * N + tid) * 2], (double)t1);
      atomicAdd(&p.in_acc[((size_t)blockIdx.y * N + tid) * 2 + 1], (double)t2);
    }
  }
}

}  // namespace

bool gemm_nt_narrow_ok(const GemmNtArgs& p) {
  const ctu_epilogue& e = p.ep;
  return (p.N == 32 || p.N == 64) && p.K % 16 == 0 && (p.K == 32 || p.K == 64 || p.K == 128) && p.a2 == nullptr && p.C1 == p.K &&
         p.M % 32 == 0 && p.M >= 32768 && p.splitk <= 1 && !e.bias && e.act == 0 && !e.residual && !e.pre_out && !e.scatter &&
         e.n_split <= 0 && e.ldc == p.N && (!p.in_acc || (p.in_rows % 32 == 0 && p.M % p.in_rows == 0)) &&
         // Layers that also sum InstanceNorm statistics stay on the general kernel unless asked for (route bit): the sums are the
         // same numbers to fp32 rounding, but in another order, and CUNet-101's bf16 forward amplifies a changed last bit of a
         // statistic into a 3e-4 shift of the coarsest head's Dice term - across the gate of the whole-model test, which is set at
         // twice the reference's own bf16 error (0.87978 -> 0.88005 against 0.87968 / float64 0.87949)
         (!p.in_acc || (ctu_option_route() & CTU_ROUTE_NT_NARROW_STATS)) &&
         !(ctu_option_route() & CTU_ROUTE_NT_NO_NARROW);
}

void launch_gemm_nt_narrow(const GemmNtArgs& p, hipStream_t stream) {
  const int items = p.in_acc ? p.M / p.in_rows : 1;
  const int64_t tiles = (p.in_acc ? (int64_t)p.in_rows : (int64_t)p.M) / 32;
  // ~1 024 workgroups in all (two per CU resident, two rounds), at least eight tiles per wave
  int64_t gx = 1024 / items;
  const int64_t cap = (tiles + 31) / 32;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  const dim3 g((unsigned)gx, (unsigned)items), b(256);
#define CTU_NARROW(KS_)                                                                            \
  do {                                                                                             \
    if (p.N == 32) hipLaunchKernelGGL((gemm_nt_narrow_kernel<KS_, 1>), g, b, 0, stream, p);        \
    else hipLaunchKernelGGL((gemm_nt_narrow_kernel<KS_, 2>), g, b, 0, stream, p);                  \
  } while (0)
  switch (p.K / 16) {
    case 2: CTU_NARROW(2); break;
    case 4: CTU_NARROW(4); break;
    case 8: CTU_NARROW(8); break;
    default: break;
  }
#undef CTU_NARROW
}
