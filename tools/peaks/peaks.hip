// Measured ceilings of the MI355X this code runs on ("microbenchmarked HBM and MFMA peaks on the box,
// stated"): bf16 MFMA rate and the clock the chip sustains under that load, and HBM read / write / copy bandwidth with
// 16-B-per-lane streams on buffers beyond the 256 MB Infinity Cache.  Standalone: hipcc --offload-arch=gfx950 peaks.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every wave: `iters` x 16 back-to-back v_mfma_f32_32x32x16_bf16 on 4 independent accumulators, operands in registers
// (random bf16 data: the clock a chip holds on zeros is not the clock it holds on data)
__global__ __launch_bounds__(512) void mfma_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters,
                                                   uint64_t* __restrict__ clk) {
  const int lane = threadIdx.x & 63;
  bf16x8 a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane];
  f32x16 c[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c[3], 0, 0, 0);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += c[i][e];
  if (s == 12345.678f) out[0] = s;  // keep the chain alive
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// The inner loop of the halo convolution, both bf16 MFMA shapes at the same 64 x 128 output tile per wave, every operand re-read
// from LDS (ds_read_b128, random bf16 data) each k step: 32x32x16 = 2 A + 4 B fragments and 8 MFMAs per 16 k; 16x16x32 = 4 A +
// 8 B fragments and 32 MFMAs per 32 k - the same LDS bytes and the same pipe cycles per FLOP.  MI355X_MICROARCH.md (DVFS item 7)
// reports that the clock the chip holds under such a load depends on the shape; this measures it here.
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int SHAPE>
__global__ __launch_bounds__(512) void lds_mfma_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters,
                                                       uint64_t* __restrict__ clk) {
  __shared__ bf16x8 lds[64 * 64];  // 64 KiB of fragments
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = in[(i * 7 + (i >> 6)) & 255];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 c[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // two k16 steps = 32 k
        const int base = ((it * 2 + ks + wave * 5) & 7) * 6 * 64 + lane;
        bf16x8 a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = lds[(base + i * 64) & 4095];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = lds[(base + (2 + j) * 64) & 4095];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], c[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += c[i][j][e];
  } else {
    f32x4 c[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) c[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int base = ((it + wave * 5) & 3) * 12 * 64 + lane;
      bf16x8 a[4], b[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = lds[(base + i * 64) & 4095];
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = lds[(base + (4 + j) * 64) & 4095];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += c[i][j][e];
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    acc ^= v;
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ dst, size_t n) {
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

template <typename F> static double time_ms(F&& f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f(i);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);

  // ---- MFMA ----
  std::vector<uint16_t> h(256 * 8);
  srand(1);
  for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // bf16 in +-[0.008, 0.03]: random mantissas, both signs
  bf16x8* din; float* dout; uint64_t* dclk;
  CK(hipMalloc(&din, 256 * 16)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dclk, 16 * 4096));
  CK(hipMemcpy(din, h.data(), 256 * 16, hipMemcpyHostToDevice));
  for (int wpw : {4, 8}) {  // waves per workgroup = waves per CU: 1 or 2 per SIMD
    const int iters = 40000;
    // warm the clock governor with ~1.5 s of this load, then time
    for (int i = 0; i < 150; ++i) hipLaunchKernelGGL(mfma_kernel, dim3(cus), dim3(64 * wpw), 0, 0, din, dout, iters, dclk);
    CK(hipDeviceSynchronize());
    const double ms = time_ms([&](int) { hipLaunchKernelGGL(mfma_kernel, dim3(cus), dim3(64 * wpw), 0, 0, din, dout, iters, dclk); }, 5);
    std::vector<uint64_t> hc(2 * cus);
    CK(hipMemcpy(hc.data(), dclk, 16 * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < cus; ++b) ghz.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 32 * 32 * 16 * 16.0 * iters * wpw * cus;
    const double cyc = (double)hc[0] / (16.0 * iters * (wpw / 4));
    printf("mfma_f32_32x32x16_bf16, %d waves/CU: %.1f TFLOP/s dense bf16; in-kernel clock median %.3f GHz (min %.3f max %.3f); "
           "%.2f cycles per MFMA per SIMD\n", wpw, flop / ms / 1e9, ghz[cus / 2], ghz.front(), ghz.back(), cyc);
  }

  // ---- the two bf16 MFMA shapes fed from LDS, two waves per SIMD ----
  for (int shape : {32, 16, 32, 16}) {
    const int iters = 20000;
    auto launch = [&]() {
      if (shape == 32) hipLaunchKernelGGL(lds_mfma_kernel<32>, dim3(cus), dim3(512), 0, 0, din, dout, iters, dclk);
      else hipLaunchKernelGGL(lds_mfma_kernel<16>, dim3(cus), dim3(512), 0, 0, din, dout, iters, dclk);
    };
    for (int i = 0; i < 120; ++i) launch();
    CK(hipDeviceSynchronize());
    const double ms = time_ms([&](int) { launch(); }, 5);
    std::vector<uint64_t> hc(2 * cus);
    CK(hipMemcpy(hc.data(), dclk, 16 * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < cus; ++b) ghz.push_back((double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 64 * 128 * 32 * (double)iters * 8 * cus;
    printf("LDS-fed 64x128 wave tile, %s, 8 waves/CU: %.1f TFLOP/s; in-kernel clock median %.3f GHz; %.0f cycles per 32-k step per wave pair\n",
           shape == 32 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", flop / ms / 1e9, ghz[cus / 2], (double)hc[0] / iters);
  }

  // ---- HBM ----
  const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer, 4 buffers rotated: far beyond the 256 MB Infinity Cache
  u32x4* buf[4];
  for (auto& b : buf) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  const size_t n = bytes / 16;
  for (int wgs_per_cu : {8, 16, 32}) {
    const int grid = cus * wgs_per_cu;
    const double r = time_ms([&](int i) { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf[i & 3], sink, n); }, 12);
    const double w = time_ms([&](int i) { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, buf[i & 3], n); }, 12);
    const double c = time_ms([&](int i) { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, buf[i & 3], buf[(i + 2) & 3], n); }, 12);
    printf("HBM, %2d workgroups/CU x 256 threads, 16 B/lane: read %.2f TB/s, write %.2f TB/s, copy %.2f TB/s (read+write bytes)\n",
           wgs_per_cu, bytes / r / 1e9, bytes / w / 1e9, 2.0 * bytes / c / 1e9);
  }
  return 0;
}
