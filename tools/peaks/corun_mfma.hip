// Can an HBM-bound streaming kernel run at (nearly) its own speed beside a matrix kernel that is LIGHT on the memory path and
// leaves most of the register file free?  (the halo forward kernel is no such partner - it streams its weights
// L2 -> LDS - and the halo weight gradient, which is, fills every CU's registers.)  A pure-MFMA kernel, one 256-thread workgroup per
// CU (one wave per SIMD, ~80 VGPRs, no memory traffic), against a stream kernel shaped like in_bwd_apply (two 16-B loads, one
// 16-B store per lane and iteration), alone and together on two HIP streams.  hipcc --offload-arch=gfx950 corun_mfma.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane];
  f32x16 c[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c[2], 0, 0, 0);
      c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += c[i][e];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o,
                                                     int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    u32x4 x = a[i], y = b[i];
    o[i] = x ^ y;
  }
}

int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  bf16x8* in;
  float* out;
  CK(hipMalloc(&in, 256 * 16));
  CK(hipMemset(in, 0x3c, 256 * 16));
  CK(hipMalloc(&out, 64));
  const int64_t n = (int64_t)512 << 20 >> 4;  // 512 MiB per array
  u32x4 *a, *b, *o;
  CK(hipMalloc(&a, n * 16));
  CK(hipMalloc(&b, n * 16));
  CK(hipMalloc(&o, n * 16));
  CK(hipMemset(a, 1, n * 16));
  CK(hipMemset(b, 2, n * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto&& f) {
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    CK(hipStreamWaitEvent(s1, e0, 0));
    CK(hipStreamWaitEvent(s2, e0, 0));
    f();
    hipEvent_t d1, d2;
    CK(hipEventCreate(&d1));
    CK(hipEventCreate(&d2));
    CK(hipEventRecord(d1, s1));
    CK(hipEventRecord(d2, s2));
    CK(hipStreamWaitEvent(0, d1, 0));
    CK(hipStreamWaitEvent(0, d2, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  const int it_thin = 6000, it_fat = 3000;
  for (int variant = 0; variant < 2; ++variant) {
    auto mf = [&]() {
      if (variant == 0) hipLaunchKernelGGL(mfma_kernel<256>, dim3(256), dim3(256), 0, s1, in, out, it_thin);    // 1 wave / SIMD
      else hipLaunchKernelGGL(mfma_kernel<512>, dim3(256), dim3(512), 0, s1, in, out, it_fat);                   // 2 waves / SIMD
    };
    auto st = [&]() { for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(stream_kernel, dim3(8192), dim3(256), 0, s2, a, b, o, n); };
    const float tm = time(mf), ts = time(st);
    const float tb = time([&]() { mf(); st(); });
    const double flop = 256.0 * (variant == 0 ? 4 : 8) * (variant == 0 ? it_thin : it_fat) * 16 * 2.0 * 32 * 32 * 16;
    printf("MFMA kernel %s: alone %.3f ms (%.0f TFLOP/s); stream 6 x 1.5 GiB alone %.3f ms (%.2f TB/s); together %.3f ms  -> overlap %.2f\n",
           variant == 0 ? "1 wave/SIMD (256 thr/CU)" : "2 waves/SIMD (512 thr/CU)", tm, flop / tm / 1e9, ts, 6 * 1.5 * 1.0737 / ts,
           tb, (tm + ts - tb) / (tm < ts ? tm : ts));
  }
  return 0;
}
