// Why does a pure-MFMA kernel beside a plain stream kernel take 2.7 x the back-to-back time (profiles/r03_corun_pure_mfma_vs_stream.log)?
// Same two kernels as corun_mfma.hip, every workgroup stamped: s_memrealtime (100 MHz, constant) and s_memtime (shader cycles) at
// start and end, HW_REG_HW_ID / HW_REG_XCC_ID (which CU it ran on).  From the stamps: the shader clock each kernel saw
// (cycles / real time), the cycles one workgroup needed (issue starvation shows here at an unchanged clock), and how many MFMA
// workgroups shared a CU (the "one long workgroup per CU" grid is only balanced when the dispatcher finds every CU equally free).
// Variants: MFMA grid as 256 long workgroups or 256 x 16 short ones; 1 or 2 MFMA waves per SIMD; s_setprio 3 on the MFMA waves or on
// the stream waves; stream kernel launched first / MFMA kernel launched first (with a 300 us head start).
// hipcc -O3 --offload-arch=gfx950 corun_mech.hip -o corun_mech
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <algorithm>
#include <map>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Stamp { uint64_t rt0, rt1, mt0, mt1, hwid, xcc; };

__device__ __forceinline__ void stamp_begin(Stamp& s) {
  uint32_t h, x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  s.hwid = h;
  s.xcc = x & 15;
  s.rt0 = __builtin_amdgcn_s_memrealtime();
  s.mt0 = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void stamp_end(Stamp& s) {
  s.mt1 = __builtin_amdgcn_s_memtime();
  s.rt1 = __builtin_amdgcn_s_memrealtime();
}

template <int THREADS, int PRIO, int YIELD = 0>
__global__ __launch_bounds__(THREADS) void mfma_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, Stamp* st) {
  const int lane = threadIdx.x & 63;
  Stamp s;
  stamp_begin(s);
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
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
    if (YIELD == 1) asm volatile("s_nop 7");
    if (YIELD == 2) __builtin_amdgcn_s_sleep(1);
    if (YIELD == 3) asm volatile("v_mov_b32 %0, %0" : "+v"(a0[0]));   // one dependent VALU per 16 MFMAs
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) sum += c[i][e];
  if (sum == 12345.678f) out[0] = sum;
  stamp_end(s);
  if (threadIdx.x == 0) st[blockIdx.x] = s;
}

template <int PRIO>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o,
                                                     int64_t n, Stamp* st, int reps) {
  Stamp s;
  stamp_begin(s);
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  for (int r = 0; r < reps; ++r)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    u32x4 x = a[i], y = b[i];
    o[i] = x ^ y;
  }
  stamp_end(s);
  if (threadIdx.x == 0) st[blockIdx.x] = s;
}

static double q(std::vector<double> v, double f) {
  std::sort(v.begin(), v.end());
  return v[(size_t)(f * (v.size() - 1))];
}

static void summarize(const char* tag, const Stamp* s, int n, uint64_t rt_origin) {
  std::vector<double> dur, ghz, cyc, start, end;
  std::map<uint64_t, int> per_cu;
  for (int i = 0; i < n; ++i) {
    const double us = (double)(s[i].rt1 - s[i].rt0) * 0.01;
    dur.push_back(us);
    cyc.push_back((double)(s[i].mt1 - s[i].mt0));
    if (s[i].rt1 > s[i].rt0 + 200) ghz.push_back((double)(s[i].mt1 - s[i].mt0) / (double)(s[i].rt1 - s[i].rt0) * 0.1);
    start.push_back((double)(s[i].rt0 - rt_origin) * 0.01);
    end.push_back((double)(s[i].rt1 - rt_origin) * 0.01);
    // HW_ID (gfx9 layout): simd [5:4], cu [11:8], sh [12], se [15:13]
    const uint64_t h = s[i].hwid, key = (s[i].xcc << 16) | (((h >> 13) & 7) << 8) | (((h >> 12) & 1) << 4) | ((h >> 8) & 15);
    per_cu[key]++;
  }
  int mx = 0, mn = 1 << 30;
  for (auto& kv : per_cu) { mx = std::max(mx, kv.second); mn = std::min(mn, kv.second); }
  printf("    %-7s %6d workgroups on %3zu CUs (%d..%d per CU): span %8.1f us (first start %.1f, last end %.1f); workgroup %8.1f / %8.1f / %8.1f us "
         "(p10 / median / max), %9.0f cycles median",
         tag, n, per_cu.size(), mn, mx, *std::max_element(end.begin(), end.end()) - *std::min_element(start.begin(), start.end()),
         *std::min_element(start.begin(), start.end()), *std::max_element(end.begin(), end.end()), q(dur, 0.1), q(dur, 0.5),
         *std::max_element(dur.begin(), dur.end()), q(cyc, 0.5));
  if (!ghz.empty()) printf(", shader clock %.2f / %.2f / %.2f GHz (p10 / median / p90)", q(ghz, 0.1), q(ghz, 0.5), q(ghz, 0.9));
  printf("\n");
}

int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1));
  CK(hipStreamCreate(&s2));
  bf16x8* in;
  float* out;
  CK(hipMalloc(&in, 256 * 16));
  {  // random bf16 operands in [1, 2): the toggle rate of real data (all-equal words let the governor grant a higher clock)
    std::vector<uint16_t> h(256 * 8);
    uint32_t r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = 0x3f80 | ((r >> 9) & 0x7f); }
    CK(hipMemcpy(in, h.data(), 256 * 16, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&out, 64));
  const int64_t n = (int64_t)512 << 20 >> 4;  // 512 MiB per array
  u32x4 *a, *b, *o;
  CK(hipMalloc(&a, n * 16));
  CK(hipMalloc(&b, n * 16));
  CK(hipMalloc(&o, n * 16));
  CK(hipMemset(a, 1, n * 16));
  CK(hipMemset(b, 2, n * 16));
  const int SG = 8192, SREP = 6, MAXM = 256 * 16;
  Stamp *dm, *ds;
  CK(hipMalloc(&dm, sizeof(Stamp) * MAXM));
  CK(hipMalloc(&ds, sizeof(Stamp) * SG * SREP));
  std::vector<Stamp> hm(MAXM), hs(SG * SREP);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  struct Cfg { const char* name; int threads, split, prio_m, prio_s, order, yield, one; };
  // order: 0 = both queues released together, 1 = MFMA 300 us ahead, 2 = stream 300 us ahead
  const Cfg cfgs[] = {
      {"1 wave/SIMD, 256 long workgroups", 256, 1, 0, 0, 0},
      {"1 wave/SIMD, 256 long workgroups, MFMA first", 256, 1, 0, 0, 1},
      {"1 wave/SIMD, 256 long workgroups, stream first", 256, 1, 0, 0, 2},
      {"1 wave/SIMD, 4096 short workgroups", 256, 16, 0, 0, 0},
      {"1 wave/SIMD, long, MFMA waves s_setprio 3", 256, 1, 3, 0, 0},
      {"1 wave/SIMD, long, stream waves s_setprio 3", 256, 1, 0, 3, 0},
      {"2 waves/SIMD, 256 long workgroups", 512, 1, 0, 0, 0},
      {"2 waves/SIMD, 4096 short workgroups", 512, 16, 0, 0, 0},
      {"1 wave/SIMD, long, MFMA first, s_nop 7 per 16 MFMAs", 256, 1, 0, 0, 1, 1},
      {"1 wave/SIMD, long, MFMA first, s_sleep 1 per 16 MFMAs", 256, 1, 0, 0, 1, 2},
      {"1 wave/SIMD, long, MFMA first, one dependent v_mov per 16 MFMAs", 256, 1, 0, 0, 1, 3},
      {"1 wave/SIMD, long, MFMA first, stream as ONE launch of 6 passes", 256, 1, 0, 0, 1, 0, 1},
      {"1 wave/SIMD, long, together, stream as ONE launch of 6 passes", 256, 1, 0, 0, 0, 0, 1},
      {"2 waves/SIMD, 4096 short, stream as ONE launch of 6 passes", 512, 16, 0, 0, 0, 0, 1},
  };
  // warm the clocks up: ~2 s of MFMA + stream back to back
  for (int r = 0; r < 300; ++r) {
    hipLaunchKernelGGL((mfma_kernel<256, 0>), dim3(256), dim3(256), 0, s1, in, out, 6000, dm);
    hipLaunchKernelGGL((stream_kernel<0>), dim3(SG), dim3(256), 0, s1, a, b, o, n, ds, 1);
  }
  CK(hipDeviceSynchronize());
  for (const Cfg& c : cfgs) {
    const int iters = (c.threads == 256 ? 6000 : 3000) / c.split, grid = 256 * c.split, nst = c.one ? SG : SG * SREP;
    auto mf = [&]() {
      if (c.threads == 256) {
        if (c.yield == 1) hipLaunchKernelGGL((mfma_kernel<256, 0, 1>), dim3(grid), dim3(256), 0, s1, in, out, iters, dm);
        else if (c.yield == 2) hipLaunchKernelGGL((mfma_kernel<256, 0, 2>), dim3(grid), dim3(256), 0, s1, in, out, iters, dm);
        else if (c.yield == 3) hipLaunchKernelGGL((mfma_kernel<256, 0, 3>), dim3(grid), dim3(256), 0, s1, in, out, iters, dm);
        else if (c.prio_m) hipLaunchKernelGGL((mfma_kernel<256, 3>), dim3(grid), dim3(256), 0, s1, in, out, iters, dm);
        else hipLaunchKernelGGL((mfma_kernel<256, 0>), dim3(grid), dim3(256), 0, s1, in, out, iters, dm);
      } else {
        hipLaunchKernelGGL((mfma_kernel<512, 0>), dim3(grid), dim3(512), 0, s1, in, out, iters, dm);
      }
    };
    auto st = [&]() {
      if (c.one) {
        hipLaunchKernelGGL((stream_kernel<0>), dim3(SG), dim3(256), 0, s2, a, b, o, n, ds, SREP);
        return;
      }
      for (int r = 0; r < SREP; ++r) {
        if (c.prio_s) hipLaunchKernelGGL((stream_kernel<3>), dim3(SG), dim3(256), 0, s2, a, b, o, n, ds + r * SG, 1);
        else hipLaunchKernelGGL((stream_kernel<0>), dim3(SG), dim3(256), 0, s2, a, b, o, n, ds + r * SG, 1);
      }
    };
    auto run = [&](int what) {  // 1 = MFMA, 2 = stream, 3 = both
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {  // the second repetition is the one reported
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        CK(hipStreamWaitEvent(s1, e0, 0));
        CK(hipStreamWaitEvent(s2, e0, 0));
        if (what == 3 && c.order == 2) {
          st();
          hipEvent_t g;
          CK(hipEventCreate(&g));
          CK(hipEventRecord(g, 0));
          CK(hipEventSynchronize(g));
          struct timespec ts = {0, 300000};
          nanosleep(&ts, nullptr);
          mf();
        } else {
          if (what & 1) mf();
          if (what == 3 && c.order == 1) {
            struct timespec ts = {0, 300000};
            nanosleep(&ts, nullptr);
          }
          if (what & 2) st();
        }
        hipEvent_t d1, d2;
        CK(hipEventCreate(&d1));
        CK(hipEventCreate(&d2));
        CK(hipEventRecord(d1, s1));
        CK(hipEventRecord(d2, s2));
        CK(hipStreamWaitEvent(0, d1, 0));
        CK(hipStreamWaitEvent(0, d2, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      CK(hipMemcpy(hm.data(), dm, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hs.data(), ds, sizeof(Stamp) * SG * SREP, hipMemcpyDeviceToHost));
      uint64_t origin = ~0ull;
      if (what & 1) for (int i = 0; i < grid; ++i) origin = std::min(origin, hm[i].rt0);
      if (what & 2) for (int i = 0; i < nst; ++i) origin = std::min(origin, hs[i].rt0);
      if (what & 1) summarize("MFMA", hm.data(), grid, origin);
      if (what & 2) summarize("stream", hs.data(), nst, origin);
      if (what == 3) {  // how much of the stream's work completed while MFMA workgroups were alive
        uint64_t m0 = ~0ull, m1 = 0;
        for (int i = 0; i < grid; ++i) { m0 = std::min(m0, hm[i].rt0); m1 = std::max(m1, hm[i].rt1); }
        long inside = 0, started = 0;
        for (int i = 0; i < nst; ++i) {
          inside += hs[i].rt1 >= m0 && hs[i].rt1 <= m1 && hs[i].rt0 >= m0;
          started += hs[i].rt0 >= m0 && hs[i].rt0 <= m1;
        }
        const double bytes = (double)inside * (double)n * 48.0 / SG * (c.one ? SREP : 1);
        printf("    stream workgroups started AND finished inside the MFMA kernel's %0.f us: %ld (started: %ld) = %.2f TB/s beside the MFMA kernel\n",
               (double)(m1 - m0) * 0.01, inside, started, bytes / ((double)(m1 - m0) * 0.01) / 1e6);
      }
      return ms;
    };
    printf("== %s\n", c.name);
    printf("  MFMA alone\n");
    const float tm = run(1);
    printf("  stream alone (6 x 1.5 GiB)\n");
    const float ts = run(2);
    printf("  together\n");
    const float tb = run(3);
    const double flop = (double)grid * (c.threads / 64) * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("  => MFMA alone %.3f ms (%.0f TFLOP/s), stream alone %.3f ms (%.2f TB/s), together %.3f ms, back to back %.3f ms, overlap %.2f\n",
           tm, flop / tm / 1e9, ts, SREP * 1.5 * 1.0737 / ts, tb, tm + ts, (tm + ts - tb) / (tm < ts ? tm : ts));
  }
  return 0;
}
