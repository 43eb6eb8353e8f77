// Dropout on the MI355X path (reference sites networks/vit.py:38,40,57,63,74 and
// networks/hybrid_CTUNet.py:459-467,521-523): y = keep ? x / (1 - p) : 0 with keep regenerated from a Philox counter
// (philox.h), so forward and backward are the same kernel on different operands and no mask is stored.
#include "common.h"
#include "philox.h"

int ctu_make_drop_ctx(float p, uint64_t seed, uint64_t offset, DropCtx* d) {
  if (!(p >= 0.f) || !(p < 1.f)) return -1;
  uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  d->thr16 = thr;
  d->scale = 65536.0f / (float)(65536u - thr);
  d->k0 = (uint32_t)seed;
  d->k1 = (uint32_t)(seed >> 32);
  d->site = (uint32_t)offset;
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, const int64_t n,
                                                      const DropCtx d) {
  const int64_t nv = (n + 7) >> 3;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
    uint32_t o[4];
    philox4x32_10((uint32_t)v, (uint32_t)(v >> 32), 0xD0D0D0D0u, d.site, d.k0, d.k1, o);
    const int64_t i0 = v << 3;
    if (i0 + 8 <= n) {
      float xv[8];
      load8(x + i0, xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = philox_draw16(o, j) >= d.thr16 ? xv[j] * d.scale : 0.f;
      if (res) {
        float rv[8];
        load8(res + i0, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] += rv[j];
      }
      store8(y + i0, xv);
    } else {
      for (int j = 0; i0 + j < n; ++j) {
        float v = philox_draw16(o, j) >= d.thr16 ? (float)x[i0 + j] * d.scale : 0.f;
        if (res) v += (float)res[i0 + j];
        y[i0 + j] = (T)v;
      }
    }
  }
}

extern "C" int ctu_dropout(ctu_dtype dtype, const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed,
                           uint64_t offset, ctu_stream_t stream) {
  CTU_REQUIRE(x && y && n > 0, "dropout: null pointer / empty tensor");
  CTU_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0,
              "dropout: operands must be 16-byte aligned");
  DropCtx d;
  CTU_REQUIRE(ctu_make_drop_ctx(p, seed, offset, &d) == 0, "dropout: p must be in [0, 1)");
  const unsigned grid = grid_for((n + 7) >> 3, 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype, hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, n, d),
               hipLaunchKernelGGL(dropout_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (const bf16*)residual, (bf16*)y, n, d));
  return ctu_check_launch("dropout");
}

// verification hook: the keep flags the attention kernels use, keep[pair][query][key] (1 byte each)
__global__ __launch_bounds__(256) void attn_mask_kernel(uint8_t* __restrict__ keep, const int pairs, const int ntok, const DropCtx d) {
  const int kq_n = (ntok + 3) >> 2;
  const int64_t total = (int64_t)pairs * ntok * kq_n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int kq = (int)(i % kq_n);
    const int q = (int)((i / kq_n) % ntok);
    const int pair = (int)(i / ((int64_t)kq_n * ntok));
    const uint32_t m = attn_keep4(d, (uint32_t)pair, q, kq);
    for (int j = 0; j < 4 && 4 * kq + j < ntok; ++j) keep[((int64_t)pair * ntok + q) * ntok + 4 * kq + j] = (m >> j) & 1u;
  }
}
extern "C" int ctu_attn_dropout_mask(uint8_t* keep, int32_t pairs, int32_t ntok, float p, uint64_t seed, uint64_t offset,
                                     ctu_stream_t stream) {
  CTU_REQUIRE(keep && pairs > 0 && ntok > 0, "attn_dropout_mask: bad args");
  DropCtx d;
  CTU_REQUIRE(ctu_make_drop_ctx(p, seed, offset, &d) == 0, "attn_dropout_mask: p must be in [0, 1)");
  hipLaunchKernelGGL(attn_mask_kernel, dim3(grid_for((int64_t)pairs * ntok * ((ntok + 3) >> 2), 256)), dim3(256), 0,
                     (hipStream_t)stream, keep, pairs, ntok, d);
  return ctu_check_launch("attn_dropout_mask");
}
