// Caller-side kernels of the training step: DiceCE with fused deep-supervision target lookup (trainer_CTUNet.py:90-103,
// main_CTUNet.py:156-158) and a flat fused AdamW (main_CTUNet.py:192-193).  Plus the library's error plumbing.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// error state
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void ctu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int ctu_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ctu_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CTU_ERR_LAUNCH;
  }
  return CTU_OK;
}
extern "C" const char* ctu_last_error(void) { return g_err; }
extern "C" int ctu_abi_version(void) { return 8; }

// ---------------------------------------------------------------------------------------------------------
// DiceCE
// ---------------------------------------------------------------------------------------------------------
#define NC_MAX 16
#define NEG_INFINITY_F (-3.0e38f)
struct DiceArgs {
  const void* logits;
  const float* labels;
  const int32_t* idx_d;
  const int32_t* idx_h;
  const int32_t* idx_w;
  int ldl, B, D, H, W, LD, LH, LW, n_cls;
};

template <typename T>
__device__ __forceinline__ int dice_voxel(const DiceArgs& a, int b, int64_t s, float (&p)[NC_MAX]) {
  // returns the label; p = softmax over the first n_cls logits of voxel s of sample b
  const T* lg = reinterpret_cast<const T*>(a.logits) + ((size_t)b * a.D * a.H * a.W + s) * a.ldl;
  float mx = NEG_INFINITY_F;
#pragma unroll
  for (int v = 0; v < NC_MAX / 8; ++v)
    if (v * 8 < a.n_cls) {
      float x[8];
      load8(lg + v * 8, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[v * 8 + e] = (v * 8 + e < a.n_cls) ? x[e] : -1.0e30f;
        mx = fmaxf(mx, p[v * 8 + e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) p[v * 8 + e] = -1.0e30f;
    }
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c) { p[c] = (c < a.n_cls) ? __expf(p[c] - mx) : 0.f; sum += p[c]; }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c) p[c] *= inv;
  const int w = (int)(s % a.W);
  const int64_t t = s / a.W;
  const int h = (int)(t % a.H), d = (int)(t / a.H);
  const int sd = a.idx_d[d], sh = a.idx_h[h], sw = a.idx_w[w];
  if ((sd | sh | sw) < 0) return 0;  // scipy zoom: out-of-range coordinate reads cval = 0
  const float lab = a.labels[(((size_t)b * a.LD + sd) * a.LH + sh) * a.LW + sw];
  return (int)lab;
}

template <typename T>
__global__ __launch_bounds__(256) void dicece_fwd_kernel(const DiceArgs a, float* __restrict__ acc) {
  __shared__ float s_py[NC_MAX], s_y[NC_MAX], s_p2[NC_MAX], s_ce;
  const int b = blockIdx.y;
  if (threadIdx.x < NC_MAX) { s_py[threadIdx.x] = 0.f; s_y[threadIdx.x] = 0.f; s_p2[threadIdx.x] = 0.f; }
  if (threadIdx.x == 0) s_ce = 0.f;
  __syncthreads();
  const int64_t S = (int64_t)a.D * a.H * a.W;
  float p2[NC_MAX];
  float ce = 0.f;
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c) p2[c] = 0.f;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
    float p[NC_MAX];
    const int lab = dice_voxel<T>(a, b, s, p);
    float pl = 0.f;
#pragma unroll
    for (int c = 0; c < NC_MAX; ++c) {
      p2[c] = fmaf(p[c], p[c], p2[c]);
      pl = (c == lab) ? p[c] : pl;
    }
    if (lab >= 0 && lab < a.n_cls) {
      atomicAdd(&s_py[lab], pl);
      atomicAdd(&s_y[lab], 1.0f);
      ce -= __logf(fmaxf(pl, 1.0e-38f));
    }
  }
#pragma unroll
  for (int c = 0; c < NC_MAX; ++c) {
    const float v = wave_sum(p2[c]);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_p2[c], v);
  }
  ce = wave_sum(ce);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_ce, ce);
  __syncthreads();
  if (threadIdx.x < a.n_cls) {
    float* dst = acc + ((size_t)b * a.n_cls + threadIdx.x) * 3;
    atomicAdd(dst + 0, s_py[threadIdx.x]);
    atomicAdd(dst + 1, s_p2[threadIdx.x]);
    atomicAdd(dst + 2, s_y[threadIdx.x]);
  }
  if (threadIdx.x == 0) atomicAdd(acc + (size_t)a.B * a.n_cls * 3, s_ce);
}

__global__ void dicece_finalize_kernel(const float* __restrict__ acc, const int B, const int n_cls, const float inv_vox,
                                       const float nr, const float dr, const float weight, float* __restrict__ loss) {
  // one wave: lanes stride over (b, c)
  float d = 0.f;
  for (int i = threadIdx.x; i < B * n_cls; i += 64) {
    const float I = acc[i * 3], P2 = acc[i * 3 + 1], Y = acc[i * 3 + 2];
    d += 1.0f - (2.0f * I + nr) / (Y + P2 + dr);
  }
  d = wave_sum(d);
  if (threadIdx.x == 0) {
    const float dice = d / (float)(B * n_cls);
    const float ce = acc[B * n_cls * 3] * inv_vox;
    atomicAdd(loss, weight * (dice + ce));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dicece_bwd_kernel(const DiceArgs a, const float* __restrict__ acc, const float nr,
                                                         const float dr, const float weight,
                                                         const float* __restrict__ gscale, T* __restrict__ dlogits) {
  __shared__ float ca[NC_MAX], cb[NC_MAX];
  const int b = blockIdx.y;
  const int64_t S = (int64_t)a.D * a.H * a.W;
  const float inv_bc = 1.0f / (float)(a.B * a.n_cls);
  if (threadIdx.x < NC_MAX) {
    float va = 0.f, vb = 0.f;
    if (threadIdx.x < a.n_cls) {
      const float* src = acc + ((size_t)b * a.n_cls + threadIdx.x) * 3;
      const float I = src[0], P2 = src[1], Y = src[2];
      const float den = Y + P2 + dr;
      va = -2.0f / den * inv_bc;                           // d dice / d p_c  (y_c = 1 term)
      vb = 2.0f * (2.0f * I + nr) / (den * den) * inv_bc;  // ... + vb * p_c
    }
    ca[threadIdx.x] = va; cb[threadIdx.x] = vb;
  }
  __syncthreads();
  const float w = weight * (gscale ? gscale[0] : 1.0f);
  const float inv_ce = 1.0f / (float)((int64_t)a.B * S);
  T* out = dlogits + (size_t)b * S * a.ldl;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
    float p[NC_MAX], g[NC_MAX];
    const int lab = dice_voxel<T>(a, b, s, p);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC_MAX; ++c) {
      g[c] = cb[c] * p[c] + ((c == lab) ? ca[c] : 0.f);
      dot = fmaf(g[c], p[c], dot);
    }
#pragma unroll
    for (int v = 0; v < NC_MAX / 8; ++v)
      if (v * 8 < a.ldl) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = v * 8 + e;
          const float y = (c == lab) ? 1.0f : 0.f;
          o[e] = (c < a.n_cls) ? w * (p[c] * (g[c] - dot) + (p[c] - y) * inv_ce) : 0.f;
        }
        store8(out + (size_t)s * a.ldl + v * 8, o);
      }
  }
}

static int fill_dice(DiceArgs* a, const void* logits, int ldl, const float* labels, const int32_t* idx_d,
                     const int32_t* idx_h, const int32_t* idx_w, int B, int D, int H, int W, int LD, int LH, int LW,
                     int n_cls) {
  CTU_REQUIRE(logits && labels && idx_d && idx_h && idx_w, "dicece: null pointer");
  CTU_REQUIRE(n_cls > 0 && n_cls <= NC_MAX && ldl % 8 == 0 && ldl >= n_cls && ldl <= NC_MAX,
              "dicece: need n_cls <= %d and ldl in {8,16} >= n_cls (n_cls=%d ldl=%d)", NC_MAX, n_cls, ldl);
  CTU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && LD > 0 && LH > 0 && LW > 0, "dicece: bad dims");
  a->logits = logits; a->labels = labels; a->idx_d = idx_d; a->idx_h = idx_h; a->idx_w = idx_w;
  a->ldl = ldl; a->B = B; a->D = D; a->H = H; a->W = W; a->LD = LD; a->LH = LH; a->LW = LW; a->n_cls = n_cls;
  return CTU_OK;
}

extern "C" int ctu_dicece_fwd(ctu_dtype dtype, const void* logits, int32_t ldl, const float* labels,
                              const int32_t* idx_d, const int32_t* idx_h, const int32_t* idx_w, int32_t B, int32_t D,
                              int32_t H, int32_t W, int32_t LD, int32_t LH, int32_t LW, int32_t n_cls, float* acc,
                              ctu_stream_t stream) {
  DiceArgs a;
  if (int rc = fill_dice(&a, logits, ldl, labels, idx_d, idx_h, idx_w, B, D, H, W, LD, LH, LW, n_cls)) return rc;
  CTU_REQUIRE(acc, "dicece: null acc");
  const int64_t S = (int64_t)D * H * W;
  dim3 grid(grid_for(S, 256, 192), B);  // few workgroups: they all add into the same 43 addresses at the end
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype, hipLaunchKernelGGL(dicece_fwd_kernel<float>, grid, dim3(256), 0, s, a, acc),
               hipLaunchKernelGGL(dicece_fwd_kernel<bf16>, grid, dim3(256), 0, s, a, acc));
  return ctu_check_launch("dicece_fwd");
}

extern "C" int ctu_dicece_finalize(const float* acc, int32_t B, int32_t n_cls, int64_t S, float smooth_nr,
                                   float smooth_dr, float weight, float* loss_out, ctu_stream_t stream) {
  CTU_REQUIRE(acc && loss_out && B > 0 && n_cls > 0 && S > 0, "dicece_finalize: bad args");
  hipLaunchKernelGGL(dicece_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, B, n_cls,
                     1.0f / (float)((double)B * (double)S), smooth_nr, smooth_dr, weight, loss_out);
  return ctu_check_launch("dicece_finalize");
}

extern "C" int ctu_dicece_bwd(ctu_dtype dtype, const void* logits, int32_t ldl, const float* labels,
                              const int32_t* idx_d, const int32_t* idx_h, const int32_t* idx_w, int32_t B, int32_t D,
                              int32_t H, int32_t W, int32_t LD, int32_t LH, int32_t LW, int32_t n_cls, const float* acc,
                              float smooth_nr, float smooth_dr, float weight, const float* gscale, void* dlogits,
                              ctu_stream_t stream) {
  DiceArgs a;
  if (int rc = fill_dice(&a, logits, ldl, labels, idx_d, idx_h, idx_w, B, D, H, W, LD, LH, LW, n_cls)) return rc;
  CTU_REQUIRE(acc && dlogits, "dicece_bwd: null pointer");
  const int64_t S = (int64_t)D * H * W;
  dim3 grid(grid_for(S, 256, 2048), B);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype,
               hipLaunchKernelGGL(dicece_bwd_kernel<float>, grid, dim3(256), 0, s, a, acc, smooth_nr, smooth_dr, weight,
                                  gscale, (float*)dlogits),
               hipLaunchKernelGGL(dicece_bwd_kernel<bf16>, grid, dim3(256), 0, s, a, acc, smooth_nr, smooth_dr, weight,
                                  gscale, (bf16*)dlogits));
  return ctu_check_launch("dicece_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW, amsgrad=False, maximize=False): p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------------------
struct SkipRanges {
  int64_t r[16][2];
  int n;
};
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, bf16* __restrict__ mirror,
                                                    const int64_t n,
                                                    float lr, const float b1, const float b2, const float eps,
                                                    const float wd, float inv_bc1, float inv_sqrt_bc2,
                                                    const SkipRanges skip, const float* __restrict__ hyper) {
  if (hyper) {  // device-resident learning rate and bias corrections (ctu_adamw_tick): graph-replay safe
    lr = hyper[0];
    inv_bc1 = hyper[2];
    inv_sqrt_bc2 = hyper[3];
  }
  for (int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i4 < n; i4 += (int64_t)gridDim.x * 256 * 4) {
    bool skipped = false;
    for (int k = 0; k < skip.n; ++k) skipped = skipped || (i4 + 3 >= skip.r[k][0] && i4 < skip.r[k][1]);
    if (i4 + 3 < n && !skipped) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i4);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i4);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i4);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + i4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pp[e] *= (1.0f - lr * wd);
        mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
        vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
        pp[e] -= lr * inv_bc1 * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
      }
      *reinterpret_cast<f32x4*>(p + i4) = pp;
      *reinterpret_cast<f32x4*>(m + i4) = mm;
      *reinterpret_cast<f32x4*>(v + i4) = vv;
      if (mirror) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = (bf16)pp[e];
        *reinterpret_cast<bf16x4*>(mirror + i4) = q;  // i4 % 4 == 0: 8-byte aligned
      }
    } else {
      for (int64_t i = i4; i < n && i < i4 + 4; ++i) {
        bool sk = false;
        for (int k = 0; k < skip.n; ++k) sk = sk || (i >= skip.r[k][0] && i < skip.r[k][1]);
        if (sk) continue;
        float pp = p[i] * (1.0f - lr * wd);
        const float gg = g[i];
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        pp -= lr * inv_bc1 * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (mirror) mirror[i] = (bf16)pp;
      }
    }
  }
}

// hyper (device, 4 x 32 bit): [0] lr (float), [1] step count (int32), [2] 1 / (1 - beta1^step), [3] 1 / sqrt(1 - beta2^step).
// One thread advances the step and refreshes the two bias corrections: a captured HIP graph of a training step replays
// with the right step number, and the learning rate is a device word the host may rewrite between replays.
__global__ void adamw_tick_kernel(float* __restrict__ hyper, const float b1, const float b2) {
  const int step = __float_as_int(hyper[1]) + 1;
  hyper[1] = __int_as_float(step);
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  hyper[2] = (float)(1.0 / bc1);
  hyper[3] = (float)(1.0 / sqrt(bc2));
}

extern "C" int ctu_adamw_tick(float* hyper_dev, float beta1, float beta2, ctu_stream_t stream) {
  CTU_REQUIRE(hyper_dev, "adamw_tick: null pointer");
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper_dev, beta1, beta2);
  return ctu_check_launch("adamw_tick");
}

extern "C" int ctu_adamw(float* p, const float* g, float* m, float* v, void* mirror_bf16, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int32_t step, const int64_t* skip_host, int32_t n_skip,
                         const float* hyper_dev, ctu_stream_t stream) {
  CTU_REQUIRE(p && g && m && v && n > 0 && (step > 0 || hyper_dev), "adamw: bad args");
  if (hyper_dev) step = 1;  // (lr and the bias corrections come from the device words)
  CTU_REQUIRE(n_skip >= 0 && n_skip <= 16 && (n_skip == 0 || skip_host), "adamw: at most 16 skip ranges");
  CTU_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
              "adamw: buffers must be 16-byte aligned");
  SkipRanges sk;
  sk.n = n_skip;
  for (int k = 0; k < n_skip; ++k) { sk.r[k][0] = skip_host[2 * k]; sk.r[k][1] = skip_host[2 * k + 1]; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     reinterpret_cast<bf16*>(mirror_bf16), n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), sk, hyper_dev);
  return ctu_check_launch("adamw");
}
