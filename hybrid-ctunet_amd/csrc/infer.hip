// Inference-side callers of forward: on-device accumulation of sliding-window predictions with an
// importance map, normalisation by the accumulated weight, and the hybrid complementation of two models' logits.
//
// Replaces (reference): the accumulate / normalise lines of sliding_window_inference (trainer_CTUNet.py:538-548,
// trainer_CUNet.py:386-392) and torch.softmax / average / torch.argmax at test_CTUNet_final.py:545-551,
// trainer_CTUNet.py:287-292.  HBM-bound elementwise kernels; fp32 outputs in the reference's NCDHW layout.
#include "common.h"

struct SwArgs {
  const void* logits;       // one window: element (c, d, h, w) at logits[c*sc + d*sd + h*sh + w*sw]
  const float* imp;         // [rd][rh][rw] importance map
  float* out;               // [B][C][D][H][W] += imp * logits
  float* cnt;               // [B][D][H][W] += imp (NULL: skip - a second output shares the first one's weights)
  int64_t sc, sd, sh, sw;
  int C, rd, rh, rw, b, d0, h0, w0, D, H, W;
};

template <typename T>
__global__ __launch_bounds__(256) void sw_accumulate_kernel(const SwArgs a) {
  const int64_t nvox = (int64_t)a.rd * a.rh * a.rw;
  const int64_t total = nvox * a.C;
  const T* lg = reinterpret_cast<const T*>(a.logits);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / nvox);
    int64_t v = i - (int64_t)c * nvox;
    const int w = (int)(v % a.rw); v /= a.rw;
    const int h = (int)(v % a.rh);
    const int d = (int)(v / a.rh);
    const float wgt = a.imp[i - (int64_t)c * nvox];
    const float x = (float)lg[c * a.sc + d * a.sd + h * a.sh + w * a.sw];
    const int64_t o = (((int64_t)a.d0 + d) * a.H + a.h0 + h) * a.W + a.w0 + w;
    const int64_t S = (int64_t)a.D * a.H * a.W;
    a.out[((int64_t)a.b * a.C + c) * S + o] += wgt * x;
    if (c == 0 && a.cnt) a.cnt[(int64_t)a.b * S + o] += wgt;
  }
}

extern "C" int ctu_sw_accumulate(ctu_dtype dtype, const void* logits, int64_t sc, int64_t sd, int64_t sh, int64_t sw,
                                 const float* importance, float* out, float* count, int32_t C, int32_t rd, int32_t rh,
                                 int32_t rw, int32_t b, int32_t d0, int32_t h0, int32_t w0, int32_t D, int32_t H, int32_t W,
                                 ctu_stream_t stream) {
  CTU_REQUIRE(logits && importance && out, "sw_accumulate: null pointer");
  CTU_REQUIRE(C > 0 && rd > 0 && rh > 0 && rw > 0 && b >= 0 && d0 >= 0 && h0 >= 0 && w0 >= 0 && d0 + rd <= D &&
                  h0 + rh <= H && w0 + rw <= W,
              "sw_accumulate: window outside the volume");
  SwArgs a;
  a.logits = logits; a.imp = importance; a.out = out; a.cnt = count;
  a.sc = sc; a.sd = sd; a.sh = sh; a.sw = sw;
  a.C = C; a.rd = rd; a.rh = rh; a.rw = rw; a.b = b; a.d0 = d0; a.h0 = h0; a.w0 = w0; a.D = D; a.H = H; a.W = W;
  const unsigned grid = grid_for((int64_t)C * rd * rh * rw, 256);
  hipStream_t s = (hipStream_t)stream;
  CTU_DISPATCH(dtype, hipLaunchKernelGGL(sw_accumulate_kernel<float>, dim3(grid), dim3(256), 0, s, a),
               hipLaunchKernelGGL(sw_accumulate_kernel<bf16>, dim3(grid), dim3(256), 0, s, a));
  return ctu_check_launch("sw_accumulate");
}

__global__ __launch_bounds__(256) void sw_normalize_kernel(float* __restrict__ out, const float* __restrict__ cnt,
                                                           const int B, const int C, const int64_t S) {
  const int64_t total = (int64_t)B * C * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = i % S;
    const int64_t b = i / (S * C);
    out[i] = out[i] / cnt[b * S + s];
  }
}
extern "C" int ctu_sw_normalize(float* out, const float* count, int32_t B, int32_t C, int64_t S, ctu_stream_t stream) {
  CTU_REQUIRE(out && count && B > 0 && C > 0 && S > 0, "sw_normalize: bad args");
  hipLaunchKernelGGL(sw_normalize_kernel, dim3(grid_for((int64_t)B * C * S, 256)), dim3(256), 0, (hipStream_t)stream, out,
                     count, B, C, S);
  return ctu_check_launch("sw_normalize");
}

// labels of one case from two models' logits [C][S] (fp32): argmax of each softmax and of their average
// (first maximum wins, like torch.argmax).  C <= 32.
__global__ __launch_bounds__(256) void hybrid_argmax_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                            const int C, const int64_t S, int64_t* __restrict__ l1,
                                                            int64_t* __restrict__ l2, int64_t* __restrict__ lh) {
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
    float a[32], b[32];
    float ma = -3.0e38f, mb = -3.0e38f;
    for (int c = 0; c < C; ++c) {
      a[c] = p1[(int64_t)c * S + s];
      b[c] = p2[(int64_t)c * S + s];
      ma = fmaxf(ma, a[c]);
      mb = fmaxf(mb, b[c]);
    }
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; ++c) {
      a[c] = expf(a[c] - ma);
      b[c] = expf(b[c] - mb);
      sa += a[c];
      sb += b[c];
    }
    int ia = 0, ib = 0, ih = 0;
    float va = -1.f, vb = -1.f, vh = -1.f;
    for (int c = 0; c < C; ++c) {
      const float qa = a[c] / sa, qb = b[c] / sb, qh = (qa + qb) / 2.0f;
      if (qa > va) { va = qa; ia = c; }
      if (qb > vb) { vb = qb; ib = c; }
      if (qh > vh) { vh = qh; ih = c; }
    }
    if (l1) l1[s] = ia;
    if (l2) l2[s] = ib;
    lh[s] = ih;
  }
}
extern "C" int ctu_hybrid_argmax(const float* p1, const float* p2, int32_t C, int64_t S, int64_t* labels1, int64_t* labels2,
                                 int64_t* labels_hybrid, ctu_stream_t stream) {
  CTU_REQUIRE(p1 && p2 && labels_hybrid && C > 0 && C <= 32 && S > 0, "hybrid_argmax: bad args (C <= 32)");
  hipLaunchKernelGGL(hybrid_argmax_kernel, dim3(grid_for(S, 256)), dim3(256), 0, (hipStream_t)stream, p1, p2, C, S, labels1,
                     labels2, labels_hybrid);
  return ctu_check_launch("hybrid_argmax");
}
