// elementwise.hip — HBM-bound streaming kernels of the EAST/TRBA path for gfx950.
// All kernels move 16 bytes per lane where the layout allows and are grid-stride
// with grids capped at 256 CUs x 8 blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msocr.h"

// Clear any stale (sticky) HIP error left by earlier runtime calls of the host process before a launch,
// so that the status read back after it belongs to this launch.
#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

static inline int grid_for(long work, int block) {
  long g = (work + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH)

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<uint16_t*>(&b);
}
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

// ---------------------------------------------------------------------------------------------
// Normalise u8 HWC pages/crops into NHWC with C padded 3->4 inside a zero canvas [Hp][Wp]
// (image placed at (pad_t, pad_l)); the zero border IS the convolution padding of the stem.
//   mode 0 (EAST, infer.py:127-132,305): ToTensor then Normalize -> (x/255 - .5)/.5, two f32 roundings
//   mode 1 (TRBA, transforms.py:185-193): A.Normalize(.5,.5,255) -> (x - 127.5) * f32(1/127.5)
template <typename T, int CP>
__global__ void normalize_u8_kernel(const uint8_t* __restrict__ src, int N, int H, int W, int pad_t, int pad_l, int Hp, int Wp,
                                    int mode, T* __restrict__ dst) {
  const long total = (long)N * Hp * Wp;
  const float inv = 1.0f / 127.5f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    const long t = i / Wp;
    const int yp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int x = xp - pad_l, y = yp - pad_t;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
      const uint8_t* s = src + (((long)n * H + y) * W + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (mode == 0) {
          const float f = (float)s[c] / 255.0f;
          v[c] = (f - 0.5f) / 0.5f;
        } else {
          v[c] = ((float)s[c] - 127.5f) * inv;
        }
      }
    }
    if (sizeof(T) == 4) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + i * CP) = o;
      if (CP == 8) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(dst + i * CP + 4) = z;
      }
    } else {
      u32x2 o = {(uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16), (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16)};
      *reinterpret_cast<u32x2*>(dst + i * CP) = o;
      if (CP == 8) {
        u32x2 z = {0u, 0u};
        *reinterpret_cast<u32x2*>(dst + i * CP + 4) = z;
      }
    }
  }
}

extern "C" int msocr_normalize_u8(const uint8_t* src, int N, int H, int W, int pad_t, int pad_l, int Hp, int Wp, int cpad, int mode,
                                  int dtype, void* out, void* stream) {
  if (!src || !out || N <= 0 || H <= 0 || W <= 0 || pad_t < 0 || pad_l < 0 || Hp < H + pad_t || Wp < W + pad_l) return MSOCR_E_ARG;
  if ((mode != 0 && mode != 1) || (cpad != 4 && cpad != 8)) return MSOCR_E_ARG;
  const long total = (long)N * Hp * Wp;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(grid_for(total, 256)), b(256);
  if (dtype == MSOCR_F32 && cpad == 4)
    MSOCR_LAUNCH((normalize_u8_kernel<float, 4>), g, b, 0, s, src, N, H, W, pad_t, pad_l, Hp, Wp, mode, (float*)out);
  else if (dtype == MSOCR_F32)
    MSOCR_LAUNCH((normalize_u8_kernel<float, 8>), g, b, 0, s, src, N, H, W, pad_t, pad_l, Hp, Wp, mode, (float*)out);
  else if (dtype == MSOCR_BF16 && cpad == 4)
    MSOCR_LAUNCH((normalize_u8_kernel<uint16_t, 4>), g, b, 0, s, src, N, H, W, pad_t, pad_l, Hp, Wp, mode, (uint16_t*)out);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH((normalize_u8_kernel<uint16_t, 8>), g, b, 0, s, src, N, H, W, pad_t, pad_l, Hp, Wp, mode, (uint16_t*)out);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// cv2.resize INTER_LINEAR, u8, 3 channels (OpenCV resize.cpp fixed-point: 11-bit coefficients,
// horizontal pass in int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).
__device__ __forceinline__ int rint_short(float v) {
  int r = __float2int_rn(v);
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}
__global__ void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int N, int sh, int sw, uint8_t* __restrict__ dst,
                                        int dh, int dw, double scale_x, double scale_y, int area2x) {
  const long total = (long)N * dh * dw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int dx = (int)(i % dw);
    const long t = i / dw;
    const int dy = (int)(t % dh);
    const int n = (int)(t / dh);
    const uint8_t* s = src + (long)n * sh * sw * 3;
    uint8_t* d = dst + i * 3;
    if (area2x) {  // exact 2x decimation: INTER_LINEAR == INTER_AREA fast path
      const uint8_t* p0 = s + ((long)(2 * dy) * sw + 2 * dx) * 3;
      const uint8_t* p1 = p0 + (long)sw * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c] = (uint8_t)((p0[c] + p0[c + 3] + p1[c] + p1[c + 3] + 2) >> 2);
      continue;
    }
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
    const int a0 = rint_short((1.f - fx) * 2048.f), a1 = rint_short(fx * 2048.f);
    const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = rint_short((1.f - fy) * 2048.f), b1 = rint_short(fy * 2048.f);
    const int r0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
    const int r1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
    const uint8_t* q0 = s + (long)r0 * sw * 3;
    const uint8_t* q1 = s + (long)r1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int S0 = q0[sx * 3 + c] * a0 + q0[sx1 * 3 + c] * a1;
      const int S1 = q1[sx * 3 + c] * a0 + q1[sx1 * 3 + c] * a1;
      int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

extern "C" int msocr_resize_linear_u8(const uint8_t* src, int N, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream) {
  if (!src || !dst || N <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return MSOCR_E_ARG;
  const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
  const int area2x = (sw == 2 * dw && sh == 2 * dh) ? 1 : 0;
  const long total = (long)N * dh * dw;
  MSOCR_LAUNCH(resize_linear_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, N, sh, sw, dst,
                     dh, dw, scale_x, scale_y, area2x);
  return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d on NHWC, vectorised over channels (4 per lane).
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ in, int N, int H, int W, int C, long in_ld, int k, int s, int p,
                               T* __restrict__ out, int Ho, int Wo, long out_ld) {
  const int C4 = C / 4;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int kh = 0; kh < k; ++kh) {
      const int hi = ho * s - p + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int wi = wo * s - p + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const T* q = in + (((long)n * H + hi) * W + wi) * in_ld + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], ld<T>(q + e));
      }
    }
    T* o = out + (((long)n * Ho + ho) * Wo + wo) * out_ld + c4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) st<T>(o + e, m[e]);
  }
}

extern "C" int msocr_maxpool2d(const void* in, int N, int H, int W, int C, int64_t in_ld, int k, int s, int p, int dtype,
                               void* out, int Ho, int Wo, int64_t out_ld, void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || C % 4 || k <= 0 || s <= 0 || Ho <= 0 || Wo <= 0) return MSOCR_E_ARG;
  if ((Ho - 1) * s - p >= H || (Wo - 1) * s - p >= W) return MSOCR_E_ARG;
  const long total = (long)N * Ho * Wo * (C / 4);
  hipStream_t st_ = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(maxpool_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st_, (const float*)in, N, H, W, C, (long)in_ld, k, s, p,
                       (float*)out, Ho, Wo, (long)out_ld);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(maxpool_kernel<uint16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st_, (const uint16_t*)in, N, H, W, C, (long)in_ld, k,
                       s, p, (uint16_t*)out, Ho, Wo, (long)out_ld);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// Bilinear x2, align_corners=False (east.py:87-91).  PyTorch's CPU kernel computes, per axis,
// src = (dst+0.5)*0.5-0.5 clamped at 0, i0=floor, lambda1 = src-i0, lambda0 = 1-lambda1, and
// out = w_y0*(w_x0*a + w_x1*b) + w_y1*(w_x0*c + w_x1*d).  Weights are exactly .25/.75 (or 0/1 at borders).
template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ in, int N, int H, int W, int C, long in_ld, T* __restrict__ out, long out_ld) {
  const int C4 = C / 4, Ho = 2 * H, Wo = 2 * W;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float sy = (yo + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = (xo + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
    const T* base = in + (long)n * H * W * in_ld + c4 * 4;
    const T* pa = base + ((long)y0 * W + x0) * in_ld;
    const T* pb = base + ((long)y0 * W + x1) * in_ld;
    const T* pc = base + ((long)y1 * W + x0) * in_ld;
    const T* pd = base + ((long)y1 * W + x1) * in_ld;
    T* o = out + (((long)n * Ho + yo) * Wo + xo) * out_ld + c4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = ly0 * (lx0 * ld<T>(pa + e) + lx1 * ld<T>(pb + e)) + ly1 * (lx0 * ld<T>(pc + e) + lx1 * ld<T>(pd + e));
      st<T>(o + e, v);
    }
  }
}

extern "C" int msocr_upsample2x_bilinear(const void* in, int N, int H, int W, int C, int64_t in_ld, int dtype, void* out,
                                         int64_t out_ld, void* stream) {
  if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || in_ld < C || out_ld < C) return MSOCR_E_ARG;
  const long total = (long)N * 4 * H * W * (C / 4);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(upsample2x_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)in, N, H, W, C, (long)in_ld,
                       (float*)out, (long)out_ld);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(upsample2x_kernel<uint16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const uint16_t*)in, N, H, W, C,
                       (long)in_ld, (uint16_t*)out, (long)out_ld);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// OutputHead (east.py:96-105): 32 -> 1 (+sigmoid) and 32 -> 8, one pixel per lane.
template <typename T>
__global__ void east_head_kernel(const T* __restrict__ h1, long npix, long in_ld, const float* __restrict__ w9,
                                 const float* __restrict__ b9, float* __restrict__ score, float* __restrict__ geo) {
  __shared__ float sw[9 * 32 + 9];
  for (int i = threadIdx.x; i < 9 * 32 + 9; i += blockDim.x) sw[i] = i < 288 ? w9[i] : b9[i - 288];
  __syncthreads();
  for (long px = blockIdx.x * (long)blockDim.x + threadIdx.x; px < npix; px += (long)gridDim.x * blockDim.x) {
    float x[32];
    const T* p = h1 + px * in_ld;
    if (sizeof(T) == 4) {
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        x[c] = v[0]; x[c + 1] = v[1]; x[c + 2] = v[2]; x[c + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int c = 0; c < 32; c += 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[c + e] = bf2f((uint16_t)(v[e >> 1] >> ((e & 1) * 16)));
      }
    }
    float o[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) a = fmaf(sw[k * 32 + c], x[c], a);
      o[k] = a + sw[288 + k];
    }
    score[px] = 1.f / (1.f + expf(-o[0]));
    f32x4 g0 = {o[1], o[2], o[3], o[4]}, g1 = {o[5], o[6], o[7], o[8]};
    *reinterpret_cast<f32x4*>(geo + px * 8) = g0;
    *reinterpret_cast<f32x4*>(geo + px * 8 + 4) = g1;
  }
}

extern "C" int msocr_east_head(const void* h1, int64_t npix, int64_t in_ld, int dtype, const float* w9, const float* b9,
                               float* score_out, float* geo_out, void* stream) {
  if (!h1 || !w9 || !b9 || !score_out || !geo_out || npix <= 0 || in_ld < 32) return MSOCR_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(east_head_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, s, (const float*)h1, (long)npix, (long)in_ld, w9, b9,
                       score_out, geo_out);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(east_head_kernel<uint16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, s, (const uint16_t*)h1, (long)npix, (long)in_ld, w9,
                       b9, score_out, geo_out);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// layout helpers (test / API boundary only): NCHW f32 <-> NHWC dtype
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int N, int C, int H, int W, T* __restrict__ out, long out_ld) {
  const long total = (long)N * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    st<T>(out + (((long)n * H + y) * W + x) * out_ld + c, in[(((long)n * C + c) * H + y) * W + x]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, int N, int C, int H, int W, long in_ld, float* __restrict__ out) {
  const long total = (long)N * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long t = i / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    out[i] = ld<T>(in + (((long)n * H + y) * W + x) * in_ld + c);
  }
}
extern "C" int msocr_nchw_f32_to_nhwc(const float* in, int N, int C, int H, int W, int dtype, void* out, int64_t out_ld, void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || out_ld < C) return MSOCR_E_ARG;
  const long total = (long)N * C * H * W;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(nchw_to_nhwc_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, in, N, C, H, W, (float*)out, (long)out_ld);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(nchw_to_nhwc_kernel<uint16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, in, N, C, H, W, (uint16_t*)out, (long)out_ld);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}
extern "C" int msocr_nhwc_to_nchw_f32(const void* in, int N, int C, int H, int W, int64_t in_ld, int dtype, float* out, void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || in_ld < C) return MSOCR_E_ARG;
  const long total = (long)N * C * H * W;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)in, N, C, H, W, (long)in_ld, out);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(nhwc_to_nchw_kernel<uint16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const uint16_t*)in, N, C, H, W, (long)in_ld, out);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

extern "C" const char* msocr_version(void) { return "msocr 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------
// Crop + ResizeAndPadA on the device (recognizers/_trba/data/transforms.py:85-120 applied to the clamped
// AABB views of _pipeline.py:204-221): one workgroup per crop, one canvas pixel (3 channels) per thread
// iteration.  cv2.resize semantics restated from OpenCV (parity unpinned, cv2 absent): INTER_AREA =
// float area tables accumulated in table order (x inside y), integer-scale fast path; INTER_LINEAR =
// 11-bit fixed point.  Descriptor per crop (8 x int32): page, x1, y1, x2, y2, new_w, new_h, y0.
__device__ __forceinline__ uint8_t sat_u8_rint(float v) {
  const float r = rintf(v);
  return (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

struct AreaTab {  // up to 3 kinds of entries for one destination index: head partial, full run, tail partial
  int s_head, s1, s2;
  float a_head, a_full, a_tail;
  bool has_head, has_tail;
};
__device__ __forceinline__ AreaTab area_tab(int d, double scale, int ssize) {
  AreaTab t;
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < ssize - 1 ? s2 : ssize - 1;
  s1 = s1 < s2 ? s1 : s2;
  t.s1 = s1;
  t.s2 = s2;
  t.has_head = (s1 - f1) > 1e-3;
  t.s_head = s1 - 1;
  t.a_head = (float)((s1 - f1) / cell);
  t.a_full = (float)(1.0 / cell);
  t.has_tail = (f2 - s2) > 1e-3;
  t.a_tail = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
  return t;
}

__global__ __launch_bounds__(256) void crop_resize_pad_kernel(const uint8_t* __restrict__ pages, int N, int H, int W,
                                                               const int32_t* __restrict__ desc, int img_h, int img_w,
                                                               uint8_t* __restrict__ out) {
  const int m = blockIdx.x;
  const int32_t* d = desc + (long)m * 8;
  const int pg = d[0], x1 = d[1], y1 = d[2], x2 = d[3], y2 = d[4], nw = d[5], nh = d[6], y0 = d[7];
  // the same bounds the host wrapper checks (descriptors written by msocr_reading_order_crops never fail them)
  if (pg < 0 || pg >= N || x1 < 0 || y1 < 0 || x2 > W || y2 > H || x2 <= x1 || y2 <= y1 || nw < 1 || nw > img_w || nh < 1 ||
      nh > img_h || y0 < 0 || y0 + nh > img_h) {
    for (int p = threadIdx.x; p < img_h * img_w * 3; p += blockDim.x) out[(long)m * img_h * img_w * 3 + p] = 255;
    return;
  }
  const int sw = x2 - x1, sh = y2 - y1;
  const uint8_t* src = pages + ((long)pg * H + y1) * W * 3 + (long)x1 * 3;
  const long rs = (long)W * 3;  // source row stride in bytes
  uint8_t* o = out + (long)m * img_h * img_w * 3;
  const bool shrink = nh < sh || nw < sw;
  const bool copy = nw == sw && nh == sh;
  const bool area = shrink && !copy && !(nw > sw || nh > sh);
  const bool area_int = area && (sw % nw == 0) && (sh % nh == 0);
  const double scale_x = (double)sw / (double)nw, scale_y = (double)sh / (double)nh;
  const double lin_sx = 1.0 / ((double)nw / (double)sw), lin_sy = 1.0 / ((double)nh / (double)sh);
  for (int p = threadIdx.x; p < img_h * img_w; p += blockDim.x) {
    const int cy = p / img_w, cx = p - cy * img_w;
    const int dy = cy - y0, dx = cx;
    uint8_t r[3] = {255, 255, 255};
    if (dy >= 0 && dy < nh && dx < nw) {
      if (copy) {
        const uint8_t* s = src + dy * rs + dx * 3;
        r[0] = s[0]; r[1] = s[1]; r[2] = s[2];
      } else if (area_int) {
        const int kx = sw / nw, ky = sh / nh;
        int acc[3] = {0, 0, 0};
        for (int yy = 0; yy < ky; ++yy) {
          const uint8_t* s = src + (long)(dy * ky + yy) * rs + (long)dx * kx * 3;
          for (int xx = 0; xx < kx; ++xx)
            for (int c = 0; c < 3; ++c) acc[c] += s[xx * 3 + c];
        }
        if (kx == 2 && ky == 2) {
          for (int c = 0; c < 3; ++c) r[c] = (uint8_t)((acc[c] + 2) >> 2);
        } else {
          const float inv = (float)(1.0 / (double)(kx * ky));
          for (int c = 0; c < 3; ++c) r[c] = sat_u8_rint((float)acc[c] * inv);
        }
      } else if (area) {
        const AreaTab tx = area_tab(dx, scale_x, sw), ty = area_tab(dy, scale_y, sh);
        float sum[3] = {0.f, 0.f, 0.f};
        auto row = [&](int sy, float beta) {
          const uint8_t* s = src + (long)sy * rs;
          float buf[3] = {0.f, 0.f, 0.f};
          if (tx.has_head)
            for (int c = 0; c < 3; ++c) buf[c] += (float)s[tx.s_head * 3 + c] * tx.a_head;
          for (int sx = tx.s1; sx < tx.s2; ++sx)
            for (int c = 0; c < 3; ++c) buf[c] += (float)s[sx * 3 + c] * tx.a_full;
          if (tx.has_tail)
            for (int c = 0; c < 3; ++c) buf[c] += (float)s[tx.s2 * 3 + c] * tx.a_tail;
          for (int c = 0; c < 3; ++c) sum[c] += buf[c] * beta;
        };
        if (ty.has_head) row(ty.s_head, ty.a_head);
        for (int sy = ty.s1; sy < ty.s2; ++sy) row(sy, ty.a_full);
        if (ty.has_tail) row(ty.s2, ty.a_tail);
        for (int c = 0; c < 3; ++c) r[c] = sat_u8_rint(sum[c]);
      } else {  // INTER_LINEAR fixed point
        float fx = (float)((dx + 0.5) * lin_sx - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        const int a0 = rint_short((1.f - fx) * 2048.f), a1 = rint_short(fx * 2048.f);
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
        float fy = (float)((dy + 0.5) * lin_sy - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        const int b0 = rint_short((1.f - fy) * 2048.f), b1 = rint_short(fy * 2048.f);
        const int r0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        const int r1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t* q0 = src + (long)r0 * rs;
        const uint8_t* q1 = src + (long)r1 * rs;
        for (int c = 0; c < 3; ++c) {
          const int S0 = q0[sx * 3 + c] * a0 + q0[sx1 * 3 + c] * a1;
          const int S1 = q1[sx * 3 + c] * a0 + q1[sx1 * 3 + c] * a1;
          const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
          r[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
    o[(long)p * 3] = r[0];
    o[(long)p * 3 + 1] = r[1];
    o[(long)p * 3 + 2] = r[2];
  }
}

extern "C" int msocr_crop_resize_pad(const uint8_t* pages, int N, int H, int W, const int32_t* desc_dev, const int32_t* desc_host,
                                     int M, int img_h, int img_w, uint8_t* canvases, void* stream) {
  if (!pages || !desc_dev || !canvases || N <= 0 || H <= 0 || W <= 0 || M <= 0 || img_h <= 0 || img_w <= 0) return MSOCR_E_ARG;
  for (int m = 0; desc_host && m < M; ++m) {  // validate on the host what the kernel's indexing assumes (the kernel re-checks)
    const int32_t* d = desc_host + (long)m * 8;
    if (d[0] < 0 || d[0] >= N || d[1] < 0 || d[2] < 0 || d[3] > W || d[4] > H || d[3] <= d[1] || d[4] <= d[2]) return MSOCR_E_ARG;
    if (d[5] < 1 || d[5] > img_w || d[6] < 1 || d[6] > img_h || d[7] < 0 || d[7] + d[6] > img_h) return MSOCR_E_ARG;
  }
  MSOCR_LAUNCH(crop_resize_pad_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, pages, N, H, W, desc_dev, img_h, img_w, canvases);
  return LAUNCH_OK();
}
