// winograd.hip — 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2,3x3) in f32 for gfx950 (MI355X).
//
// Why here: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s, 1/16 of the bf16 rate, while HBM3E
// delivers 8 TB/s — so on this chip an f32 3x3 convolution is worth trading 2.25x fewer matrix FLOPs for two
// streaming transform passes.  out = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A  per 2x2 output tile:
//   1. wino_input_kernel : V[xi*4+nu][tile][c]  = (B^T d B)[xi][nu]      (HBM-bound, 16-B coalesced)
//   2. 16 GEMMs in ONE launch of conv_igemm_kernel: Mw[p][tile][co] = sum_c V[p][tile][c] * U[p][co][c]  (MFMA)
//   3. wino_output_kernel: out[2x2 tile][co] = act(A^T Mw A + bias (+ residual))  (HBM-bound)
// U = G g G^T is computed once at weight-load time in f64 (msocr_winograd_weights_host).
// The transforms only add/subtract and scale by 1/2 (exact), so the result differs from the direct f32
// convolution by rounding order only (Lavin & Gray 2016 measure F(2x2,3x3) at or below direct-conv error).
//
// Replaces nn.Conv2d(3x3, stride 1, padding 1)+BatchNorm2d(+ReLU)(+add) of
//   recognizers/_trba/model/seresnet31.py:37-45,81-89 ; detectors/_east/east.py:13-30 (conv3x3 of DecoderBlock)
//   and the stride-1 Bottleneck conv2 of torchvision ResNet-50 (detectors/_east/east.py:56-67).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "internal.h"
#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 1-D transforms of F(4,3) on the interpolation points {0, 3/2, -3/2, 2/3, -2/3, inf} (round 4; the matrices and why these points:
// the comment block in front of wino44_input_kernel).  Used on both axes of the square form and on the H axis of the tall form.
__device__ __forceinline__ void wino44_bt(const f32x4 d[6], f32x4 r[6]) {
  constexpr float k97_36 = 97.0f / 36.0f, k4_9 = 4.0f / 9.0f, k9_4 = 2.25f, k3_2 = 1.5f, k2_3 = 2.0f / 3.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r[0][e] = fmaf(-k97_36, d[2][e], d[0][e]) + d[4][e];
    r[5][e] = fmaf(-k97_36, d[3][e], d[1][e]) + d[5][e];
    const float e1 = fmaf(-k4_9, d[2][e], d[4][e]), o1 = fmaf(k3_2, d[3][e], -k2_3 * d[1][e]);
    const float e2 = fmaf(-k9_4, d[2][e], d[4][e]), o2 = fmaf(k2_3, d[3][e], -k3_2 * d[1][e]);
    r[1][e] = e1 + o1; r[2][e] = e1 - o1;
    r[3][e] = e2 + o2; r[4][e] = e2 - o2;
  }
}
__device__ __forceinline__ void wino44_at(const f32x4 m[6], f32x4 y[4]) {
  constexpr float k3_2 = 1.5f, k2_3 = 2.0f / 3.0f, k9_4 = 2.25f, k4_9 = 4.0f / 9.0f, k27_8 = 3.375f, k8_27 = 8.0f / 27.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float p12 = m[1][e] + m[2][e], d12 = m[1][e] - m[2][e], p34 = m[3][e] + m[4][e], d34 = m[3][e] - m[4][e];
    y[0][e] = (m[0][e] + p12) + p34;
    y[1][e] = fmaf(k3_2, d12, k2_3 * d34);
    y[2][e] = fmaf(k9_4, p12, k4_9 * p34);
    y[3][e] = fmaf(k27_8, d12, k8_27 * d34) + m[5][e];
  }
}


struct WinoGeom {
  int N, H, W;      // image extent (output extent is the same: stride 1, pad 1)
  int TH, TW;       // 2x2-output tiles per image
  long Mt;          // N * TH * TW
};

// ---- 1. input transform: one thread = one tile x 4 channels ------------------------------------------------
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ in, long sN, long sH, long sW, int C,
                                                          WinoGeom g, float* __restrict__ V) {
  const int cch = C >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
  const float* base = in + (long)n * sN + c;
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hi = h0 + i;
    const bool okh = (unsigned)hi < (unsigned)g.H;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wi = w0 + j;
      const bool ok = okh && (unsigned)wi < (unsigned)g.W;
      // branch-free: padded taps read the (always valid) first vector of the image row block and are masked to zero
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? base + (long)hi * sH + (long)wi * sW : base);
      d[i][j] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
  f32x4 q[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    q[0][j] = d[0][j] - d[2][j];
    q[1][j] = d[1][j] + d[2][j];
    q[2][j] = d[2][j] - d[1][j];
    q[3][j] = d[1][j] - d[3][j];
  }
  // (B^T d) B : columns likewise
  const long plane = g.Mt * (long)C;
  float* o = V + t * (long)C + c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(o + (i * 4 + 0) * plane) = q[i][0] - q[i][2];
    *reinterpret_cast<f32x4*>(o + (i * 4 + 1) * plane) = q[i][1] + q[i][2];
    *reinterpret_cast<f32x4*>(o + (i * 4 + 2) * plane) = q[i][2] - q[i][1];
    *reinterpret_cast<f32x4*>(o + (i * 4 + 3) * plane) = q[i][1] - q[i][3];
  }
}

// ---- 3. output transform: one thread = one tile x 4 output channels ---------------------------------------
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mw, int Cout, WinoGeom g,
                                                           const float* __restrict__ bias, const float* __restrict__ res, long res_ld,
                                                           int relu, float* __restrict__ out, long out_ld) {
  const int cch = Cout >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const long plane = g.Mt * (long)Cout;
  const float* mp = Mw + t * (long)Cout + c;
  f32x4 m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(mp + (i * 4 + j) * plane);
  // A^T m : rows (m0+m1+m2, m1-m2-m3)
  f32x4 s[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = (m[0][j] + m[1][j]) + m[2][j];
    s[1][j] = (m[1][j] - m[2][j]) - m[3][j];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias) b = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ho = 2 * th + i;
    if (ho >= g.H) continue;
    f32x4 y[2];
    y[0] = (s[i][0] + s[i][1]) + s[i][2];
    y[1] = (s[i][1] - s[i][2]) - s[i][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wo = 2 * tw + j;
      if (wo >= g.W) continue;
      const long pix = ((long)n * g.H + ho) * g.W + wo;
      f32x4 v = y[j] + b;
      if (res) v += *reinterpret_cast<const f32x4*>(res + pix * res_ld + c);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
  }
}

static bool wino_geom(const msocr_conv_desc* d, WinoGeom* g) {
  if (!d || d->dtype != MSOCR_F32) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin <= 0 || d->Cin % 16 || d->Cout <= 0 || d->Cout % 32) return false;
  g->N = d->N; g->H = d->H; g->W = d->W;
  g->TH = (d->H + 1) / 2; g->TW = (d->W + 1) / 2;
  g->Mt = (long)d->N * g->TH * g->TW;
  return g->Mt <= 0x7fffffffL;
}

extern "C" int64_t msocr_conv3x3_winograd_workspace_bytes(const msocr_conv_desc* d) {
  WinoGeom g;
  if (!wino_geom(d, &g)) return -1;
  return 16 * g.Mt * ((int64_t)d->Cin + d->Cout) * (int64_t)sizeof(float);
}

static int wino_check(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino_geom(d, g)) return MSOCR_E_ARG;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;
  return MSOCR_OK;
}

// The three stages of msocr_conv3x3_winograd as separate entry points (same kernels): tests and the per-kernel roofline of
// bench.py time them one by one.  V = workspace, Mw = workspace + 16 * tiles * Cin floats.
extern "C" int msocr_winograd_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream) {
  WinoGeom g;
  if (wino_check(d, &g) != MSOCR_OK || !in || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const long nb_in = (g.Mt * (d->Cin / 4) + 255) / 256;
  if (nb_in > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino_input_kernel, dim3((unsigned)nb_in), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN,
               (long)d->in_sH, (long)d->in_sW, d->Cin, g, (float*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_winograd_gemm(const msocr_conv_desc* d, const float* u_weight, void* workspace, void* stream) {
  WinoGeom g;
  if (wino_check(d, &g) != MSOCR_OK || !u_weight || !workspace) return MSOCR_E_ARG;
  float* V = (float*)workspace;
  return msocr_internal_gemm_f32_batched(V, u_weight, V + 16 * g.Mt * (long)d->Cin, g.Mt, d->Cout, d->Cin, 16, (hipStream_t)stream);
}

extern "C" int msocr_winograd_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias, const void* residual,
                                               void* out, void* stream) {
  WinoGeom g;
  if (wino_check(d, &g) != MSOCR_OK || !out || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)out | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  if (bias && ((uintptr_t)bias & 15)) return MSOCR_E_ARG;
  const long nb_out = (g.Mt * (d->Cout / 4) + 255) / 256;
  if (nb_out > 0x7fffffffL) return MSOCR_E_ARG;
  const float* Mw = (const float*)workspace + 16 * g.Mt * (long)d->Cin;
  MSOCR_LAUNCH(wino_output_kernel, dim3((unsigned)nb_out), dim3(256), 0, (hipStream_t)stream, Mw, d->Cout, g, bias,
               has_res ? (const float*)residual : nullptr, (long)d->res_ld, (d->flags & MSOCR_CONV_RELU) ? 1 : 0, (float*)out,
               (long)d->out_ld);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_conv3x3_winograd(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                                      const void* residual, void* out, void* workspace, void* stream) {
  if (!u_weight) return MSOCR_E_ARG;
  int rc = msocr_winograd_input_transform(d, in, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  rc = msocr_winograd_gemm(d, u_weight, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  return msocr_winograd_output_transform(d, workspace, bias, residual, out, stream);
}

// =====================================================================================================================
// The TALL form: F(4,3) along H x F(2,3) along W.  A 6x4 input tile gives a 4x2 output tile through 24 transform-domain
// products: 3 multiplies per output instead of 4 (direct: 9), V / Mw expanded 3x instead of 4x.  Only the H axis takes the
// 6-point transform: on the textbook points {0, +-1, +-2, inf} its constants (4, 5, 2, 8; 1/6, 1/24 in the weights) grew the f32
// rounding error of the layer by ~2.5x rms over F(2x2) (measured against f64: DESIGN.md section 4), where the 6x6 form would grow
// it 6x.  Since round 4 the H axis runs on the points {0, +-3/2, +-2/3, inf} (wino44_bt / wino44_at above): half that error, and
// the 6x6 form on them (wino44_*, below) comes out where this form used to be.  ops.conv2d() picks the tall form for the Cin = 64
// layers (fused kernels) and where the square form does not pay; the 2x2 form when 24 * ceil(H/4) >= 16 * ceil(H/2).
//   V[(xi*4+nu)][tile][c], xi = 0..5 (H axis), nu = 0..3 (W axis); Mw likewise; U from msocr_winograd42_weights_host.
// =====================================================================================================================
__global__ __launch_bounds__(256) void wino42_input_kernel(const float* __restrict__ in, long sN, long sH, long sW, int C,
                                                            WinoGeom g, float* __restrict__ V) {
  const int cch = C >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const int h0 = 4 * th - 1, w0 = 2 * tw - 1;
  const float* base = in + (long)n * sN + c;
  // W axis first (B4^T per row as the row is loaded), then the 6-point H transform per column
  f32x4 q[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int hi = h0 + i;
    const bool okh = (unsigned)hi < (unsigned)g.H;
    f32x4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wi = w0 + j;
      const bool ok = okh && (unsigned)wi < (unsigned)g.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? base + (long)hi * sH + (long)wi * sW : base);
      d[j] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    q[i][0] = d[0] - d[2];
    q[i][1] = d[1] + d[2];
    q[i][2] = d[2] - d[1];
    q[i][3] = d[1] - d[3];
  }
  const long plane = g.Mt * (long)C;
  float* o = V + t * (long)C + c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 col[6] = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j]};
    f32x4 v[6];
    wino44_bt(col, v);   // H axis on the accuracy-chosen points (round 4: half the rounding error of {0, +-1, +-2})
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(o + (i * 4 + j) * plane) = v[i];
  }
}

__global__ __launch_bounds__(256) void wino42_output_kernel(const float* __restrict__ Mw, int Cout, WinoGeom g,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             long res_ld, int relu, float* __restrict__ out, long out_ld) {
  const int cch = Cout >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const long plane = g.Mt * (long)Cout;
  const float* mp = Mw + t * (long)Cout + c;
  // H axis first (A6^T per column as the column is loaded), then A4^T along W
  f32x4 s[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(mp + (i * 4 + j) * plane);
    f32x4 y6[4];
    wino44_at(m, y6);
    s[0][j] = y6[0]; s[1][j] = y6[1]; s[2][j] = y6[2]; s[3][j] = y6[3];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias) b = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ho = 4 * th + i;
    if (ho >= g.H) continue;
    f32x4 y[2];
    y[0] = (s[i][0] + s[i][1]) + s[i][2];
    y[1] = (s[i][1] - s[i][2]) - s[i][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wo = 2 * tw + j;
      if (wo >= g.W) continue;
      const long pix = ((long)n * g.H + ho) * g.W + wo;
      f32x4 v = y[j] + b;
      if (res) v += *reinterpret_cast<const f32x4*>(res + pix * res_ld + c);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
  }
}

static bool wino42_geom(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino_geom(d, g)) return false;
  g->TH = (d->H + 3) / 4;
  g->Mt = (long)d->N * g->TH * g->TW;
  return 24 * g->Mt * (long)(d->Cin > d->Cout ? d->Cin : d->Cout) <= 0x3fffffffffffL;
}

extern "C" int64_t msocr_conv3x3_winograd42_workspace_bytes(const msocr_conv_desc* d) {
  WinoGeom g;
  if (!wino42_geom(d, &g)) return -1;
  return 24 * g.Mt * ((int64_t)d->Cin + d->Cout) * (int64_t)sizeof(float);
}

static int wino42_check(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino42_geom(d, g)) return MSOCR_E_ARG;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;
  return MSOCR_OK;
}

extern "C" int msocr_winograd42_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream) {
  WinoGeom g;
  if (wino42_check(d, &g) != MSOCR_OK || !in || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const long nb_in = (g.Mt * (d->Cin / 4) + 255) / 256;
  if (nb_in > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino42_input_kernel, dim3((unsigned)nb_in), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN,
               (long)d->in_sH, (long)d->in_sW, d->Cin, g, (float*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_winograd42_gemm(const msocr_conv_desc* d, const float* u_weight, void* workspace, void* stream) {
  WinoGeom g;
  if (wino42_check(d, &g) != MSOCR_OK || !u_weight || !workspace) return MSOCR_E_ARG;
  float* V = (float*)workspace;
  return msocr_internal_gemm_f32_batched(V, u_weight, V + 24 * g.Mt * (long)d->Cin, g.Mt, d->Cout, d->Cin, 24, (hipStream_t)stream);
}

// The 24 transform-domain GEMMs on the bf16 matrix pipes with exactly split operands (conv_split.hip): V split in registers,
// U given as three bf16 planes [3][24][Cout][Cin] (msocr_split_bf16x3_host of msocr_winograd42_weights_host's output).
extern "C" int msocr_winograd42_gemm_split(const msocr_conv_desc* d, const void* u_planes, void* workspace, void* stream) {
  WinoGeom g;
  if (wino42_check(d, &g) != MSOCR_OK || !u_planes || !workspace || d->Cin % 32 || d->Cout % 64) return MSOCR_E_ARG;
  float* V = (float*)workspace;
  return msocr_internal_gemm_split_batched(V, (const uint16_t*)u_planes, V + 24 * g.Mt * (long)d->Cin, g.Mt, d->Cout, d->Cin, 24,
                                           (hipStream_t)stream);
}

extern "C" int msocr_winograd42_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias,
                                                 const void* residual, void* out, void* stream) {
  WinoGeom g;
  if (wino42_check(d, &g) != MSOCR_OK || !out || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)out | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  if (bias && ((uintptr_t)bias & 15)) return MSOCR_E_ARG;
  const long nb_out = (g.Mt * (d->Cout / 4) + 255) / 256;
  if (nb_out > 0x7fffffffL) return MSOCR_E_ARG;
  const float* Mw = (const float*)workspace + 24 * g.Mt * (long)d->Cin;
  MSOCR_LAUNCH(wino42_output_kernel, dim3((unsigned)nb_out), dim3(256), 0, (hipStream_t)stream, Mw, d->Cout, g, bias,
               has_res ? (const float*)residual : nullptr, (long)d->res_ld, (d->flags & MSOCR_CONV_RELU) ? 1 : 0, (float*)out,
               (long)d->out_ld);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_conv3x3_winograd42(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                                        const void* residual, void* out, void* workspace, void* stream) {
  if (!u_weight) return MSOCR_E_ARG;
  int rc = msocr_winograd42_input_transform(d, in, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  rc = msocr_winograd42_gemm(d, u_weight, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  return msocr_winograd42_output_transform(d, workspace, bias, residual, out, stream);
}

extern "C" int msocr_conv3x3_winograd42_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                              const void* residual, void* out, void* workspace, void* stream) {
  if (!u_planes) return MSOCR_E_ARG;
  int rc = msocr_winograd42_input_transform(d, in, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  rc = msocr_winograd42_gemm_split(d, u_planes, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  return msocr_winograd42_output_transform(d, workspace, bias, residual, out, stream);
}

// =====================================================================================================================
// F(4,3) x F(4,3) on ACCURACY-CHOSEN interpolation points (round 4): 36 transform points per 4 x 4 outputs = 2.25 multiplies and
// workspace words per output instead of the tall form's 3.  With the textbook points {0, +-1, +-2, inf} the square form's rounding
// error is 4.7x the tall form's and cannot pass the f64 arbitration of the tolerances (tests/test_gpu_f64.py); with
// {0, +-3/2, +-2/3, inf} — reciprocal pairs keep the Vandermonde entries within [8/27, 27/8] — an f32 simulation of the whole
// pipeline of transforms measures 1.05x the tall form's error (and 0.5x for the tall form itself on these points;
// tools/winograd_points.py, profiles/r04_winograd_points.txt).  Matrices (Cook-Toom, wincnn scaling: G carries 1 / prod(a_j - a_l)):
//   B^T d:  r0 = d0 - 97/36 d2 + d4          r5 = d1 - 97/36 d3 + d5
//           e1 = d4 - 4/9 d2,  o1 = 3/2 d3 - 2/3 d1:   r1 = e1 + o1,  r2 = e1 - o1        (points +-3/2)
//           e2 = d4 - 9/4 d2,  o2 = 2/3 d3 - 3/2 d1:   r3 = e2 + o2,  r4 = e2 - o2        (points +-2/3)
//   A^T m:  y0 = m0 + (m1 + m2) + (m3 + m4)             y1 = 3/2 (m1 - m2) + 2/3 (m3 - m4)
//           y2 = 9/4 (m1 + m2) + 4/9 (m3 + m4)          y3 = 27/8 (m1 - m2) + 8/27 (m3 - m4) + m5
//   G (host, f64): rows [1,0,0], [8,+-12,18]/65, [-81/2,-+27,-18]/65, [0,0,1].
//   V[(xi*6+nu)][tile][c], xi = 0..5 (H axis), nu = 0..5 (W axis); Mw likewise; U from msocr_winograd44_weights_host.
// Reference layers: the 3x3 / stride 1 / pad 1 convolutions of seresnet31.py:37-67 and of torchvision's Bottleneck (east.py:13-30).
// =====================================================================================================================
__global__ __launch_bounds__(256) void wino44_input_kernel(const float* __restrict__ in, long sN, long sH, long sW, int C,
                                                            WinoGeom g, float* __restrict__ V) {
  const int cch = C >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const int h0 = 4 * th - 1, w0 = 4 * tw - 1;
  const float* base = in + (long)n * sN + c;
  // W axis first (B^T per row as the row is loaded), then the H transform per column
  f32x4 q[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int hi = h0 + i;
    const bool okh = (unsigned)hi < (unsigned)g.H;
    f32x4 d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int wi = w0 + j;
      const bool ok = okh && (unsigned)wi < (unsigned)g.W;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? base + (long)hi * sH + (long)wi * sW : base);
      d[j] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    wino44_bt(d, q[i]);
  }
  const long plane = g.Mt * (long)C;
  float* o = V + t * (long)C + c;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const f32x4 col[6] = {q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j]};
    f32x4 v[6];
    wino44_bt(col, v);
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(o + (i * 6 + j) * plane) = v[i];
  }
}

__global__ __launch_bounds__(256) void wino44_output_kernel(const float* __restrict__ Mw, int Cout, WinoGeom g,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             long res_ld, int relu, float* __restrict__ out, long out_ld) {
  const int cch = Cout >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.Mt) return;
  const int tw = (int)(t % g.TW);
  const long r = t / g.TW;
  const int th = (int)(r % g.TH);
  const int n = (int)(r / g.TH);
  const long plane = g.Mt * (long)Cout;
  const float* mp = Mw + t * (long)Cout + c;
  // H axis first (A^T per column as the column is loaded), then A^T along W
  f32x4 s[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    f32x4 m[6], y[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(mp + (i * 6 + j) * plane);
    wino44_at(m, y);
#pragma unroll
    for (int a = 0; a < 4; ++a) s[a][j] = y[a];
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias) b = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int ho = 4 * th + a;
    if (ho >= g.H) continue;
    f32x4 y[4];
    wino44_at(s[a], y);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wo = 4 * tw + j;
      if (wo >= g.W) continue;
      const long pix = ((long)n * g.H + ho) * g.W + wo;
      f32x4 v = y[j] + b;
      if (res) v += *reinterpret_cast<const f32x4*>(res + pix * res_ld + c);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
  }
}

static bool wino44_geom(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino_geom(d, g)) return false;
  g->TH = (d->H + 3) / 4;
  g->TW = (d->W + 3) / 4;
  g->Mt = (long)d->N * g->TH * g->TW;
  return 36 * g->Mt * (long)(d->Cin > d->Cout ? d->Cin : d->Cout) <= 0x3fffffffffffL;
}

extern "C" int64_t msocr_conv3x3_winograd44_workspace_bytes(const msocr_conv_desc* d) {
  WinoGeom g;
  if (!wino44_geom(d, &g)) return -1;
  return 36 * g.Mt * ((int64_t)d->Cin + d->Cout) * (int64_t)sizeof(float);
}

static int wino44_check(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino44_geom(d, g)) return MSOCR_E_ARG;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;
  return MSOCR_OK;
}

extern "C" int msocr_winograd44_input_transform(const msocr_conv_desc* d, const void* in, void* workspace, void* stream) {
  WinoGeom g;
  if (wino44_check(d, &g) != MSOCR_OK || !in || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const long nb_in = (g.Mt * (d->Cin / 4) + 255) / 256;
  if (nb_in > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino44_input_kernel, dim3((unsigned)nb_in), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN,
               (long)d->in_sH, (long)d->in_sW, d->Cin, g, (float*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

// The 36 transform-domain GEMMs on the bf16 matrix pipes with exactly split operands (conv_split_pp.hip): V split in registers,
// U given as K-tile-major bf16 planes [3][36][Cin/32][Cout][32] of msocr_winograd44_weights_host's output.
extern "C" int msocr_winograd44_gemm_split(const msocr_conv_desc* d, const void* u_planes, void* workspace, void* stream) {
  WinoGeom g;
  if (wino44_check(d, &g) != MSOCR_OK || !u_planes || !workspace || d->Cin % 32 || d->Cout % 64) return MSOCR_E_ARG;
  float* V = (float*)workspace;
  return msocr_internal_gemm_split_batched(V, (const uint16_t*)u_planes, V + 36 * g.Mt * (long)d->Cin, g.Mt, d->Cout, d->Cin, 36,
                                           (hipStream_t)stream);
}

extern "C" int msocr_winograd44_output_transform(const msocr_conv_desc* d, const void* workspace, const float* bias,
                                                 const void* residual, void* out, void* stream) {
  WinoGeom g;
  if (wino44_check(d, &g) != MSOCR_OK || !out || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)out | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  if (bias && ((uintptr_t)bias & 15)) return MSOCR_E_ARG;
  const long nb_out = (g.Mt * (d->Cout / 4) + 255) / 256;
  if (nb_out > 0x7fffffffL) return MSOCR_E_ARG;
  const float* Mw = (const float*)workspace + 36 * g.Mt * (long)d->Cin;
  MSOCR_LAUNCH(wino44_output_kernel, dim3((unsigned)nb_out), dim3(256), 0, (hipStream_t)stream, Mw, d->Cout, g, bias,
               has_res ? (const float*)residual : nullptr, (long)d->res_ld, (d->flags & MSOCR_CONV_RELU) ? 1 : 0, (float*)out,
               (long)d->out_ld);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_conv3x3_winograd44_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                              const void* residual, void* out, void* workspace, void* stream) {
  if (!u_planes) return MSOCR_E_ARG;
  int rc = msocr_winograd44_input_transform(d, in, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  rc = msocr_winograd44_gemm_split(d, u_planes, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  return msocr_winograd44_output_transform(d, workspace, bias, residual, out, stream);
}

// U[xi*6+nu][co][c] = sum_{kh,kw} G[xi][kh] G[nu][kw] w[co][kh][kw][c] for the points {0, 3/2, -3/2, 2/3, -2/3, inf}, f64, rounded once.
// HOST function like msocr_winograd42_weights_host.
extern "C" int msocr_winograd44_weights_host(const float* w_khwc, int Cout, int Cin, float* u_out) {
  if (!w_khwc || !u_out || Cout <= 0 || Cin <= 0) return MSOCR_E_ARG;
  static const double G[6][3] = {{1.0, 0.0, 0.0},
                                 {8.0 / 65, 12.0 / 65, 18.0 / 65},     {8.0 / 65, -12.0 / 65, 18.0 / 65},
                                 {-81.0 / 130, -27.0 / 65, -18.0 / 65}, {-81.0 / 130, 27.0 / 65, -18.0 / 65},
                                 {0.0, 0.0, 1.0}};
  const long plane = (long)Cout * Cin;
  for (int co = 0; co < Cout; ++co) {
    const float* w = w_khwc + (long)co * 9 * Cin;
    for (int c = 0; c < Cin; ++c) {
      double gw[6][3];  // G g
      for (int xi = 0; xi < 6; ++xi)
        for (int kw = 0; kw < 3; ++kw)
          gw[xi][kw] = G[xi][0] * w[(0 * 3 + kw) * Cin + c] + G[xi][1] * w[(1 * 3 + kw) * Cin + c] + G[xi][2] * w[(2 * 3 + kw) * Cin + c];
      for (int xi = 0; xi < 6; ++xi)
        for (int nu = 0; nu < 6; ++nu)
          u_out[(xi * 6 + nu) * plane + (long)co * Cin + c] =
              (float)(gw[xi][0] * G[nu][0] + gw[xi][1] * G[nu][1] + gw[xi][2] * G[nu][2]);
    }
  }
  return MSOCR_OK;
}

// =====================================================================================================================
// Cin == 64: the tall form with the 24 GEMMs AND the output transform in ONE kernel (Mw never reaches HBM).
// With 64 input channels the transform-domain GEMMs have K = 64 and the unfused form is HBM-bound on Mw (3x the layer's output:
// TRBA conv0b would move 132 GB per step against 19 GB for the direct convolution).  Here a workgroup owns 32 tiles x 32 output
// channels: for each of the 24 points it stages V[p][32 tiles][64] and U[p][32 couts][64] through LDS (double-buffered, 16-B chunks
// XOR-swizzled by row), one 16x16 accumulator tile per wave and point (v_mfma_f32_16x16x4_f32, k = 64 -> 16 MFMAs); after the 24th
// point every lane holds all 24 transform-domain values of its 4 (tile, cout) positions and applies A6^T . A4 in registers, then
// bias / residual / ReLU and — MSOCR_CONV_POOL2 — the 2x2/2 max-pool that closes conv0 in SE-ResNet31 (seresnet31.py:81-89: ... ReLU,
// MaxPool2d(2, 2)), so the pooled map is the only thing written.  HBM traffic: V once (3x the input) + the (pooled) output.
// =====================================================================================================================
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
#ifndef WINO_FUSED_PFD
#define WINO_FUSED_PFD 3
#endif

// SPLIT = true (round 3): the 24 K = 64 GEMMs on the bf16 matrix pipes with exactly split operands (conv_split.hip has the
// arithmetic): V is split in registers on the way into LDS, U arrives as three bf16 planes [3][24][Cout][64]; per point and wave
// 12 x v_mfma_f32_16x16x32_bf16 (192 cycles) replace 16 x v_mfma_f32_16x16x4_f32 (512 cycles).  Same accumulator layout, so the
// output transform below is shared.
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int wino_split_step(float& x, float& y) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x, y};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const unsigned int pk = __builtin_bit_cast(unsigned int, h);
  x -= __uint_as_float(pk << 16);
  y -= __uint_as_float(pk & 0xffff0000u);
  return pk;
}

template <bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256, 3) void wino42_fused64_kernel(const float* __restrict__ V, const void* __restrict__ Uv, int Cout,
                                                                 WinoGeom g, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, long res_ld, int relu,
                                                                 float* __restrict__ out, long out_ld) {
  constexpr int K = 64, BM = 32, BN = 32, ROWB = K * 4;  // exact form: 256-byte tile rows, 16 chunks of 16 B
  constexpr int ROWS = 128, PLANE = BM * ROWS;           // split form: six [32 rows][64 bf16 = 128 B] planes per stage
  constexpr int STAGE_BYTES = SPLIT ? 6 * PLANE : (BM + BN) * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave >> 1, nj = wave & 1;
  const int r16 = lane & 15, kg = lane >> 4;
  // XCD-aware order: blocks b, b+8, ... share an XCD (its L2): give each XCD a contiguous range of logical blocks, cout blocks
  // fastest, so the Cout/32 workgroups that read one V tile run on the same L2
  const int nbn = Cout / BN;
  const long nblk = (long)gridDim.x;
  long bid = blockIdx.x;
  {
    const long q = nblk >> 3, r = nblk & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int nb = (int)(bid % nbn);
  const long m0 = (bid / nbn) * BM;
  const int n0 = nb * BN;
  f32x4 acc[24];

  if constexpr (SPLIT) {
    const unsigned short* U = reinterpret_cast<const unsigned short*>(Uv);
    // staging: V as the exact form (row = tid / 16 (+16), 16-B chunk of 4 f32 = tid % 16); U planes: row = tid / 8, 16-B chunk of 8 bf16 = tid % 8
    const int chunk = tid & 15, row0 = tid >> 4;
    const int bchunk = tid & 7, brow = tid >> 3;
    const long planeV = g.Mt * (long)K, planeU = (long)Cout * K, uplane = 24 * planeU;
    const float* pa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long m = m0 + row0 + 16 * i;
      if (m >= g.Mt) m = g.Mt - 1;  // valid memory; the rows are never stored
      pa[i] = V + m * K + chunk * 4;
    }
    const unsigned short* pb = U + (long)(n0 + brow) * K + bchunk * 8;
    constexpr int PFD = 2;
    u32x4w ra[PFD][2], rb[PFD][3];
    auto gload = [&](int p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) ra[p % PFD][i] = *reinterpret_cast<const u32x4w*>(pa[i] + p * planeV);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) rb[p % PFD][pl] = *reinterpret_cast<const u32x4w*>(pb + pl * uplane + p * planeU);
    };
    auto sstore = [&](int p) {
      unsigned char* st = &smem[p & 1][0];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + 16 * i;
        float x0 = __uint_as_float(ra[p % PFD][i][0]), x1 = __uint_as_float(ra[p % PFD][i][1]);
        float x2 = __uint_as_float(ra[p % PFD][i][2]), x3 = __uint_as_float(ra[p % PFD][i][3]);
        unsigned char* dst = st + row * ROWS + (((chunk >> 1) ^ (row & 7)) << 4) + ((chunk & 1) << 3);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          u32x2w v;
          v[0] = wino_split_step(x0, x1);
          v[1] = wino_split_step(x2, x3);
          *reinterpret_cast<u32x2w*>(dst + pl * PLANE) = v;
        }
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4w*>(st + (3 + pl) * PLANE + brow * ROWS + ((bchunk ^ (brow & 7)) << 4)) = rb[p % PFD][pl];
    };
#pragma unroll
    for (int q = 0; q < PFD; ++q) gload(q);
    sstore(0);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 24; ++p) {
      if (p + PFD < 24) gload(p + PFD);
      const unsigned char* st = &smem[p & 1][0];
      const int rowa = mi * 16 + r16, rowb = nj * 16 + r16;
      f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // two k32 steps; lane (r16, kg) takes k = 32 ks + 8 kg + 0..7 of its row for both operands
        const int c = 4 * ks + kg;
        bf16x8w fa[3], fb[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          fa[pl] = *reinterpret_cast<const bf16x8w*>(st + pl * PLANE + rowa * ROWS + ((c ^ (rowa & 7)) << 4));
          fb[pl] = *reinterpret_cast<const bf16x8w*>(st + (3 + pl) * PLANE + rowb * ROWS + ((c ^ (rowb & 7)) << 4));
        }
        // smallest terms first
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0], c4, 0, 0, 0);
      }
      acc[p] = c4;
      if (p + 1 < 24) sstore(p + 1);
      __syncthreads();
    }
  } else {
  const float* U = reinterpret_cast<const float*>(Uv);
  // staging coordinates: thread -> (row = tid / 16 (+16 on the second pass), 16-B chunk = tid % 16)
  const int chunk = tid & 15, row0 = tid >> 4;
  const long planeV = g.Mt * (long)K, planeU = (long)Cout * K;
  const float* pa[2];
  const float* pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long m = m0 + row0 + 16 * i;
    if (m >= g.Mt) m = g.Mt - 1;  // valid memory; the rows are never stored
    pa[i] = V + m * K + chunk * 4;
    pb[i] = U + (long)(n0 + row0 + 16 * i) * K + chunk * 4;
  }
  // register prefetch PFD points ahead (slot = point % PFD): one point's MFMAs (16 per wave) are shorter than an HBM round trip
  constexpr int PFD = WINO_FUSED_PFD;
  u32x4w ra[PFD][2], rb[PFD][2];
  auto gload = [&](int p) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[p % PFD][i] = *reinterpret_cast<const u32x4w*>(pa[i] + p * planeV);
      rb[p % PFD][i] = *reinterpret_cast<const u32x4w*>(pb[i] + p * planeU);
    }
  };
  auto sstore = [&](int p) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = row0 + 16 * i;
      *reinterpret_cast<u32x4w*>(&smem[p & 1][row * ROWB + ((chunk ^ (row & 15)) << 4)]) = ra[p % PFD][i];
      *reinterpret_cast<u32x4w*>(&smem[p & 1][(BM + row) * ROWB + ((chunk ^ (row & 15)) << 4)]) = rb[p % PFD][i];
    }
  };
#pragma unroll
  for (int q = 0; q < PFD; ++q) gload(q);
  sstore(0);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 24; ++p) {
    if (p + PFD < 24) gload(p + PFD);  // slot p % PFD held point p, stored to LDS at the end of point p - 1
    const unsigned char* sa = &smem[p & 1][0];
    const unsigned char* sb = sa + BM * ROWB;
    // lane (r16, kg) takes k = 16 * kg + 0..15 of its row for BOTH operands (a consistent permutation of the reduction index)
    const int rowa = mi * 16 + r16, rowb = nj * 16 + r16;
    u32x4w fa[4], fb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = kg * 4 + q;
      fa[q] = *reinterpret_cast<const u32x4w*>(sa + rowa * ROWB + ((c ^ (rowa & 15)) << 4));
      fb[q] = *reinterpret_cast<const u32x4w*>(sb + rowb * ROWB + ((c ^ (rowb & 15)) << 4));
    }
    // one dependent accumulation chain per point (two chains double the live registers: measured, dropped; the U fragments straight
    // from L2 instead of LDS: 1.5x slower)
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[q][e]), __uint_as_float(fb[q][e]), c4, 0, 0, 0);
    acc[p] = c4;
    if (p + 1 < 24) sstore(p + 1);
    __syncthreads();
  }

  }

  // ---- output transform in registers: lane holds rows (tiles) 4 * kg + e, column (cout) r16 of its wave's 16x16 block ----
  const int co = n0 + nj * 16 + r16;
  const float b = bias ? bias[co] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long t = m0 + mi * 16 + 4 * kg + e;
    if (t >= g.Mt) continue;
    const int tw = (int)(t % g.TW);
    const long r = t / g.TW;
    const int th = (int)(r % g.TH);
    const int n = (int)(r / g.TH);
    float s[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float m0_ = acc[0 * 4 + j][e], m1 = acc[1 * 4 + j][e], m2 = acc[2 * 4 + j][e], m3 = acc[3 * 4 + j][e], m4 = acc[4 * 4 + j][e],
                  m5 = acc[5 * 4 + j][e];
      const float p12 = m1 + m2, d12 = m1 - m2, p34 = m3 + m4, d34 = m3 - m4;   // A^T of wino44_at, scalar
      s[0][j] = (m0_ + p12) + p34;
      s[1][j] = fmaf(1.5f, d12, (2.0f / 3.0f) * d34);
      s[2][j] = fmaf(2.25f, p12, (4.0f / 9.0f) * p34);
      s[3][j] = fmaf(3.375f, d12, (8.0f / 27.0f) * d34) + m5;
    }
    float y[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[i][0] = ((s[i][0] + s[i][1]) + s[i][2]) + b;
      y[i][1] = ((s[i][1] - s[i][2]) - s[i][3]) + b;
    }
    if constexpr (POOL) {
      // H and W are even (host check): the 2x2 windows are (rows 4th+{0,1} | 4th+{2,3}) x (cols 2tw+{0,1})
      const int Hp = g.H >> 1, Wp = g.W >> 1;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int hp = 2 * th + i2;
        if (hp >= Hp || tw >= Wp) continue;
        float v0 = y[2 * i2][0], v1 = y[2 * i2][1], v2 = y[2 * i2 + 1][0], v3 = y[2 * i2 + 1][1];
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        out[(((long)n * Hp + hp) * Wp + tw) * out_ld + co] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ho = 4 * th + i;
        if (ho >= g.H) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int wo = 2 * tw + j;
          if (wo >= g.W) continue;
          const long pix = ((long)n * g.H + ho) * g.W + wo;
          float v = y[i][j];
          if (res) v += res[pix * res_ld + co];
          if (relu) v = fmaxf(v, 0.f);
          out[pix * out_ld + co] = v;
        }
      }
    }
  }
}

// ---- fused Cin = 64 layer, second form (round 3, split operands, Cout % 64 == 0) --------------------------------------------------
// The kernel above holds the 24 products of a (tile, cout) pair in 24 accumulators and transforms them at the end, which pins a
// wave to one 16 x 16 block (96 accumulator registers) and a workgroup to 32 tiles x 32 couts: per point 12 short MFMAs between two
// barriers, and every workgroup pulls all of U for its 32 couts (288 KB) from L2 for 32 tiles (PMC: matrix pipes 0.25 busy on conv0b,
// 13 ms per step).  The output transform is linear in the 24 products, so this form accumulates it on the fly:
//   Y[a][b] += AT6[a][i] * AT4[b][j] * M[i][j]   right after point (i, j) is multiplied
// (8 output accumulators + the current product; 108 FMAs per element and kernel instead of 24 x 64 MACs on the pipes).  A wave then
// owns a 32 x 32 block (v_mfma_f32_32x32x16_bf16, 24 per point), a workgroup 64 tiles x 64 couts: 4x the matrix work per barrier and
// half the L2 traffic per MAC.  Same V / U layouts, same epilogue semantics (bias, residual, ReLU, MaxPool2d(2, 2)).
template <bool POOL>
__global__ __launch_bounds__(256, 2) void wino42_fused64_v2_kernel(const float* __restrict__ V, const unsigned short* __restrict__ U, int Cout,
                                                                    WinoGeom g, const float* __restrict__ bias,
                                                                    const float* __restrict__ res, long res_ld, int relu,
                                                                    float* __restrict__ out, long out_ld) {
  constexpr int K = 64, BM = 64, BN = 64, ROWS = 128, PLANE = BM * ROWS;  // six [64 rows][64 bf16 = 128 B] planes, one stage
  typedef float f32x16w __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const int nbn = Cout / BN;
  const long nblk = (long)gridDim.x;
  long bid = blockIdx.x;
  {  // XCD-aware order, cout blocks fastest (the workgroups that read one V tile share an L2)
    const long q = nblk >> 3, r = nblk & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int nb = (int)(bid % nbn);
  const long m0 = (bid / nbn) * BM;
  const int n0 = nb * BN;

  // staging: V rows = tid / 16 + 16 i, 16-B chunk of 4 f32 = tid % 16; U planes: rows = tid / 8 + 32 j, 16-B chunk of 8 bf16 = tid % 8
  const int chunk = tid & 15, row0 = tid >> 4;
  const int bchunk = tid & 7, brow = tid >> 3;
  const long planeV = g.Mt * (long)K, planeU = (long)Cout * K, uplane = 24 * planeU;
  const float* pa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + row0 + 16 * i;
    if (m >= g.Mt) m = g.Mt - 1;  // valid memory; the rows are never stored
    pa[i] = V + m * K + chunk * 4;
  }
  const unsigned short* pb = U + (long)(n0 + brow) * K + bchunk * 8;
  u32x4w ra[4], rb[3][2];
  auto gload = [&](int p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const u32x4w*>(pa[i] + p * planeV);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < 2; ++j) rb[pl][j] = *reinterpret_cast<const u32x4w*>(pb + pl * uplane + p * planeU + (long)j * 32 * K);
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 16 * i;
      float x0 = __uint_as_float(ra[i][0]), x1 = __uint_as_float(ra[i][1]), x2 = __uint_as_float(ra[i][2]), x3 = __uint_as_float(ra[i][3]);
      unsigned char* dst = smem + row * ROWS + (((chunk >> 1) ^ (row & 7)) << 4) + ((chunk & 1) << 3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        u32x2w v;
        v[0] = wino_split_step(x0, x1);
        v[1] = wino_split_step(x2, x3);
        *reinterpret_cast<u32x2w*>(dst + pl * PLANE) = v;
      }
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = brow + 32 * j;
        *reinterpret_cast<u32x4w*>(smem + (3 + pl) * PLANE + row * ROWS + ((bchunk ^ (row & 7)) << 4)) = rb[pl][j];
      }
  };

  f32x16w Y[8];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int e = 0; e < 16; ++e) Y[o][e] = 0.f;

  gload(0);
  sstore();
  __syncthreads();
  // coefficient of point p = 4 i + j in output o = 2 a + b: AT6[a][i] * AT4[b][j], AT6 = A^T of wino44_at (the points
  // {0, +-3/2, +-2/3, inf} since round 4), AT4 = [[1,1,1,0],[0,1,-1,-1]] — a run-time table and a ROLLED loop over the points: unrolled, the 24 points'
  // loads and products are hoisted across each other and the kernel spills (375 registers fully unrolled, 128 with four points
  // per iteration); the price is 8 FMAs per element and point including the 84 zero coefficients (192 instead of 108)
  __shared__ float s_cf[24][8];
  if (tid < 192) {
    const float t6[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1.5f, -1.5f, 2.0f / 3.0f, -2.0f / 3.0f, 0}, {0, 2.25f, 2.25f, 4.0f / 9.0f, 4.0f / 9.0f, 0},
                            {0, 3.375f, -3.375f, 8.0f / 27.0f, -8.0f / 27.0f, 1}};   // A^T on the points {0, +-3/2, +-2/3, inf}
    const float t4[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    const int pp = tid >> 3, o = tid & 7;
    s_cf[pp][o] = t6[o >> 1][pp >> 2] * t4[o & 1][pp & 3];
  }
  __syncthreads();
  const int rowa = wm * 32 + r32, rowb = wn * 32 + r32;
#pragma unroll 1
  for (int p = 0; p < 24; ++p) {
    if (p + 1 < 24) gload(p + 1);  // in flight under this point's MFMAs
    f32x16w m;
#pragma unroll
    for (int e = 0; e < 16; ++e) m[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // k16 steps; lane (r32, half) takes k = 16 ks + 8 half + 0..7 of its row for both operands
      const int c = 2 * ks + half;
      bf16x8w fa[3], fb[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[pl] = *reinterpret_cast<const bf16x8w*>(smem + pl * PLANE + rowa * ROWS + ((c ^ (rowa & 7)) << 4));
        fb[pl] = *reinterpret_cast<const bf16x8w*>(smem + (3 + pl) * PLANE + rowb * ROWS + ((c ^ (rowb & 7)) << 4));
      }
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], m, 0, 0, 0);  // smallest terms first
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], m, 0, 0, 0);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const float cf = s_cf[p][o];
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[o][e] = fmaf(cf, m[e], Y[o][e]);
    }
    __syncthreads();  // every wave has read the stage
    if (p + 1 < 24) sstore();
    __syncthreads();
  }

  // ---- epilogue: element e of this lane = tile m0 + 32 wm + (e & 3) + 8 (e >> 2) + 4 half, cout n0 + 32 wn + r32 ----
  const int co = n0 + wn * 32 + r32;
  const float bv = bias ? bias[co] : 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long t = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    if (t >= g.Mt) continue;
    const int tw = (int)(t % g.TW);
    const long r = t / g.TW;
    const int th = (int)(r % g.TH);
    const int n = (int)(r / g.TH);
    if constexpr (POOL) {
      const int Hp = g.H >> 1, Wp = g.W >> 1;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int hp = 2 * th + i2;
        if (hp >= Hp || tw >= Wp) continue;
        float v0 = Y[4 * i2 + 0][e] + bv, v1 = Y[4 * i2 + 1][e] + bv, v2 = Y[4 * i2 + 2][e] + bv, v3 = Y[4 * i2 + 3][e] + bv;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        out[(((long)n * Hp + hp) * Wp + tw) * out_ld + co] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
      }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int ho = 4 * th + a;
        if (ho >= g.H) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int wo = 2 * tw + b;
          if (wo >= g.W) continue;
          const long pix = ((long)n * g.H + ho) * g.W + wo;
          float v = Y[a * 2 + b][e] + bv;
          if (res) v += res[pix * res_ld + co];
          if (relu) v = fmaxf(v, 0.f);
          out[pix * out_ld + co] = v;
        }
      }
    }
  }
}

static int wino42_fused_check(const msocr_conv_desc* d, WinoGeom* g) {
  if (!wino42_geom(d, g) || d->Cin != 64) return MSOCR_E_ARG;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;
  if (d->flags & MSOCR_CONV_POOL2) {
    if ((d->H & 1) || (d->W & 1) || (d->flags & MSOCR_CONV_RESIDUAL)) return MSOCR_E_ARG;
  }
  return MSOCR_OK;
}

extern "C" int64_t msocr_conv3x3_winograd42_fused_workspace_bytes(const msocr_conv_desc* d) {
  WinoGeom g;
  if (wino42_fused_check(d, &g) != MSOCR_OK) return -1;
  return 24 * g.Mt * (int64_t)d->Cin * (int64_t)sizeof(float);
}

// stage 2 of msocr_conv3x3_winograd42_fused (stage 1 is msocr_winograd42_input_transform): V in the workspace -> out
static int wino42_fused_launch(const msocr_conv_desc* d, const void* u_weight, bool split, const void* workspace, const float* bias,
                               const void* residual, void* out, void* stream);

extern "C" int msocr_winograd42_fused_gemm_output(const msocr_conv_desc* d, const float* u_weight, const void* workspace,
                                                  const float* bias, const void* residual, void* out, void* stream) {
  return wino42_fused_launch(d, u_weight, false, workspace, bias, residual, out, stream);
}

// the same with U as three bf16 planes [3][24][Cout][64] (msocr_split_bf16x3_host of msocr_winograd42_weights_host's output): the
// GEMMs run on the bf16 matrix pipes with exactly split operands
extern "C" int msocr_winograd42_fused_gemm_output_split(const msocr_conv_desc* d, const void* u_planes, const void* workspace,
                                                        const float* bias, const void* residual, void* out, void* stream) {
  return wino42_fused_launch(d, u_planes, true, workspace, bias, residual, out, stream);
}

static int wino42_fused_launch(const msocr_conv_desc* d, const void* u_weight, bool split, const void* workspace, const float* bias,
                               const void* residual, void* out, void* stream) {
  WinoGeom g;
  if (wino42_fused_check(d, &g) != MSOCR_OK || !u_weight || !workspace || !out) return MSOCR_E_ARG;
  if (((uintptr_t)workspace | (uintptr_t)u_weight) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld < d->Cout)) return MSOCR_E_ARG;
  const long nblk = ((g.Mt + 31) / 32) * (long)(d->Cout / 32);
  if (nblk <= 0 || nblk > 0x7fffffffL) return MSOCR_E_ARG;
  const int relu = (d->flags & MSOCR_CONV_RELU) ? 1 : 0;
  const float* rp = has_res ? (const float*)residual : nullptr;
  const dim3 grid((unsigned)nblk), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const float* ws = (const float*)workspace;
  // split operands and Cout % 64 == 0: the on-the-fly output transform with 64 x 64 workgroup tiles (MSOCR_WINO_FUSED_V2=0: the
  // 24-accumulator kernel, kept for Cout % 64 != 0 and as the cross-check of the tests)
  static const bool v2 = !(getenv("MSOCR_WINO_FUSED_V2") && getenv("MSOCR_WINO_FUSED_V2")[0] == '0');
  if (split && v2 && d->Cout % 64 == 0) {
    const long nblk2 = ((g.Mt + 63) / 64) * (long)(d->Cout / 64);
    if (nblk2 <= 0 || nblk2 > 0x7fffffffL) return MSOCR_E_ARG;
    const dim3 grid2((unsigned)nblk2);
    const unsigned short* up = (const unsigned short*)u_weight;
    if (d->flags & MSOCR_CONV_POOL2)
      MSOCR_LAUNCH((wino42_fused64_v2_kernel<true>), grid2, blk, 0, st, ws, up, d->Cout, g, bias, (const float*)nullptr, 0L, relu, (float*)out, (long)d->out_ld);
    else
      MSOCR_LAUNCH((wino42_fused64_v2_kernel<false>), grid2, blk, 0, st, ws, up, d->Cout, g, bias, rp, (long)d->res_ld, relu, (float*)out, (long)d->out_ld);
    return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
  }
  if (d->flags & MSOCR_CONV_POOL2) {
    if (split) MSOCR_LAUNCH((wino42_fused64_kernel<true, true>), grid, blk, 0, st, ws, u_weight, d->Cout, g, bias, (const float*)nullptr, 0L, relu, (float*)out, (long)d->out_ld);
    else MSOCR_LAUNCH((wino42_fused64_kernel<true, false>), grid, blk, 0, st, ws, u_weight, d->Cout, g, bias, (const float*)nullptr, 0L, relu, (float*)out, (long)d->out_ld);
  } else {
    if (split) MSOCR_LAUNCH((wino42_fused64_kernel<false, true>), grid, blk, 0, st, ws, u_weight, d->Cout, g, bias, rp, (long)d->res_ld, relu, (float*)out, (long)d->out_ld);
    else MSOCR_LAUNCH((wino42_fused64_kernel<false, false>), grid, blk, 0, st, ws, u_weight, d->Cout, g, bias, rp, (long)d->res_ld, relu, (float*)out, (long)d->out_ld);
  }
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_conv3x3_winograd42_fused(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                                              const void* residual, void* out, void* workspace, void* stream) {
  WinoGeom g;
  if (wino42_fused_check(d, &g) != MSOCR_OK || !in || !workspace) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const long nb_in = (g.Mt * (d->Cin / 4) + 255) / 256;
  if (nb_in > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino42_input_kernel, dim3((unsigned)nb_in), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN,
               (long)d->in_sH, (long)d->in_sW, d->Cin, g, (float*)workspace);
  if (hipGetLastError() != hipSuccess) return MSOCR_E_LAUNCH;
  return msocr_winograd42_fused_gemm_output(d, u_weight, workspace, bias, residual, out, stream);
}

extern "C" int msocr_conv3x3_winograd42_fused_split(const msocr_conv_desc* d, const void* in, const void* u_planes, const float* bias,
                                                    const void* residual, void* out, void* workspace, void* stream) {
  WinoGeom g;
  if (wino42_fused_check(d, &g) != MSOCR_OK || !in || !workspace || !u_planes) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)workspace) & 15) return MSOCR_E_ARG;
  const long nb_in = (g.Mt * (d->Cin / 4) + 255) / 256;
  if (nb_in > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino42_input_kernel, dim3((unsigned)nb_in), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN,
               (long)d->in_sH, (long)d->in_sW, d->Cin, g, (float*)workspace);
  if (hipGetLastError() != hipSuccess) return MSOCR_E_LAUNCH;
  return msocr_winograd42_fused_gemm_output_split(d, u_planes, workspace, bias, residual, out, stream);
}

// U[xi*4+nu][co][c] = sum_{kh,kw} G6[xi][kh] G4[nu][kw] w[co][kh][kw][c] (xi = 0..5 on the kernel's H axis), f64, rounded once.
// HOST function like msocr_winograd_weights_host.
extern "C" int msocr_winograd42_weights_host(const float* w_khwc, int Cout, int Cin, float* u_out) {
  if (!w_khwc || !u_out || Cout <= 0 || Cin <= 0) return MSOCR_E_ARG;
  // H axis: the points {0, 3/2, -3/2, 2/3, -2/3, inf} (round 4), as msocr_winograd44_weights_host
  static const double G6[6][3] = {{1.0, 0.0, 0.0},
                                  {8.0 / 65, 12.0 / 65, 18.0 / 65},     {8.0 / 65, -12.0 / 65, 18.0 / 65},
                                  {-81.0 / 130, -27.0 / 65, -18.0 / 65}, {-81.0 / 130, 27.0 / 65, -18.0 / 65},
                                  {0.0, 0.0, 1.0}};
  static const double G4[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const long plane = (long)Cout * Cin;
  for (int co = 0; co < Cout; ++co) {
    const float* w = w_khwc + (long)co * 9 * Cin;
    for (int c = 0; c < Cin; ++c) {
      double gw[6][3];  // G6 g
      for (int xi = 0; xi < 6; ++xi)
        for (int kw = 0; kw < 3; ++kw)
          gw[xi][kw] = G6[xi][0] * w[(0 * 3 + kw) * Cin + c] + G6[xi][1] * w[(1 * 3 + kw) * Cin + c] + G6[xi][2] * w[(2 * 3 + kw) * Cin + c];
      for (int xi = 0; xi < 6; ++xi)
        for (int nu = 0; nu < 4; ++nu)
          u_out[(xi * 4 + nu) * plane + (long)co * Cin + c] =
              (float)(gw[xi][0] * G4[nu][0] + gw[xi][1] * G4[nu][1] + gw[xi][2] * G4[nu][2]);
    }
  }
  return MSOCR_OK;
}

// U[xi*4+nu][co][c] = sum_{kh,kw} G[xi][kh] G[nu][kw] w[co][kh][kw][c], evaluated in f64 and rounded once to f32.
// HOST function (runs at weight-load time): w_khwc and u_out are host pointers.
extern "C" int msocr_winograd_weights_host(const float* w_khwc, int Cout, int Cin, float* u_out) {
  if (!w_khwc || !u_out || Cout <= 0 || Cin <= 0) return MSOCR_E_ARG;
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  const long plane = (long)Cout * Cin;
  for (int co = 0; co < Cout; ++co) {
    const float* w = w_khwc + (long)co * 9 * Cin;
    for (int c = 0; c < Cin; ++c) {
      double gw[4][3];  // G g
      for (int xi = 0; xi < 4; ++xi)
        for (int kw = 0; kw < 3; ++kw)
          gw[xi][kw] = G[xi][0] * w[(0 * 3 + kw) * Cin + c] + G[xi][1] * w[(1 * 3 + kw) * Cin + c] + G[xi][2] * w[(2 * 3 + kw) * Cin + c];
      for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu)
          u_out[(xi * 4 + nu) * plane + (long)co * Cin + c] =
              (float)(gw[xi][0] * G[nu][0] + gw[xi][1] * G[nu][1] + gw[xi][2] * G[nu][2]);
    }
  }
  return MSOCR_OK;
}
