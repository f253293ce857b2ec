// trba_kernels.hip — the non-convolutional part of the TRBA recogniser on gfx950:
// squeeze-excite tail, mean over H, BiLSTM recurrence and the attention decoder
// (greedy and beam).  These stages are bandwidth/latency bound: they run
// on the VALU in exact f32 with coalesced weight streams (weights pre-transposed so lane j
// reads column j), LDS-resident per-row state and wave-level reductions; no MFMA.
// The sequential loops (T encoder steps, <= 26 decoder steps) live INSIDE one launch:
// rows are independent, so one workgroup owns a row (or a few) for the whole loop and no
// inter-workgroup hand-off exists.
//
//   msocr_se_residual       <- recognizers/_trba/model/seresnet31.py:5-20, 61-66
//   msocr_mean_over_h       <- recognizers/_trba/model/model.py:388-390
//   msocr_bilstm_recurrent  <- model.py:9-21 (nn.LSTM, gate order i,f,g,o)
//   msocr_attn_greedy       <- model.py:34-46 + 227-259
//   msocr_attn_beam(+_finalize) <- model.py:34-46 + 92-225
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "internal.h"
#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<uint16_t*>(&b);
}
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// --------------------------------------------------------------------------------------------- SE tail
// One workgroup per sample: mean over HW (coalesced over channels), two tiny FCs, then
// out = relu(x * gate + identity).  x is re-read from L2 for the last phase.
// 4 channels per lane (16-byte f32 / 8-byte bf16 accesses); the 256 threads split into PG = 1024/C pixel groups whose
// partial channel sums are combined through LDS in a fixed order (deterministic).
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p);
template <>
__device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 ld4<uint16_t>(const uint16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  f32x4 r = {bf2f((uint16_t)(v.x & 0xffff)), bf2f((uint16_t)(v.x >> 16)), bf2f((uint16_t)(v.y & 0xffff)), bf2f((uint16_t)(v.y >> 16))};
  return r;
}
template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void st4<uint16_t>(uint16_t* p, f32x4 v) {
  uint2 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = o;
}

// Measured and dropped (round 3): keeping the summed x values of a 4 x 13 map in registers (26 x 16 B per thread) so that the scale
// pass does not read x again — 0.118 against 0.091 ms per 960 crops: the second read hits L2, and 163 registers cost occupancy.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void se_residual_kernel(const T* __restrict__ x, const T* __restrict__ idt, int HW, int C,
                                                           const float* __restrict__ w1, const float* __restrict__ w2,
                                                           float* __restrict__ gate_ws, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // part[PG][C] | mean[C] | hid[C/16] | gate[C]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int C4 = C / 4, PG = NT / C4;  // host guarantees C in {64,128,256,512,1024}: C4 divides NT
  float* part = sm;
  float* mean = sm + PG * C;
  float* hid = mean + C;
  float* gate = hid + C / 16;
  const T* xs = x + (long)n * HW * C;
  const int cg = tid % C4, pg = tid / C4;
  {
    // four loads in flight per thread, added in the order of the one-at-a-time loop (same sums): with one load per iteration the
    // kernel sat at 0.45 of the HBM rate — too few bytes in flight per CU for a 2 us round trip
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = pg;
    for (; p + 3 * PG < HW; p += 4 * PG) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4<T>(xs + (long)(p + u * PG) * C + cg * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; s[3] += v[u][3]; }
    }
    for (; p < HW; p += PG) {
      const f32x4 v = ld4<T>(xs + (long)p * C + cg * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    *reinterpret_cast<f32x4*>(&part[pg * C + cg * 4]) = s;
  }
  __syncthreads();
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += NT) {
    float s = 0.f;
    for (int g = 0; g < PG; ++g) s += part[g * C + c];
    mean[c] = s * inv;
  }
  __syncthreads();
  const int Cr = C / 16;
  {  // hid = relu(W1 mean): one wave per output, lanes over C
    const int lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < Cr; j += NT / 64) {
      float a = 0.f;
      for (int c = lane; c < C; c += 64) a = fmaf(w1[(long)j * C + c], mean[c], a);
      a = wave_sum(a);
      if (lane == 0) hid[j] = fmaxf(a, 0.f);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    float a = 0.f;
    for (int j = 0; j < Cr; ++j) a = fmaf(w2[(long)c * Cr + j], hid[j], a);
    const float g = sigmoidf_(a);
    gate[c] = g;
    gate_ws[(long)n * C + c] = g;
  }
  __syncthreads();
  const T* is = idt + (long)n * HW * C;
  T* os = out + (long)n * HW * C;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(&gate[cg * 4]);
  int p = pg;
  for (; p + 3 * PG < HW; p += 4 * PG) {
    f32x4 v[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long o = (long)(p + u * PG) * C + cg * 4;
      v[u] = ld4<T>(xs + o);
      r[u] = ld4<T>(is + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaxf(v[u][e] * g4[e] + r[u][e], 0.f);
      st4<T>(os + (long)(p + u * PG) * C + cg * 4, y);
    }
  }
  for (; p < HW; p += PG) {
    const long o = (long)p * C + cg * 4;
    const f32x4 v = ld4<T>(xs + o), r = ld4<T>(is + o);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaxf(v[e] * g4[e] + r[e], 0.f);
    st4<T>(os + o, y);
  }
}

extern "C" int msocr_se_residual(const void* x, const void* identity, int N, int HW, int C, int dtype, const float* w1, const float* w2,
                                 float* gate_ws, void* out, void* stream) {
  if (!x || !identity || !w1 || !w2 || !gate_ws || !out || N <= 0 || HW <= 0 || C < 64 || C > 1024 || (C & (C - 1))) return MSOCR_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  // f32: 1024-thread workgroups (2 per CU: 512 crops in flight).  A crop is read twice (mean, then scale); with 256-thread workgroups
  // all 1920 crops of a sub-batch are resident at once — 0.8 GB between the two reads of a crop, nothing of it left in L2 / MALL.
  // Measured per 1920 crops: 0.587 / 0.311 / 0.173 ms (16x50x128 / 8x25x256 / 4x13x512) -> 0.541 / 0.249 / 0.151.  The size is the
  // same for every batch size (the order of the mean's additions depends on it, and results must not depend on the batch
  // composition); MSOCR_SE_NT=256 selects the small workgroups.
  static const int nt_env = getenv("MSOCR_SE_NT") ? atoi(getenv("MSOCR_SE_NT")) : 0;
  const int NT = nt_env == 256 ? 256 : 1024;
  const size_t lds = (size_t)((NT / (C / 4)) * C + 2 * C + C / 16) * sizeof(float);
  if (dtype == MSOCR_F32) {
    if (NT == 1024)
      MSOCR_LAUNCH((se_residual_kernel<float, 1024>), dim3(N), dim3(1024), lds, s, (const float*)x, (const float*)identity, HW, C, w1, w2, gate_ws, (float*)out);
    else
      MSOCR_LAUNCH((se_residual_kernel<float, 256>), dim3(N), dim3(256), lds, s, (const float*)x, (const float*)identity, HW, C, w1, w2, gate_ws, (float*)out);
  } else if (dtype == MSOCR_BF16) {
    MSOCR_LAUNCH((se_residual_kernel<uint16_t, 256>), dim3(N), dim3(256), (size_t)((1024 / C) * C + 2 * C + C / 16) * sizeof(float), s,
                 (const uint16_t*)x, (const uint16_t*)identity, HW, C, w1, w2, gate_ws, (uint16_t*)out);
  } else {
    return MSOCR_E_ARG;
  }
  return LAUNCH_OK();
}

// --------------------------------------------------------------------------------------------- mean over H
template <typename T>
__global__ void mean_over_h_kernel(const T* __restrict__ in, int N, int H, int W, int C, float* __restrict__ out) {
  const long total = (long)N * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long t = i / C;
    const int w = (int)(t % W);
    const int n = (int)(t / W);
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += ldf<T>(in + (((long)n * H + h) * W + w) * C + c);
    out[i] = s / (float)H;
  }
}
extern "C" int msocr_mean_over_h(const void* in, int N, int H, int W, int C, int dtype, float* out, void* stream) {
  if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MSOCR_E_ARG;
  const long total = (long)N * W * C;
  long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MSOCR_F32)
    MSOCR_LAUNCH(mean_over_h_kernel<float>, dim3((unsigned)g), dim3(256), 0, s, (const float*)in, N, H, W, C, out);
  else if (dtype == MSOCR_BF16)
    MSOCR_LAUNCH(mean_over_h_kernel<uint16_t>, dim3((unsigned)g), dim3(256), 0, s, (const uint16_t*)in, N, H, W, C, out);
  else
    return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// --------------------------------------------------------------------------------------------- BiLSTM recurrence
// grid (ceil(B/RB), 2 directions), 256 threads = hidden units (H == 256).  Thread j owns unit j's
// four gates for RB batch rows; h_{t-1} of the RB rows sits in LDS as [k][r] so one
// ds_read_b128 pair broadcasts the 8 row values of column k; W_hh^T rows stream from L2,
// coalesced over j.  xproj already holds x W_ih^T + b_ih + b_hh.
#define LSTM_RB 4
template <int H>
__global__ __launch_bounds__(H) void bilstm_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_t, int B, int T,
                                                    float* __restrict__ hcat) {
  constexpr int G = 4 * H, RB = LSTM_RB;
  __shared__ __attribute__((aligned(16))) float hs[2][H][RB];
  const int j = threadIdx.x, d = blockIdx.y;
  const int b0 = blockIdx.x * RB;
  const float* wt = whh_t + (long)d * H * G;
  float c[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    c[r] = 0.f;
    hs[0][j][r] = 0.f;
  }
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    const int cur = s & 1;
    float acc[4][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int b = b0 + r < B ? b0 + r : B - 1;
      const float* xp = xproj + (((long)b * T + t) * 2 + d) * G;
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g][r] = xp[g * H + j];
    }
#pragma unroll 1
    for (int k0 = 0; k0 < H; k0 += 8) {
      f32x4 wq[8];  // gates i,f,g,o of unit j: 8 x 16-B loads in flight per lane
#pragma unroll
      for (int u = 0; u < 8; ++u) wq[u] = *reinterpret_cast<const f32x4*>(&wt[((long)(k0 + u) * H + j) * 4]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float hv[RB];
#pragma unroll
        for (int r4 = 0; r4 < RB; r4 += 4) {
          const f32x4 hq = *reinterpret_cast<const f32x4*>(&hs[cur][k0 + u][r4]);
          hv[r4] = hq[0]; hv[r4 + 1] = hq[1]; hv[r4 + 2] = hq[2]; hv[r4 + 3] = hq[3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int r = 0; r < RB; ++r) acc[g][r] = fmaf(wq[u][g], hv[r], acc[g][r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]), gg = tanhf(acc[2][r]), og = sigmoidf_(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      const float h = og * tanhf(c[r]);
      hs[cur ^ 1][j][r] = h;
      if (b0 + r < B) hcat[((long)(b0 + r) * T + t) * (2 * H) + d * H + j] = h;
    }
    __syncthreads();
  }
}

extern "C" int msocr_bilstm_recurrent(const float* xproj, const float* w_hh_t, int B, int T, int H, float* hcat_out, void* stream) {
  if (!xproj || !w_hh_t || !hcat_out || B <= 0 || T <= 0) return MSOCR_E_ARG;
  const dim3 grid((B + LSTM_RB - 1) / LSTM_RB, 2);
  // hidden_size comes from the checkpoint's config (recognizers/_trba/__init__.py:142-151): 256 is the reference default
  if (H == 256) MSOCR_LAUNCH(bilstm_kernel<256>, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 128) MSOCR_LAUNCH(bilstm_kernel<128>, grid, dim3(128), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 512) MSOCR_LAUNCH(bilstm_kernel<512>, grid, dim3(512), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 64) MSOCR_LAUNCH(bilstm_kernel<64>, grid, dim3(64), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 192) MSOCR_LAUNCH(bilstm_kernel<192>, grid, dim3(192), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 320) MSOCR_LAUNCH(bilstm_kernel<320>, grid, dim3(320), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 384) MSOCR_LAUNCH(bilstm_kernel<384>, grid, dim3(384), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else if (H == 448) MSOCR_LAUNCH(bilstm_kernel<448>, grid, dim3(448), 0, (hipStream_t)stream, xproj, w_hh_t, B, T, hcat_out);
  else return MSOCR_E_ARG;
  return LAUNCH_OK();
}

// --------------------------------------------------------------------------------------------- attention decoder
// One workgroup (256 threads, H == 256) owns NB batch rows for the whole step loop; KR = NB*RPB state rows
// (RPB = 1 greedy, 8 beam).  Thread j owns hidden unit j (4 gates x KR rows in registers); weights are
// pre-transposed so lane j streams column j (gate-interleaved: one 16-B load per k); every weight element a
// workgroup fetches from L2 is used for KR rows, so NB = 2 halves the L2 traffic that bounds this kernel.
// LDS: batch_H / proj_H of the NB rows [NB][T][H], h and ctx as [k][row], ph[row][j], alpha[row][t], logits[row][v].
#define ATT_H 256
#define ATT_KMAX 8


template <int KR, int RPB, int PF = 4>
__device__ __forceinline__ void attention_cell_step(const AttnArgs& a, int tid, const float* sH, const float* sP, float (*sh)[KR],
                                                    float (*sctx)[KR], float (*sph)[ATT_H], float (*salpha)[64],
                                                    float (*slog)[256], float (&c)[KR], const int* tok, int T, int V) {
  constexpr int H = ATT_H;
  const int j = tid;
  // (a) ph[r][j] = h2h(h)[j]
  {
    float acc[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) acc[r] = a.w.h2h_b[j];
#pragma unroll 1
    for (int k0 = 0; k0 < H; k0 += 8) {
      float wq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wq[u] = a.w.h2h_wt[(long)(k0 + u) * H + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < KR; ++r) acc[r] = fmaf(wq[u], sh[k0 + u][r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) sph[r][j] = acc[r];
  }
  __syncthreads();
  // (b) e[r][t] = score . tanh(proj_H[t] + ph[r]) : one wave per (r,t) pair, lanes over j
  {
    const int lane = tid & 63, wv = tid >> 6;
    for (int p = wv; p < KR * T; p += 4) {
      const int r = p / T, t = p - r * T;
      const float* pP = sP + (r / RPB) * T * H;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < H / 64; ++q) {
        const int jj = lane + 64 * q;
        s = fmaf(a.w.score_w[jj], tanhf(pP[t * H + jj] + sph[r][jj]), s);
      }
      s = wave_sum(s);
      if (lane == 0) salpha[r][t] = s;
    }
  }
  __syncthreads();
  // (c) softmax over t (T <= 64): thread r
  if (tid < KR) {
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, salpha[tid][t]);
    float sum = 0.f;
    for (int t = 0; t < T; ++t) {
      const float e = expf(salpha[tid][t] - m);
      salpha[tid][t] = e;
      sum += e;
    }
    for (int t = 0; t < T; ++t) salpha[tid][t] = salpha[tid][t] / sum;
  }
  __syncthreads();
  // (d) ctx[r][j] = sum_t alpha[r][t] * batch_H[t][j]
  {
    float acc[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) acc[r] = 0.f;
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int nb = 0; nb < KR / RPB; ++nb) {
        const float hv = sH[(nb * T + t) * H + j];
#pragma unroll
        for (int q = 0; q < RPB; ++q) acc[nb * RPB + q] = fmaf(salpha[nb * RPB + q][t], hv, acc[nb * RPB + q]);
      }
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) sctx[j][r] = acc[r];
  }
  __syncthreads();
  // (e) LSTMCell gates: W_ih[:, :H] ctx + W_ih[:, H+tok] + W_hh h + (b_ih + b_hh); thread j owns unit j
  float g4[4][KR];
  {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(&a.w.b_gates[j * 4]);
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(&a.w.wih_tok[((long)tok[r] * H + j) * 4]);
#pragma unroll
      for (int g = 0; g < 4; ++g) g4[g][r] = b4[g] + t4[g];
    }
  }
  // weights stream from L2: keep PF 16-byte loads in flight per lane (the FMAs of a k-group run under the next group's loads)
  auto gate_pass = [&](const float* __restrict__ wt, float (*x)[KR]) {
    f32x4 wq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) wq[u] = *reinterpret_cast<const f32x4*>(&wt[((long)u * H + j) * 4]);
#pragma unroll 1
    for (int k0 = 0; k0 < H; k0 += PF) {
      f32x4 wn[PF];
      if (k0 + PF < H) {
#pragma unroll
        for (int u = 0; u < PF; ++u) wn[u] = *reinterpret_cast<const f32x4*>(&wt[((long)(k0 + PF + u) * H + j) * 4]);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < KR; ++r) g4[g][r] = fmaf(wq[u][g], x[k0 + u][r], g4[g][r]);
      if (k0 + PF < H) {
#pragma unroll
        for (int u = 0; u < PF; ++u) wq[u] = wn[u];
      }
    }
  };
  gate_pass(a.w.wih_ctx_t, sctx);
  gate_pass(a.w.whh_t, sh);
  __syncthreads();  // everyone finished reading the old h
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const float ig = sigmoidf_(g4[0][r]), fg = sigmoidf_(g4[1][r]), gg = tanhf(g4[2][r]), og = sigmoidf_(g4[3][r]);
    c[r] = fg * c[r] + ig * gg;
    sh[j][r] = og * tanhf(c[r]);
  }
  __syncthreads();
  // (f) generator logits[r][v]
  if (tid < V) {
    float acc[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) acc[r] = a.w.gen_b[tid];
#pragma unroll 1
    for (int k0 = 0; k0 < H; k0 += 8) {
      float wq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wq[u] = a.w.gen_wt[(long)(k0 + u) * V + tid];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int r = 0; r < KR; ++r) acc[r] = fmaf(wq[u], sh[k0 + u][r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) slog[r][tid] = (tid == a.blank_id) ? -1e4f : acc[r];
  }
  __syncthreads();
}

// block-wide arg-max over (value, index) pairs with "larger value, then smaller index" order
__device__ __forceinline__ void block_argmax(float v, int idx, float* s_val, int* s_idx, int tid, float& out_v, int& out_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(idx, o);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = idx; }
  __syncthreads();
  out_v = s_val[0];
  out_i = s_idx[0];
  for (int w = 1; w < 4; ++w)
    if (s_val[w] > out_v || (s_val[w] == out_v && s_idx[w] < out_i)) { out_v = s_val[w]; out_i = s_idx[w]; }
  __syncthreads();
}

__global__ __launch_bounds__(256) void attn_greedy_kernel(AttnArgs a) {
  constexpr int H = ATT_H, K = 1;
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // batch_H[T][H] | proj_H[T][H]
  __shared__ __attribute__((aligned(16))) float sh[H][K], sctx[H][K];
  __shared__ float sph[K][H], salpha[K][64], slog[K][256];
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  const int b = blockIdx.x, tid = threadIdx.x, T = a.T, V = a.V;
  float* sH = dyn;
  float* sP = dyn + T * H;
  for (int i = tid; i < T * H; i += 256) {
    sH[i] = a.batch_H[(long)b * T * H + i];
    sP[i] = a.proj_H[(long)b * T * H + i];
  }
  float c[K] = {0.f};
  sh[tid][0] = 0.f;
  int tok[K] = {a.sos_id};
  __syncthreads();
  for (int s = 0; s < a.steps; ++s) {
    attention_cell_step<K, 1>(a, tid, sH, sP, sh, sctx, sph, salpha, slog, c, tok, T, V);
    float v = -INFINITY;
    int idx = 0x7fffffff;
    if (tid < V) {
      v = slog[0][tid];
      idx = tid;
      a.logits_out[((long)b * a.steps + s) * V + tid] = v;
    }
    float bv;
    int bi;
    block_argmax(v, idx, s_val, s_idx, tid, bv, bi);
    tok[0] = bi;
    if (tid == 0) a.ids_out[(long)b * a.steps + s] = bi;
  }
}

// DIAGNOSTIC kernel (MSOCR_BEAM_MFMA=0), off the default path: the default beam decoder is attn_beam_mfma_kernel (attn_beam_mfma.hip);
// this VALU form stays as its independent cross-check (tests/test_gpu_trba.py: MFMA vs VALU agreement, three-way parity test).
// HB = number of 256-thread halves per workgroup.  With HB = 2 two batch rows share one workgroup: both halves run
// the same instruction stream between the same barriers, so the second half's weight loads hit the lines the first
// half just pulled into the CU's L1 — the L2 -> L1 weight stream that bounds this kernel is paid once for two rows,
// at unchanged registers per thread and waves per CU.
template <int NB, int HB, int PF = 4>
__global__ __launch_bounds__(256 * HB, (NB == 1 && HB == 1) ? 2 : 1) void attn_beam_kernel(AttnArgs a) {
  constexpr int H = ATT_H, K = ATT_KMAX, KR = NB * K;
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // per half: batch_H[NB][T][H] | proj_H[NB][T][H]
  __shared__ __attribute__((aligned(16))) float sh_[HB][H][KR], sctx_[HB][H][KR];
  __shared__ float sph_[HB][KR][H], salpha_[HB][KR][64], slog_[HB][KR][256];
  __shared__ float s_val_[HB][4];
  __shared__ int s_idx_[HB][4];
  __shared__ float s_score_[HB][KR], s_lse_[HB][KR], s_top_[HB][KR];
  __shared__ int s_tok_[HB][KR], s_done_[HB][KR], s_src_[HB][KR], s_nxt_[HB][KR];
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
  float (*sh)[KR] = sh_[half];
  float (*sctx)[KR] = sctx_[half];
  float (*sph)[H] = sph_[half];
  float (*salpha)[64] = salpha_[half];
  float (*slog)[256] = slog_[half];
  float* s_val = s_val_[half];
  int* s_idx = s_idx_[half];
  float *s_score = s_score_[half], *s_lse = s_lse_[half], *s_top = s_top_[half];
  int *s_tok = s_tok_[half], *s_done = s_done_[half], *s_src = s_src_[half], *s_nxt = s_nxt_[half];
  const int b0 = (blockIdx.x * HB + half) * NB, T = a.T, V = a.V, KB = a.K;
  float* sH = dyn + half * (2 * NB * T * H);
  float* sP = sH + NB * T * H;
  for (int nb = 0; nb < NB; ++nb) {
    const int b = min(b0 + nb, a.B - 1);  // a ragged last workgroup recomputes row B-1 and does not store it
    for (int i = tid; i < T * H; i += 256) {
      sH[nb * T * H + i] = a.batch_H[(long)b * T * H + i];
      sP[nb * T * H + i] = a.proj_H[(long)b * T * H + i];
    }
  }
  float c[KR];
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    c[r] = 0.f;
    sh[tid][r] = 0.f;
  }
  if (tid < KR) {
    s_score[tid] = (tid % K) == 0 ? 0.f : -INFINITY;
    s_tok[tid] = a.sos_id;
    s_done[tid] = 0;
  }
  __syncthreads();
  int fin[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) fin[nb] = a.steps;
  const float temp = fmaxf(a.temperature, 1e-6f);
  for (int s = 0; s < a.steps; ++s) {
    int tok[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) tok[r] = s_tok[r];
    attention_cell_step<KR, K, PF>(a, tid, sH, sP, sh, sctx, sph, salpha, slog, c, tok, T, V);
    // temperature (true f32 division, model.py:135-137), keep the scaled logits for the trace
    if (tid < V) {
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        float v = slog[r][tid];
        if (a.temperature != 1.0f) v = v / temp;
        slog[r][tid] = v;
        const int b = b0 + r / K, rb = r % K;
        if (rb < KB && b < a.B) a.logits_out[(((long)b * a.steps + s) * KB + rb) * V + tid] = v;
      }
    }
    __syncthreads();
    // log_softmax per state row: one wave per row
    {
      const int lane = tid & 63, wv = tid >> 6;
      for (int r = wv; r < KR; r += 4) {
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, slog[r][v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int v = lane; v < V; v += 64) sum += expf(slog[r][v] - m);
        sum = wave_sum(sum);
        if (lane == 0) s_lse[r] = m + logf(sum);
      }
    }
    __syncthreads();
    // candidates: cand[r][v] = (score[r] + logp[r][v]) / lp ; finished beams: only EOS with logp 0
    const float lp = a.lp ? a.lp[s] : 1.0f;
    float cv[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      cv[r] = -INFINITY;
      if (tid < V && (r % K) < KB) {
        float logp = slog[r][tid] - s_lse[r];  // log_softmax = x - (max + log(sum exp(x - max)))
        if (s_done[r]) logp = (tid == a.eos_id) ? 0.f : -INFINITY;
        float tot = s_score[r] + logp;
        if (a.lp) tot = tot / lp;
        cv[r] = tot;
      }
    }
    // top-K per batch row by K rounds of block arg-max over its K*V candidates (flat index = beam*V + v)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      for (int kk = 0; kk < KB; ++kk) {
        float bvv = -INFINITY;
        int bii = 0x7fffffff;
#pragma unroll
        for (int rb = 0; rb < K; ++rb) {
          const int fi = rb * V + tid;
          const float x = cv[nb * K + rb];
          if (tid < V && rb < KB && (x > bvv || (x == bvv && fi < bii))) { bvv = x; bii = fi; }
        }
        float wv_;
        int wi_;
        block_argmax(bvv, bii, s_val, s_idx, tid, wv_, wi_);
        if (wi_ == 0x7fffffff) wi_ = 0;  // every candidate NaN: degenerate input
        if (tid == 0) {
          s_top[nb * K + kk] = wv_;
          s_src[nb * K + kk] = wi_ / V;
          s_nxt[nb * K + kk] = wi_ % V;
        }
        // remove the winner: NaN never wins a comparison again
        if (tid < V) {
          const int wr = wi_ / V, wc = wi_ - wr * V;
#pragma unroll
          for (int rb = 0; rb < K; ++rb)
            if (rb == wr && tid == wc) cv[nb * K + rb] = __int_as_float(0x7fc00000);
        }
      }
    }
    __syncthreads();
    // reorder beam state by src (within each batch row)
    float cn[KR], hn[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int nb = r / K, rb = r % K;
      const int src = rb < KB ? s_src[r] : rb;
      float cc = 0.f, hh = 0.f;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        cc = (q == src) ? c[nb * K + q] : cc;
        hh = (q == src) ? sh[tid][nb * K + q] : hh;
      }
      cn[r] = cc;
      hn[r] = hh;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      c[r] = cn[r];
      sh[tid][r] = hn[r];
    }
    int nd[KR];
    int alldone[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) alldone[nb] = 1;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int nb = r / K, rb = r % K;
      nd[r] = rb < KB ? (s_done[nb * K + s_src[r]] | (s_nxt[r] == a.eos_id)) : 1;
      alldone[nb] &= nd[r];
    }
    __syncthreads();
    if (tid < KR) {
      const int nb = tid / K, rb = tid % K, b = b0 + nb;
      if (rb < KB) {
        if (b < a.B) {
          const long o = ((long)b * a.steps + s) * KB + rb;
          a.back[o] = s_src[tid];
          a.tokv[o] = s_nxt[tid];
        }
        s_score[tid] = a.lp ? s_top[tid] * lp : s_top[tid];  // f32 round trip of the reference (model.py:188-192)
        s_tok[tid] = s_nxt[tid];
        s_done[tid] = nd[tid];
      }
    }
    __syncthreads();
    if (tid < NB && b0 + tid < a.B) {
      // best beam if the loop stopped after this step: argmax of the un-normalised sums, first maximum
      int best = 0;
      float bs = s_score[tid * K];
      for (int r = 1; r < KB; ++r)
        if (s_score[tid * K + r] > bs) { bs = s_score[tid * K + r]; best = r; }
      a.best_at[(long)(b0 + tid) * a.steps + s] = best;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if (alldone[nb] && fin[nb] == a.steps) fin[nb] = s + 1;
    __syncthreads();
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    if (tid == 0 && b0 + nb < a.B) a.fin_step[b0 + nb] = fin[nb];
}

// finalize: walk the back-pointers from (t_run-1, best_at[t_run-1]) and gather the path's logits
__global__ void attn_beam_finalize_kernel(const float* __restrict__ ws_logits, const int32_t* __restrict__ back,
                                          const int32_t* __restrict__ tokv, const int32_t* __restrict__ best_at,
                                          const int32_t* __restrict__ trun, int V, int steps, int K, float* __restrict__ logits_out,
                                          int32_t* __restrict__ ids_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ int path[64];
  const int tr = trun[b];
  if (tid == 0) {
    int cur = best_at[(long)b * steps + tr - 1];
    for (int t = tr - 1; t >= 0; --t) {
      const long o = ((long)b * steps + t) * K + cur;
      ids_out[(long)b * steps + t] = tokv[o];
      path[t] = back[o];  // row of the PARENT beam whose logits produced this token (model.py:198-201)
      cur = back[o];
    }
    for (int t = tr; t < steps; ++t) ids_out[(long)b * steps + t] = -1;
  }
  __syncthreads();
  for (int t = 0; t < tr; ++t)
    for (int v = tid; v < V; v += blockDim.x)
      logits_out[((long)b * steps + t) * V + v] = ws_logits[(((long)b * steps + t) * K + path[t]) * V + v];
}

// shapes of the two fast kernels (one hidden unit per thread, logits of one row in 256 LDS floats)
static bool attn_fast_shape(int T, int H, int V) { return H == ATT_H && V <= 256 && T <= 48; }
static int check_attn(const float* bh, const float* ph, const msocr_attn_weights* w, int B, int T, int H, int V, int steps) {
  if (!bh || !ph || !w || B <= 0 || T <= 0 || T > 64 || H < 64 || H > 512 || H % 64 || V <= 0 || V > 512 || steps <= 0 || steps > 64)
    return MSOCR_E_ARG;
  if (!w->h2h_wt || !w->h2h_b || !w->score_w || !w->wih_ctx_t || !w->wih_tok || !w->whh_t || !w->b_gates || !w->gen_wt || !w->gen_b)
    return MSOCR_E_ARG;
  return MSOCR_OK;
}

extern "C" int msocr_attn_greedy(const float* batch_H, const float* proj_H, const msocr_attn_weights* w, int B, int T, int H, int V,
                                 int steps, int sos_id, int eos_id, int blank_id, float* logits_out, int32_t* ids_out, void* stream) {
  if (check_attn(batch_H, proj_H, w, B, T, H, V, steps) || !logits_out || !ids_out) return MSOCR_E_ARG;
  if (sos_id < 0 || sos_id >= V) return MSOCR_E_ARG;
  AttnArgs a{};
  a.batch_H = batch_H; a.proj_H = proj_H; a.w = *w;
  a.B = B; a.T = T; a.V = V; a.steps = steps; a.K = 1;
  a.sos_id = sos_id; a.eos_id = eos_id; a.blank_id = blank_id; a.temperature = 1.0f;
  a.logits_out = logits_out; a.ids_out = ids_out;
  if (!attn_fast_shape(T, H, V)) return msocr_internal_attn_general(a, H, false, (hipStream_t)stream);
  const size_t lds = (size_t)2 * T * ATT_H * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_greedy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 48 * ATT_H * 4) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  MSOCR_LAUNCH(attn_greedy_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, a);
  return LAUNCH_OK();
}

// mode="greedy" on the matrix cores (attn_greedy_mfma_kernel, csrc/attn_beam_mfma.hip): 32 crops per workgroup, the three per-step
// products in the split-operand form, the context half of the gate product hoisted (ctx_gates = batch_H x W_ih[:, :H]^T, [B][T][H][4],
// computed once per call by a GEMM).  Same outputs as msocr_attn_greedy; hidden 256, V <= 256, T <= 48 only.
extern "C" int msocr_attn_greedy_hoisted(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                                         const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int sos_id, int eos_id,
                                         int blank_id, float* logits_out, int32_t* ids_out, void* stream) {
  if (check_attn(batch_H, proj_H, w, B, T, H, V, steps) || !logits_out || !ids_out) return MSOCR_E_ARG;
  if (sos_id < 0 || sos_id >= V || !attn_fast_shape(T, H, V)) return MSOCR_E_ARG;
  if (!ctx_gates || ((uintptr_t)ctx_gates & 15)) return MSOCR_E_ARG;
  if (!ws || !ws->h2h_p || !ws->whh_p || !ws->gen_p || (((uintptr_t)ws->h2h_p | (uintptr_t)ws->whh_p | (uintptr_t)ws->gen_p) & 15))
    return MSOCR_E_ARG;
  AttnArgs a{};
  a.batch_H = batch_H; a.proj_H = proj_H; a.w = *w; a.ctx_gates = ctx_gates;
  a.h2h_p = ws->h2h_p; a.whh_p = ws->whh_p; a.gen_p = ws->gen_p;
  a.B = B; a.T = T; a.V = V; a.steps = steps; a.K = 1;
  a.sos_id = sos_id; a.eos_id = eos_id; a.blank_id = blank_id; a.temperature = 1.0f;
  a.logits_out = logits_out; a.ids_out = ids_out;
  return msocr_internal_attn_greedy_mfma(a, (hipStream_t)stream);
}

// workspace: logits [B][steps][K][V] f32 | back [B][steps][K] i32 | tokv [B][steps][K] i32 | best_at [B][steps] i32
static inline int64_t beam_ws_logits(int B, int steps, int K, int V) { return (int64_t)B * steps * K * V * 4; }
extern "C" int64_t msocr_attn_beam_workspace_bytes(int B, int steps, int beam, int V) {
  if (B <= 0 || steps <= 0 || beam <= 0 || V <= 0) return 0;
  return beam_ws_logits(B, steps, beam, V) + (int64_t)B * steps * beam * 8 + (int64_t)B * steps * 4 + 256;
}

static int attn_beam_impl(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                          const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int beam, const float* lp_dev, float temperature, int sos_id, int eos_id, int blank_id,
                          int32_t* fin_step_out, void* workspace, const int32_t* chunk_id_dev, const int32_t* chunk_size_dev,
                          int32_t* chunk_state_dev, void* stream);

extern "C" int msocr_attn_beam(const float* batch_H, const float* proj_H, const msocr_attn_weights* w, int B, int T, int H, int V,
                               int steps, int beam, const float* lp_dev, float temperature, int sos_id, int eos_id, int blank_id,
                               int32_t* fin_step_out, void* workspace, const int32_t* chunk_id_dev, const int32_t* chunk_size_dev,
                               int32_t* chunk_state_dev, void* stream) {
  return attn_beam_impl(batch_H, proj_H, nullptr, w, nullptr, B, T, H, V, steps, beam, lp_dev, temperature, sos_id, eos_id, blank_id, fin_step_out,
                        workspace, chunk_id_dev, chunk_size_dev, chunk_state_dev, stream);
}

extern "C" int msocr_attn_beam_hoisted(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                                       const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int beam,
                                       const float* lp_dev, float temperature, int sos_id, int eos_id, int blank_id, int32_t* fin_step_out,
                                       void* workspace, const int32_t* chunk_id_dev, const int32_t* chunk_size_dev,
                                       int32_t* chunk_state_dev, void* stream) {
  if (!ctx_gates || ((uintptr_t)ctx_gates & 15)) return MSOCR_E_ARG;
  if (ws && (!ws->h2h_p || !ws->whh_p || !ws->gen_p || (((uintptr_t)ws->h2h_p | (uintptr_t)ws->whh_p | (uintptr_t)ws->gen_p) & 15)))
    return MSOCR_E_ARG;
  return attn_beam_impl(batch_H, proj_H, ctx_gates, w, ws, B, T, H, V, steps, beam, lp_dev, temperature, sos_id, eos_id, blank_id,
                        fin_step_out, workspace, chunk_id_dev, chunk_size_dev, chunk_state_dev, stream);
}

static int attn_beam_impl(const float* batch_H, const float* proj_H, const float* ctx_gates, const msocr_attn_weights* w,
                          const msocr_attn_split_weights* ws, int B, int T, int H, int V, int steps, int beam, const float* lp_dev, float temperature, int sos_id, int eos_id, int blank_id,
                          int32_t* fin_step_out, void* workspace, const int32_t* chunk_id_dev, const int32_t* chunk_size_dev,
                          int32_t* chunk_state_dev, void* stream) {
  if (check_attn(batch_H, proj_H, w, B, T, H, V, steps) || !fin_step_out || !workspace) return MSOCR_E_ARG;
  if (beam < 1 || beam > 16 || sos_id < 0 || sos_id >= V || ((uintptr_t)workspace & 15)) return MSOCR_E_ARG;
  AttnArgs a{};
  a.batch_H = batch_H; a.proj_H = proj_H; a.w = *w; a.ctx_gates = ctx_gates;
  if (ws) { a.h2h_p = ws->h2h_p; a.whh_p = ws->whh_p; a.gen_p = ws->gen_p; }
  a.B = B; a.T = T; a.V = V; a.steps = steps; a.K = beam;
  a.sos_id = sos_id; a.eos_id = eos_id; a.blank_id = blank_id; a.temperature = temperature; a.lp = lp_dev;
  char* p = (char*)workspace;
  a.logits_out = (float*)p; p += beam_ws_logits(B, steps, beam, V);
  a.back = (int32_t*)p; p += (int64_t)B * steps * beam * 4;
  a.tokv = (int32_t*)p; p += (int64_t)B * steps * beam * 4;
  a.best_at = (int32_t*)p;
  a.fin_step = fin_step_out;
  if (chunk_id_dev && chunk_size_dev && chunk_state_dev) { a.chunk_id = chunk_id_dev; a.chunk_size = chunk_size_dev; a.chunk_state = chunk_state_dev; }
  if (!attn_fast_shape(T, H, V) || beam > ATT_KMAX) {  // other hidden sizes, charsets above 256, beams above 8: the general kernel
    if (ctx_gates) return MSOCR_E_ARG;
    return msocr_internal_attn_general(a, H, true, (hipStream_t)stream);
  }
  // NB = 2 batch rows per workgroup when their encoder rows fit in LDS beside the 68 KB of state (T <= 20), else 1
  // two batch rows per workgroup (two 256-thread halves sharing the weight stream through L1) when both rows' encoder
  // tiles fit in LDS beside 2 x 34 KB of state (T <= 20); MSOCR_BEAM_HB=1 forces one row per workgroup
  int HB = 1;  // measured: two rows per workgroup (MSOCR_BEAM_HB=2) runs at the same speed (6.6 ms per 960 rows), so the
  {            // kernel is bound by per-wave issue/latency, not by the L2 -> L1 weight stream; one row stays the default
    const char* e = getenv("MSOCR_BEAM_HB");
    if (e && e[0] == '2' && T <= 20 && B >= 2) HB = 2;
  }
  // default: the matrix-core kernel (4 crops x 8 beams per workgroup); MSOCR_BEAM_MFMA=0 selects the VALU kernel below
  const char* em = getenv("MSOCR_BEAM_MFMA");  // read per call: tests switch kernels inside one process
  const bool use_mfma = !(em && em[0] == '0');
  if (use_mfma && HB == 1) return msocr_internal_attn_beam_mfma(a, (hipStream_t)stream);
  a.ctx_gates = nullptr;  // the VALU kernels form the context vector themselves
  const size_t lds = (size_t)2 * HB * T * ATT_H * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_beam_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 48 * ATT_H * 4) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_beam_kernel<1, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 48 * ATT_H * 4) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_beam_kernel<1, 1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 48 * ATT_H * 4) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_beam_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 20 * ATT_H * 4) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  static const int pf = getenv("MSOCR_BEAM_PF") ? atoi(getenv("MSOCR_BEAM_PF")) : 4;
  if (HB == 2)
    MSOCR_LAUNCH((attn_beam_kernel<1, 2>), dim3((B + 1) / 2), dim3(512), lds, (hipStream_t)stream, a);
  else if (pf == 8)
    MSOCR_LAUNCH((attn_beam_kernel<1, 1, 8>), dim3(B), dim3(256), lds, (hipStream_t)stream, a);
  else if (pf == 16)
    MSOCR_LAUNCH((attn_beam_kernel<1, 1, 16>), dim3(B), dim3(256), lds, (hipStream_t)stream, a);
  else
    MSOCR_LAUNCH((attn_beam_kernel<1, 1>), dim3(B), dim3(256), lds, (hipStream_t)stream, a);
  return LAUNCH_OK();
}

extern "C" int msocr_attn_beam_finalize(const void* workspace, int B, int V, int steps, int beam, const int32_t* trun_dev,
                                        float* logits_out, int32_t* ids_out, void* stream) {
  if (!workspace || !trun_dev || !logits_out || !ids_out || B <= 0 || V <= 0 || steps <= 0 || steps > 64 || beam < 1 || beam > 16)
    return MSOCR_E_ARG;
  const char* p = (const char*)workspace;
  const float* wl = (const float*)p; p += beam_ws_logits(B, steps, beam, V);
  const int32_t* back = (const int32_t*)p; p += (int64_t)B * steps * beam * 4;
  const int32_t* tokv = (const int32_t*)p; p += (int64_t)B * steps * beam * 4;
  const int32_t* best_at = (const int32_t*)p;
  MSOCR_LAUNCH(attn_beam_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, wl, back, tokv, best_at, trun_dev, V, steps, beam,
               logits_out, ids_out);
  return LAUNCH_OK();
}

// --------------------------------------------------------------------------------------------- confidence
// TRBA.predict's confidence (recognizers/_trba/__init__.py:413-431): log_softmax over V of the returned logits,
// exp of the chosen token's log-prob, mean over ALL t_run generated positions.  One wave per row.
__global__ __launch_bounds__(64) void seq_confidence_kernel(const float* __restrict__ logits, const int32_t* __restrict__ ids,
                                                             const int32_t* __restrict__ trun, int V, int steps,
                                                             float* __restrict__ conf) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int tr = trun[b];
  float acc = 0.f;
  for (int t = 0; t < tr; ++t) {
    const float* x = logits + ((long)b * steps + t) * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
    s = wave_sum(s);
    const int id = ids[(long)b * steps + t];
    const float logp = (x[id] - m) - logf(s);
    acc += expf(logp);
  }
  if (lane == 0) conf[b] = tr > 0 ? acc / (float)tr : 0.f;
}

extern "C" int msocr_seq_confidence(const float* logits, const int32_t* ids, const int32_t* trun_dev, int B, int V, int steps,
                                    float* conf_out, void* stream) {
  if (!logits || !ids || !trun_dev || !conf_out || B <= 0 || V <= 0 || steps <= 0) return MSOCR_E_ARG;
  MSOCR_LAUNCH(seq_confidence_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ids, trun_dev, V, steps, conf_out);
  return LAUNCH_OK();
}
