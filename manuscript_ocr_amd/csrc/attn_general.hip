// attn_general.hip — attention decode of the TRBA recogniser for the shapes the two fast kernels do not take:
// hidden sizes other than 256 (the reference reads hidden_size from the checkpoint's config, recognizers/_trba/__init__.py:142-151),
// charsets of up to 512 tokens, beam widths of up to 16 (TRBA.predict exposes beam_size, :295-299; the reference's own Optuna
// script sweeps 2..12).  Same arithmetic, outputs, workspace layout and tie rules as attn_greedy_kernel / attn_beam_kernel
// (trba_kernels.hip): one 256-thread workgroup per batch row runs the whole step loop, the K beam rows' state (h, c, context,
// h2h projection, logits) lives in LDS, every phase is a strided loop over (row, unit) or (row, token) pairs, weights stream from
// L2.  Written for generality, not speed: the default configuration (hidden 256, 194 tokens, beam 8) never comes here.
//
// Replaces recognizers/_trba/model/model.py:34-46 (AttentionCell.forward), :227-259 (_greedy_decode), :92-225 (_beam_decode).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "internal.h"
#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NTH = 256;
constexpr int KCAP = 16;  // beam rows

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide arg-max, "larger value first, then smaller index"; NaN never wins
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
  for (int w = 1; w < NTH / 64; ++w)
    if (s_val[w] > out_v || (s_val[w] == out_v && s_idx[w] < out_i)) { out_v = s_val[w]; out_i = s_idx[w]; }
  __syncthreads();
}

// dynamic LDS: h [K][H] | c [K][H] | ctx [K][H] | ph [K][H] | logits [K][V] | alpha [K][64]
template <bool BEAM>
__global__ __launch_bounds__(NTH) void attn_general_kernel(AttnArgs a, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = BEAM ? a.K : 1, T = a.T, V = a.V;
  float* sh = lds;
  float* sc = sh + K * H;
  float* sctx = sc + K * H;
  float* sph = sctx + K * H;
  float* slog = sph + K * H;
  float* salpha = slog + K * V;
  __shared__ float s_val[NTH / 64];
  __shared__ int s_idx[NTH / 64];
  __shared__ float s_score[KCAP], s_lse[KCAP], s_top[KCAP];
  __shared__ int s_tok[KCAP], s_done[KCAP], s_src[KCAP], s_nxt[KCAP];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* bH = a.batch_H + (long)b * T * H;
  const float* pH = a.proj_H + (long)b * T * H;
  for (int i = tid; i < K * H; i += NTH) { sh[i] = 0.f; sc[i] = 0.f; }
  if (tid < K) {
    s_score[tid] = tid == 0 ? 0.f : -INFINITY;
    s_tok[tid] = a.sos_id;
    s_done[tid] = 0;
  }
  __syncthreads();
  int fin = a.steps;
  const float temp = fmaxf(a.temperature, 1e-6f);
  for (int s = 0; s < a.steps; ++s) {
    // (a) ph[r][j] = h2h_b[j] + sum_k h[r][k] * h2h_wt[k][j]
    for (int i = tid; i < K * H; i += NTH) {
      const int r = i / H, j = i - r * H;
      float acc = a.w.h2h_b[j];
      for (int k = 0; k < H; ++k) acc = fmaf(a.w.h2h_wt[(long)k * H + j], sh[r * H + k], acc);
      sph[i] = acc;
    }
    __syncthreads();
    // (b) e[r][t] = sum_j score_w[j] * tanh(proj_H[t][j] + ph[r][j]) : one wave per (r, t)
    for (int p = wv; p < K * T; p += NTH / 64) {
      const int r = p / T, t = p - r * T;
      float e = 0.f;
      for (int j = lane; j < H; j += 64) e = fmaf(a.w.score_w[j], tanhf(pH[t * H + j] + sph[r * H + j]), e);
      e = wsum(e);
      if (lane == 0) salpha[r * 64 + t] = e;
    }
    __syncthreads();
    // (c) softmax over t
    if (tid < K) {
      float m = -INFINITY;
      for (int t = 0; t < T; ++t) m = fmaxf(m, salpha[tid * 64 + t]);
      float sum = 0.f;
      for (int t = 0; t < T; ++t) {
        const float ev = expf(salpha[tid * 64 + t] - m);
        salpha[tid * 64 + t] = ev;
        sum += ev;
      }
      for (int t = 0; t < T; ++t) salpha[tid * 64 + t] = salpha[tid * 64 + t] / sum;
    }
    __syncthreads();
    // (d) ctx[r][j] = sum_t alpha[r][t] * batch_H[t][j]
    for (int i = tid; i < K * H; i += NTH) {
      const int r = i / H, j = i - r * H;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc = fmaf(salpha[r * 64 + t], bH[t * H + j], acc);
      sctx[i] = acc;
    }
    __syncthreads();
    // (e) LSTMCell: gates = W_ih[:, :H] ctx + W_ih[:, H + tok] + W_hh h + b ; new h into ph (dead since (b)), c in place
    for (int i = tid; i < K * H; i += NTH) {
      const int r = i / H, j = i - r * H;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&a.w.b_gates[j * 4]);
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(&a.w.wih_tok[((long)s_tok[r] * H + j) * 4]);
      f32x4 g = b4 + t4;
      for (int k = 0; k < H; ++k) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&a.w.wih_ctx_t[((long)k * H + j) * 4]);
        const float x = sctx[r * H + k];
        g[0] = fmaf(w4[0], x, g[0]); g[1] = fmaf(w4[1], x, g[1]); g[2] = fmaf(w4[2], x, g[2]); g[3] = fmaf(w4[3], x, g[3]);
      }
      for (int k = 0; k < H; ++k) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&a.w.whh_t[((long)k * H + j) * 4]);
        const float x = sh[r * H + k];
        g[0] = fmaf(w4[0], x, g[0]); g[1] = fmaf(w4[1], x, g[1]); g[2] = fmaf(w4[2], x, g[2]); g[3] = fmaf(w4[3], x, g[3]);
      }
      const float ig = sigm(g[0]), fg = sigm(g[1]), gg = tanhf(g[2]), og = sigm(g[3]);
      const float cn = fg * sc[i] + ig * gg;
      sc[i] = cn;
      sph[i] = og * tanhf(cn);
    }
    __syncthreads();
    for (int i = tid; i < K * H; i += NTH) sh[i] = sph[i];
    __syncthreads();
    // (f) logits[r][v] = gen_b[v] + sum_k h'[r][k] * gen_wt[k][v]
    for (int i = tid; i < K * V; i += NTH) {
      const int r = i / V, v = i - r * V;
      float acc = a.w.gen_b[v];
      for (int k = 0; k < H; ++k) acc = fmaf(a.w.gen_wt[(long)k * V + v], sh[r * H + k], acc);
      slog[i] = (v == a.blank_id) ? -1e4f : acc;
    }
    __syncthreads();
    if constexpr (!BEAM) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int v = tid; v < V; v += NTH) {
        const float x = slog[v];
        a.logits_out[((long)b * a.steps + s) * V + v] = x;
        if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
      }
      float wv_;
      int wi_;
      block_argmax(bv, bi, s_val, s_idx, tid, wv_, wi_);
      if (wi_ == 0x7fffffff) wi_ = 0;
      if (tid == 0) {
        s_tok[0] = wi_;
        a.ids_out[(long)b * a.steps + s] = wi_;
      }
      __syncthreads();
    } else {
      const int KB = K;
      // temperature (true f32 division, model.py:135-137) + trace
      for (int i = tid; i < K * V; i += NTH) {
        float x = slog[i];
        if (a.temperature != 1.0f) x = x / temp;
        slog[i] = x;
        const int r = i / V, v = i - r * V;
        a.logits_out[(((long)b * a.steps + s) * KB + r) * V + v] = x;
      }
      __syncthreads();
      for (int r = wv; r < K; r += NTH / 64) {  // log-sum-exp per row
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, slog[r * V + v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int v = lane; v < V; v += 64) sum += expf(slog[r * V + v] - m);
        sum = wsum(sum);
        if (lane == 0) s_lse[r] = m + logf(sum);
      }
      __syncthreads();
      // candidates in place: (score[r] + logp[r][v]) / lp ; finished beams: only EOS with logp 0 (model.py:140-160)
      const float lp = a.lp ? a.lp[s] : 1.0f;
      for (int i = tid; i < K * V; i += NTH) {
        const int r = i / V, v = i - r * V;
        float logp = slog[i] - s_lse[r];
        if (s_done[r]) logp = (v == a.eos_id) ? 0.f : -INFINITY;
        float tot = s_score[r] + logp;
        if (a.lp) tot = tot / lp;
        slog[i] = tot;
      }
      __syncthreads();
      for (int kk = 0; kk < KB; ++kk) {  // top-K by K rounds of block arg-max over the flat index r * V + v
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < K * V; i += NTH) {
          const float x = slog[i];
          if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
        }
        float wv_;
        int wi_;
        block_argmax(bv, bi, s_val, s_idx, tid, wv_, wi_);
        if (wi_ == 0x7fffffff) wi_ = 0;  // every candidate NaN: degenerate input
        if (tid == 0) {
          s_top[kk] = wv_;
          s_src[kk] = wi_ / V;
          s_nxt[kk] = wi_ % V;
          slog[wi_] = __int_as_float(0x7fc00000);  // the winner never wins again
        }
        __syncthreads();
      }
      // reorder the beam state by src through ctx / ph (both dead here)
      for (int i = tid; i < K * H; i += NTH) {
        const int r = i / H, j = i - r * H;
        sctx[i] = sh[s_src[r] * H + j];
        sph[i] = sc[s_src[r] * H + j];
      }
      __syncthreads();
      for (int i = tid; i < K * H; i += NTH) { sh[i] = sctx[i]; sc[i] = sph[i]; }
      int nd = 1;
      if (tid < K) nd = s_done[s_src[tid]] | (s_nxt[tid] == a.eos_id);
      __syncthreads();
      if (tid < K) {
        const long o = ((long)b * a.steps + s) * KB + tid;
        a.back[o] = s_src[tid];
        a.tokv[o] = s_nxt[tid];
        s_score[tid] = a.lp ? s_top[tid] * lp : s_top[tid];  // f32 round trip of the reference (model.py:188-192)
        s_tok[tid] = s_nxt[tid];
        s_done[tid] = nd;
      }
      __syncthreads();
      if (tid == 0) {
        int best = 0, all = 1;
        float bs = s_score[0];
        for (int r = 0; r < KB; ++r) {
          if (r && s_score[r] > bs) { bs = s_score[r]; best = r; }
          all &= s_done[r];
        }
        a.best_at[(long)b * a.steps + s] = best;
        s_idx[0] = all;
      }
      __syncthreads();
      if (s_idx[0] && fin == a.steps) fin = s + 1;
      __syncthreads();
    }
  }
  if (BEAM && tid == 0) a.fin_step[b] = fin;
}

}  // namespace

// Shapes taken: H % 64 == 0, 64 <= H <= 512, V <= 512, T <= 64, steps <= 64, beam <= 16, beam * H <= 4096 (the state of the beam
// rows must fit in LDS).  beam = 0 selects the greedy loop.
int msocr_internal_attn_general(const AttnArgs& a, int H, bool beam, hipStream_t s) {
  const int K = beam ? a.K : 1;
  if (H < 64 || H > 512 || H % 64 || a.V <= 0 || a.V > 512 || a.T <= 0 || a.T > 64 || a.steps <= 0 || a.steps > 64 || K < 1 || K > KCAP ||
      K * H > 4096)
    return MSOCR_E_ARG;
  const size_t ldsz = (size_t)(4 * K * H + K * a.V + K * 64) * sizeof(float);
  auto kb = attn_general_kernel<true>;
  auto kg = attn_general_kernel<false>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  if (ldsz > 112 * 1024) return MSOCR_E_ARG;
  if (beam)
    MSOCR_LAUNCH(kb, dim3(a.B), dim3(NTH), ldsz, s, a, H);
  else
    MSOCR_LAUNCH(kg, dim3(a.B), dim3(NTH), ldsz, s, a, H);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}
