// attn_beam_mfma.hip — beam-search attention decode of the TRBA recogniser with the step's three matrix products on
// the matrix cores: in the split-operand form (default, precision "fp32": every f32 operand as three bf16 terms, six
// v_mfma_f32_32x32x16_bf16 partial products per 16 k, f32 accumulation) or on the exact-f32 pipe (v_mfma_f32_32x32x2_f32,
// precision "fp32-exact" or no msocr_attn_split_weights).
//
// One 512-thread workgroup owns NB = 4 crops x 8 beam slots = 32 state rows for the whole step loop (rows of different
// crops are independent: no inter-workgroup hand-off).  32 rows are exactly one MFMA row block, so every weight element a
// workgroup pulls from L2 now feeds 32 rows instead of the 8 of the VALU kernel (trba_kernels.hip): the L2 -> CU weight
// stream that bounded that kernel (2.5 MB per step per workgroup) is shared by 4x more work, and the gate arithmetic
// moves off the VALU.  Per step:
//   (a) ph    = h2h(h)                     [32x256] x [256x256]        MFMA  (wave w: columns 32w..32w+31)
//   (b) e     = score . tanh(proj_H + ph)  32 x T dot products          VALU  (one wave per (row, t) pair)
//   (c) alpha = softmax_t(e)                                            VALU
//   (d) ctx   = alpha . batch_H                                         VALU
//   (e) gates = [ctx | h] x [W_ih_ctx ; W_hh]^T + W_ih_tok[token] + b   [32x512] x [512x1024]  MFMA
//       (wave w owns hidden units 32w..32w+31; one 16-byte load per lane = the unit's 4 gates = B operands of 4 MFMAs,
//        so the LSTM cell update is lane-local in the accumulator layout)
//   (f) logits = generator(h')             [32x256] x [256xV]          MFMA
//   (g-j) temperature, log-softmax, top-8 of 8*V candidates per crop, back-pointers, beam state permutation   VALU/LDS
// Same outputs, workspace layout and tie rules as attn_beam_kernel (larger value first, then smaller flat index).
//
// Replaces recognizers/_trba/model/model.py:34-46 (AttentionCell.forward) + :92-225 (Attention._beam_decode).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "internal.h"
#include "msocr.h"
#include "split_rows32.h"

using split_rows32::acc_row;
using split_rows32::f32x16;
using split_rows32::mfma_cols32_split;
using split_rows32::mfma_gates_split;
using split_rows32::PPL;
using split_rows32::PSB;
using split_rows32::split_pair;
using split_rows32::u32x4;

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Per-phase timestamps of workgroup 0 (dev builds only: tools/attn_phase_times.sh compiles this file with -DMSOCR_ATTN_TIMING)
#ifdef MSOCR_ATTN_TIMING
__device__ unsigned long long msocr_attn_timing[64 * 16];
extern "C" int msocr_attn_timing_read(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(msocr_attn_timing), sizeof(msocr_attn_timing)) == hipSuccess ? 0 : -2;
}
#define TSTAMP(ph) do { if (blockIdx.x == 0 && threadIdx.x == 0 && s < 64) msocr_attn_timing[s * 16 + (ph)] = wall_clock64(); } while (0)
#else
#define TSTAMP(ph) do { } while (0)
#endif


namespace {

constexpr int H = split_rows32::H;  // hidden size (ATT_H)
constexpr int KB8 = 8;        // beam slots per crop (ATT_KMAX)
constexpr int NB = 4;         // crops per workgroup
constexpr int R = NB * KB8;   // 32 state rows = one MFMA row block
static_assert(R == split_rows32::R, "row block");
constexpr int XS = 2 * H + 4; // row stride of X = [ctx | h] in floats: 516 -> ds_read_b128 of 32 rows is conflict-free
constexpr int NT = 512;       // threads per workgroup

// Hardware-rate transcendentals (v_exp_f32 / v_rcp_f32, ~1-2 ulp each): the decode step evaluates 32 x T x 256 tanh and
// 5 x 32 x 256 gate activations on the VALU between the matrix phases; libm-grade expf/tanhf made that the longest phase.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fexp(2.0f * x) + 1.0f); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Wave-wide reductions on the DPP data path (quad_perm / row_half_mirror / row_mirror / row_bcast15 / row_bcast31: a few cycles per
// step) instead of ds_bpermute shuffles (an LDS round trip per step): the decode step runs ~400 of them per wave.  The result is
// complete in lane 63 (every lane of the last row for sums of full rows); *_bcast return it to all lanes through an SGPR.
template <int CTRL, int RMASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, RMASK, 0xf, false));
}
template <int CTRL, int RMASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RMASK, 0xf, false); }
#define DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_HMIRROR 0x141    // row_half_mirror: lane i <-> 7 - i inside each group of 8
#define DPP_MIRROR 0x140     // row_mirror: lane i <-> 15 - i inside each row of 16
#define DPP_BCAST15 0x142    // lane 15 of every row -> the next row (row_mask 0xA: rows 1 and 3 take it)
#define DPP_BCAST31 0x143    // lane 31 -> rows 2 and 3 (row_mask 0xC)
__device__ __forceinline__ float wave_sum63(float v) {  // total in lane 63
  v += dpp_f<DPP_XOR1>(0.f, v);
  v += dpp_f<DPP_XOR2>(0.f, v);
  v += dpp_f<DPP_HMIRROR>(0.f, v);
  v += dpp_f<DPP_MIRROR>(0.f, v);
  v += dpp_f<DPP_BCAST15, 0xA>(0.f, v);
  v += dpp_f<DPP_BCAST31, 0xC>(0.f, v);
  return v;
}
__device__ __forceinline__ float wave_max63(float v) {
  v = fmaxf(v, dpp_f<DPP_XOR1>(v, v));
  v = fmaxf(v, dpp_f<DPP_XOR2>(v, v));
  v = fmaxf(v, dpp_f<DPP_HMIRROR>(v, v));
  v = fmaxf(v, dpp_f<DPP_MIRROR>(v, v));
  v = fmaxf(v, dpp_f<DPP_BCAST15, 0xA>(v, v));
  v = fmaxf(v, dpp_f<DPP_BCAST31, 0xC>(v, v));
  return v;
}
__device__ __forceinline__ float bcast63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// arg-max with the beam search's tie rule (larger value first, then smaller flat index; NaN never wins): result in lane 63
template <int CTRL, int RMASK = 0xf>
__device__ __forceinline__ void argmax_step(float& v, int& i) {
  const float ov = dpp_f<CTRL, RMASK>(v, v);
  const int oi = dpp_i<CTRL, RMASK>(i, i);
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void wave_argmax63(float& v, int& i) {
  argmax_step<DPP_XOR1>(v, i);
  argmax_step<DPP_XOR2>(v, i);
  argmax_step<DPP_HMIRROR>(v, i);
  argmax_step<DPP_MIRROR>(v, i);
  argmax_step<DPP_BCAST15, 0xA>(v, i);
  argmax_step<DPP_BCAST31, 0xC>(v, i);
}

// Packed-f32 VALU beside bf16 MFMAs.  The VALU work between the softmax barrier and the gate MFMAs (token-row initialisation, the
// hoisted context sum) runs while the other wave of the SIMD is inside its v_mfma_f32_32x32x16_bf16 loop.  Round 3 found that with
// the packed form the compiler chose for that sum — v_pk_fma_f32 ... op_sel:[0,1,0], the LOW result taking the HIGH dword of a
// source pair — about 0.5 % of the state rows came out with the low result of lanes 48..63 wrong (tools/attn_packed_probe.sh,
// profiles/r03_attn_packed_probe.txt).  Round 4 settled the cause: the disassembly of that build has hundreds of cycles (a branch,
// twelve buffer loads, an s_waitcnt vmcnt) between the last v_pk_fma_f32 that writes an accumulator and the first MFMA that reads it
// as SrcC, so it is no missing VALU -> MFMA wait state; and the 90-line stand-alone tools/microbench/pk_fma_beside_mfma.hip
// reproduces it with nothing else in the kernel — 22 880 wrong low results, all in lanes 48..63, out of 2.6e9 with the partner waves
// running v_mfma_f32_32x32x16_bf16; 0 with idle partners; 0 for the same instruction without op_sel (profiles/r04_pk_fma_probe.txt).
// It is a property of the instruction beside MFMAs, not of this kernel's code.  The remedy is at build level: this file, like every
// translation unit whose kernels run beside bf16 MFMAs, is compiled without packed-f32 instructions (csrc/Makefile, NOPK_OBJS), the
// sums below are plain C (the compiler's hazard recognizer sees them — round 3's inline asm hid them from it), and
// tests/test_host_cpu.py::test_no_packed_f32_valu_beside_mfma disassembles the built objects.  tests/test_gpu_trba.py keeps the
// split-against-exact comparison of every beam's logits as the run-time guard.
__device__ __forceinline__ float add_np(float a, float b) { return a + b; }
__device__ __forceinline__ float fmac_np(float a, float b, float c) { return fmaf(a, b, c); }


// Weight streams use buffer loads: ONE per-lane byte offset in a VGPR (loop-invariant) + a scalar byte offset per load, so
// the 12-16 loads in flight cost no address VGPRs (a global_load needs a 64-bit VGPR address each); out-of-range lanes
// (generator columns >= V) read 0 by the buffer range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int nfloats) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nfloats * 4, 0x00020000);
}

// D[32 rows][32 columns of this wave] += X[:, k0 : k0+256] * W[k][col], W row-major with `ldw` floats per k, one dword per lane.
__device__ __forceinline__ void mfma_cols32(const float* __restrict__ sX, int k0, const float* __restrict__ W, int ldw, int col,
                                            bool col_ok, int r32, int half, f32x16& acc) {
  constexpr int PFQ = 4;  // k-groups (of 8) of weight loads kept in flight
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(W, H * ldw);
  const int voff = col_ok ? (4 * half * ldw + col) * 4 : 0x7ffffff0;
  const int kstep = ldw * 4;  // bytes between consecutive k
  float wb[PFQ][4];
#pragma unroll
  for (int pq = 0; pq < PFQ; ++pq)
#pragma unroll
    for (int e = 0; e < 4; ++e) wb[pq][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (8 * pq + e) * kstep, 0));
#pragma unroll 1
  for (int q0 = 0; q0 < H / 8; q0 += PFQ) {
#pragma unroll
    for (int pq = 0; pq < PFQ; ++pq) {
      const int q = q0 + pq;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sX[r32 * XS + k0 + 8 * q + 4 * half]);
      float cur[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[e] = wb[pq][e];
      const int qn = q + PFQ;
      if (qn < H / 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wb[pq][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (8 * qn + e) * kstep, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], cur[e], acc, 0, 0, 0);
    }
  }
}

// gates of hidden units 32w..32w+31: acc[g] += X[:, k0 : k0+256] * Wt[k][j][g]   (Wt gate-interleaved: [k][j][4])
__device__ __forceinline__ void mfma_gates(const float* __restrict__ sX, int k0, const float* __restrict__ Wt, int j, int r32, int half,
                                           f32x16 (&acc)[4]) {
  constexpr int PFQ = 4;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(Wt, H * H * 4);
  const int voff = (4 * half * H + j) * 16;
  constexpr int kstep = H * 16;
  f32x4 wb[PFQ][4];
#pragma unroll
  for (int pq = 0; pq < PFQ; ++pq)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      wb[pq][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (8 * pq + e) * kstep, 0));
#pragma unroll 1
  for (int q0 = 0; q0 < H / 8; q0 += PFQ) {
#pragma unroll
    for (int pq = 0; pq < PFQ; ++pq) {
      const int q = q0 + pq;
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sX[r32 * XS + k0 + 8 * q + 4 * half]);
      f32x4 cur[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[e] = wb[pq][e];
      const int qn = q + PFQ;
      if (qn < H / 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          wb[pq][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (8 * qn + e) * kstep, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], cur[e][g], acc[g], 0, 0, 0);
    }
  }
}

// HOIST: the context half of the LSTMCell input product is hoisted out of the step loop.  The reference computes
// gates = W_ih [ctx ; onehot] + W_hh h with ctx = sum_t alpha_t batch_H_t (model.py:40-45); since W_ih[:, :H] ctx =
// sum_t alpha_t (W_ih[:, :H] batch_H_t), the products P_t = W_ih[:, :H] batch_H_t are computed ONCE per crop by a GEMM before the
// kernel (a.ctx_gates, [B][T][H][4]) and a step only forms sum_t alpha_t P_t on the VALU (13 x 1024 FMAs per row instead of
// 256 x 1024 MACs): half of the step's matrix work, 1 of its 2.5 MB of weights and the ctx phase (d) disappear.  Same arithmetic
// up to the order of the f32 summation.
// SPLITW (with HOIST): the three matrix products in the split-operand form above; h lives only as its three bf16 planes.
template <bool HOIST, bool SPLITW>
__global__ __launch_bounds__(NT, 1) void attn_beam_mfma_kernel(AttnArgs a) {
  static_assert(HOIST || !SPLITW, "the split form is built for the hoisted kernel");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sX = lds;                 // [R][XS]   ctx (0..255) | h (256..511)        (exact form)
  unsigned char* sP = reinterpret_cast<unsigned char*>(lds);  // [3][R][PSB] bf16 planes of h  (split form)
  float* sbuf = SPLITW ? lds + 3 * PPL / 4 : sX + R * XS;     // [R][H]    ph, then logits, then scratch of the state permutation
  float* salpha = sbuf + R * H;    // [R][64]
  __shared__ float s_score[R], s_lse[R], s_top[R];
  __shared__ int s_tok[R], s_done[R], s_src[R], s_nxt[R], s_fin[NB], s_exit;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const int T = a.T, V = a.V, KB = a.K;
  const int b0 = blockIdx.x * NB;
  const int ju = 32 * wv + r32;    // hidden unit / output column owned in the MFMA phases

  if constexpr (SPLITW) {
    for (int i = tid; i < 3 * PPL / 4; i += NT) lds[i] = 0.f;  // h = 0
  } else {
    for (int i = tid; i < R * H; i += NT) sX[(i >> 8) * XS + H + (i & 255)] = 0.f;
  }
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  if (tid < R) {
    s_score[tid] = (tid % KB8) == 0 ? 0.f : -INFINITY;
    s_tok[tid] = a.sos_id;
    s_done[tid] = 0;
  }
  if (tid < NB) s_fin[tid] = a.steps;
  __syncthreads();
  const float temp = fmaxf(a.temperature, 1e-6f);

  for (int s = 0; s < a.steps; ++s) {
    TSTAMP(0);
    // ---- (a) ph[r][j] = h2h_b[j] + sum_k h[r][k] * h2h_wt[k][j]
    {
      f32x16 acc;
      const float bj = a.w.h2h_b[ju];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = bj;
      if constexpr (SPLITW) mfma_cols32_split(sP, a.h2h_p, H, ju, true, r32, half, acc);
      else mfma_cols32(sX, H, a.w.h2h_wt, H, ju, true, r32, half, acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) sbuf[acc_row(e, half) * H + ju] = acc[e];
    }
    __syncthreads();
    TSTAMP(1);
    // ---- (b) e[r][t] = sum_j score_w[j] * tanh(proj_H[crop][t][j] + ph[r][j]) : one wave per (crop, t) group — the
    //      proj_H row is read once for the crop's 8 beams; the rows of up to GB groups are in flight together
    {
      constexpr int GB = 4;
      float sw[H / 64];
#pragma unroll
      for (int q = 0; q < H / 64; ++q) sw[q] = a.w.score_w[lane + 64 * q];
      const int ngroups = NB * T;
      for (int g0 = wv; g0 < ngroups; g0 += GB * (NT / 64)) {
        float pr[GB][H / 64];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int g = g0 + u * (NT / 64);
          if (g < ngroups) {
            const int nb = g / T, t = g - nb * T;
            const float* pP = a.proj_H + ((long)min(b0 + nb, a.B - 1) * T + t) * H;
#pragma unroll
            for (int q = 0; q < H / 64; ++q) pr[u][q] = pP[lane + 64 * q];
          }
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int g = g0 + u * (NT / 64);
          if (g < ngroups) {
            const int nb = g / T, t = g - nb * T;
            float sacc[KB8];
#pragma unroll
            for (int rb = 0; rb < KB8; ++rb) {
              const int r = nb * KB8 + rb;
              sacc[rb] = 0.f;
#pragma unroll
              for (int q = 0; q < H / 64; ++q) sacc[rb] = fmaf(sw[q], ftanh(pr[u][q] + sbuf[r * H + lane + 64 * q]), sacc[rb]);
            }
            // 8 independent wave sums on the DPP path, totals in lane 63
#pragma unroll
            for (int rb = 0; rb < KB8; ++rb) sacc[rb] = wave_sum63(sacc[rb]);
            if (lane == 63) {
#pragma unroll
              for (int rb = 0; rb < KB8; ++rb) salpha[(nb * KB8 + rb) * 64 + t] = sacc[rb];
            }
          }
        }
      }
    }
    __syncthreads();
    TSTAMP(2);
    // ---- (c) softmax over t: wave w handles rows 4w..4w+3, lane = t (T <= 64)
    for (int r = 4 * wv; r < 4 * wv + 4; ++r) {
      const float ev0 = lane < T ? salpha[r * 64 + lane] : -INFINITY;
      const float m = bcast63(wave_max63(ev0));
      const float ev = lane < T ? expf(ev0 - m) : 0.f;
      const float sum = bcast63(wave_sum63(ev));
      if (lane < T) salpha[r * 64 + lane] = ev / sum;
    }
    __syncthreads();
    TSTAMP(3);
    // ---- (d) ctx[r][j] = sum_t alpha[r][t] * batch_H[crop(r)][t][j] : thread = (j, half of the crops)
    if constexpr (!HOIST) {
      const int j = tid & 255, ch = tid >> 8;
#pragma unroll
      for (int n2 = 0; n2 < NB / 2; ++n2) {
        const int nb = ch * (NB / 2) + n2;
        const int b = min(b0 + nb, a.B - 1);
        const float* pH = a.batch_H + (long)b * T * H + j;
        float accd[KB8];
#pragma unroll
        for (int q = 0; q < KB8; ++q) accd[q] = 0.f;
        for (int t0 = 0; t0 < T; t0 += 16) {  // 16 independent loads in flight, then the FMAs
          float hv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) hv[u] = (t0 + u < T) ? pH[(long)(t0 + u) * H] : 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (t0 + u < T) {
#pragma unroll
              for (int q = 0; q < KB8; ++q) accd[q] = fmaf(salpha[(nb * KB8 + q) * 64 + t0 + u], hv[u], accd[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < KB8; ++q) sX[(nb * KB8 + q) * XS + j] = accd[q];
      }
    }
    if constexpr (!HOIST) __syncthreads();
    TSTAMP(4);
    // ---- (e) gates + LSTM cell for units ju, rows acc_row(e, half)
    {
      f32x16 acc[4];
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&a.w.b_gates[ju * 4]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tk = s_tok[acc_row(e, half)];
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&a.w.wih_tok[((long)tk * H + ju) * 4]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][e] = add_np(b4[g], t4[g]);
      }
      // + sum_t alpha[row][t] * P[crop(row)][t][ju][gate]; this lane's rows of crop nb are beams 4 * half + 0..3 = acc elements 4 nb + i
      auto ctx_sum = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float* pP = a.ctx_gates + ((long)min(b0 + nb, a.B - 1) * T * H + ju) * 4;
          const float* pa = salpha + (nb * KB8 + 4 * half) * 64;
          for (int t0 = 0; t0 < T; t0 += 8) {  // 8 loads in flight, then the FMAs
            f32x4 pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (t0 + u < T) pv[u] = *reinterpret_cast<const f32x4*>(pP + (long)(t0 + u) * H * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (t0 + u < T) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float al = pa[i * 64 + t0 + u];
#pragma unroll
                  for (int g = 0; g < 4; ++g) acc[g][4 * nb + i] = fmac_np(al, pv[u][g], acc[g][4 * nb + i]);
                }
              }
          }
        }
      };
#if defined(MSOCR_ATTN_SUM_BARRIER) || defined(MSOCR_ATTN_NO_STAGGER)
      constexpr bool STAGGER = false;
#else
      constexpr bool STAGGER = HOIST && SPLITW;
#endif
      if constexpr (STAGGER) {
        // The context sum is load-latency and VALU work, the recurrent product matrix-pipe work, and the two are independent: the two
        // waves of a SIMD (w and w + 4) take them in opposite orders, so one's loads and FMAs run under the other's MFMAs instead of
        // both waiting for memory and then both queueing on the pipe (one-register FMAs: see add_np / fmac_np above).
        const bool mfma_first = __builtin_amdgcn_readfirstlane(wv) >= 4;
        if (!mfma_first) ctx_sum();
        mfma_gates_split(sP, a.whh_p, ju, r32, half, acc);
        if (mfma_first) ctx_sum();
      } else {
        if constexpr (HOIST) {
          ctx_sum();
#ifdef MSOCR_ATTN_SUM_BARRIER  // dev builds: no wave enters the MFMA loop while another is still in the sum above
          __syncthreads();
#endif
        } else {
          mfma_gates(sX, 0, a.w.wih_ctx_t, ju, r32, half, acc);
        }
        if constexpr (SPLITW) mfma_gates_split(sP, a.whh_p, ju, r32, half, acc);
        else mfma_gates(sX, H, a.w.whh_t, ju, r32, half, acc);
      }
      __syncthreads();  // every wave has read the old h
      float hv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ig = sigmoidf_(acc[0][e]), fg = sigmoidf_(acc[1][e]), gg = ftanh(acc[2][e]), og = sigmoidf_(acc[3][e]);
        c[e] = fg * c[e] + ig * gg;
        hv[e] = og * ftanh(c[e]);
      }
      if constexpr (SPLITW) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) {  // acc_row(e + 1) == acc_row(e) + 1
          unsigned char* d = sP + acc_row(e, half) * PSB + ju * 2;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const uint32_t pk = split_pair(hv[e], hv[e + 1]);
            *reinterpret_cast<uint16_t*>(d + pl * PPL) = (uint16_t)pk;
            *reinterpret_cast<uint16_t*>(d + pl * PPL + PSB) = (uint16_t)(pk >> 16);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) sX[acc_row(e, half) * XS + H + ju] = hv[e];
      }
    }
    __syncthreads();
    TSTAMP(5);
    // ---- (f) logits[r][v] = gen_b[v] + sum_k h'[r][k] * gen_wt[k][v]; temperature; trace store
    {
      const bool vok = ju < V;
      f32x16 acc;
      const float bv = vok ? a.w.gen_b[ju] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = bv;
      if (32 * wv < V) {
        if constexpr (SPLITW) mfma_cols32_split(sP, a.gen_p, (V + 31) & ~31, ju, true, r32, half, acc);  // padded columns are zeros
        else mfma_cols32(sX, H, a.w.gen_wt, V, ju, vok, r32, half, acc);
      }
      if (vok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = acc_row(e, half);
          float v = (ju == a.blank_id) ? -1e4f : acc[e];
          if (a.temperature != 1.0f) v = v / temp;  // true f32 division (model.py:135-137)
          sbuf[r * H + ju] = v;
          const int b = b0 + r / KB8, rb = r % KB8;
          if (rb < KB && b < a.B) a.logits_out[(((long)b * a.steps + s) * KB + rb) * V + ju] = v;
        }
      }
    }
    __syncthreads();
    TSTAMP(6);
    // ---- (g) log-sum-exp per row: wave w handles rows 4w..4w+3
    for (int r = 4 * wv; r < 4 * wv + 4; ++r) {
      float m = -INFINITY;
      for (int v = lane; v < V; v += 64) m = fmaxf(m, sbuf[r * H + v]);
      m = bcast63(wave_max63(m));
      float sum = 0.f;
      for (int v = lane; v < V; v += 64) sum += expf(sbuf[r * H + v] - m);
      sum = wave_sum63(sum);
      if (lane == 63) s_lse[r] = m + logf(sum);  // libm-grade exp / log: the scores of a beam search are sums of these
    }
    __syncthreads();
    TSTAMP(7);
    // ---- (h) top-K of the K*V candidates of every crop: ONE wave per crop (lane owns v = lane + 64 i), the K rounds of
    //      arg-max + winner removal need no workgroup barrier.  Order: larger value, then smaller flat index beam*V + v.
    const float lp = a.lp ? a.lp[s] : 1.0f;
    if (wv < NB) {
      const int nb = wv;
      constexpr int VI = 4;  // V <= 256
      float cv[VI][KB8];
      float lse_r[KB8], sc_r[KB8];
      int dn_r[KB8];
#pragma unroll
      for (int rb = 0; rb < KB8; ++rb) {  // per-beam values once, not once per candidate
        lse_r[rb] = s_lse[nb * KB8 + rb];
        sc_r[rb] = s_score[nb * KB8 + rb];
        dn_r[rb] = s_done[nb * KB8 + rb];
      }
#pragma unroll
      for (int i = 0; i < VI; ++i) {
        const int v = lane + 64 * i;
#pragma unroll
        for (int rb = 0; rb < KB8; ++rb) {
          cv[i][rb] = -INFINITY;
          const int r = nb * KB8 + rb;
          if (v < V && rb < KB) {
            float logp = sbuf[r * H + v] - lse_r[rb];
            if (dn_r[rb]) logp = (v == a.eos_id) ? 0.f : -INFINITY;
            float tot = sc_r[rb] + logp;
            if (a.lp) tot = tot / lp;
            cv[i][rb] = tot;
          }
        }
      }
      // per-lane cache: best candidate of every v-slot over the beams, and the lane's best over its slots.  A round then costs one
      // wave arg-max + the rescan of the winner's slot (8 entries) instead of a scan of all 32 entries per lane.
      float sv[VI], lv;
      int sf[VI], lf;
      auto rescan_slot = [&](int i) {
        float bv = -INFINITY;
        int bf = 0x7fffffff;
        const int v = lane + 64 * i;
#pragma unroll
        for (int rb = 0; rb < KB8; ++rb) {
          const float x = cv[i][rb];
          const int fi = rb * V + v;
          if (v < V && rb < KB && (x > bv || (x == bv && fi < bf))) { bv = x; bf = fi; }
        }
        sv[i] = bv; sf[i] = bf;
      };
      auto rescan_lane = [&]() {
        lv = sv[0]; lf = sf[0];
#pragma unroll
        for (int i = 1; i < VI; ++i)
          if (sv[i] > lv || (sv[i] == lv && sf[i] < lf)) { lv = sv[i]; lf = sf[i]; }
      };
#pragma unroll
      for (int i = 0; i < VI; ++i) rescan_slot(i);
      rescan_lane();
      for (int kk = 0; kk < KB; ++kk) {
        float bvv = lv;
        int bii = lf;
        wave_argmax63(bvv, bii);
        bvv = bcast63(bvv);
        bii = __builtin_amdgcn_readlane(bii, 63);
        int wi_ = bii;
        if (wi_ == 0x7fffffff) wi_ = 0;  // every candidate NaN: degenerate input
        const int wr = wi_ / V, wc = wi_ - wr * V;
        if (lane == 0) {
          s_top[nb * KB8 + kk] = bvv;
          s_src[nb * KB8 + kk] = wr;
          s_nxt[nb * KB8 + kk] = wc;
        }
        // the winner never wins again (NaN); wr, wc are wave-uniform, so is the slot wc >> 6: one slot is rescanned
        const int wslot = wc >> 6;
        const bool mine = (wc & 63) == lane;
#pragma unroll
        for (int i = 0; i < VI; ++i)
          if (i == wslot) {
#pragma unroll
            for (int rb = 0; rb < KB8; ++rb)
              if (mine && rb == wr) cv[i][rb] = __int_as_float(0x7fc00000);
            rescan_slot(i);
          }
        rescan_lane();
      }
    }
    __syncthreads();
    TSTAMP(8);
    // ---- (i) bookkeeping per state row
    int nd = 1;
    if (tid < R) {
      const int nb = tid / KB8, rb = tid % KB8;
      if (rb < KB) nd = s_done[nb * KB8 + s_src[tid]] | (s_nxt[tid] == a.eos_id);
    }
    __syncthreads();
    if (tid < R) {
      const int nb = tid / KB8, rb = tid % KB8, b = b0 + nb;
      if (rb < KB) {
        if (b < a.B) {
          const long o = ((long)b * a.steps + s) * KB + rb;
          a.back[o] = s_src[tid];
          a.tokv[o] = s_nxt[tid];
        }
        s_score[tid] = a.lp ? s_top[tid] * lp : s_top[tid];  // f32 round trip of the reference (model.py:188-192)
        s_tok[tid] = s_nxt[tid];
      }
      s_done[tid] = nd;
    }
    __syncthreads();
    if (tid < NB) {
      int best = 0, all = 1;
      float bs = s_score[tid * KB8];
      for (int r = 0; r < KB; ++r) {
        if (r && s_score[tid * KB8 + r] > bs) { bs = s_score[tid * KB8 + r]; best = r; }
        all &= s_done[tid * KB8 + r];
      }
      if (b0 + tid < a.B) a.best_at[(long)(b0 + tid) * a.steps + s] = best;
      if (all && s_fin[tid] == a.steps) {
        s_fin[tid] = s + 1;
        if (a.chunk_state && b0 + tid < a.B) {  // publish: max finish step first, then the count that readers test
          const int ch = a.chunk_id[b0 + tid];
          atomicMax(&a.chunk_state[2 * ch + 1], s + 1);
          __threadfence();
          atomicAdd(&a.chunk_state[2 * ch], 1);
        }
      }
    }
    TSTAMP(9);
    // ---- (j) permute beam state by src within every crop: c through sbuf (logits are consumed), h through registers
    {
#pragma unroll
      for (int e = 0; e < 16; ++e) sbuf[acc_row(e, half) * H + ju] = c[e];
      if constexpr (SPLITW) {
        const int j2 = tid & 127, nb = tid >> 7;  // column pair, crop: the crop's 8 rows of the three planes
        uint32_t hp[3][KB8];
#pragma unroll
        for (int rb = 0; rb < KB8; ++rb) {
          const int src = rb < KB ? s_src[nb * KB8 + rb] : rb;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) hp[pl][rb] = *reinterpret_cast<const uint32_t*>(sP + pl * PPL + (nb * KB8 + src) * PSB + j2 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = acc_row(e, half), cb = r / KB8, rb = r % KB8;
          const int src = rb < KB ? s_src[r] : rb;
          c[e] = sbuf[(cb * KB8 + src) * H + ju];
        }
#pragma unroll
        for (int rb = 0; rb < KB8; ++rb)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(sP + pl * PPL + (nb * KB8 + rb) * PSB + j2 * 4) = hp[pl][rb];
      } else {
        const int j = tid & 255, ch = tid >> 8;
        float hn[R / 2];
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
          const int r = ch * (R / 2) + q, nb = r / KB8, rb = r % KB8;
          const int src = rb < KB ? s_src[r] : rb;
          hn[q] = sX[(nb * KB8 + src) * XS + H + j];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = acc_row(e, half), nb = r / KB8, rb = r % KB8;
          const int src = rb < KB ? s_src[r] : rb;
          c[e] = sbuf[(nb * KB8 + src) * H + ju];
        }
#pragma unroll
        for (int q = 0; q < R / 2; ++q) sX[(ch * (R / 2) + q) * XS + H + j] = hn[q];
      }
    }
    __syncthreads();
    TSTAMP(10);
    // ---- early exit (model.py:215 breaks the loop once every beam of every row of the chunk is finished): leave after
    //      step s when every chunk this workgroup's crops belong to has all its crops finished at steps <= s + 1, i.e. the
    //      chunk's run length T_run = max finish step is covered.  Nobody waits: a workgroup that cannot see the others'
    //      flags yet just runs further steps, whose outputs are then unused.
    if (a.chunk_state) {
      if (tid == 0) {
        int ex = 1;
        for (int nb = 0; nb < NB && ex; ++nb) {
          const int b = b0 + nb;
          if (b >= a.B) continue;
          if (s_fin[nb] == a.steps) { ex = 0; break; }
          const int ch = a.chunk_id[b];
          if (__hip_atomic_load(&a.chunk_state[2 * ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.chunk_size[ch]) { ex = 0; break; }
          __threadfence();
          if (__hip_atomic_load(&a.chunk_state[2 * ch + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > s + 1) ex = 0;
        }
        s_exit = ex;
      }
      __syncthreads();
      if (s_exit) break;
    }
  }
  if (tid < NB && b0 + tid < a.B) a.fin_step[b0 + tid] = s_fin[tid];
}

// ------------------------------------------------------------------------------------------------- greedy (round 4)
// mode="greedy" (Attention._greedy_decode, model.py:227-259) in the same row-block form: one 512-thread workgroup = 32 CROPS = 32
// state rows = one MFMA row block, all `steps` steps in one launch.  The rows are independent (the reference keeps every row
// running after its own EOS and only stops a chunk when all rows emit EOS in the same step; the host derives those run lengths
// from the ids, recognizers/_trba/__init__.py), so a step is the beam kernel's (a)-(f) without any beam bookkeeping — no
// log-softmax, no top-k, no state permutation — followed by an arg-max per row (larger value, then smaller index; blank masked
// to -1e4 as in attn_greedy_kernel).  Split-operand products and hoisted context gates (a.ctx_gates) as in
// attn_beam_mfma_kernel<true, true>; gate and score nonlinearities libm-grade (see sigmoid_libm).  Each row reads ITS crop's proj_H / ctx_gates frames
// (8x the beam kernel's traffic per row, from L2 / the Infinity Cache).
// libm-grade nonlinearities for the greedy kernel: its logits are held to the VALU kernel's bound (2e-4 of max |logit| against the
// oracle's decoder over 41 chained steps of the all-random decoder, tests/test_gpu_trba.py DECODER_LOGIT_RTOL), which the hardware-
// rate v_exp_f32 / v_rcp_f32 forms of the beam kernel miss by 5 % on one row of 96; greedy is not the pipeline's default mode.
__device__ __forceinline__ float sigmoid_libm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(NT, 1) void attn_greedy_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned char* sP = reinterpret_cast<unsigned char*>(lds);  // [3][R][PSB] bf16 planes of h
  float* sbuf = lds + 3 * PPL / 4;                             // [R][H]    ph, then logits
  float* salpha = sbuf + R * H;                                // [R][64]
  __shared__ int s_tok[R];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const int T = a.T, V = a.V;
  const int b0 = blockIdx.x * R;
  const int ju = 32 * wv + r32;
  for (int i = tid; i < 3 * PPL / 4; i += NT) lds[i] = 0.f;   // h = 0
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  if (tid < R) s_tok[tid] = a.sos_id;
  __syncthreads();

  for (int s = 0; s < a.steps; ++s) {
    // ---- (a) ph[r][j] = h2h_b[j] + sum_k h[r][k] * h2h_wt[k][j]
    {
      f32x16 acc;
      const float bj = a.w.h2h_b[ju];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = bj;
      mfma_cols32_split(sP, a.h2h_p, H, ju, true, r32, half, acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) sbuf[acc_row(e, half) * H + ju] = acc[e];
    }
    __syncthreads();
    // ---- (b) e[r][t] = sum_j score_w[j] * tanh(proj_H[crop r][t][j] + ph[r][j]): one wave per (row, t), GB rows of proj_H in flight
    {
      constexpr int GB = 4;
      float sw[H / 64];
#pragma unroll
      for (int q = 0; q < H / 64; ++q) sw[q] = a.w.score_w[lane + 64 * q];
      const int ngroups = R * T;
      for (int g0 = wv; g0 < ngroups; g0 += GB * (NT / 64)) {
        float pr[GB][H / 64];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int g = g0 + u * (NT / 64);
          if (g < ngroups) {
            const int r = g / T, t = g - r * T;
            const float* pP = a.proj_H + ((long)min(b0 + r, a.B - 1) * T + t) * H;
#pragma unroll
            for (int q = 0; q < H / 64; ++q) pr[u][q] = pP[lane + 64 * q];
          }
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int g = g0 + u * (NT / 64);
          if (g < ngroups) {
            const int r = g / T, t = g - r * T;
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < H / 64; ++q) sacc = fmaf(sw[q], tanhf(pr[u][q] + sbuf[r * H + lane + 64 * q]), sacc);
            sacc = wave_sum63(sacc);
            if (lane == 63) salpha[r * 64 + t] = sacc;
          }
        }
      }
    }
    __syncthreads();
    // ---- (c) softmax over t: wave w handles rows 4w..4w+3, lane = t (T <= 64)
    for (int r = 4 * wv; r < 4 * wv + 4; ++r) {
      const float ev0 = lane < T ? salpha[r * 64 + lane] : -INFINITY;
      const float m = bcast63(wave_max63(ev0));
      const float ev = lane < T ? expf(ev0 - m) : 0.f;
      const float sum = bcast63(wave_sum63(ev));
      if (lane < T) salpha[r * 64 + lane] = ev / sum;
    }
    __syncthreads();
    // ---- (e) gates = b + W_ih_tok[token] + sum_t alpha_t P[crop][t] + W_hh h ; LSTM cell for units ju, rows acc_row(e, half)
    {
      f32x16 acc[4];
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&a.w.b_gates[ju * 4]);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = acc_row(e, half);
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&a.w.wih_tok[((long)s_tok[row] * H + ju) * 4]);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g][e] = add_np(b4[g], t4[g]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {  // one row's frames (T x 16 B per lane, 8 in flight), then its FMAs; rows do not overlap
        const float* pP = a.ctx_gates + ((long)min(b0 + acc_row(e, half), a.B - 1) * T * H + ju) * 4;
        const float* pa = salpha + acc_row(e, half) * 64;
        for (int t0 = 0; t0 < T; t0 += 8) {
          f32x4 pv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (t0 + u < T) pv[u] = *reinterpret_cast<const f32x4*>(pP + (long)(t0 + u) * H * 4);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (t0 + u < T) {
              const float al = pa[t0 + u];
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[g][e] = fmac_np(al, pv[u][g], acc[g][e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the rows' loads from piling up across iterations (registers)
      }
      mfma_gates_split(sP, a.whh_p, ju, r32, half, acc);
      __syncthreads();  // every wave has read the old h
      float hv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ig = sigmoid_libm(acc[0][e]), fg = sigmoid_libm(acc[1][e]), gg = tanhf(acc[2][e]), og = sigmoid_libm(acc[3][e]);
        c[e] = fg * c[e] + ig * gg;
        hv[e] = og * tanhf(c[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; e += 2) {  // acc_row(e + 1) == acc_row(e) + 1
        unsigned char* d = sP + acc_row(e, half) * PSB + ju * 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint32_t pk = split_pair(hv[e], hv[e + 1]);
          *reinterpret_cast<uint16_t*>(d + pl * PPL) = (uint16_t)pk;
          *reinterpret_cast<uint16_t*>(d + pl * PPL + PSB) = (uint16_t)(pk >> 16);
        }
      }
    }
    __syncthreads();
    // ---- (f) logits[r][v] = gen_b[v] + sum_k h'[r][k] * gen_wt[k][v]
    {
      const bool vok = ju < V;
      f32x16 acc;
      const float bv = vok ? a.w.gen_b[ju] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = bv;
      if (32 * wv < V) mfma_cols32_split(sP, a.gen_p, (V + 31) & ~31, ju, true, r32, half, acc);  // padded columns are zeros
      if (vok) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = acc_row(e, half);
          const float v = (ju == a.blank_id) ? -1e4f : acc[e];
          sbuf[r * H + ju] = v;
          if (b0 + r < a.B) a.logits_out[((long)(b0 + r) * a.steps + s) * V + ju] = v;
        }
      }
    }
    __syncthreads();
    // ---- (g) arg-max per row (larger value, then smaller index): wave w handles rows 4w..4w+3
    for (int r = 4 * wv; r < 4 * wv + 4; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      for (int v = lane; v < V; v += 64) {
        const float x = sbuf[r * H + v];
        if (x > bv || (x == bv && v < bi)) { bv = x; bi = v; }
      }
      wave_argmax63(bv, bi);
      if (lane == 63) {
        if (bi == 0x7fffffff) bi = 0;  // every logit NaN: degenerate input
        s_tok[r] = bi;
        if (b0 + r < a.B) a.ids_out[(long)(b0 + r) * a.steps + s] = bi;
      }
    }
    __syncthreads();
  }
}

}  // namespace

int msocr_internal_attn_beam_mfma(const AttnArgs& a, hipStream_t s) {
  const size_t ldsz = (size_t)(R * XS + R * H + R * 64) * sizeof(float);
  const size_t ldsz_split = (size_t)3 * PPL + (size_t)(R * H + R * 64) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_beam_mfma_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_beam_mfma_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_beam_mfma_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz_split) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  const dim3 grid((a.B + NB - 1) / NB);
  if (a.ctx_gates && a.h2h_p)
    MSOCR_LAUNCH((attn_beam_mfma_kernel<true, true>), grid, dim3(NT), ldsz_split, s, a);
  else if (a.ctx_gates)
    MSOCR_LAUNCH((attn_beam_mfma_kernel<true, false>), grid, dim3(NT), ldsz, s, a);
  else
    MSOCR_LAUNCH((attn_beam_mfma_kernel<false, false>), grid, dim3(NT), ldsz, s, a);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

int msocr_internal_attn_greedy_mfma(const AttnArgs& a, hipStream_t s) {
  if (!a.ctx_gates || !a.h2h_p || !a.whh_p || !a.gen_p) return MSOCR_E_ARG;
  const size_t ldsz_split = (size_t)3 * PPL + (size_t)(R * H + R * 64) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_greedy_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz_split) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  MSOCR_LAUNCH(attn_greedy_mfma_kernel, dim3((a.B + R - 1) / R), dim3(NT), ldsz_split, s, a);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

// HOST helper: a transposed f32 weight matrix of the decoder ([256][N] row-major: h2h_wt, gen_wt; or gate-interleaved
// [256][N / 4][4]: whh_t) -> the packed split form the matrix-core beam kernel reads: out[plane][k / 16][column][k % 16] bf16 with
// w == p0 + p1 + p2 exactly; columns padded with zeros to a multiple of 32; gate-interleaved input: column = gate * (N / 4) + unit.
extern "C" int64_t msocr_attn_pack_split_elems(int N) { return N > 0 ? (int64_t)3 * H * ((N + 31) & ~31) : 0; }
extern "C" int msocr_attn_pack_split_host(const float* wt_host, int N, int gate_interleaved, uint16_t* out_host) {
  if (!wt_host || !out_host || N <= 0 || (gate_interleaved && N % 4)) return MSOCR_E_ARG;
  const int Np = (N + 31) & ~31;
  const int64_t plane = (int64_t)H * Np;
  for (int64_t i = 0; i < 3 * plane; ++i) out_host[i] = 0;
  auto rne = [](float f) -> uint16_t {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  for (int k = 0; k < H; ++k)
    for (int n = 0; n < N; ++n) {
      const int col = gate_interleaved ? (n & 3) * (N / 4) + (n >> 2) : n;
      float r = wt_host[(int64_t)k * N + n];
      for (int pl = 0; pl < 3; ++pl) {
        const uint16_t hb = rne(r);
        out_host[pl * plane + ((int64_t)(k >> 4) * Np + col) * 16 + (k & 15)] = hb;
        r -= __builtin_bit_cast(float, (uint32_t)hb << 16);
      }
    }
  return MSOCR_OK;
}
