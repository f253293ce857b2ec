// conv_igemm.hip — im2col-free implicit-GEMM convolution on MFMA for gfx950 (MI355X).
//
// GEMM view: D[M = N*Ho*Wo][Cout] = A[M][K = KH*KW*Cin] * W^T, NHWC activations, weights
// [Cout][KH][KW][Cin].  Both operands are K-contiguous, so both LDS tiles are [rows][BK]
// with 16-byte chunks XOR-swizzled by row (ds_read_b128 conflict-free), and one code path
// serves f32 (v_mfma_f32_32x32x2_f32, exact f32) and bf16 (v_mfma_f32_32x32x16_bf16).
// A K-tile never straddles a filter tap (Cin % BK == 0), so an A row is one 16-B-aligned
// contiguous run of the input: plain global_load_dwordx4 with a zero fill for padding.
// 256 threads = 4 waves, each wave owns a (WM x WN) sub-tile as 32x32 MFMA tiles.
// Epilogue: accumulators -> LDS -> (bias, residual, ReLU) -> 16-byte coalesced stores.
//
// Replaces nn.Conv2d+BatchNorm2d(+ReLU)(+residual add) on the reference hot path:
//   detectors/_east/east.py:13-30,56-67 ; recognizers/_trba/model/seresnet31.py:37-45,81-89,129-155
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "internal.h"
#include "msocr.h"

// 16 zero bytes: padded (out-of-image) taps load from here, so the operand tile needs no masking
__device__ __attribute__((aligned(16))) uint32_t msocr_zero16[4] = {0u, 0u, 0u, 0u};

// LEAN: 1x1 kernel without padding (every 1x1 convolution and the batched GEMMs of the Winograd path) — a K-tile is a plain
// pointer increment, no tap decoding, no bounds masks (rows >= M load valid garbage that the epilogue never stores).  PMC on
// the Winograd GEMM counted 1.9 VALU + 1.2 SALU instructions per MFMA in the general loader; they share the SIMD's issue port.
// WPE: workgroups per CU the register allocation is held to (0 = 3 for the one-stage 128-byte-row form, else 2).
template <typename T, int BM, int BN, int BKB, int WM, int WN, int STAGES = 2, bool LEAN = false, int MT = 32, int WPE = 0>
__global__ __launch_bounds__(256, WPE ? WPE : ((STAGES == 1 && BKB <= 128) ? 3 : 2)) void conv_igemm_kernel(ConvParams p) {
  constexpr int ES = sizeof(T);
  constexpr int CPR = BKB / 16;  // 16-B chunks per tile row
  constexpr int EPC = 16 / ES;   // elements per chunk
  constexpr int BK = BKB / ES;
  constexpr int WAVES_N = BN / WN;
  static_assert(MT == 32 || (MT == 16 && sizeof(T) == 4), "16x16x4 is the f32 shape");
  constexpr int TM = WM / MT, TN = WN / MT;
  constexpr int AE = MT == 32 ? 16 : 4;
  constexpr int CQ = 64 / MT;
  using AccT = typename std::conditional<MT == 32, f32x16, f32x4>::type;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  constexpr int RPP = 256 / CPR;  // tile rows covered per pass of 256 threads
  constexpr int A_IT = BM / RPP;
  constexpr int B_IT = (BN + RPP - 1) / RPP;
  constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;
  constexpr int STAGE = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware tile mapping: blocks b, b+8, ... share an XCD/L2 -> give each XCD a contiguous
  // range of logical tiles (all N-tiles of neighbouring M-tiles: shared A rows + 3x3 halos).
  const int nblk1 = p.tilesM * p.tilesN;
  const int nblk = nblk1 * p.nbatch;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int batch = bid / nblk1;
  bid -= batch * nblk1;
  const int tile_n = bid % p.tilesN;
  const int tile_m = bid / p.tilesN;
  const char* const g_in = p.in + (long)batch * p.bsA * ES;
  const char* const g_w = p.w + (long)batch * p.bsW * ES;
  char* const g_out = p.out + (long)batch * p.bsO * ES;

  // ---- per-thread staging coordinates (fixed over the K loop) ----
  const int chunk = tid % CPR;
  const int row0 = tid / CPR;
  long a_base[A_IT];
  int a_hi0[A_IT], a_wi0[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const long m = (long)tile_m * BM + row0 + i * RPP;
    if (m < p.M) {
      const long hw = (long)p.Ho * p.Wo;
      const int n = (int)(m / hw);
      const int rem = (int)(m - (long)n * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.SH - p.PH;
      a_wi0[i] = wo * p.SW - p.PW;
      a_base[i] = (long)n * p.sN + (long)a_hi0[i] * p.sH + (long)a_wi0[i] * p.sW + chunk * EPC;
    } else {
      a_hi0[i] = -0x40000000;  // never in range
      a_wi0[i] = 0;
      a_base[i] = 0;
    }
  }
  const char* b_ptr[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int co = tile_n * BN + row0 + j * RPP;
    b_ptr[j] = g_w + ((long)co * p.Ktot + chunk * EPC) * ES;
  }

  u32x4 ra[A_IT], rb[B_IT];
  int t_kh = 0, t_kw = 0, t_c0 = 0;  // tap and channel offset of the NEXT K-tile load_tile() will fetch

  const char* a_ptr[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) a_ptr[i] = g_in + a_base[i] * ES;
  auto load_tile = [&](int kt) {
    if constexpr (LEAN) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + (long)kt * BKB);
#pragma unroll
      for (int j = 0; j < B_IT; ++j) {
        if (BN % RPP == 0 || row0 + j * RPP < BN) rb[j] = *reinterpret_cast<const u32x4*>(b_ptr[j] + (long)kt * BKB);
      }
      return;
    }
    // (kh, kw, c0) of K-tile kt, advanced incrementally (tiles are visited in order 0, 1, 2, ...): no divisions in the loop
    const long koff = (long)t_kh * p.sH + (long)t_kw * p.sW + t_c0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int hi = a_hi0[i] + t_kh, wi = a_wi0[i] + t_kw;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      // branch-free: out-of-image taps read 16 zero bytes
      const char* src = ok ? g_in + (a_base[i] + koff) * ES : reinterpret_cast<const char*>(msocr_zero16);
      ra[i] = *reinterpret_cast<const u32x4*>(src);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      if (BN % RPP == 0 || row0 + j * RPP < BN) rb[j] = *reinterpret_cast<const u32x4*>(b_ptr[j] + (long)kt * BKB);
    }
    t_c0 += BK;
    if (t_c0 == p.Cin) {
      t_c0 = 0;
      if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
    }
  };
  auto store_tile = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = row0 + i * RPP;
      *reinterpret_cast<u32x4*>(sa + row * BKB + ((chunk ^ swz<BKB>(row)) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      const int row = row0 + j * RPP;
      if (BN % RPP == 0 || row < BN) *reinterpret_cast<u32x4*>(sb + row * BKB + ((chunk ^ swz<BKB>(row)) << 4)) = rb[j];
    }
  };

  AccT acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;

  const int r32 = lane & (MT - 1), half = lane / MT;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  auto read_frags = [&](const unsigned char* sa, const unsigned char* sb, int q, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) {
    const int c = CQ * q + half;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WM + i * MT + r32;
      fa[i] = *reinterpret_cast<const u32x4*>(sa + row * BKB + ((c ^ swz<BKB>(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * WN + j * MT + r32;
      fb[j] = *reinterpret_cast<const u32x4*>(sb + row * BKB + ((c ^ swz<BKB>(row)) << 4));
    }
  };
  constexpr int NQ = CPR / CQ;
  for (int kt = 0; kt < p.ktiles; ++kt) {
    const int cur = STAGES == 1 ? 0 : (kt & 1);
    if (kt + 1 < p.ktiles) load_tile(kt + 1);  // global loads in flight under the MFMAs
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + A_BYTES;
    if constexpr (MT == 16) {
      // 16x16x4: 16 accumulator tiles per wave; fragments are read just in time (8 x b128 per 16 k), no second fragment buffer
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        u32x4 fa[TM], fb[TN];
        read_frags(sa, sb, q, fa, fb);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i][e]), __uint_as_float(fb[j][e]), acc[i][j], 0, 0, 0);
      }
    } else {
    u32x4 fa[2][TM], fb[2][TN];
      read_frags(sa, sb, 0, fa[0], fb[0]);
  #pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {  // LDS reads of the next k-group are issued BEFORE this group's MFMAs and stay pinned there
          read_frags(sa, sb, q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (sizeof(T) == 4) {
          // k-element outermost: consecutive MFMAs hit different accumulators (no back-to-back dependent issue)
  #pragma unroll
          for (int e = 0; e < 4; ++e)
  #pragma unroll
            for (int i = 0; i < TM; ++i)
  #pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[q & 1][i][e]), __uint_as_float(fb[q & 1][j][e]),
                                                                 acc[i][j], 0, 0, 0);
        } else {
  #pragma unroll
          for (int i = 0; i < TM; ++i)
  #pragma unroll
            for (int j = 0; j < TN; ++j) Mma<T>::run(fa[q & 1][i], fb[q & 1][j], acc[i][j]);
        }
      }
    }
    if (STAGES == 1) {  // one LDS stage (3 workgroups per CU): everyone must be done reading before it is overwritten
      __syncthreads();
      if (kt + 1 < p.ktiles) store_tile(0);
    } else if (kt + 1 < p.ktiles) {
      store_tile(cur ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue: TM passes of (acc row-block -> LDS [PR][BN] f32 -> bias/residual/ReLU -> 16-B stores) ----
  constexpr int PR = (BM / WM) * 32;  // tile rows handled per pass
  float* sc = reinterpret_cast<float*>(smem);
  constexpr int VPR = BN / EPC;       // 16-B output vectors per tile row
  constexpr int ROWS_PP = 256 / VPR;  // rows per sweep of 256 threads
  const int vcol = (tid % VPR) * EPC;
  const int vrow0 = tid / VPR;
  const int co = tile_n * BN + vcol;
  float bias[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bias[e] = p.bias ? p.bias[co + e] : 0.f;

  constexpr int TPP = 32 / MT;
#pragma unroll
  for (int i = 0; i < WM / 32; ++i) {
    if (i) __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TPP; ++ti)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < AE; ++e) {
          const int rit = MT == 32 ? (e & 3) + 8 * (e >> 2) + 4 * half : 4 * half + e;
          sc[(wm * 32 + ti * MT + rit) * BN + wn * WN + j * MT + r32] = acc[i * TPP + ti][j][e];
        }
    __syncthreads();
    for (int lr = vrow0; lr < PR; lr += ROWS_PP) {
      const int trow = (lr >> 5) * WM + i * 32 + (lr & 31);
      const long m = (long)tile_m * BM + trow;
      if (m >= p.M) continue;
      float v[EPC];
#pragma unroll
      for (int e4 = 0; e4 < EPC; e4 += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(&sc[lr * BN + vcol + e4]);
        v[e4] = t[0]; v[e4 + 1] = t[1]; v[e4 + 2] = t[2]; v[e4 + 3] = t[3];
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] += bias[e];
      if (p.has_res) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + (m * p.res_ld + co) * ES);
        if constexpr (ES == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(rv[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((uint16_t)(rv[e >> 1] >> ((e & 1) * 16)));
        }
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      u32x4 o;
      if constexpr (ES == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
      }
      *reinterpret_cast<u32x4*>(g_out + (m * p.out_ld + co) * ES) = o;
    }
  }
}

template <typename T, int BM, int BN, int BKB, int WM, int WN, int STAGES = 2, bool LEAN = false, int MT = 32, int WPE = 0>
static int launch_cfg(ConvParams& p, hipStream_t s) {
  p.tilesM = (int)((p.M + BM - 1) / BM);
  p.tilesN = p.Cout / BN;
  constexpr int BK = BKB / (int)sizeof(T);
  p.cin_tiles = p.Cin / BK;
  p.ktiles = p.KH * p.KW * p.cin_tiles;
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int EPI = (BM / WM) * 32 * BN * 4;
  constexpr int LDS = STAGES * STAGE > EPI ? STAGES * STAGE : EPI;
  auto kern = conv_igemm_kernel<T, BM, BN, BKB, WM, WN, STAGES, LEAN, MT, WPE>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr_set = true;
  }
  const long nblk = (long)p.tilesM * p.tilesN * p.nbatch;
  if (nblk <= 0 || nblk > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(kern, dim3((unsigned)nblk), dim3(256), LDS, s, p);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

template <typename T>
static int launch_typed(ConvParams& p, hipStream_t s) {
  constexpr int ES = sizeof(T);
  // row bytes: 128 B when Cin allows (f32: BK 32, bf16: BK 64), else 64 B
  const bool wide = (p.Cin * ES) % 128 == 0;
  // The f32 instances keep ONE LDS stage at 3-4 workgroups per CU (a resident neighbour covers the others' prologue / epilogue: worth
  // 3-5 % at short K); bf16 keeps the register-staged double buffer at 2 per CU.  `lean` = 1x1 kernel without padding.
  const bool lean = p.KH == 1 && p.KW == 1 && p.PH == 0 && p.PW == 0;
  if (p.Cout % 128 == 0) {
    if constexpr (sizeof(T) == 4) {
      // lean: v_mfma_f32_16x16x4_f32 tiles, K-tiles of 16 (64-byte rows), 128 VGPRs, FOUR workgroups per CU (round 2: +2.3 % on the
      // pipeline against K-tiles of 32 at 3 per CU; K-tiles of 64, staggered workgroup starts, two LDS stages and a direct-store
      // epilogue all measured within 1 % of these two and are gone)
      if (wide && lean) return launch_cfg<T, 128, 128, 64, 64, 64, 1, true, 16, 4>(p, s);
      if (wide) return launch_cfg<T, 128, 128, 128, 64, 64, 1>(p, s);
    }
    return wide ? launch_cfg<T, 128, 128, 128, 64, 64>(p, s) : launch_cfg<T, 128, 128, 64, 64, 64>(p, s);
  } else if (p.Cout % 64 == 0) {
    if constexpr (sizeof(T) == 4) {
      if (wide && lean) return launch_cfg<T, 128, 64, 128, 64, 32, 2, true>(p, s);
    }
    return wide ? launch_cfg<T, 128, 64, 128, 64, 32>(p, s) : launch_cfg<T, 128, 64, 64, 64, 32>(p, s);
  } else {
    return wide ? launch_cfg<T, 256, 32, 128, 64, 32>(p, s) : launch_cfg<T, 256, 32, 64, 64, 32>(p, s);
  }
}

extern "C" int msocr_conv2d(const msocr_conv_desc* d, const void* in, const void* weight, const float* bias,
                            const void* residual, void* out, void* stream) {
  if (!d || !in || !weight || !out) return MSOCR_E_ARG;
  const int ES = d->dtype == MSOCR_F32 ? 4 : (d->dtype == MSOCR_BF16 ? 2 : 0);
  if (!ES) return MSOCR_E_ARG;
  const int EPC = 16 / ES;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return MSOCR_E_ARG;
  if (d->Cin <= 0 || (d->Cin * ES) % 64 != 0) return MSOCR_E_ARG;           // K-tile inside one tap
  if (d->Cout <= 0 || d->Cout % 32 != 0) return MSOCR_E_ARG;
  if (d->KH <= 0 || d->KW <= 0 || d->stride_h <= 0 || d->stride_w <= 0) return MSOCR_E_ARG;
  // 16-byte alignment of every vector access
  if (d->in_sN % EPC || d->out_ld % EPC || d->out_ld < d->Cout) return MSOCR_E_ARG;
  // rows/pixels the kernel can touch: hi = ho*sh - ph + kh, wi = wo*sw - pw + kw; every one must start 16-B aligned
  if (d->in_sH % EPC) {
    if ((d->stride_h * d->in_sH) % EPC) return MSOCR_E_ARG;
    for (int kh = 0; kh < d->KH; ++kh)
      if (((kh - d->pad_h) * d->in_sH) % EPC) return MSOCR_E_ARG;
  }
  if (d->in_sW % EPC) {  // e.g. the C=4 stem canvas: pixels are 8 B in bf16, only even pixels are read
    if ((d->stride_w * d->in_sW) % EPC) return MSOCR_E_ARG;
    for (int kw = 0; kw < d->KW; ++kw)
      if (((kw - d->pad_w) * d->in_sW) % EPC) return MSOCR_E_ARG;
  }
  if (((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % EPC || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  // output extent must agree with the conv arithmetic (guards the kernel's indexing)
  if ((d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 < d->Ho || (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1 < d->Wo)
    return MSOCR_E_ARG;

  ConvParams p;
  p.in = (const char*)in; p.w = (const char*)weight; p.bias = bias; p.res = (const char*)residual; p.out = (char*)out;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.sN = d->in_sN; p.sH = d->in_sH; p.sW = d->in_sW;
  p.KH = d->KH; p.KW = d->KW; p.SH = d->stride_h; p.SW = d->stride_w; p.PH = d->pad_h; p.PW = d->pad_w;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.M = (long)d->N * d->Ho * d->Wo;
  p.Ktot = (long)d->KH * d->KW * d->Cin;
  p.out_ld = d->out_ld; p.res_ld = d->res_ld;
  p.relu = (d->flags & MSOCR_CONV_RELU) ? 1 : 0;
  p.has_res = has_res ? 1 : 0;
  p.nbatch = 1; p.bsA = p.bsW = p.bsO = 0;
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == MSOCR_F32 ? launch_typed<float>(p, s) : launch_typed<__bf16>(p, s);
}

// nbatch independent f32 GEMMs of one shape in ONE launch: C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]
// (A [nbatch][M][K], B [nbatch][N][K], C [nbatch][M][N], all dense).  Used by the Winograd path (winograd.hip).
int msocr_internal_gemm_f32_batched(const float* A, const float* B, float* C, long M, int N, int K, int nbatch, hipStream_t s) {
  if (!A || !B || !C || M <= 0 || N <= 0 || N % 32 || K <= 0 || K % 16 || nbatch <= 0) return MSOCR_E_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return MSOCR_E_ARG;
  if (M > 0x7fffffffL) return MSOCR_E_ARG;
  ConvParams p;
  p.in = (const char*)A; p.w = (const char*)B; p.bias = nullptr; p.res = nullptr; p.out = (char*)C;
  p.N = 1; p.H = (int)M; p.W = 1; p.Cin = K;
  p.sN = M * (long)K; p.sH = K; p.sW = K;
  p.KH = p.KW = 1; p.SH = p.SW = 1; p.PH = p.PW = 0;
  p.Ho = (int)M; p.Wo = 1; p.Cout = N;
  p.M = M; p.Ktot = K;
  p.out_ld = N; p.res_ld = 0;
  p.relu = 0; p.has_res = 0;
  p.nbatch = nbatch; p.bsA = M * (long)K; p.bsW = (long)N * K; p.bsO = M * (long)N;
  return launch_typed<float>(p, s);
}
