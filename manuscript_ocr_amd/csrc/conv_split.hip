// conv_split.hip — f32 GEMM / 1x1 convolution on the bf16 matrix pipes with EXACTLY split operands ("bf16x3").
//
// gfx950 runs exact-f32 MFMA (v_mfma_f32_32x32x2_f32) at 1/16 of the bf16 rate.  An f32 value splits exactly into three bf16
// terms, a = a0 + a1 + a2 (round-to-nearest residuals: |a1| <= 2^-9 |a|, |a2| <= 2^-18 |a|), so
//     a * b = a0b0 + (a0b1 + a1b0) + (a1b1 + a0b2 + a2b0) + [a1b2 + a2b1 + a2b2],
// and the bracket is <= 2^-25 |ab|: below the rounding of the f32 accumulation both paths share.  The six kept products run on
// v_mfma_f32_32x32x16_bf16 (products of bf16 values are exact in f32; f32 accumulate), smallest terms first: 6 bf16 MFMAs
// replace 8 f32 MFMAs of 1/16 the rate each — the K loop costs 6 x 32 cycles per 32x32x16 block instead of 8 x 64.
//
// Operands: A = activations (or the Winograd-domain V) as plain f32 in HBM, split IN REGISTERS on the way into LDS (three
// v_cvt_pk_bf16_f32 + two subtractions per pair); B = weights (or the Winograd U), split ONCE at load time into three bf16
// planes [3][Cout][K] (msocr_split_bf16x3_host / ops.py).  Same tile as conv_igemm.hip's lean kernel: 128 x 128 (or 128 x 64) per
// workgroup, 4 waves of 64 x 64 (64 x 32), K-tiles of 32, one LDS stage of six [rows][64 B] planes (16-B chunks XOR-swizzled by
// row), 3 workgroups per CU; the epilogue (bias, residual, ReLU, 16-byte stores) is the exact-f32 kernel's.
//
// Used for every launch the lean exact-f32 GEMM served (1x1 / stride 1 / no padding convolutions, the LSTM / linear GEMMs and
// the batched Winograd-domain GEMMs) unless the caller asks for precision = "fp32-exact".  Reference layers:
//   recognizers/_trba/model/seresnet31.py:37-67 ; detectors/_east/east.py:13-30 ; torchvision Bottleneck conv1 / conv3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"
#include "internal.h"
#include "msocr.h"

namespace {

// 16 zero bytes: padded (out-of-image) taps of the general loader read from here
__device__ __attribute__((aligned(16))) uint32_t msocr_split_zero16[4] = {0u, 0u, 0u, 0u};

// BN in {128, 64}; wave tile 64 x WN (WN = BN / 2); K-tile = 32 elements (A rows 128 B of f32 in HBM, 64 B per bf16 plane in LDS).
// GEN = false: 1x1 / stride 1 / no padding over a dense pixel sequence and the batched GEMMs — a K-tile is a plain pointer
// increment.  GEN = true: any kernel size / stride / padding (the implicit-GEMM loader of conv_igemm.hip: a K-tile lies inside one
// filter tap because Cin % 32 == 0; out-of-image taps read zeros).
template <int BN, int WPE, bool GEN>
__global__ __launch_bounds__(256, WPE) void conv_split_kernel(ConvParams p) {
  constexpr int BM = 128, WM = 64, WN = BN / 2, BK = 32;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int ROWB = BK * 2;             // bytes per LDS plane row (BK bf16)
  constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
  constexpr int STAGE_B = 3 * (A_PLANE + B_PLANE);
  constexpr int ACH = BK / 4;              // A: 16-B chunks per f32 row
  constexpr int ARP = 256 / ACH;           //    rows per pass of 256 threads
  constexpr int A_IT = BM / ARP;
  constexpr int BCH = BK / 8;              // B: 16-B chunks per bf16 row
  constexpr int BRP = 256 / BCH;
  constexpr int B_IT = (BN + BRP - 1) / BRP;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile mapping (as conv_igemm_kernel): each XCD gets a contiguous range of logical tiles
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
  const char* const g_in = p.in + (long)batch * p.bsA * 4;
  const char* const g_w = p.w + (long)batch * p.bsW * 2;
  char* const g_out = p.out + (long)batch * p.bsO * 4;

  // ---- staging coordinates ----
  // Lean form: every address is a wave-uniform base (SGPRs: batch + K-tile + plane) plus a 32-bit per-thread offset (one VGPR per
  // load), so the loop holds 4 + B_IT address registers instead of ten 64-bit pointers — the 168-register budget of 3 workgroups per
  // CU then leaves no spill in the loop (a spilled, freshly loaded register made every K-tile wait for a memory round trip).
  const int a_chunk = tid % ACH, a_row0 = tid / ACH;
  uint32_t a_off[A_IT];  // GEN: signed byte offset of the row's (ho*SH - PH, wo*SW - PW) pixel, < 2 GB in magnitude (host check)
  int a_hi0[GEN ? A_IT : 1], a_wi0[GEN ? A_IT : 1];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    long m = (long)tile_m * BM + a_row0 + i * ARP;
    if (m >= p.M) m = p.M - 1;  // rows past the end: valid addresses, values never stored
    if constexpr (GEN) {
      const long hw = (long)p.Ho * p.Wo;
      const int n = (int)(m / hw);
      const int rem = (int)(m - (long)n * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.SH - p.PH;
      a_wi0[i] = wo * p.SW - p.PW;
      a_off[i] = (uint32_t)(int32_t)(((long)n * p.sN + (long)a_hi0[i] * p.sH + (long)a_wi0[i] * p.sW + a_chunk * 4) * 4);
    } else {
      // 1x1 / stride 1 / no padding: row m of the GEMM is pixel m of the NHWC input (row stride sW elements); < 4 GB (host check)
      a_off[i] = (uint32_t)((m * p.sW + a_chunk * 4) * 4);
    }
  }
  int t_kh = 0, t_kw = 0, t_c0 = 0;  // GEN: tap and channel offset of the NEXT K-tile load_tile() will fetch
  const int b_chunk = tid % BCH, b_row0 = tid / BCH;
  uint32_t b_off[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    int co = tile_n * BN + b_row0 + j * BRP;
    if (co >= p.Cout) co = p.Cout - 1;
    b_off[j] = (uint32_t)(co * 64 + b_chunk * 16);   // K-tile-major planes: [k/32][Cout][32]
  }
  const long wplane_b = p.wplane * 2;

  u32x4 ra[A_IT], rb[3][B_IT];
  auto load_tile = [&](int kt) {
    if constexpr (GEN) {
      const int32_t koff = (int32_t)(((long)t_kh * p.sH + (long)t_kw * p.sW + t_c0) * 4);  // uniform
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int hi = a_hi0[i] + t_kh, wi = a_wi0[i] + t_kw;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        // 32-bit offsets kept in registers, the 64-bit address formed per load: four address registers less than four pointers, which
        // is what keeps this loop free of spill reloads at 168 registers
        const char* src = g_in + (long)(int32_t)(a_off[i] + (uint32_t)koff);
        ra[i] = *reinterpret_cast<const u32x4*>(ok ? src : reinterpret_cast<const char*>(msocr_split_zero16));
      }
      t_c0 += BK;
      if (t_c0 == p.Cin) {
        t_c0 = 0;
        if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
      }
    } else {
      const char* const ga = g_in + (long)kt * (BK * 4);  // uniform
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ga + a_off[i]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const char* const gb = g_w + pl * wplane_b + (long)kt * p.w_kt_b;  // uniform
#pragma unroll
      for (int j = 0; j < B_IT; ++j)
        if (BN % BRP == 0 || b_row0 + j * BRP < BN) rb[pl][j] = *reinterpret_cast<const u32x4*>(gb + b_off[j]);
    }
  };
  auto store_tile = [&](int stage) {
    unsigned char* const sA = smem + stage * STAGE_B;  // [3][BM][ROWB] (one stage: stage == 0)
    unsigned char* const sB = sA + 3 * A_PLANE;        // [3][BN][ROWB]
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = a_row0 + i * ARP;
      float x0 = __uint_as_float(ra[i][0]), x1 = __uint_as_float(ra[i][1]), x2 = __uint_as_float(ra[i][2]), x3 = __uint_as_float(ra[i][3]);
      // this thread's 4 elements are bf16 positions 4 * a_chunk .. + 3 of the row: half of 16-B chunk a_chunk / 2
      unsigned char* dst = sA + row * ROWB + (((a_chunk >> 1) ^ swz<ROWB>(row)) << 4) + ((a_chunk & 1) << 3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        u32x2 v;
        v[0] = split_step(x0, x1);
        v[1] = split_step(x2, x3);
        *reinterpret_cast<u32x2*>(dst + pl * A_PLANE) = v;
      }
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < B_IT; ++j) {
        const int row = b_row0 + j * BRP;
        if (BN % BRP == 0 || row < BN)
          *reinterpret_cast<u32x4*>(sB + pl * B_PLANE + row * ROWB + ((b_chunk ^ swz<ROWB>(row)) << 4)) = rb[pl][j];
      }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int r32 = lane & 31, half = lane >> 5;

  auto compute = [&](int stage) {
    const unsigned char* const sA = smem + stage * STAGE_B;
    const unsigned char* const sB = sA + 3 * A_PLANE;
#pragma unroll
    for (int q = 0; q < BK / 16; ++q) {        // k16 steps per K-tile
      const int c = 2 * q + half;
      bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WM + i * 32 + r32;
          fa[pl][i] = *reinterpret_cast<const bf16x8*>(sA + pl * A_PLANE + row * ROWB + ((c ^ swz<ROWB>(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WN + j * 32 + r32;
          fb[pl][j] = *reinterpret_cast<const bf16x8*>(sB + pl * B_PLANE + row * ROWB + ((c ^ swz<ROWB>(row)) << 4));
        }
      }
      // smallest terms first; consecutive MFMAs of one product class hit different accumulators
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[PB[t]][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  // One LDS stage, two barriers per K-tile, 3 workgroups per CU.  Measured alternatives (round 3, gpurun_out/r3_split_variants.txt,
  // f32-equivalent TFLOP/s at K = 512 / 4096 on M = 161280, N = 512): this loop 156 / 182; two LDS stages with K-tiles of 16 (one
  // barrier per tile, 3 per CU) 145 / 148; two stages with K-tiles of 32 (1 per CU) 120 / 148; K-tiles of 16 at 4 per CU 126 / 94;
  // 2 per CU 154 / 176; the weight planes straight into LDS (global_load_lds_dwordx4, double-buffered, 2 per CU) 158 / 189 against
  // 166 / 191 for this loop on the same box.  Timing ablations of this loop: MFMAs + fragment reads alone 266, + restaging 228,
  // + loads that always hit the cache 190.
  for (int kt = 0; kt < p.ktiles; ++kt) {
    if (kt + 1 < p.ktiles) load_tile(kt + 1);  // global loads in flight under the MFMAs
    compute(0);
    __syncthreads();  // everyone is done reading the stage before it is overwritten
    if (kt + 1 < p.ktiles) store_tile(0);
    __syncthreads();
  }

  // ---- epilogue (conv_igemm_kernel's): TM passes of (acc row-block -> LDS [64][BN] f32 -> bias/residual/ReLU -> 16-B stores) ----
  constexpr int PR = (BM / WM) * 32;
  float* sc = reinterpret_cast<float*>(smem);
  constexpr int VPR = BN / 4;
  constexpr int ROWS_PP = 256 / VPR;
  const int vcol = (tid % VPR) * 4;
  const int vrow0 = tid / VPR;
  const int co = tile_n * BN + vcol;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rit = (e & 3) + 8 * (e >> 2) + 4 * half;
        sc[(wm * 32 + rit) * BN + wn * WN + j * 32 + r32] = acc[i][j][e];
      }
    __syncthreads();
    for (int lr = vrow0; lr < PR; lr += ROWS_PP) {
      const int trow = (lr >> 5) * WM + i * 32 + (lr & 31);
      const long m = (long)tile_m * BM + trow;
      if (m >= p.M) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(&sc[lr * BN + vcol]) + bias;
      if (p.has_res) v += *reinterpret_cast<const f32x4*>(p.res + (m * p.res_ld + co) * 4);
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      *reinterpret_cast<f32x4*>(g_out + (m * p.out_ld + co) * 4) = v;
    }
  }
}

template <int BN, int WPE, bool GEN>
int launch_split(ConvParams& p, hipStream_t s) {
  p.tilesM = (int)((p.M + 127) / 128);
  p.tilesN = p.Cout / BN;
  p.ktiles = (int)(p.Ktot / 32);
  constexpr int STAGE = 3 * (128 + BN) * 64;
  constexpr int EPI = 64 * BN * 4;
  constexpr int LDS = STAGE > EPI ? STAGE : EPI;
  auto kern = conv_split_kernel<BN, WPE, GEN>;
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

// Kernel choice.  MSOCR_SPLIT_PP (default 1): shapes the producer / consumer kernel of conv_split_pp.hip has an instance for
// (Cout % 128 == 0) go there; 0 keeps everything on conv_split_kernel (diagnostics, A/B timing).
int launch_split_one(ConvParams& p, hipStream_t s, bool general) {
  static int use_pp = -1;
  if (use_pp < 0) {
    const char* e = getenv("MSOCR_SPLIT_PP");
    use_pp = e ? atoi(e) : 1;
  }
  if (use_pp && msocr_internal_split_pp_takes(p)) return msocr_internal_split_pp_launch(p, s, general);
  if (general) return p.Cout % 128 == 0 ? launch_split<128, 3, true>(p, s) : launch_split<64, 3, true>(p, s);
  return p.Cout % 128 == 0 ? launch_split<128, 3, false>(p, s) : launch_split<64, 3, false>(p, s);
}

// Both loaders address with a per-thread 32-bit byte offset from a wave-uniform base: the lean one unsigned (rows of ONE problem
// must span less than 4 GB), the general one signed (an image range below 2 GB: msocr_conv2d_split splits the batch).  A lean
// single-problem launch whose rows span more is cut into row ranges here (rows are independent; bases advance in 64 bits).
int launch_split_any(ConvParams& p, hipStream_t s, bool general) {
  if (((long)p.Cout * p.Ktot + 32) * 2 >= (1L << 32)) return MSOCR_E_ARG;
  p.w_kt_b = (long)p.Cout * 64;
  if (general) return launch_split_one(p, s, true);
  const long row_b = p.sW * 4;
  const long limit = (1L << 32) - 4096;
  if ((p.M * p.sW + 32) * 4 < limit) return launch_split_one(p, s, false);
  if (p.nbatch != 1 || row_b <= 0 || row_b * 256 >= limit) return MSOCR_E_ARG;
  const long rows_per = (limit / row_b) & ~255L;   // a multiple of every M-tile
  for (long m0 = 0; m0 < p.M; m0 += rows_per) {
    ConvParams q = p;
    q.M = p.M - m0 < rows_per ? p.M - m0 : rows_per;
    q.in = p.in + m0 * row_b;
    q.out = p.out + m0 * p.out_ld * 4;
    if (p.has_res) q.res = p.res + m0 * p.res_ld * 4;
    const int rc = launch_split_one(q, s, false);
    if (rc != MSOCR_OK) return rc;
  }
  return MSOCR_OK;
}

}  // namespace

// HOST helper: w [n] f32 -> planes [3][n] bf16 with w == p0 + p1 + p2 exactly (round-to-nearest-even residual chain, the same
// arithmetic the kernel applies to its activation operand).
extern "C" int msocr_split_bf16x3_host(const float* w_host, int64_t n, uint16_t* planes_out_host) {
  if (!w_host || !planes_out_host || n <= 0) return MSOCR_E_ARG;
  auto rne = [](float f) -> uint16_t {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  auto up = [](uint16_t h) -> float {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
  };
  for (int64_t i = 0; i < n; ++i) {
    float r = w_host[i];
    for (int pl = 0; pl < 3; ++pl) {
      const uint16_t h = rne(r);
      planes_out_host[pl * n + i] = h;
      r -= up(h);
    }
  }
  return MSOCR_OK;
}

// HOST helper: the same split of w [nbatch][rows][k], planes written K-tile-major [3][nbatch][k/32][rows][32] (include/msocr.h).
extern "C" int msocr_split_bf16x3_ktile_host(const float* w_host, int64_t nbatch, int64_t rows, int64_t k, uint16_t* planes_out_host) {
  if (!w_host || !planes_out_host || nbatch <= 0 || rows <= 0 || k <= 0 || k % 32) return MSOCR_E_ARG;
  const int64_t n = nbatch * rows * k;
  uint16_t* tmp = (uint16_t*)malloc((size_t)n * 3 * sizeof(uint16_t));
  if (!tmp) return MSOCR_E_ARG;
  const int rc = msocr_split_bf16x3_host(w_host, n, tmp);
  if (rc == MSOCR_OK) {
    const int64_t kt = k / 32;
    for (int pl = 0; pl < 3; ++pl)
      for (int64_t b = 0; b < nbatch; ++b)
        for (int64_t r = 0; r < rows; ++r)
          for (int64_t t = 0; t < kt; ++t)
            memcpy(planes_out_host + pl * n + ((b * kt + t) * rows + r) * 32, tmp + pl * n + (b * rows + r) * k + t * 32, 64);
  }
  free(tmp);
  return rc;
}

// 1x1 / stride 1 / no padding convolution (a GEMM over pixels) with the weight operand given as three bf16 planes,
// K-tile-major [3][Cin/32][Cout][32] (plane stride Cout * Cin).  Same descriptor, epilogue flags and error behaviour as msocr_conv2d.
extern "C" int msocr_conv1x1_split(const msocr_conv_desc* d, const void* in, const void* weight_planes, const float* bias,
                                   const void* residual, void* out, void* stream) {
  if (!d || !in || !weight_planes || !out) return MSOCR_E_ARG;
  if (d->dtype != MSOCR_F32 || d->KH != 1 || d->KW != 1 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h || d->pad_w) return MSOCR_E_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho != d->H || d->Wo != d->W) return MSOCR_E_ARG;
  if (d->Cin <= 0 || d->Cin % 32 || d->Cout <= 0 || d->Cout % 64) return MSOCR_E_ARG;
  // the pixels must form ONE dense row sequence: pixel stride sW, rows and images contiguous in units of it
  if (d->in_sW % 4 || d->in_sW < d->Cin || (d->H > 1 && d->in_sH != (int64_t)d->W * d->in_sW) ||
      (d->N > 1 && d->in_sN != (int64_t)d->H * d->W * d->in_sW)) return MSOCR_E_ARG;
  if (d->out_ld % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;
  if (((uintptr_t)in | (uintptr_t)weight_planes | (uintptr_t)out) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  ConvParams p = {};
  p.in = (const char*)in; p.w = (const char*)weight_planes; p.bias = bias; p.res = (const char*)residual; p.out = (char*)out;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin;
  p.sN = d->in_sN; p.sH = d->in_sH; p.sW = d->in_sW;
  p.KH = p.KW = 1; p.SH = p.SW = 1; p.PH = p.PW = 0;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.M = (long)d->N * d->Ho * d->Wo;
  p.Ktot = d->Cin;
  p.out_ld = d->out_ld; p.res_ld = d->res_ld;
  p.relu = (d->flags & MSOCR_CONV_RELU) ? 1 : 0;
  p.has_res = has_res ? 1 : 0;
  p.nbatch = 1; p.bsA = p.bsW = p.bsO = 0;
  p.wplane = (long)d->Cout * d->Cin;
  return launch_split_any(p, (hipStream_t)stream, false);
}

// Any kernel size / stride / padding with the weight operand as three K-tile-major bf16 planes of [Cout][KH*KW*Cin] (the strided 3x3 and 1x1
// convolutions of the ResNet trunks that have no Winograd form).  Same descriptor rules as msocr_conv2d for MSOCR_F32, plus
// Cin % 32 == 0 and Cout % 64 == 0.
extern "C" int msocr_conv2d_split(const msocr_conv_desc* d, const void* in, const void* weight_planes, const float* bias,
                                  const void* residual, void* out, void* stream) {
  if (!d || !in || !weight_planes || !out || d->dtype != MSOCR_F32) return MSOCR_E_ARG;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return MSOCR_E_ARG;
  if (d->Cin <= 0 || d->Cin % 32 || d->Cout <= 0 || d->Cout % 64) return MSOCR_E_ARG;
  if (d->KH <= 0 || d->KW <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0) return MSOCR_E_ARG;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld % 4 || d->out_ld < d->Cout) return MSOCR_E_ARG;  // 16-byte vector accesses
  if (((uintptr_t)in | (uintptr_t)weight_planes | (uintptr_t)out) & 15) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  // output extent must agree with the conv arithmetic (guards the kernel's indexing)
  if ((d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 < d->Ho || (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1 < d->Wo)
    return MSOCR_E_ARG;
  ConvParams p = {};
  p.in = (const char*)in; p.w = (const char*)weight_planes; p.bias = bias; p.res = (const char*)residual; p.out = (char*)out;
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
  p.wplane = (long)d->Cout * p.Ktot;
  // the general loader keeps signed 32-bit byte offsets from the input pointer: a batch whose input extent reaches 2 GB is launched
  // image range by image range (outputs of different images are independent)
  const long img_bytes = (long)d->in_sN * 4;
  if (img_bytes >= (1L << 31)) return MSOCR_E_ARG;
  const int per = (int)(((1L << 31) - 1) / (img_bytes > 0 ? img_bytes : 1));
  if (d->N <= per) return launch_split_any(p, (hipStream_t)stream, true);
  const long out_img = (long)d->Ho * d->Wo * d->out_ld * 4, res_img = (long)d->Ho * d->Wo * d->res_ld * 4;
  for (int n0 = 0; n0 < d->N; n0 += per) {
    ConvParams q = p;
    q.N = d->N - n0 < per ? d->N - n0 : per;
    q.M = (long)q.N * d->Ho * d->Wo;
    q.in = p.in + (long)n0 * img_bytes;
    q.out = p.out + (long)n0 * out_img;
    if (has_res) q.res = p.res + (long)n0 * res_img;
    const int rc = launch_split_any(q, (hipStream_t)stream, true);
    if (rc != MSOCR_OK) return rc;
  }
  return MSOCR_OK;
}

// nbatch independent GEMMs of one shape in ONE launch, C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]: A f32 [nbatch][M][K],
// B as three K-tile-major bf16 planes [3][nbatch][K/32][N][32], C f32 [nbatch][M][N].  The Winograd-domain GEMMs (winograd.hip).
int msocr_internal_gemm_split_batched(const float* A, const uint16_t* Bplanes, float* C, long M, int N, int K, int nbatch, hipStream_t s) {
  if (!A || !Bplanes || !C || M <= 0 || N <= 0 || N % 64 || K <= 0 || K % 32 || nbatch <= 0) return MSOCR_E_ARG;
  if (((uintptr_t)A | (uintptr_t)Bplanes | (uintptr_t)C) & 15) return MSOCR_E_ARG;
  if (M > 0x7fffffffL) return MSOCR_E_ARG;
  ConvParams p = {};
  p.in = (const char*)A; p.w = (const char*)Bplanes; p.bias = nullptr; p.res = nullptr; p.out = (char*)C;
  p.N = 1; p.H = (int)M; p.W = 1; p.Cin = K;
  p.sN = M * (long)K; p.sH = K; p.sW = K;
  p.KH = p.KW = 1; p.SH = p.SW = 1; p.PH = p.PW = 0;
  p.Ho = (int)M; p.Wo = 1; p.Cout = N;
  p.M = M; p.Ktot = K;
  p.out_ld = N; p.res_ld = 0;
  p.relu = 0; p.has_res = 0;
  p.nbatch = nbatch; p.bsA = M * (long)K; p.bsW = (long)N * K; p.bsO = M * (long)N;
  p.wplane = (long)nbatch * N * K;
  return launch_split_any(p, s, false);
}
