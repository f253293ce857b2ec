// winograd_rs.hip — Winograd F(2x2,3x3) with the transforms SPLIT BY AXIS around a 4-point GEMM ("row-split" form), f32, gfx950.
//
// The plain form (winograd.hip) materialises V = B^T d B (16 values per 2x2-output tile and channel: 4x the input) and
// Mw = sum_c U (.) V (16 per tile and output channel: 4x the output), and streams both through HBM twice: 18 units of traffic
// per layer against 2 for the convolution's own input + output.  Here a GEMM workgroup owns the FOUR transform points of one
// transform ROW xi (points (xi, nu), nu = 0..3) of its tiles, so that
//   * the input transform's column pass runs in the GEMM's operand loader: the loader reads Q[xi] = (B^T d)[xi] — the row pass,
//     a dense array over (tile row, xi, input column) = 2x the input, shared by horizontally adjacent tiles — and forms the four
//     V[xi][nu] in registers (the same subtractions / additions in the same order as wino_input_kernel: V is bit-identical);
//   * the output transform's column pass runs in the GEMM's epilogue: the four accumulators of a (tile, cout) element are
//     lane-local, R[xi][j] = (M A)[xi][j] is 2 values instead of 4, so the array handed to the last kernel is 2x the output.
// Traffic per layer: input 1 + Q (2 + 2) + R (2 + 2) + output 1 = 10 units instead of 18; the two streaming kernels move 3 units
// each instead of 5.  The output's additions happen in the other order (columns first), i.e. the result differs from the plain
// form by rounding only.
//   1. wino_rows_in_kernel : Q[xi][n*TH+th][wq][c] = (B^T d)[xi] at padded input column wq        (HBM-bound)
//   2. wino_gemm4_kernel   : R[xi*2+j][tile][co] = sum_nu A^T-column-pass of ( sum_c V[xi][nu][tile][c] * U[xi*4+nu][co][c] )
//                            64 tiles x 128 couts x 4 points per workgroup, v_mfma_f32_16x16x4_f32, K-tiles of 16 channels
//   3. wino_rows_out_kernel: out[2x2 tile][co] = act(A^T-row-pass(R) + bias (+ residual))          (HBM-bound)
// Same reference code as winograd.hip: every 3x3 / stride 1 / pad 1 convolution of seresnet31.py:37-45,81-89 and east.py:13-30,
// 56-67 with Cin >= 128.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct RsGeom {
  int N, H, W, TH, TW, Wq;  // Wq = 2*TW + 2 padded input columns per tile row
  long Mt;                  // N * TH * TW tiles
  long rowsQ;               // N * TH tile rows
};

// ---- 1. row pass of the input transform: one thread = (tile row, padded column, 4 channels) ---------------------------------
__global__ __launch_bounds__(256) void wino_rows_in_kernel(const float* __restrict__ in, long sN, long sH, long sW, int C, RsGeom g,
                                                            float* __restrict__ Q) {
  const int cch = C >> 2;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid / cch;
  const int c = (int)(gid - t * cch) << 2;
  if (t >= g.rowsQ * g.Wq) return;
  const int wq = (int)(t % g.Wq);
  const long rq = t / g.Wq;
  const int th = (int)(rq % g.TH);
  const int n = (int)(rq / g.TH);
  const int w = wq - 1, h0 = 2 * th - 1;
  const bool okw = (unsigned)w < (unsigned)g.W;
  const float* base = in + (long)n * sN + c;
  f32x4 d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hi = h0 + i;
    const bool ok = okw && (unsigned)hi < (unsigned)g.H;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? base + (long)hi * sH + (long)w * sW : base);
    d[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const long plane = g.rowsQ * g.Wq * (long)C;
  float* o = Q + (rq * g.Wq + wq) * (long)C + c;
  *reinterpret_cast<f32x4*>(o) = d[0] - d[2];
  *reinterpret_cast<f32x4*>(o + plane) = d[1] + d[2];
  *reinterpret_cast<f32x4*>(o + 2 * plane) = d[2] - d[1];
  *reinterpret_cast<f32x4*>(o + 3 * plane) = d[1] - d[3];
}

// ---- 2. the 4-point GEMM ---------------------------------------------------------------------------------------------------------
struct Gemm4Params {
  const float* Q;   // [4][rowsQ][Wq][Cin]
  const float* U;   // [16][Cout][Cin]
  float* R;         // [8][Mt][Cout]
  RsGeom g;
  int Cin, Cout, ktiles, tilesM, tilesN;
};

constexpr int RS_BN = 128, RS_BKB = 64;  // BM tiles x 128 couts per workgroup, K-tiles of 16 f32 (64-byte rows)
constexpr int RS_B_BYTES = 4 * RS_BN * RS_BKB;
template <int BM> constexpr int rs_lds() { return (4 * BM * RS_BKB + RS_B_BYTES) > BM * RS_BN * 4 ? (4 * BM * RS_BKB + RS_B_BYTES) : BM * RS_BN * 4; }

__device__ __forceinline__ int rs_swz(int row) { return (row >> 2) & 3; }  // 64-byte rows: 4 rows per 256-byte bank line

// RS_BM = 64: 128 accumulator registers per lane, 2 workgroups per CU; RS_BM = 32: 64 registers, 3 per CU, twice the weight re-reads
template <int RS_BM>
__global__ __launch_bounds__(256, RS_BM == 64 ? 2 : 3) void wino_gemm4_kernel(Gemm4Params p) {
  constexpr int RS_A_BYTES = 4 * RS_BM * RS_BKB;
  constexpr int TMR = RS_BM / 16;           // 16-row MFMA tiles per wave along the tile dimension
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sa = smem;                 // [4 points][BM rows][64 B]
  unsigned char* sb = smem + RS_A_BYTES;    // [4 points][128 rows][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;

  // XCD-aware order: each XCD gets a contiguous range of logical blocks; xi outermost, then the cout block, tile block innermost,
  // so neighbouring workgroups share one 4-point weight slice [4][128][Cin] in their L2
  const int per_xi = p.tilesM * p.tilesN, nblk = 4 * per_xi;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int xi = bid / per_xi;
  bid -= xi * per_xi;
  const int tile_n = bid / p.tilesM, tile_m = bid - tile_n * p.tilesM;

  // ---- operand staging coordinates
  const int a_row = tid >> 2, chunk = tid & 3;  // A: one (tile row, 16-byte chunk) per thread (the first 4 * BM threads)
  const bool a_on = a_row < RS_BM;
  const float* a_ptr;
  {
    long m = (long)tile_m * RS_BM + (a_on ? a_row : 0);
    if (m >= p.g.Mt) m = p.g.Mt - 1;  // rows past the end load valid data that the epilogue never stores
    const long rq = m / p.g.TW;
    const int tw = (int)(m - rq * p.g.TW);
    a_ptr = p.Q + (long)xi * p.g.rowsQ * p.g.Wq * p.Cin + (rq * p.g.Wq + 2 * tw) * (long)p.Cin + chunk * 4;
  }
  const float* b_ptr[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = a_row + 64 * it;  // 0..511 = point * 128 + cout
    const int pt = row >> 7, co = row & 127;
    b_ptr[it] = p.U + ((long)(xi * 4 + pt) * p.Cout + (long)tile_n * RS_BN + co) * p.Cin + chunk * 4;
  }
  f32x4 qa[4];
  u32x4 rb[8];
  auto load_tile = [&](int kt) {
    if (a_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) qa[j] = *reinterpret_cast<const f32x4*>(a_ptr + (long)j * p.Cin + kt * 16);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) rb[it] = *reinterpret_cast<const u32x4*>(b_ptr[it] + kt * 16);
  };
  auto store_tile = [&]() {
    // column pass of the input transform (the second half of B^T d B), same operations as wino_input_kernel
    if (a_on) {
      const f32x4 v0 = qa[0] - qa[2], v1 = qa[1] + qa[2], v2 = qa[2] - qa[1], v3 = qa[1] - qa[3];
      unsigned char* dst = sa + a_row * RS_BKB + ((chunk ^ rs_swz(a_row)) << 4);
      *reinterpret_cast<f32x4*>(dst) = v0;
      *reinterpret_cast<f32x4*>(dst + RS_BM * RS_BKB) = v1;
      *reinterpret_cast<f32x4*>(dst + 2 * RS_BM * RS_BKB) = v2;
      *reinterpret_cast<f32x4*>(dst + 3 * RS_BM * RS_BKB) = v3;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = a_row + 64 * it;
      *reinterpret_cast<u32x4*>(sb + row * RS_BKB + ((chunk ^ rs_swz(row & 127)) << 4)) = rb[it];
    }
  };

  f32x4 acc[4][TMR][2];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int i = 0; i < TMR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[pt][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int kt = 0; kt < p.ktiles; ++kt) {
    if (kt + 1 < p.ktiles) load_tile(kt + 1);  // global loads in flight under the MFMAs
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      u32x4 fa[TMR], fb[2];
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        const int row = i * 16 + r16;
        fa[i] = *reinterpret_cast<const u32x4*>(sa + pt * RS_BM * RS_BKB + row * RS_BKB + ((kq ^ rs_swz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wave * 32 + j * 16 + r16;
        fb[j] = *reinterpret_cast<const u32x4*>(sb + pt * RS_BN * RS_BKB + row * RS_BKB + ((kq ^ rs_swz(row)) << 4));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TMR; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[pt][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fa[i][e]), __uint_as_float(fb[j][e]), acc[pt][i][j], 0, 0, 0);
    }
    __syncthreads();  // everyone is done reading the stage
    if (kt + 1 < p.ktiles) store_tile();
    __syncthreads();
  }

  // ---- epilogue: column pass of the output transform (A^T M A's "M A"), then the two R planes through LDS -> 16-byte stores
  float* sc = reinterpret_cast<float*>(smem);  // [BM rows][128 couts] f32
  const int vcol = (tid & 31) * 4, vrow0 = tid >> 5;  // 32 threads per row of 128 couts, 8 rows per sweep
#pragma unroll
  for (int jo = 0; jo < 2; ++jo) {
    if (jo) __syncthreads();
#pragma unroll
    for (int i = 0; i < TMR; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m0 = acc[0][i][j][e], m1 = acc[1][i][j][e], m2 = acc[2][i][j][e], m3 = acc[3][i][j][e];
          const float r = jo == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
          sc[(i * 16 + 4 * kq + e) * RS_BN + wave * 32 + j * 16 + r16] = r;
        }
    __syncthreads();
    float* plane = p.R + (long)(xi * 2 + jo) * p.g.Mt * p.Cout;
    for (int lr = vrow0; lr < RS_BM; lr += 8) {
      const long m = (long)tile_m * RS_BM + lr;
      if (m >= p.g.Mt) continue;
      *reinterpret_cast<f32x4*>(plane + m * p.Cout + (long)tile_n * RS_BN + vcol) = *reinterpret_cast<const f32x4*>(&sc[lr * RS_BN + vcol]);
    }
  }
}

// ---- 3. row pass of the output transform: one thread = one tile x 4 output channels -------------------------------------------
__global__ __launch_bounds__(256) void wino_rows_out_kernel(const float* __restrict__ R, int Cout, RsGeom g, const float* __restrict__ bias,
                                                             const float* __restrict__ res, long res_ld, int relu, float* __restrict__ out,
                                                             long out_ld) {
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
  const float* rp = R + t * (long)Cout + c;
  f32x4 s[4][2];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int j = 0; j < 2; ++j) s[xi][j] = *reinterpret_cast<const f32x4*>(rp + (xi * 2 + j) * plane);
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias) b = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ho = 2 * th + i;
    if (ho >= g.H) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wo = 2 * tw + j;
      if (wo >= g.W) continue;
      const f32x4 y = i == 0 ? (s[0][j] + s[1][j]) + s[2][j] : (s[1][j] - s[2][j]) - s[3][j];
      const long pix = ((long)n * g.H + ho) * g.W + wo;
      f32x4 v = y + b;
      if (res) v += *reinterpret_cast<const f32x4*>(res + pix * res_ld + c);
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
  }
}

bool rs_geom(const msocr_conv_desc* d, RsGeom* g) {
  if (!d || d->dtype != MSOCR_F32) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin <= 0 || d->Cin % 16 || d->Cout <= 0 || d->Cout % RS_BN) return false;
  if (d->in_sN % 4 || d->in_sH % 4 || d->in_sW % 4 || d->out_ld % 4 || d->out_ld < d->Cout) return false;
  g->N = d->N; g->H = d->H; g->W = d->W;
  g->TH = (d->H + 1) / 2; g->TW = (d->W + 1) / 2;
  g->Wq = 2 * g->TW + 2;
  g->rowsQ = (long)d->N * g->TH;
  g->Mt = g->rowsQ * g->TW;
  return g->Mt <= 0x7fffffffL && g->rowsQ * g->Wq <= 0x7fffffffL;
}
inline long rs_q_floats(const msocr_conv_desc* d, const RsGeom& g) { return 4L * g.rowsQ * g.Wq * d->Cin; }

}  // namespace

extern "C" int64_t msocr_conv3x3_winograd_rs_workspace_bytes(const msocr_conv_desc* d) {
  RsGeom g;
  if (!rs_geom(d, &g)) return -1;
  return (rs_q_floats(d, g) + 8L * g.Mt * d->Cout) * (int64_t)sizeof(float);
}

extern "C" int msocr_winograd_rs_rows_in(const msocr_conv_desc* d, const void* in, void* workspace, void* stream) {
  RsGeom g;
  if (!rs_geom(d, &g) || !in || !workspace || (((uintptr_t)in | (uintptr_t)workspace) & 15)) return MSOCR_E_ARG;
  const long nthr = g.rowsQ * g.Wq * (d->Cin / 4), nb = (nthr + 255) / 256;
  if (nb > 0x7fffffffL) return MSOCR_E_ARG;
  MSOCR_LAUNCH(wino_rows_in_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)in, (long)d->in_sN, (long)d->in_sH,
               (long)d->in_sW, d->Cin, g, (float*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_winograd_rs_gemm(const msocr_conv_desc* d, const float* u_weight, void* workspace, void* stream) {
  RsGeom g;
  if (!rs_geom(d, &g) || !u_weight || !workspace || (((uintptr_t)u_weight | (uintptr_t)workspace) & 15)) return MSOCR_E_ARG;
  Gemm4Params p;
  p.Q = (const float*)workspace;
  p.U = u_weight;
  p.R = (float*)workspace + rs_q_floats(d, g);
  p.g = g;
  p.Cin = d->Cin; p.Cout = d->Cout;
  p.ktiles = d->Cin / 16;
  const int bm = getenv("MSOCR_WINO_RS_BM") ? atoi(getenv("MSOCR_WINO_RS_BM")) : 64;  // 32: 3 workgroups per CU (diagnostic)
  p.tilesM = (int)((g.Mt + bm - 1) / bm);
  p.tilesN = d->Cout / RS_BN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm4_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, rs_lds<64>()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm4_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, rs_lds<32>()) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr_set = true;
  }
  const long nblk = 4L * p.tilesM * p.tilesN;
  if (nblk > 0x7fffffffL) return MSOCR_E_ARG;
  if (bm == 32)
    MSOCR_LAUNCH(wino_gemm4_kernel<32>, dim3((unsigned)nblk), dim3(256), rs_lds<32>(), (hipStream_t)stream, p);
  else
    MSOCR_LAUNCH(wino_gemm4_kernel<64>, dim3((unsigned)nblk), dim3(256), rs_lds<64>(), (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_winograd_rs_rows_out(const msocr_conv_desc* d, const void* workspace, const float* bias, const void* residual, void* out,
                                          void* stream) {
  RsGeom g;
  if (!rs_geom(d, &g) || !out || !workspace || (((uintptr_t)out | (uintptr_t)workspace) & 15)) return MSOCR_E_ARG;
  const bool has_res = (d->flags & MSOCR_CONV_RESIDUAL) != 0;
  if (has_res && (!residual || d->res_ld % 4 || d->res_ld < d->Cout || ((uintptr_t)residual & 15))) return MSOCR_E_ARG;
  if (bias && ((uintptr_t)bias & 15)) return MSOCR_E_ARG;
  const long nb = (g.Mt * (d->Cout / 4) + 255) / 256;
  if (nb > 0x7fffffffL) return MSOCR_E_ARG;
  const float* R = (const float*)workspace + rs_q_floats(d, g);
  MSOCR_LAUNCH(wino_rows_out_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, R, d->Cout, g, bias,
               has_res ? (const float*)residual : nullptr, (long)d->res_ld, (d->flags & MSOCR_CONV_RELU) ? 1 : 0, (float*)out, (long)d->out_ld);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

extern "C" int msocr_conv3x3_winograd_rs(const msocr_conv_desc* d, const void* in, const float* u_weight, const float* bias,
                                         const void* residual, void* out, void* workspace, void* stream) {
  int rc = msocr_winograd_rs_rows_in(d, in, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  rc = msocr_winograd_rs_gemm(d, u_weight, workspace, stream);
  if (rc != MSOCR_OK) return rc;
  return msocr_winograd_rs_rows_out(d, workspace, bias, residual, out, stream);
}
