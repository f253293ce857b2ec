// conv_split_pp.hip — the split-operand ("bf16x3") f32 GEMM / convolution of conv_split.hip as a PRODUCER / CONSUMER workgroup.
//
// Same arithmetic idea as conv_split_kernel (an f32 value is the exact sum of three bf16 values; six of the nine cross products on
// the bf16 matrix pipes, smallest terms first, f32 accumulate), different machine mapping (round 4):
//
//   * 512-thread workgroup, TWO waves on every SIMD with fixed roles (MI355X_MICROARCH.md, "Two waves per SIMD"):
//       waves 0-3  CONSUMERS: ds_read_b128 + MFMA only.  2 x 2 waves, wave tile 64 x 128 (or 128 x 64) = 32 blocks of 16 x 16, the
//                  128 accumulator registers live across the whole K loop.
//       waves 4-7  PRODUCERS: global -> registers (two K-tiles in flight) -> exact split in registers -> LDS.  No packed-f32 VALU
//                  (the Makefile compiles this file without them): 16 v_pk_add_f32 per K-tile beside the partner's MFMAs cost 0.13
//                  of the matrix pipes' busy fraction (profiles/r04_pp_ablations.txt).
//     Two LDS stages of six [rows][64 B] planes (72 KB each), ONE barrier per K-tile (conv_split_kernel: two, and half the MFMAs
//     between them).  Persistent: a static, XCD-contiguous list of output tiles per workgroup; the producers run ahead across tile
//     boundaries.
//   * v_mfma_f32_16x16x32_bf16 instead of 32x32x16: on random data the chip holds a higher clock on it — 337 against 299 TFLOP/s
//     f32-equivalent for this very loop (tools/microbench/mfma_energy.hip, profiles/r04_mfma_energy_microbench.txt); the kernel is
//     power-limited (0.83 pipe-busy at 1.42 GHz at K = 4096), so that is wall time.
//   * CHUNKED ACCUMULATION: the six products of one 32-deep K-tile are chained from C = 0 into a 4-register partial sum (roundings
//     at the size of a 32-term partial dot product) and that partial is added to the accumulator ONCE (v_add_f32 in the MFMAs'
//     shadow).  conv_split_kernel rounds at the accumulator's magnitude 12 times per 32 k, this kernel once: measured against an
//     f64 evaluation of the whole network the device's error drops accordingly (tests/test_gpu_f64.py).
//   * The epilogue overlaps the next tile: block (i, j) of the finished tile is stored (bias, ReLU; 16-byte stores, each lane 4
//     consecutive output channels — the MFMA runs with the weight fragment as its A operand, so a lane holds D[n .. n+3][m]) just
//     before the next tile's first K-tile overwrites it; the stores drain under that tile's K loop.  Bias values reach the
//     consumers through LDS (written by the producers with every K-tile), so the consumers issue no vector-memory loads at all.
//
// Tile 128 x 256 (Cout % 256 == 0) or 256 x 128 (Cout % 128 == 0); launches with a residual operand and all other shapes stay on
// conv_split_kernel.  Reference layers: recognizers/_trba/model/seresnet31.py:37-67 ; detectors/_east/east.py:13-30 ; torchvision
// Bottleneck.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "internal.h"
#include "msocr.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t msocr_pp_zero16[4] = {0u, 0u, 0u, 0u};

template <int I>
using ic = std::integral_constant<int, I>;
template <int U, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (U < N) {
    f(ic<U>{});
    static_for<U + 1, N>(f);
  }
}

// every LDS access issued so far has completed (producers: stores landed; consumers: fragments in registers), then the workgroup
// barrier.  Inline asm on purpose: __syncthreads() also drains vmcnt, i.e. the producers' global loads of the NEXT K-tiles.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// 16 bytes per lane to sbase (uniform, SGPR pair) + voff (per lane, 32 bits) + OFF: the saddr form costs no 64-bit VGPR address
// per store.  Inline asm: the compiler's vmcnt bookkeeping does not see these stores — the consumers never wait on vector memory.
template <int OFF>
__device__ __forceinline__ void store_f32x4_saddr(const char* sbase, uint32_t voff, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}

// LDS rows are 64 bytes (one K-tile of 32 bf16); a 16x16x32 fragment read takes 16-byte chunk lane / 16 of row lane % 16.  This
// XOR of the chunk index by row makes every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) hit 16
// distinct 16-byte slots (tools/microbench/mfma_energy.hip uses the same map; checked with SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int swz16(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// TM x TN = 32-row x 32-column units per consumer wave (2 x 4 or 4 x 2).  GEN as in conv_split_kernel: false = a K-tile is a pointer
// increment (1x1 / stride 1 / no padding over a dense pixel sequence, batched GEMMs); true = taps / stride / padding.
template <int TM, int TN, bool GEN>
__global__ __launch_bounds__(512, 2) void conv_split_pp_kernel(ConvParams p, int prio) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, ROWB = 64;
  constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
  constexpr int STAGE_B = 3 * (A_PLANE + B_PLANE);
  constexpr int BIAS_OFF = 2 * STAGE_B;        // 4 slots of BN floats behind the two stages
  constexpr int RM = 2 * TM, RN = 2 * TN;      // 16 x 16 blocks per consumer wave
  constexpr bool STREAM_B = TN >= TM;          // the operand with more blocks streams through, the other is held for the K-tile
  constexpr int RH = STREAM_B ? RM : RN, RS = STREAM_B ? RN : RM;   // held / streamed blocks: 4 / 8

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: role, wave tile and epilogue bases are wave-uniform

  // ---- static tile list: XCD x owns a contiguous range of logical tiles (neighbours share A rows / all share B in its L2);
  //      the workgroups of an XCD (blockIdx.x % 8 == x under round-robin placement; speed only) take them round-robin ----
  const int nblk1 = p.tilesM * p.tilesN;
  const int nblk = nblk1 * p.nbatch;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_x = gridDim.x >> 3;
  const int xq = nblk >> 3, xr = nblk & 7;
  const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int x_cnt = xq + (xcd < xr ? 1 : 0);
  const int my_tiles = slot < x_cnt ? (x_cnt - slot + per_x - 1) / per_x : 0;
  const int G = my_tiles * p.ktiles;           // K-tiles this workgroup stages and multiplies
  if (G == 0) return;                          // uniform over the workgroup, before any barrier

  auto tile_coords = [&](int n, int& batch, int& tile_m, int& tile_n) __attribute__((always_inline)) {
    int t = x_start + slot + n * per_x;
    batch = t / nblk1;
    t -= batch * nblk1;
    tile_n = t % p.tilesN;
    tile_m = t / p.tilesN;
  };

  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    // Loads are issued two K-tiles ahead of the store that consumes them, B before A inside an iteration.  (Deeper prefetch was
    // measured and changes nothing: the loads were never the stall, profiles/r04_pp_ablations.txt.)
    constexpr int D = 2;
    const int ltid = tid & 255;
    constexpr int ACH = 8, ARP = 32, A_IT = BM / ARP;   // A: 8 x 16-B chunks per 128-B f32 row, 32 rows per pass
    constexpr int BCH = 4, BRP = 64, B_IT = BN / BRP;   // B: 4 x 16-B chunks per 64-B bf16 row, 64 rows per pass
    const int a_chunk = ltid % ACH, a_row0 = ltid / ACH;
    const int b_chunk = ltid % BCH, b_row0 = ltid / BCH;
    const long wplane_b = p.wplane * 2;
    const int bias_i = ltid < BN ? ltid : BN - 1;       // this thread's column of the tile's bias slice

    // two positions in the workgroup's (tile, K-tile) sequence: the next A tile and the next B tile (+ bias) to fetch
    uint32_t a_off[A_IT], b_off[B_IT];
    int a_hi0[GEN ? A_IT : 1], a_wi0[GEN ? A_IT : 1];
    const char* ga_base = nullptr;
    const char* gb_base = nullptr;
    const float* gbias = nullptr;
    int an = 0, akt = 0, bn = 0, bkt = 0;
    int sn = 0, skt = 0;                       // tile / K-tile of the next store
    int t_kh = 0, t_kw = 0, t_c0 = 0;          // GEN: tap and channel offset of the next A tile

    auto set_tile_a = [&](int n) __attribute__((always_inline)) {
      int batch, tile_m, tile_n;
      tile_coords(n, batch, tile_m, tile_n);
      ga_base = p.in + (long)batch * p.bsA * 4;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        long m = (long)tile_m * BM + a_row0 + i * ARP;
        if (m >= p.M) m = p.M - 1;             // rows past the end: valid addresses, values never stored
        if constexpr (GEN) {
          const long hw = (long)p.Ho * p.Wo;
          const int n_img = (int)(m / hw);
          const int rem = (int)(m - (long)n_img * hw);
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          a_hi0[i] = ho * p.SH - p.PH;
          a_wi0[i] = wo * p.SW - p.PW;
          a_off[i] = (uint32_t)(int32_t)(((long)n_img * p.sN + (long)a_hi0[i] * p.sH + (long)a_wi0[i] * p.sW + a_chunk * 4) * 4);
        } else {
          a_off[i] = (uint32_t)((m * p.sW + a_chunk * 4) * 4);   // < 4 GB (host check)
        }
      }
      t_kh = t_kw = t_c0 = 0;
    };
    auto set_tile_b = [&](int n) __attribute__((always_inline)) {
      int batch, tile_m, tile_n;
      tile_coords(n, batch, tile_m, tile_n);
      gb_base = p.w + (long)batch * p.bsW * 2;
      // no bias: every lane reads the same zero word (the load stays unconditional, see below)
      gbias = p.bias ? p.bias + tile_n * BN + bias_i : reinterpret_cast<const float*>(msocr_pp_zero16);
#pragma unroll
      for (int j = 0; j < B_IT; ++j) {
        int co = tile_n * BN + b_row0 + j * BRP;
        if (co >= p.Cout) co = p.Cout - 1;
        b_off[j] = (uint32_t)(co * 64 + b_chunk * 16);   // K-tile-major planes [k/32][Cout][32]: the tile's rows are one dense block
      }
    };

    // Loads are issued UNCONDITIONALLY — past the last K-tile of the last tile the position wraps to that tile's first K-tile (valid
    // addresses, values never multiplied) — so that the number of loads in flight at every wait is a compile-time constant: with a
    // conditional load the compiler must assume the shorter queue and its s_waitcnt vmcnt(N) drains the prefetch.
    u32x4 ra[D][A_IT], rb[D][3][B_IT];
    float rbias[D];
    auto load_a = [&](auto buf_c) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_c)::value;
      if constexpr (GEN) {
        const int32_t koff = (int32_t)(((long)t_kh * p.sH + (long)t_kw * p.sW + t_c0) * 4);   // uniform
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
          const int hi = a_hi0[i] + t_kh, wi = a_wi0[i] + t_kw;
          const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
          const char* src = ga_base + (long)(int32_t)(a_off[i] + (uint32_t)koff);
          ra[BUF][i] = *reinterpret_cast<const u32x4*>(ok ? src : reinterpret_cast<const char*>(msocr_pp_zero16));
        }
        t_c0 += BK;
        if (t_c0 == p.Cin) {
          t_c0 = 0;
          if (++t_kw == p.KW) { t_kw = 0; ++t_kh; }
        }
      } else {
        const char* const ga = ga_base + (long)akt * (BK * 4);   // uniform
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[BUF][i] = *reinterpret_cast<const u32x4*>(ga + a_off[i]);
      }
      if (++akt == p.ktiles) {
        akt = 0;
        if (an + 1 < my_tiles) set_tile_a(++an);
        else t_kh = t_kw = t_c0 = 0;
      }
    };
    auto load_b = [&](auto buf_c) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_c)::value;
      rbias[BUF] = *gbias;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const char* const gb = gb_base + pl * wplane_b + (long)bkt * p.w_kt_b;   // uniform
#pragma unroll
        for (int j = 0; j < B_IT; ++j) rb[BUF][pl][j] = *reinterpret_cast<const u32x4*>(gb + b_off[j]);
      }
      if (++bkt == p.ktiles) {
        bkt = 0;
        if (bn + 1 < my_tiles) set_tile_b(++bn);
      }
    };
    auto store = [&](auto buf_c, int stage) __attribute__((always_inline)) {
      constexpr int BUF = decltype(buf_c)::value;
      unsigned char* const sA = smem + stage * STAGE_B;   // [3][BM][ROWB]
      unsigned char* const sB = sA + 3 * A_PLANE;         // [3][BN][ROWB]
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int row = a_row0 + i * ARP;
        float x0 = __uint_as_float(ra[BUF][i][0]), x1 = __uint_as_float(ra[BUF][i][1]);
        float x2 = __uint_as_float(ra[BUF][i][2]), x3 = __uint_as_float(ra[BUF][i][3]);
        // this thread's 4 elements are bf16 positions 4 * a_chunk .. + 3 of the row: half of 16-B chunk a_chunk / 2
        unsigned char* dst = sA + row * ROWB + (((a_chunk >> 1) ^ swz16(row)) << 4) + ((a_chunk & 1) << 3);
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
          *reinterpret_cast<u32x4*>(sB + pl * B_PLANE + row * ROWB + ((b_chunk ^ swz16(row)) << 4)) = rb[BUF][pl][j];
        }
      // the bias slice of this K-tile's output tile, slot (tile number) % 4: rewritten with every K-tile of the tile (same values),
      // read by the consumers one tile later, overwritten four tiles later
      if (BN == 256 || ltid < BN) *reinterpret_cast<float*>(smem + BIAS_OFF + ((sn & 3) * BN + ltid) * 4) = rbias[BUF];
      if (++skt == p.ktiles) { skt = 0; ++sn; }
    };

    set_tile_a(0);
    set_tile_b(0);
    load_b(ic<0>{});
    load_a(ic<0>{});
    load_b(ic<1>{});
    load_a(ic<1>{});
    store(ic<0>{}, 0);
    lds_barrier();                                         // #0: stage 0 is full
    // iteration g (K-tile g is being multiplied from stage g % 2): fetch K-tile g + 2 into the registers K-tile g used, store
    // K-tile g + 1 into the other stage (after the last K-tile: a stage nobody reads any more)
    for (int g = 0;;) {
      load_b(ic<0>{});
      load_a(ic<0>{});
      store(ic<1>{}, 1);
      lds_barrier();                                       // #(g + 1)
      if (++g >= G) break;
      load_b(ic<1>{});
      load_a(ic<1>{});
      store(ic<0>{}, 0);
      lds_barrier();
      if (++g >= G) break;
    }
    return;
  }

  // =========================================== CONSUMERS ===========================================
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kg = lane >> 4;
  // fragment address inside a stage: row * 64 + ((chunk ^ swz16(row)) << 4), chunk = lane / 16; rows are r16 + multiples of 16,
  // so the swizzle depends on the lane only
  const uint32_t c16 = (uint32_t)((kg ^ swz16(r16)) << 4);
  const uint32_t offA = (uint32_t)((wm * 32 * TM + r16) * ROWB) + c16;
  const uint32_t offB = (uint32_t)(3 * A_PLANE + (wn * 32 * TN + r16) * ROWB) + c16;
  const uint32_t offH = STREAM_B ? offA : offB, offS = STREAM_B ? offB : offA;
  constexpr int PLANE_H = STREAM_B ? A_PLANE : B_PLANE, PLANE_S = STREAM_B ? B_PLANE : A_PLANE;

  bf16x8 fh[3][RH], fs[2][3];
  auto read_held = [&](auto h_c, uint32_t so) __attribute__((always_inline)) {
    constexpr int H = decltype(h_c)::value;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fh[pl][H] = *reinterpret_cast<const bf16x8*>(smem + so + offH + pl * PLANE_H + H * 16 * ROWB);
  };
  auto read_stream = [&](auto s_c, uint32_t so) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fs[S & 1][pl] = *reinterpret_cast<const bf16x8*>(smem + so + offS + pl * PLANE_S + S * 16 * ROWB);
  };

  f32x4 acc[RM][RN];
  // One 16 x 16 block, one K-tile: chain() runs the six products from C = 0, smallest terms first, into a 4-register partial sum;
  // commit() adds that partial to the accumulator — ONE rounding at the accumulator's magnitude per 32 k (FIRST: the block starts a
  // new output tile, the partial IS the accumulator).  The weight fragment is the MFMA's A operand: the lane holds D[n .. n+3][m],
  // four consecutive output channels of pixel m.  Commits trail the chains by ONE UNIT (two sets of partial sums), so the
  // additions issue in the shadow of the next unit's MFMAs instead of waiting for their own.  The empty asm pins each addition
  // where it is written: left alone, the compiler sinks all 32 additions of a K-tile to its end (128 live registers, spilled).
  f32x4 part[2][RH];
  auto chain = [&](auto s_c, auto h_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value, H = decltype(h_c)::value;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if constexpr (STREAM_B) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fs[S & 1][PB[k]], fh[PA[k]][H], t, 0, 0, 0);
      else t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[PB[k]][H], fs[S & 1][PA[k]], t, 0, 0, 0);
    }
    part[S & 1][H] = t;
  };
  auto commit = [&](auto s_c, auto h_c, auto first_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value, H = decltype(h_c)::value;
    f32x4& a = STREAM_B ? acc[H][S] : acc[S][H];
    if constexpr (decltype(first_c)::value != 0) a = part[S & 1][H];
    else a += part[S & 1][H];
    asm volatile("" : "+v"(a));
  };

  // ---- epilogue state of the tile whose accumulators are complete ----
  char* e_ob = nullptr;            // uniform: &out[batch][row0][col0] of the wave tile
  int e_lim = 0;                   // per lane: rows of the wave tile inside M, minus the lane's row inside a block
  bool e_full = true;
  int e_slot = 0;                  // LDS bias slot of that tile
  const float lo = p.relu ? 0.f : -__builtin_inff();
  const uint32_t vo = (uint32_t)((r16 * p.out_ld + 4 * kg) * 4);
  auto set_epilogue = [&](int n) __attribute__((always_inline)) {
    int batch, tile_m, tile_n;
    tile_coords(n, batch, tile_m, tile_n);
    const long row0 = (long)tile_m * BM + wm * 32 * TM;
    const int col0 = tile_n * BN + wn * 32 * TN;
    e_ob = p.out + ((long)batch * p.bsO + row0 * p.out_ld + col0) * 4;
    const long rows = p.M - row0;
    e_full = rows >= 32 * TM;
    e_lim = (int)(rows < 32 * TM ? rows : 32 * TM) - r16;
    e_slot = n & 3;
  };
  // blocks of streamed index S: (h, S) for h < RH when B streams, (S, h) when A streams; block (i, j) = rows 16 i .., columns 16 j ..
  auto store_blocks = [&](auto s_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value;
    const long ld16 = p.out_ld * 64;   // bytes per 16 rows
    const float* const sbias = reinterpret_cast<const float*>(smem + BIAS_OFF) + e_slot * BN + wn * 32 * TN + 4 * kg;
    auto body = [&](auto masked_c) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_c)::value != 0;
      static_for<0, RH>([&](auto h_c) __attribute__((always_inline)) {
        constexpr int H = decltype(h_c)::value;
        constexpr int I = STREAM_B ? H : S, J = STREAM_B ? S : H;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + J * 16);
        f32x4 v = acc[I][J] + b4;
        v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
        const char* const rowp = e_ob + I * ld16;          // uniform
        if (!MASKED || I * 16 < e_lim) store_f32x4_saddr<J * 64>(rowp, vo, v);
      });
    };
    if (e_full) body(ic<0>{});
    else body(ic<1>{});
  };

  uint32_t so = 0;                                         // byte offset of the stage being read
  // one K-tile = RS units (one streamed block x all held blocks, 6 RH MFMAs).  FIRST = the first K-tile of an output tile: unit s
  // first stores the previous tile's blocks of streamed index s (prev), then starts them anew.  PEND_FIRST = the K-tile BEFORE
  // this one was such a first K-tile (its last unit's partial sums are committed in this one's unit 0).
  auto ktile = [&](auto first_c, auto pend_first_c, bool prev) __attribute__((always_inline)) {
    constexpr int FIRST = decltype(first_c)::value;
    static_for<0, RS - 1>([&](auto u_c) __attribute__((always_inline)) {
      constexpr int U = decltype(u_c)::value;
      read_stream(ic<U + 1>{}, so);
      __builtin_amdgcn_sched_barrier(0);     // keep the reads AHEAD of this unit's MFMAs (the scheduler sinks them otherwise)
      if constexpr (FIRST != 0) {
        if (prev) store_blocks(u_c);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<0, RH>([&](auto h_c) __attribute__((always_inline)) {
        chain(u_c, h_c);
        if constexpr (U > 0) commit(ic<U - 1>{}, h_c, first_c);
        else if (FIRST == 0 || prev) commit(ic<RS - 1>{}, h_c, pend_first_c);   // the previous K-tile's last unit
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    // last unit: its streamed fragments are in registers -> the stage can be handed back.  The next K-tile's first streamed block
    // goes out at once; its held blocks are re-read one by one, each right behind the last MFMAs that used the old one.  (After
    // the workgroup's last K-tile these reads fetch stale LDS bytes nobody uses: unconditional, so the loop body has no branch.)
    lds_barrier();                                         // #(g + 1)
    so ^= (uint32_t)STAGE_B;
    read_stream(ic<0>{}, so);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FIRST != 0) {
      if (prev) store_blocks(ic<RS - 1>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<0, RH>([&](auto h_c) __attribute__((always_inline)) {
      chain(ic<RS - 1>{}, h_c);
      commit(ic<RS - 2>{}, h_c, first_c);
      __builtin_amdgcn_sched_barrier(0);
      read_held(h_c, so);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  lds_barrier();                                           // #0
  static_for<0, RH>([&](auto h_c) __attribute__((always_inline)) { read_held(h_c, so); });
  read_stream(ic<0>{}, so);
  for (int n = 0; n < my_tiles; ++n) {                     // ktiles >= 2 (host check)
    ktile(ic<1>{}, ic<0>{}, n > 0);
    set_epilogue(n);
    ktile(ic<0>{}, ic<1>{}, false);
    for (int kt = 2; kt < p.ktiles; ++kt) ktile(ic<0>{}, ic<0>{}, false);
  }
  static_for<0, RH>([&](auto h_c) __attribute__((always_inline)) { commit(ic<RS - 1>{}, h_c, ic<0>{}); });
  static_for<0, RS>([&](auto s_c) __attribute__((always_inline)) { store_blocks(s_c); });
}

template <int TM, int TN, bool GEN>
int launch_pp(ConvParams& p, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.tilesM = (int)((p.M + BM - 1) / BM);
  p.tilesN = p.Cout / BN;
  p.ktiles = (int)(p.Ktot / 32);
  constexpr int LDS = 2 * 3 * (BM + BN) * 64 + 4 * BN * 4;
  auto kern = conv_split_pp_kernel<TM, TN, GEN>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr_set = true;
  }
  const long nblk = (long)p.tilesM * p.tilesN * p.nbatch;
  if (nblk <= 0 || nblk > 0x7fffffffL) return MSOCR_E_ARG;
  static int n_cu = 0, prio = -1;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MSOCR_E_LAUNCH;
    n_cu = prop.multiProcessorCount > 8 ? prop.multiProcessorCount & ~7 : 8;
  }
  if (prio < 0) {
    const char* e = getenv("MSOCR_PP_PRIO");
    prio = e ? atoi(e) : 1;
  }
  long grid = (nblk + 7) & ~7L;
  if (grid > n_cu) grid = n_cu;
  MSOCR_LAUNCH(kern, dim3((unsigned)grid), dim3(512), LDS, s, p, prio);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}

}  // namespace

// Whether this launch has a producer / consumer instance: Cout % 128 == 0, no residual operand, at least two K-tiles (the caller has
// checked Ktot % 32).
bool msocr_internal_split_pp_takes(const ConvParams& p) { return p.Cout % 128 == 0 && !p.has_res && p.Ktot >= 64; }

int msocr_internal_split_pp_launch(ConvParams& p, hipStream_t s, bool general) {
  if (p.Cout % 256 == 0) return general ? launch_pp<2, 4, true>(p, s) : launch_pp<2, 4, false>(p, s);
  if (p.Cout % 128 == 0) return general ? launch_pp<4, 2, true>(p, s) : launch_pp<4, 2, false>(p, s);
  return MSOCR_E_ARG;
}
