// conv_split_pp.hip — the split-operand ("bf16x3") f32 GEMM / convolution of conv_split.hip as a PRODUCER / CONSUMER workgroup.
//
// Same arithmetic as conv_split_kernel (an f32 value is the exact sum of three bf16 values; six of the nine cross products on
// v_mfma_f32_32x32x16_bf16, smallest terms first, f32 accumulate, the same order of additions per output element), different
// machine mapping (round 4).  conv_split_kernel's four waves each load, split, stage, multiply and meet at two barriers per
// K-tile; its matrix pipes idle while a workgroup splits and stages (PMC: 0.48 busy).  Here a 512-thread workgroup puts TWO
// waves on every SIMD with fixed roles (MI355X_MICROARCH.md, "Two waves per SIMD"):
//   waves 0-3  CONSUMERS: nothing but ds_read_b128 + MFMA.  2 x 2 waves, wave tile (32 TM) x (32 TN) = 64 x 128 or 128 x 64, the
//              128 accumulator registers of a wave live across the whole K loop.  Fragments are double-buffered in registers: the
//              reads of the next unit (one 32-wide block of the streamed operand x all blocks of the held operand, 12 MFMAs) are
//              in flight under the MFMAs of the current one, across the barrier too.
//   waves 4-7  PRODUCERS: global -> registers (two K-tiles in flight) -> split in registers (v_cvt_pk_bf16_f32 + exact
//              residuals) -> LDS.  Their VALU and memory instructions issue in the gaps of the partner wave's MFMAs.
// Two LDS stages of six [rows][64 B] planes (3 x (BM + BN) x 64 B = 72 KB each), ONE barrier per K-tile of 96 MFMAs per consumer
// wave (conv_split_kernel: two per 48).  The workgroup is persistent: a static, XCD-contiguous list of output tiles per
// workgroup; the producers run ahead across tile boundaries, so the first K-tiles of the next tile load while the consumers
// store the finished one (straight from the accumulators: 2 x 128-byte segments per store instruction, no LDS pass).
//
// Tile 128 x 256 (Cout % 256 == 0) halves the activation re-splits and re-reads per output column block against the 128 x 128 of
// conv_split_kernel; 256 x 128 serves Cout % 128 == 0.  Everything else stays on conv_split_kernel.
//
// Reference layers: recognizers/_trba/model/seresnet31.py:37-67 ; detectors/_east/east.py:13-30 ; torchvision Bottleneck.
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

// One dword per lane to sbase (uniform, SGPR pair) + voff (per lane, 32 bits) + OFF: the saddr form costs no 64-bit VGPR address per
// store (written as C the compiler builds 64-bit per-row addresses in VGPRs and spills them beside the 200 live registers).
template <int OFF>
__device__ __forceinline__ void store_f32_saddr(const char* sbase, uint32_t voff, float v) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(sbase), "n"(OFF) : "memory");
}

// TM x TN = 32x32 blocks per consumer wave (2 x 4 or 4 x 2).  GEN as in conv_split_kernel: false = a K-tile is a pointer
// increment (1x1 / stride 1 / no padding over a dense pixel sequence, batched GEMMs); true = taps / stride / padding.
// DBG (timing ablations, wrong results by construction, MSOCR_PP_DBG): 1 = no global loads after the prologue, 2 = producers only
// meet the barriers, 4 = consumers issue no MFMA, 8 = consumers read no fragments after the first, 16 / 32 = no B / no A loads after the prologue, 64 = no output stores, 128 = no split arithmetic (raw bits stored).
template <int TM, int TN, bool GEN, int DBG = 0, int DA_ = 2>
__global__ __launch_bounds__(512, 2) void conv_split_pp_kernel(ConvParams p, int prio) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, ROWB = 64;
  constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
  constexpr int STAGE_B = 3 * (A_PLANE + B_PLANE);
  constexpr bool STREAM_B = TN >= TM;          // the operand with more blocks streams through, the other is held per k16 step
  constexpr int TH = STREAM_B ? TM : TN, TS = STREAM_B ? TN : TM;
  constexpr int UNITS = 2 * TS;                // per K-tile: 2 k16 steps x TS streamed blocks, 6 * TH MFMAs each

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

  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    // Prefetch distances: the activation operand streams from HBM — its loads are issued DA = 4 K-tiles ahead of the store that
    // consumes them; the weight planes come from L2 — DB = 2 ahead.  (With both at 2 the producers arrived late at every barrier:
    // matrix pipes 0.62 busy against 0.80 with the loads removed, profiles/r04_pp_ablations.txt.)  vmcnt retires in issue order, so
    // inside an iteration B is issued before A: waiting for a B tile then never waits for the younger A tiles.
    constexpr int DA = DA_, DB = 2;
    constexpr int UNR = DA % 2 ? 2 * DA : DA;   // lcm(DA, DB): the register buffers cycle with the iteration number
    const int ltid = tid & 255;
    constexpr int ACH = 8, ARP = 32, A_IT = BM / ARP;   // A: 8 x 16-B chunks per 128-B f32 row, 32 rows per pass
    constexpr int BCH = 4, BRP = 64, B_IT = BN / BRP;   // B: 4 x 16-B chunks per 64-B bf16 row, 64 rows per pass
    const int a_chunk = ltid % ACH, a_row0 = ltid / ACH;
    const int b_chunk = ltid % BCH, b_row0 = ltid / BCH;
    const long wplane_b = p.wplane * 2;

    // two independent positions in the workgroup's (tile, K-tile) sequence: the next A tile and the next B tile to fetch
    uint32_t a_off[A_IT], b_off[B_IT];
    int a_hi0[GEN ? A_IT : 1], a_wi0[GEN ? A_IT : 1];
    const char* ga_base = nullptr;
    const char* gb_base = nullptr;
    int an = 0, akt = 0, bn = 0, bkt = 0;
    int t_kh = 0, t_kw = 0, t_c0 = 0;          // GEN: tap and channel offset of the next A tile

    auto tile_coords = [&](int n, int& batch, int& tile_m, int& tile_n) __attribute__((always_inline)) {
      int t = x_start + slot + n * per_x;
      batch = t / nblk1;
      t -= batch * nblk1;
      tile_n = t % p.tilesN;
      tile_m = t / p.tilesN;
    };
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
    u32x4 ra[DA][A_IT], rb[DB][3][B_IT];
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
    auto store = [&](auto abuf_c, auto bbuf_c, int stage) __attribute__((always_inline)) {
      constexpr int AB = decltype(abuf_c)::value, BB = decltype(bbuf_c)::value;
      unsigned char* const sA = smem + stage * STAGE_B;   // [3][BM][ROWB]
      unsigned char* const sB = sA + 3 * A_PLANE;         // [3][BN][ROWB]
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int row = a_row0 + i * ARP;
        float x0 = __uint_as_float(ra[AB][i][0]), x1 = __uint_as_float(ra[AB][i][1]);
        float x2 = __uint_as_float(ra[AB][i][2]), x3 = __uint_as_float(ra[AB][i][3]);
        // this thread's 4 elements are bf16 positions 4 * a_chunk .. + 3 of the row: half of 16-B chunk a_chunk / 2
        unsigned char* dst = sA + row * ROWB + (((a_chunk >> 1) ^ swz<ROWB>(row)) << 4) + ((a_chunk & 1) << 3);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          u32x2 v;
          if constexpr (DBG & 128) {
            v[0] = ra[AB][i][pl];
            v[1] = ra[AB][i][pl + 1];
          } else {
            v[0] = split_step(x0, x1);
            v[1] = split_step(x2, x3);
          }
          *reinterpret_cast<u32x2*>(dst + pl * A_PLANE) = v;
        }
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
          const int row = b_row0 + j * BRP;
          *reinterpret_cast<u32x4*>(sB + pl * B_PLANE + row * ROWB + ((b_chunk ^ swz<ROWB>(row)) << 4)) = rb[BB][pl][j];
        }
    };

    set_tile_a(0);
    set_tile_b(0);
    load_b(ic<0>{});
    static_for<0, DA - 1>([&](auto d) __attribute__((always_inline)) { load_a(d); });      // A tiles 0 .. DA-2
    load_b(ic<1>{});
    load_a(ic<DA - 1>{});
    store(ic<0>{}, ic<0>{}, 0);
    lds_barrier();                                         // #0: stage 0 is full
    // iteration g (K-tile g is being multiplied from stage g % 2): fetch B tile g + DB and A tile g + DA into the registers K-tile g
    // used, store K-tile g + 1 into the other stage (after the last K-tile: a stage nobody reads any more)
    for (int g = 0;;) {
      bool done = false;
      static_for<0, UNR>([&](auto r_c) __attribute__((always_inline)) {
        constexpr int R = decltype(r_c)::value;
        if (done) return;
        if (!(DBG & 3)) {
          if (!(DBG & 16)) load_b(ic<R % DB>{});
          if (!(DBG & 32)) load_a(ic<R % DA>{});
        }
        if (!(DBG & 2)) store(ic<(R + 1) % DA>{}, ic<(R + 1) % DB>{}, (R + 1) & 1);
        lds_barrier();                                     // #(g + 1)
        if (++g >= G) done = true;
      });
      if (done) break;
    }
    return;
  }

  // =========================================== CONSUMERS ===========================================
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 3) __builtin_amdgcn_s_setprio(3);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  // fragment addresses inside a stage: row * 64 + ((chunk ^ swz(row)) << 4), chunk = 2 q + half; rows are r32 + multiples of 32,
  // so the swizzle depends on the lane only
  const int sw = swz<ROWB>(r32);
  uint32_t offA[2], offB[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c16 = ((2 * q + half) ^ sw) << 4;
    offA[q] = (uint32_t)((wm * 32 * TM + r32) * ROWB + c16);
    offB[q] = (uint32_t)(3 * A_PLANE + (wn * 32 * TN + r32) * ROWB + c16);
  }

  bf16x8 fh[2][3][TH], fs[2][3];
  // reads of unit u (k16 step q = u / TS, streamed block s = u % TS) from the stage at byte offset so
  auto issue_reads = [&](auto u_c, uint32_t so) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    constexpr int q = U / TS, s = U % TS;
    const unsigned char* const base = smem + so;
    if constexpr (s == 0) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int h = 0; h < TH; ++h)
          fh[q][pl][h] = *reinterpret_cast<const bf16x8*>(base + (STREAM_B ? offA[q] + pl * A_PLANE : offB[q] + pl * B_PLANE) + h * 32 * ROWB);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      fs[U & 1][pl] = *reinterpret_cast<const bf16x8*>(base + (STREAM_B ? offB[q] + pl * B_PLANE : offA[q] + pl * A_PLANE) + s * 32 * ROWB);
  };

  f32x16 acc[TM][TN];
  // FIRST: the unit's blocks start a new output tile — their first MFMA takes C = 0 instead of the accumulator
  auto mfma_unit = [&](auto u_c, auto first_c) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;
    constexpr int q = U / TS, s = U % TS;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // smallest terms first (conv_split_kernel's order per accumulator); the held blocks alternate
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int h = 0; h < TH; ++h) {
        if constexpr (STREAM_B)
          acc[h][s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[q][PA[t]][h], fs[U & 1][PB[t]], FIRST && t == 0 ? zero : acc[h][s], 0, 0, 0);
        else
          acc[s][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fs[U & 1][PA[t]], fh[q][PB[t]][h], FIRST && t == 0 ? zero : acc[s][h], 0, 0, 0);
      }
    }
  };

  // ---- epilogue state of the tile whose accumulators are complete: stored block by block from the registers (register e of
  //      a block is row (e & 3) + 8 (e >> 2) + 4 half, column r32: one store instruction = two 128-byte row segments), each
  //      block just before the NEXT tile's first MFMA overwrites it — the stores drain under that tile's K loop ----
  char* e_ob = nullptr;            // uniform: &out[batch][row0][col0] of the wave tile
  int e_lim = 0;                   // per lane: rows of the wave tile inside M, minus the lane's 4 half (>= 32 TM: no masking)
  bool e_full = true;
  float e_bias[TN];
  const float lo = p.relu ? 0.f : -__builtin_inff();
  const uint32_t vo = (uint32_t)((4 * half * p.out_ld + r32) * 4);
  auto set_epilogue = [&](int n) __attribute__((always_inline)) {
    int t = x_start + slot + n * per_x;
    const int batch = t / nblk1;
    t -= batch * nblk1;
    const int tile_n = t % p.tilesN, tile_m = t / p.tilesN;
    const long row0 = (long)tile_m * BM + wm * 32 * TM;
    const int col0 = tile_n * BN + wn * 32 * TN;
    e_ob = p.out + ((long)batch * p.bsO + row0 * p.out_ld + col0) * 4;
    const long rows = p.M - row0;
    e_full = rows >= 32 * TM;
    e_lim = (int)(rows < 32 * TM ? rows : 32 * TM) - 4 * half;
#pragma unroll
    for (int j = 0; j < TN; ++j) e_bias[j] = p.bias ? p.bias[col0 + j * 32 + r32] : 0.f;
  };
  // blocks of streamed index s: (h, s) for h < TH when B streams, (s, h) when A streams
  auto store_blocks = [&](auto s_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value;
    const long ld_b = p.out_ld * 4;
    auto body = [&](auto masked_c) __attribute__((always_inline)) {
      constexpr bool MASKED = decltype(masked_c)::value != 0;
      static_for<0, TH>([&](auto h_c) __attribute__((always_inline)) {
        constexpr int H = decltype(h_c)::value;
        constexpr int I = STREAM_B ? H : S, J = STREAM_B ? S : H;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rr = I * 32 + (e & 3) + 8 * (e >> 2);  // + 4 half per lane
          const float v = fmaxf(acc[I][J][e] + e_bias[J], lo);
          const char* const rowp = e_ob + rr * ld_b;        // uniform
          if (!MASKED || rr < e_lim) store_f32_saddr<J * 128>(rowp, vo, v);
        }
      });
    };
    if (e_full) body(ic<0>{});
    else body(ic<1>{});
  };

  uint32_t so = 0;                                         // byte offset of the stage being read
  int g = 0;
  // one K-tile.  FIRST = the first of an output tile: its first TS units finish the previous tile (prev) block by block
  auto ktile = [&](auto first_c, bool prev) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value != 0;
    static_for<0, UNITS - 1>([&](auto u_c) __attribute__((always_inline)) {
      constexpr int U = decltype(u_c)::value;
      if (!(DBG & 8)) issue_reads(ic<U + 1>{}, so);
      __builtin_amdgcn_sched_barrier(0);     // keep the reads AHEAD of this unit's MFMAs (the scheduler sinks them otherwise)
      if constexpr (FIRST && U < TS) {
        if constexpr (U == 0) {
          // the bias values were requested a whole K loop ago: make the compiler wait for them HERE, before the first store — its
          // vmcnt bookkeeping does not see the inline-asm stores, and a wait placed between two blocks' stores would drain them
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(e_bias[j]));
        }
        if (prev && (!(DBG & 64) || p.relu == 12345)) store_blocks(ic<U>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(DBG & 4)) mfma_unit(u_c, ic<(FIRST && U < TS) ? 1 : 0>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // last unit: its fragments are in registers -> the stage can be handed back; the first reads of the next stage cover
    // their latency under this unit's MFMAs
    lds_barrier();                                         // #(g + 1)
    so ^= (uint32_t)STAGE_B;
    if (!(DBG & 8) && g + 1 < G) issue_reads(ic<0>{}, so);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) mfma_unit(ic<UNITS - 1>{}, ic<0>{});
    __builtin_amdgcn_sched_barrier(0);
    ++g;
  };

  lds_barrier();                                           // #0
  issue_reads(ic<0>{}, so);
  for (int n = 0; n < my_tiles; ++n) {
    ktile(ic<1>{}, n > 0);
    set_epilogue(n);
    for (int kt = 1; kt < p.ktiles; ++kt) ktile(ic<0>{}, false);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(e_bias[j]));
  if (!(DBG & 64) || p.relu == 12345) static_for<0, TS>([&](auto s_c) __attribute__((always_inline)) { store_blocks(s_c); });
}

template <int TM, int TN, bool GEN, int DBG = 0, int DA_ = 2>
int launch_pp(ConvParams& p, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  p.tilesM = (int)((p.M + BM - 1) / BM);
  p.tilesN = p.Cout / BN;
  p.ktiles = (int)(p.Ktot / 32);
  constexpr int LDS = 2 * 3 * (BM + BN) * 64;
  auto kern = conv_split_pp_kernel<TM, TN, GEN, DBG, DA_>;
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

// Whether this shape has a producer / consumer instance (Cout % 128 == 0; the caller has checked Ktot % 32 == 0).
bool msocr_internal_split_pp_takes(const ConvParams& p) { return p.Cout % 128 == 0 && !p.has_res; }

int msocr_internal_split_pp_launch(ConvParams& p, hipStream_t s, bool general) {
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("MSOCR_PP_DBG");
    dbg = e ? atoi(e) : 0;
  }
  if (dbg && p.Cout % 256 == 0 && !general) {
    switch (dbg) {
      case 1: return launch_pp<2, 4, false, 1>(p, s);
      case 2: return launch_pp<2, 4, false, 2>(p, s);
      case 4: return launch_pp<2, 4, false, 4>(p, s);
      case 8: return launch_pp<2, 4, false, 8>(p, s);
      case 10: return launch_pp<2, 4, false, 10>(p, s);
      case 12: return launch_pp<2, 4, false, 12>(p, s);
      case 16: return launch_pp<2, 4, false, 16>(p, s);
      case 32: return launch_pp<2, 4, false, 32>(p, s);
      case 64: return launch_pp<2, 4, false, 64>(p, s);
      case 128: return launch_pp<2, 4, false, 128>(p, s);
      case 130: return launch_pp<2, 4, false, 130>(p, s);
      case 80: return launch_pp<2, 4, false, 80>(p, s);
      case 96: return launch_pp<2, 4, false, 96>(p, s);
      case 316: return launch_pp<2, 4, false, 16, 3>(p, s);
      case 416: return launch_pp<2, 4, false, 16, 4>(p, s);
      case 616: return launch_pp<2, 4, false, 16, 6>(p, s);
      case 300: return launch_pp<2, 4, false, 0, 3>(p, s);
      case 400: return launch_pp<2, 4, false, 0, 4>(p, s);
    }
  }
  if (p.Cout % 256 == 0) return general ? launch_pp<2, 4, true>(p, s) : launch_pp<2, 4, false>(p, s);
  if (p.Cout % 128 == 0) return general ? launch_pp<4, 2, true>(p, s) : launch_pp<4, 2, false>(p, s);
  return MSOCR_E_ARG;
}
