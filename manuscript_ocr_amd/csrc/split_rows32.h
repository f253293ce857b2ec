// split_rows32.h — device helpers shared by the recurrent kernels that keep a 32-row hidden state (one MFMA row block, 256 units) in
// LDS and multiply it with a weight matrix streamed from L2 in the split-operand form: attn_beam_mfma.hip (beam decode) and
// bilstm_mfma.hip (encoder BiLSTM).  Not part of the C ABI.
#ifndef MSOCR_SPLIT_ROWS32_H
#define MSOCR_SPLIT_ROWS32_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace split_rows32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256;  // hidden units = reduction length
constexpr int R = 32;   // state rows = one MFMA row block

// row of accumulator register e in the 32x32 MFMA output layout (lane half = lane >> 5)
__device__ __forceinline__ int acc_row(int e, int half) { return (e & 3) + 8 * (e >> 2) + 4 * half; }

// ---- split-operand form of the three matrix products : h is kept in LDS as three bf16 planes with h == p0 + p1 + p2
// exactly (the residual chain of conv_split.hip), the weights come pre-split and packed [plane][k / 16][column][16] bf16
// (msocr_attn_pack_split_host), and every f32 product a * b is the six bf16 products a2b0 + a0b2 + a1b1 + a1b0 + a0b1 + a0b0 on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation (dropped terms <= 2^-25 |a b|): 6 MFMAs of 8 passes per 16 k instead of 8 MFMAs of
// 16 passes on the exact-f32 pipe, i.e. 2.7x less matrix-pipe time for the same f32 result up to summation order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PSB = H * 2 + 16;       // bytes per plane row: 528 = 132 dwords, rows shift 4 banks -> ds_read_b128 of 32 rows is conflict-free
constexpr int PPL = R * PSB;          // bytes per plane

__device__ __forceinline__ uint32_t split_pair(float& x, float& y) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x, y};
  const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, RNE
  x -= __uint_as_float(pk << 16);
  y -= __uint_as_float(pk & 0xffff0000u);
  return pk;
}
__device__ __forceinline__ void mma6(const bf16x8 (&fa)[3], const u32x4 (&wb)[3], f32x16& acc) {
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, wb[0]), b1 = __builtin_bit_cast(bf16x8, wb[1]), b2 = __builtin_bit_cast(bf16x8, wb[2]);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], b0, acc, 0, 0, 0);
}
__device__ __forceinline__ void read_a3(const unsigned char* sP, int kb, int r32, int half, bf16x8 (&fa)[3]) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const bf16x8*>(sP + pl * PPL + r32 * PSB + kb * 32 + half * 16);
}

// D[32 rows][32 columns of this wave] += h * W, W packed [3][16][ncols][16] bf16
__device__ __forceinline__ void mfma_cols32_split(const unsigned char* __restrict__ sP, const uint16_t* __restrict__ Wp, int ncols, int col,
                                                  bool col_ok, int r32, int half, f32x16& acc) {
  constexpr int PF = 4;  // k-blocks (of 16) of weight loads kept in flight
  const int kbstep = ncols * 32, plstep = 16 * kbstep;  // bytes
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, 3 * plstep, 0x00020000);
  const int voff = col_ok ? col * 32 + half * 16 : 0x7ffffff0;
  u32x4 wb[PF][3];
#pragma unroll
  for (int pq = 0; pq < PF; ++pq)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wb[pq][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, pl * plstep + pq * kbstep, 0);
#pragma unroll 1
  for (int kb0 = 0; kb0 < H / 16; kb0 += PF) {
#pragma unroll
    for (int pq = 0; pq < PF; ++pq) {
      const int kb = kb0 + pq;
      bf16x8 fa[3];
      read_a3(sP, kb, r32, half, fa);
      u32x4 cur[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) cur[pl] = wb[pq][pl];
      if (kb + PF < H / 16) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wb[pq][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, pl * plstep + (kb + PF) * kbstep, 0);
      }
      mma6(fa, cur, acc);
    }
  }
}

// gates of hidden units 32w..32w+31: acc[g] += h * W_hh, packed [3][16][4 H (column g * H + j)][16] bf16
__device__ __forceinline__ void mfma_gates_split(const unsigned char* __restrict__ sP, const uint16_t* __restrict__ Wp, int j, int r32, int half,
                                                 f32x16 (&acc)[4]) {
  constexpr int kbstep = 4 * H * 32, plstep = 16 * kbstep, gstep = H * 32;  // bytes
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, 3 * plstep, 0x00020000);
  const int voff = j * 32 + half * 16;
  u32x4 wb[4][3];  // one k-block of the four gates; a gate's next block is requested as soon as its six MFMAs are issued
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wb[g][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, pl * plstep + g * gstep, 0);
#pragma unroll 2
  for (int kb = 0; kb < H / 16; ++kb) {
    bf16x8 fa[3];
    read_a3(sP, kb, r32, half, fa);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      mma6(fa, wb[g], acc[g]);
      if (kb + 1 < H / 16) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          wb[g][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, pl * plstep + (kb + 1) * kbstep + g * gstep, 0);
      }
    }
  }
}

}  // namespace split_rows32
#endif
