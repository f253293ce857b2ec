// bilstm_mfma.hip — the recurrence of one bidirectional LSTM layer of the TRBA encoder (hidden size 256) with the step's
// [32 x 256] x [256 x 1024] product h_{t-1} W_hh^T on the bf16 matrix pipes in the split-operand form (split_rows32.h).
//
// One 512-thread workgroup owns 32 crops of one direction for all T steps; h lives in LDS as three bf16 planes, W_hh comes from L2
// pre-split and packed (msocr_attn_pack_split_host, one [3][16][1024][16] block per direction), wave w owns hidden units
// 32w..32w+31 with their four gates in its four accumulators, so the cell update is lane-local.  Against bilstm_kernel
// (trba_kernels.hip: 4 crops per workgroup on the VALU, W_hh streamed once per 4 crops) every weight element pulled from L2 feeds
// 32 rows instead of 4 and the multiply-adds leave the VALU.  xproj already holds x W_ih^T + b_ih + b_hh (a GEMM before this launch).
//
// Replaces recognizers/_trba/model/model.py:9-21 (BidirectionalLSTM.forward: nn.LSTM(bidirectional=True), the recurrent part).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "internal.h"
#include "msocr.h"
#include "split_rows32.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace {

using namespace split_rows32;

constexpr int NT = 512;
constexpr int G = 4 * H;

// Hardware-rate transcendentals (v_exp_f32 / v_rcp_f32, 1-2 ulp each), as in the beam kernel: 80 activations per lane and step made
// the libm forms the longest phase of the step (measured: 28 -> 23 us per step and workgroup with these and the prefetch below).
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + fexp(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fexp(2.0f * x) + 1.0f); }

__global__ __launch_bounds__(NT, 1) void bilstm_split_kernel(const float* __restrict__ xproj, const uint16_t* __restrict__ whh_p, int B, int T,
                                                            float* __restrict__ hcat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sP[];  // [3][R][PSB] bf16 planes of h
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r32 = lane & 31, half = lane >> 5;
  const int ju = 32 * wv + r32;  // hidden unit owned in the MFMA phase
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * R;
  const uint16_t* const wp = whh_p + (long)d * 3 * H * G;

  for (int i = tid; i < 3 * PPL / 4; i += NT) reinterpret_cast<float*>(sP)[i] = 0.f;  // h = 0
  f32x16 c;
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  // this lane's 16 rows: crops b0 + acc_row(e, half); rows past B repeat the last crop (computed, never stored)
  uint32_t xoff[16];  // element offsets into xproj: below 2^32 (host check)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int b = min(b0 + acc_row(e, half), B - 1);
    xoff[e] = (uint32_t)(((long)b * T * 2 + d) * G + ju);
  }
  __syncthreads();

  f32x16 xn[4];  // the input projections of the NEXT step, requested before the cell update of the current one
  auto load_x = [&](int s) {
    const int t = d == 0 ? s : T - 1 - s;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* xp = xproj + (xoff[e] + (uint32_t)t * (2 * G));
#pragma unroll
      for (int g = 0; g < 4; ++g) xn[g][e] = xp[g * H];
    }
  };
  load_x(0);
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = xn[g];
    mfma_gates_split(sP, wp, ju, r32, half, acc);
    if (s + 1 < T) load_x(s + 1);
    __syncthreads();  // every wave has read h_{t-1}
    float hv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float ig = sigm(acc[0][e]), fg = sigm(acc[1][e]), gg = ftanh(acc[2][e]), og = sigm(acc[3][e]);
      c[e] = fg * c[e] + ig * gg;
      hv[e] = og * ftanh(c[e]);
      const int row = acc_row(e, half);
      if (b0 + row < B) hcat[((long)(b0 + row) * T + t) * (2 * H) + d * H + ju] = hv[e];
    }
#pragma unroll
    for (int e = 0; e < 16; e += 2) {  // acc_row(e + 1) == acc_row(e) + 1
      unsigned char* dst = sP + acc_row(e, half) * PSB + ju * 2;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const uint32_t pk = split_pair(hv[e], hv[e + 1]);
        *reinterpret_cast<uint16_t*>(dst + pl * PPL) = (uint16_t)pk;
        *reinterpret_cast<uint16_t*>(dst + pl * PPL + PSB) = (uint16_t)(pk >> 16);
      }
    }
    __syncthreads();
  }
}

}  // namespace

// msocr_bilstm_recurrent with W_hh given in the packed split form: whh_planes = two blocks (forward, reverse) of
// msocr_attn_pack_split_elems(4 H) uint16 each, packed by msocr_attn_pack_split_host(w_hh_t[dir], 4 H, gate_interleaved = 1, .).
extern "C" int msocr_bilstm_recurrent_split(const float* xproj, const uint16_t* whh_planes, int B, int T, int H_, float* hcat_out, void* stream) {
  if (!xproj || !whh_planes || !hcat_out || B <= 0 || T <= 0 || H_ != H || ((uintptr_t)whh_planes & 15)) return MSOCR_E_ARG;
  if ((long)B * T * 2 * G >= (1L << 32)) return MSOCR_E_ARG;  // 32-bit element offsets into xproj (the VALU kernel has no such limit)
  const size_t ldsz = (size_t)3 * PPL;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)bilstm_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz) != hipSuccess) return MSOCR_E_LAUNCH;
    attr = true;
  }
  MSOCR_LAUNCH(bilstm_split_kernel, dim3((B + R - 1) / R, 2), dim3(NT), ldsz, (hipStream_t)stream, xproj, whh_planes, B, T, hcat_out);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}
