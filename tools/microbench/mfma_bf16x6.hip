// Micro-benchmark for the "considered, not built" split-operand GEMM of DESIGN.md section 7: an f32 GEMM whose operands are split
// exactly into three bf16 terms (a = a0 + a1 + a2) and whose products are 6 of the 9 cross terms on v_mfma_f32_32x32x16_bf16
// (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0: the dropped terms are <= 2^-26 of the product), f32 accumulate.  Same tile as the conv
// kernel: 128 x 128 per workgroup, 64 x 64 per wave, K-tiles of 32.  Reported as f32-EQUIVALENT TFLOP/s (2*M*N*K / time); the bf16
// pipes execute 6x that.
//   MODE 0: LDS fragment reads (3 + 3 planes) + MFMAs only
//   MODE 1: + per K-tile [barrier; restage the 6 planes from registers (12 x ds_write_b128 per thread); barrier]
//   MODE 2: MODE 1 + the planes re-fetched from global memory every K-tile (12 x 16-B loads per thread, pre-split operands)
//   MODE 3: MODE 1 + the A operand fetched as f32 (8 x 16-B loads per thread) and split in registers (the activation side of a real
//           kernel: round-to-nearest bf16 three times), B planes pre-split
// Compare with tools/microbench/mfma_pipeline.hip (exact-f32 MFMA: 137-152 TFLOP/s in the same structure).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rn(float f) { __bf16 b = (__bf16)f; return *reinterpret_cast<unsigned short*>(&b); }
__device__ __forceinline__ float bf16_f(unsigned h) { return __uint_as_float(h << 16); }
// a -> (a0, a1, a2) with a0 + a1 + a2 == a exactly
__device__ __forceinline__ void split3(float a, unsigned& h0, unsigned& h1, unsigned& h2) {
  h0 = bf16_rn(a);
  const float r1 = a - bf16_f(h0);
  h1 = bf16_rn(r1);
  const float r2 = r1 - bf16_f(h1);
  h2 = bf16_rn(r2);
}

template <int MODE, int WPC>
__global__ __launch_bounds__(256, WPC) void k(const float* in, float* out, int iters) {
  // 6 planes of [128 rows][32 bf16 = 64 B]: A0 A1 A2 B0 B1 B2
  __shared__ __attribute__((aligned(16))) unsigned char s[6][128 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 6 * 128 * 16; i += 256) reinterpret_cast<unsigned*>(s)[i] = __float_as_uint(in[i]) & 0xffffu ? __float_as_uint(in[i]) : 0x3f803f80u;
  __syncthreads();
  u32x4 stage[12];
  for (int i = 0; i < 12; ++i) stage[i] = *(const u32x4*)&in[(tid * 12 + i) * 4];
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* gsrc = in + ((size_t)(blockIdx.x % 40) << 20) + tid * 48;  // 40 windows of 4 MB + 128 K-tiles of 48 KB stay inside the 256 MB buffer
  const int row0 = tid >> 2, ch = tid & 3;  // staging: 4 chunks of 16 B per 64-byte row, 64 rows per pass
  for (int it = 0; it < iters; ++it) {
    u32x4 nxt[12];
    f32x4 na[4];
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) nxt[i] = *(const u32x4*)&gsrc[((it & 127) * 12288) + i * 4];
    }
    if (MODE == 3) {  // A tile 128 x 32 f32 = 16 f32 per thread; B planes pre-split = 6 x 16 B per thread
#pragma unroll
      for (int i = 0; i < 4; ++i) na[i] = *(const f32x4*)&gsrc[((it & 127) * 12288) + i * 4];
#pragma unroll
      for (int i = 0; i < 6; ++i) nxt[6 + i] = *(const u32x4*)&gsrc[((it & 127) * 12288) + 16 + i * 4];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // two k16 steps per K-tile of 32
      bf16x8 fa[3][2], fb[3][2];
      const int c = 2 * q + half;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ra = wm * 64 + i * 32 + r32, rb = wn * 64 + i * 32 + r32;
          fa[p][i] = *(const bf16x8*)&s[p][ra * 64 + ((c ^ ((ra >> 2) & 3)) << 4)];
          fb[p][i] = *(const bf16x8*)&s[3 + p][rb * 64 + ((c ^ ((rb >> 2) & 3)) << 4)];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 12; ++i) stage[i] = nxt[i];
    }
    if (MODE == 3) {  // exact 3-way split of the 16 f32: plane p gets two 16-B chunks (8 bf16 each)
      unsigned w[3][8];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        unsigned h0, h1, h2;
        split3(na[e >> 2][e & 3], h0, h1, h2);
        if (e & 1) { w[0][e >> 1] |= h0 << 16; w[1][e >> 1] |= h1 << 16; w[2][e >> 1] |= h2 << 16; }
        else { w[0][e >> 1] = h0; w[1][e >> 1] = h1; w[2][e >> 1] = h2; }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        stage[2 * p] = (u32x4){w[p][0], w[p][1], w[p][2], w[p][3]};
        stage[2 * p + 1] = (u32x4){w[p][4], w[p][5], w[p][6], w[p][7]};
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) stage[6 + i] = nxt[6 + i];
    }
    if (MODE >= 1) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = row0 + 64 * i;
          *(u32x4*)&s[p][row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)] = stage[p * 2 + i];
        }
      __syncthreads();
    }
  }
  float t = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
  out[blockIdx.x * 256 + tid] = t;
}
template <int MODE, int WPC>
static void run(const float* in, float* out, int nb, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPC>), dim3(nb), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double eq = 2.0 * 128 * 128 * 32 * (double)iters * nb / ms / 1e9;
    printf("%-58s rep %d: %.2f ms  %.1f TFLOP/s f32-equivalent (%.0f bf16-executed)\n", name, rep, ms, eq, 6 * eq);
  }
}
int main() {
  float *in, *out; const int nb = 768 * 4;
  (void)hipMalloc(&in, (size_t)64 << 22); (void)hipMalloc(&out, nb * 256 * 4);
  const size_t nfl = (size_t)64 << 20;
  float* h = (float*)malloc(nfl * 4); for (size_t i = 0; i < nfl; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f);
  (void)hipMemcpy(in, h, nfl * 4, hipMemcpyHostToDevice);
  const int iters = 6000;
  run<0, 2>(in, out, nb, iters, "LDS reads + 6-product MFMAs, 2 WG/CU");
  run<0, 3>(in, out, nb, iters, "LDS reads + 6-product MFMAs, 3 WG/CU");
  run<1, 2>(in, out, nb, iters, "+ barrier/restage every 32 k, 2 WG/CU");
  run<1, 3>(in, out, nb, iters, "+ barrier/restage every 32 k, 3 WG/CU");
  run<2, 2>(in, out, nb, iters, "+ 6 planes re-fetched from global each K-tile, 2 WG/CU");
  run<2, 3>(in, out, nb, iters, "+ 6 planes re-fetched from global each K-tile, 3 WG/CU");
  run<3, 2>(in, out, nb, iters, "+ A fetched as f32 and split in registers, 2 WG/CU");
  return 0;
}
