// Micro-benchmark (round 4): what the six-product bf16 MFMA loop of the split-operand GEMM sustains on RANDOM data, by MFMA shape,
// wave tile and fragment source.  MI355X holds its clock down under dense MFMA load (MI355X_MICROARCH.md, "DVFS give-back"), so the
// quantity that decides throughput is energy per MFMA, not issue slots: every variant runs one wave per SIMD, one workgroup per CU,
// the same FLOP per wave tile, operands = the three bf16 planes of random f32 values, fragments re-read from LDS every K-tile (or
// held in registers, READS = 0).  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_energy mfma_energy.hip ; run: ./mfma_energy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS image: planes [3][ROWS][64 B] for A and for B (one K-tile of 32), 16-B chunks swizzled per row so that ds_read_b128 is
// conflict-free for the shape's fragment pattern
__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }            // 32x32x16: lane = row % 32, chunk = 2 q + lane / 32
__device__ __forceinline__ int swz16(int row) { return (4 - ((row >> 2) & 3)) & 3; }  // 16x16x32: lane = row % 16, chunk = lane / 16

template <int SHAPE, int TM, int TN, int READS>   // TM x TN in units of 32 rows / columns per wave; 2 x 2 waves per workgroup
__global__ __launch_bounds__(256, 1) void k(const uint16_t* __restrict__ planes, float* out, int iters) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                   // [3][BM][64]
  unsigned char* sB = smem + 3 * BM * 64;     // [3][BN][64]
  // fill: plane data = random bf16 (the host made them as the exact three-term split of random f32 values)
  for (int i = threadIdx.x; i < 3 * (BM + BN) * 4; i += 256) {
    const int row = (i >> 2) % (BM + BN), pl = (i >> 2) / (BM + BN), c = i & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(planes + ((size_t)(pl * 512 + (row & 511)) * 32 + c * 8));
    unsigned char* base = row < BM ? sA + (pl * BM + row) * 64 : sB + (pl * BN + row - BM) * 64;
    const int rr = row < BM ? row : row - BM;
    *reinterpret_cast<uint4*>(base + ((c ^ (SHAPE == 32 ? swz32(rr) : swz16(rr))) << 4)) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  float accsum = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r32 = lane & 31, half = lane >> 5;
    bf16x8 fa[3][TM], fb[3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c16 = ((2 * q + half) ^ swz32(r32)) << 4;
        if (READS || it == 0) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(sA + (pl * BM + wm * 32 * TM + i * 32 + r32) * 64 + c16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (READS || it == 0) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(sB + (pl * BN + wn * 32 * TN + j * 32 + r32) * 64 + c16);
          }
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[PB[t]], acc[i][j], 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) accsum += acc[i][j][e];
  } else {
    constexpr int RM = 2 * TM, RN = 2 * TN;   // 16-row / 16-column blocks
    f32x4 acc[RM][RN];
    for (int i = 0; i < RM; ++i) for (int j = 0; j < RN; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int r16 = lane & 15, g = lane >> 4;
    const int c16 = (g ^ swz16(r16)) << 4;   // rows are r16 + multiples of 16: (row >> 2) & 3 depends on r16 only
    bf16x8 fa[3][RM], fb[3];
    for (int it = 0; it < iters; ++it) {
      if (READS || it == 0) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int i = 0; i < RM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(sA + (pl * BM + wm * 32 * TM + i * 16 + r16) * 64 + c16);
      }
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        if (READS || it == 0) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(sB + (pl * BN + wn * 32 * TN + j * 16 + r16) * 64 + c16);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < RM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[t]][i], fb[PB[t]], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < RM; ++i) for (int j = 0; j < RN; ++j) for (int e = 0; e < 4; ++e) accsum += acc[i][j][e];
  }
  out[blockIdx.x * 256 + threadIdx.x] = accsum;
}

static uint16_t rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float up(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int SHAPE, int TM, int TN, int READS>
static void run(const char* name, const uint16_t* planes, float* out, int nb, int iters) {
  constexpr int LDS = 3 * 64 * (TM + TN) * 64;
  auto kern = k<SHAPE, TM, TN, READS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), LDS, 0, planes, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * (64.0 * TM) * (64.0 * TN) * 32 * (double)iters * nb;   // f32-equivalent FLOP (6 bf16 MFMA FLOP each)
    if (rep) printf("%-44s rep %d: %8.2f ms  %6.1f TFLOP/s f32-equivalent (%5.0f on the bf16 pipes)\n", name, rep, ms, fl / ms / 1e9, 6 * fl / ms / 1e9);
  }
}

int main() {
  const int nb = 256 * 4;
  uint16_t* planes;
  float* out;
  (void)hipMalloc(&planes, 3 * 512 * 32 * 2);
  (void)hipMalloc(&out, (size_t)nb * 256 * 4);
  uint16_t* h = (uint16_t*)malloc(3 * 512 * 32 * 2);
  srand(1);
  for (int i = 0; i < 512 * 32; ++i) {
    float r = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (int pl = 0; pl < 3; ++pl) {
      const uint16_t b = rne(r);
      h[pl * 512 * 32 + i] = b;
      r -= up(b);
    }
  }
  hipMemcpy(planes, h, 3 * 512 * 32 * 2, hipMemcpyHostToDevice);
  const int iters = 3000;
  for (int round = 0; round < 2; ++round) {
    run<32, 2, 4, 1>("32x32x16  wave 64x128  fragments from LDS", planes, out, nb, iters);
    run<16, 2, 4, 1>("16x16x32  wave 64x128  fragments from LDS", planes, out, nb, iters);
    run<32, 2, 4, 0>("32x32x16  wave 64x128  fragments in registers", planes, out, nb, iters);
    run<16, 2, 4, 0>("16x16x32  wave 64x128  fragments in registers", planes, out, nb, iters);
    run<32, 4, 4, 1>("32x32x16  wave 128x128 fragments from LDS", planes, out, nb, iters / 2);
    run<16, 4, 4, 1>("16x16x32  wave 128x128 fragments from LDS", planes, out, nb, iters / 2);
    run<32, 2, 2, 1>("32x32x16  wave 64x64   fragments from LDS", planes, out, nb, iters * 2);
  }
  return 0;
}
