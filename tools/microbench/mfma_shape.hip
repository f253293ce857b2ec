// Micro-benchmark: f32 MFMA shape vs sustained clock.  Each wave accumulates a 64x64 tile from LDS-resident random operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const float* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sA[128 * 32], sB[128 * 32];
  for (int i = threadIdx.x; i < 128 * 32; i += 256) { sA[i] = in[i]; sB[i] = in[4096 + i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float accsum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r32 = lane & 31, half = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 fa[2], fb[2];
        for (int i = 0; i < 2; ++i) fa[i] = *(const f32x4*)&sA[(wm * 64 + i * 32 + r32) * 32 + (((2 * q + half) ^ (r32 & 7)) << 2)];
        for (int j = 0; j < 2; ++j) fb[j] = *(const f32x4*)&sB[(wn * 64 + j * 32 + r32) * 32 + (((2 * q + half) ^ (r32 & 7)) << 2)];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) accsum += acc[i][j][e];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int r16 = lane & 15, g = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // 16 k per q
        f32x4 fa[4], fb[4];
        for (int i = 0; i < 4; ++i) fa[i] = *(const f32x4*)&sA[(wm * 64 + i * 16 + r16) * 32 + (((4 * q + g) ^ (r16 & 7)) << 2)];
        for (int j = 0; j < 4; ++j) fb[j] = *(const f32x4*)&sB[(wn * 64 + j * 16 + r16) * 32 + (((4 * q + g) ^ (r16 & 7)) << 2)];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) accsum += acc[i][j][e];
  }
  out[blockIdx.x * 256 + threadIdx.x] = accsum;
}
int main() {
  float *in, *out; const int nb = 512 * 8;
  (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, nb * 256 * 4);
  float* h = (float*)malloc(8192 * 4); for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f);
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  const int iters = 4000;  // each iter: 32 k over a 128x128 block tile
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape : {32, 16, 32, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(nb), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k<16>, dim3(nb), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 2.0 * 128 * 128 * 32 * (double)iters * nb;
      printf("shape %dx%d rep %d: %.2f ms  %.1f TFLOP/s\n", shape, shape, rep, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
