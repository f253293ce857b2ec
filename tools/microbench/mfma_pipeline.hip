// Micro-benchmark: what the K-loop structure of the conv kernel costs, without any global traffic.
//   MODE 0: LDS fragment reads + 64 MFMAs (32x32x2 f32) per 32-k step, nothing else
//   MODE 1: + per step [__syncthreads; 8 x ds_write_b128 per thread (restaging the tile from registers); __syncthreads]
//   MODE 2: MODE 1 with 64-k steps (half as many barriers per MFMA)
//   MODE 3: MODE 1 + the restaged tile is re-fetched from global memory every step (8 x 16-B loads per thread issued before the
//           MFMAs, like the conv kernel's register-staged loader; each workgroup streams its own 4 MB window)
// WPC = workgroups per CU (launch bound), 256 threads each, every wave owns a 64x64 accumulator tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int WPC>
__global__ __launch_bounds__(256, WPC) void k(const float* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sA[128 * 32], sB[128 * 32];
  f32x4 stage[8];
  for (int i = 0; i < 8; ++i) stage[i] = *(const f32x4*)&in[(threadIdx.x * 8 + i) * 4];
  for (int i = threadIdx.x; i < 128 * 32; i += 256) { sA[i] = in[i]; sB[i] = in[4096 + i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, half = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const float* gsrc = in + ((size_t)(blockIdx.x % 64) << 20) + threadIdx.x * 32;  // MODE 3: 4 MB window per workgroup id % 64
  for (int it = 0; it < iters; ++it) {
    f32x4 nxt[8];
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) nxt[i] = *(const f32x4*)&gsrc[((it & 127) * 8192) + i * 4];
    }
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
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stage[i] = nxt[i];
    }
    if (MODE == 1 || MODE == 3 || (MODE == 2 && (it & 1))) {
      __syncthreads();
      const int row0 = threadIdx.x >> 3, ch = threadIdx.x & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(f32x4*)&sA[(row0 + 32 * i) * 32 + ((ch ^ ((row0 + 32 * i) & 7)) << 2)] = stage[i];
        *(f32x4*)&sB[(row0 + 32 * i) * 32 + ((ch ^ ((row0 + 32 * i) & 7)) << 2)] = stage[4 + i];
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int WPC>
static void run(const float* in, float* out, int nb, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPC>), dim3(nb), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s rep %d: %.2f ms  %.1f TFLOP/s\n", name, rep, ms, 2.0 * 128 * 128 * 32 * (double)iters * nb / ms / 1e9);
  }
}
int main() {
  float *in, *out; const int nb = 768 * 4;
  (void)hipMalloc(&in, (size_t)64 << 22); (void)hipMalloc(&out, nb * 256 * 4);
  const size_t nfl = (size_t)64 << 20;
  float* h = (float*)malloc(nfl * 4); for (size_t i = 0; i < nfl; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f);
  (void)hipMemcpy(in, h, nfl * 4, hipMemcpyHostToDevice);
  const int iters = 3000;
  run<0, 2>(in, out, nb, iters, "pure LDS-read + MFMA, 2 WG/CU");
  run<0, 3>(in, out, nb, iters, "pure LDS-read + MFMA, 3 WG/CU");
  run<1, 2>(in, out, nb, iters, "+ barrier/restage every 32 k, 2 WG/CU");
  run<1, 3>(in, out, nb, iters, "+ barrier/restage every 32 k, 3 WG/CU");
  run<2, 2>(in, out, nb, iters, "+ barrier/restage every 64 k, 2 WG/CU");
  run<2, 3>(in, out, nb, iters, "+ barrier/restage every 64 k, 3 WG/CU");
  run<3, 2>(in, out, nb, iters, "+ tile re-fetched from global each step, 2 WG/CU");
  run<3, 3>(in, out, nb, iters, "+ tile re-fetched from global each step, 3 WG/CU");
  return 0;
}
