// Stand-alone probe for the packed-f32 observation of attn_beam_mfma.hip: does
//     v_pk_fma_f32 D, S0, S1, D op_sel:[0,1,0]      (the LOW result takes the HIGH dword of S1)
// return wrong low results while the partner wave of the same SIMD streams v_mfma_f32_32x32x16_bf16?
// 512-thread workgroups (two waves per SIMD): waves 0-3 run the packed FMA on exactly representable integers and check every
// result, waves 4-7 run nothing (mode 0) or a bf16 MFMA loop (mode 1).  Form 0 = op_sel:[0,1,0], form 1 = the plain pairing,
// form 2 = op_sel_hi:[1,0,1] (the HIGH result takes the LOW dword of S1).
// Build: hipcc -O3 --offload-arch=gfx950 -o pk_fma_beside_mfma pk_fma_beside_mfma.hip ; prints mismatches per 16-lane group.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(512) void k(int mode, unsigned* bad /*[4] per 16-lane group*/, float* sink, int rounds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (mode == 0) return;
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.25f * ((lane + e) % 7) - 0.75f); b[e] = (__bf16)(0.5f * ((lane * 3 + e) % 5) - 1.f); }
    for (int r = 0; r < rounds * 40; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    return;
  }
  // x = (lane + 1, 2 lane + 3), y = (3, 5): 100 packed FMAs per round into a zeroed pair; every partial sum is an integer < 2^24
  const f32x2 x = {(float)(lane + 1), (float)(2 * lane + 3)};
  const f32x2 y = {3.f, 5.f};
  unsigned nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    f32x2 d = {0.f, 0.f};
    for (int i = 0; i < 100; ++i) {
      if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(d) : "v"(x), "v"(y));
      else if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y));
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d) : "v"(x), "v"(y));
    }
    // form 0: both results multiply by y.hi = 5;  form 1: lo by 3, hi by 5;  form 2: both by y.lo = 3
    const float want_lo = 100.f * x[0] * (FORM == 0 ? 5.f : 3.f), want_hi = 100.f * x[1] * (FORM == 2 ? 3.f : 5.f);
    nbad += (d[0] != want_lo) + (d[1] != want_hi);
  }
  if (nbad) atomicAdd(&bad[lane >> 4], nbad);
}

int main() {
  unsigned* bad;
  float* sink;
  (void)hipMalloc(&bad, 16);
  (void)hipMalloc(&sink, 1024 * 512 * 4);
  const int rounds = 20000;
  for (int form = 0; form < 3; ++form)
    for (int mode = 0; mode < 2; ++mode) {
      (void)hipMemset(bad, 0, 16);
      if (form == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(512), 0, 0, mode, bad, sink, rounds);
      else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(512), 0, 0, mode, bad, sink, rounds);
      else hipLaunchKernelGGL(k<2>, dim3(1024), dim3(512), 0, 0, mode, bad, sink, rounds);
      (void)hipDeviceSynchronize();
      unsigned h[4];
      (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
      printf("%-22s partner waves: %-28s mismatches by lane group 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u  (of %.3g results per group)\n",
             form == 0 ? "op_sel:[0,1,0]" : form == 1 ? "plain pairing" : "op_sel_hi:[1,0,1]", mode ? "v_mfma_f32_32x32x16_bf16 loop" : "none", h[0], h[1], h[2], h[3],
             1024.0 * 4 * 16 * 2 * rounds);
    }
  return 0;
}
