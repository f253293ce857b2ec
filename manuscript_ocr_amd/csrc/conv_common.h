// conv_common.h — types shared by the MFMA convolution / GEMM kernels (conv_igemm.hip: exact-f32 and bf16 operands;
// conv_split.hip: f32 operands split into three bf16 terms).  Not part of the C ABI.
#ifndef MSOCR_CONV_COMMON_H
#define MSOCR_CONV_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// Clear any stale (sticky) HIP error left by earlier runtime calls of the host process before a launch,
// so that the status read back after it belongs to this launch.
#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct ConvParams {
  const char* in;
  const char* w;
  const float* bias;
  const char* res;
  char* out;
  int N, H, W, Cin;
  long sN, sH, sW;
  int KH, KW, SH, SW, PH, PW;
  int Ho, Wo, Cout;
  long M;
  int ktiles, cin_tiles;
  long Ktot;
  long out_ld, res_ld;
  int relu, has_res;
  int tilesM, tilesN;
  int nbatch;           // independent problems of identical shape in one launch (Winograd: the 16 transform points)
  long bsA, bsW, bsO;   // element strides between consecutive problems (input, weight, output)
  long wplane;          // conv_split.hip: elements between the three bf16 planes of the split weight operand
  long w_kt_b;          // conv_split.hip: bytes between consecutive K-tiles of a weight plane (K-tile-major planes: Cout * 64)
};

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return *reinterpret_cast<uint16_t*>(&b);
}

template <typename T>
struct Mma;
template <>
struct Mma<float> {
  // one 16-byte chunk = 4 consecutive k for this lane's (row, k-half): 4 MFMA 32x32x2 steps
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  }
};
template <>
struct Mma<__bf16> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    bf16x8 av = __builtin_bit_cast(bf16x8, a);
    bf16x8 bv = __builtin_bit_cast(bf16x8, b);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  }
};

// two f32 -> (packed bf16 pair of the leading terms, the two residuals): one step of the exact three-term split of conv_split.hip
__device__ __forceinline__ uint32_t split_step(float& x, float& y) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {x, y};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32, round to nearest even
  const uint32_t pk = __builtin_bit_cast(uint32_t, h);
  x -= __uint_as_float(pk << 16);          // exact: the leading term shares x's exponent
  y -= __uint_as_float(pk & 0xffff0000u);
  return pk;
}

// conv_split_pp.hip: the producer / consumer form of the split-operand kernel (Cout % 128 == 0); same ConvParams as conv_split_kernel
__attribute__((visibility("hidden"))) bool msocr_internal_split_pp_takes(const ConvParams& p);
__attribute__((visibility("hidden"))) int msocr_internal_split_pp_launch(ConvParams& p, hipStream_t s, bool general);

// BKB = bytes per tile row (64 or 128).  swizzle: 16-B chunk index ^= (row / rows_per_256B) % chunks_per_row
template <int BKB>
__device__ __forceinline__ int swz(int row) {
  constexpr int CPR = BKB / 16;
  constexpr int RPB = 256 / BKB;
  return (row / RPB) & (CPR - 1);
}

#endif
