// internal.h — symbols shared between the translation units of libmsocr.so (not part of the C ABI).
#ifndef MSOCR_INTERNAL_H
#define MSOCR_INTERNAL_H
#include <hip/hip_runtime.h>

// conv_igemm.hip: nbatch independent f32 GEMMs of one shape in one launch,
// C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]  (A [nbatch][M][K], B [nbatch][N][K], C [nbatch][M][N], dense, 16-B aligned).
__attribute__((visibility("hidden"))) int msocr_internal_gemm_f32_batched(const float* A, const float* B, float* C, long M, int N,
                                                                          int K, int nbatch, hipStream_t s);
#endif
