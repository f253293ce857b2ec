// internal.h — symbols shared between the translation units of libmsocr.so (not part of the C ABI).
#ifndef MSOCR_INTERNAL_H
#define MSOCR_INTERNAL_H
#include <hip/hip_runtime.h>

// conv_igemm.hip: nbatch independent f32 GEMMs of one shape in one launch,
// C[b][m][n] = sum_k A[b][m][k] * B[b][n][k]  (A [nbatch][M][K], B [nbatch][N][K], C [nbatch][M][N], dense, 16-B aligned).
__attribute__((visibility("hidden"))) int msocr_internal_gemm_f32_batched(const float* A, const float* B, float* C, long M, int N,
                                                                          int K, int nbatch, hipStream_t s);

// conv_split.hip: the same with B given as three K-tile-major bf16 planes [3][nbatch][K/32][N][32] (B == p0 + p1 + p2 exactly), A split in
// registers: six bf16 MFMA products per f32 product, f32 accumulate (K % 32 == 0, N % 64 == 0).
__attribute__((visibility("hidden"))) int msocr_internal_gemm_split_batched(const float* A, const uint16_t* Bplanes, float* C, long M,
                                                                            int N, int K, int nbatch, hipStream_t s);

// attention decoder arguments shared by trba_kernels.hip (VALU kernels) and attn_beam_mfma.hip (matrix-core beam kernel)
#include "msocr.h"
struct AttnArgs {
  const float* batch_H;
  const float* proj_H;
  const float* ctx_gates;  // optional (matrix-core beam kernel): [B][T][H][4] = batch_H x rnn.weight_ih[:, :H]^T, hoisted out of the step loop
  msocr_attn_weights w;
  const uint16_t *h2h_p, *whh_p, *gen_p;  // optional (matrix-core beam kernel, with ctx_gates): msocr_attn_split_weights
  int B, T, V, steps, K;
  int sos_id, eos_id, blank_id;
  float temperature;
  const float* lp;      // [steps] f32 length-penalty factors (beam, alpha > 0) or nullptr
  float* logits_out;    // greedy: [B][steps][V]; beam: workspace [B][steps][K][V]
  int32_t* ids_out;     // greedy: [B][steps]
  int32_t* back;        // beam: [B][steps][K]
  int32_t* tokv;        // beam: [B][steps][K]
  int32_t* best_at;     // beam: [B][steps]
  int32_t* fin_step;    // beam: [B]
  const int32_t* chunk_id;    // beam, optional: [B] index of the reference chunk (one predict() slice of batch_size crops) of each crop
  const int32_t* chunk_size;  // [nchunks] crops per chunk
  int32_t* chunk_state;       // [2*nchunks] zeroed by the caller: {crops finished, max finish step}; enables the early exit
};
// attn_beam_mfma.hip: beam decode with 4 crops x 8 beams per workgroup on the f32 matrix cores (H == 256, beam <= 8, T <= 64)
__attribute__((visibility("hidden"))) int msocr_internal_attn_beam_mfma(const AttnArgs& a, hipStream_t s);
// attn_beam_mfma.hip: greedy decode with 32 crops per workgroup on the matrix cores (needs ctx_gates and the split weights)
__attribute__((visibility("hidden"))) int msocr_internal_attn_greedy_mfma(const AttnArgs& a, hipStream_t s);
// attn_general.hip: greedy / beam decode for hidden sizes other than 256, charsets above 256 tokens, beams above 8 (see its header
// for the shapes taken); same outputs and workspace layout as the fast kernels.
__attribute__((visibility("hidden"))) int msocr_internal_attn_general(const AttnArgs& a, int H, bool beam, hipStream_t s);
#endif
