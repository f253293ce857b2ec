// placeholder until the TRBA kernels land (replaced in the next commit)
#include <hip/hip_runtime.h>
#include "msocr.h"
extern "C" int msocr_se_residual(const void*, const void*, int, int, int, int, const float*, const float*, float*, void*, void*) { return MSOCR_E_ARG; }
extern "C" int msocr_mean_over_h(const void*, int, int, int, int, int, float*, void*) { return MSOCR_E_ARG; }
extern "C" int msocr_bilstm_recurrent(const float*, const float*, int, int, int, float*, void*) { return MSOCR_E_ARG; }
extern "C" int msocr_linear_f32(const float*, const float*, const float*, int, int, int, float*, void*) { return MSOCR_E_ARG; }
extern "C" int msocr_attn_greedy(const float*, const float*, const msocr_attn_weights*, int, int, int, int, int, int, int, int, float*, int32_t*, void*) { return MSOCR_E_ARG; }
extern "C" int msocr_attn_beam(const float*, const float*, const msocr_attn_weights*, int, int, int, int, int, int, float, float, int, int, int, float*, int32_t*, int32_t*, void*, void*) { return MSOCR_E_ARG; }
extern "C" int64_t msocr_attn_beam_workspace_bytes(int, int, int, int) { return 0; }
