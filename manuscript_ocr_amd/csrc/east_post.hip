// east_post.hip — EAST post-processing on the device: quad decode and locality-aware NMS.
// Compile with -ffp-contract=off: the fp64 geometry must round exactly like the
// reference's numba/NumPy code (no FMA contraction), results are compared bit for bit.
//
//   msocr_east_decode  <- detectors/_east/utils.py:328-381  decode_quads_from_maps
//   msocr_east_lanms   <- detectors/_east/lanms.py:7-207    polygon_* / normalize_polygon /
//                                                           standard_nms / locality_aware_nms
// One workgroup per page; every kernel has a bounded trip count (no spinning).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msocr.h"

// Clear any stale (sticky) HIP error left by earlier runtime calls of the host process before a launch,
// so that the status read back after it belongs to this launch.
#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH)

// ------------------------------------------------------------------------------------------ decode
// Cells of q x q pixels in row-major order == np.unique(axis=0) order of the quantised (y, x)
// pairs (utils.py:349-356).  A cell is a candidate iff any of its pixels has score > thr
// (strict, f32).  Row = 4 vertices (x*s + dx*s, y*s + dy*s) + score at the cell CENTRE.
__global__ __launch_bounds__(1024) void east_decode_kernel(const float* __restrict__ score, const float* __restrict__ geo, int H,
                                                            int W, float thr, double scale, int q, float* __restrict__ cand,
                                                            int32_t* __restrict__ count, int max_cand) {
  const int n = blockIdx.x;
  const float* sm = score + (long)n * H * W;
  const float* gm = geo + (long)n * H * W * 8;
  float* out = cand + (long)n * max_cand * 9;
  const int Hq = H / q, Wq = W / q;
  const int ncell = Hq * Wq;
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  const float scale_f = (float)scale;
  for (int c0 = 0; c0 < ncell; c0 += 1024) {
    const int cell = c0 + tid;
    bool on = false;
    int cy = 0, cx = 0;
    if (cell < ncell) {
      cy = cell / Wq;
      cx = cell - cy * Wq;
      for (int dy = 0; dy < q; ++dy)
        for (int dx = 0; dx < q; ++dx) on |= sm[(long)(cy * q + dy) * W + cx * q + dx] > thr;
    }
    const unsigned long long bal = __ballot(on);
    const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
    if (on) {
      const int idx = off + wpre;
      if (idx < max_cand) {
        const int y = q > 1 ? cy * q + q / 2 : cy, x = q > 1 ? cx * q + q / 2 : cx;
        const float* g = gm + ((long)y * W + x) * 8;
        float* o = out + (long)idx * 9;
        const double xs = (double)x * scale, ys = (double)y * scale;  // int64 * float -> f64
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dxs = g[2 * i] * scale_f, dys = g[2 * i + 1] * scale_f;  // f32 * f32
          o[2 * i] = (float)(xs + (double)dxs);                               // f64 add, one rounding to f32
          o[2 * i + 1] = (float)(ys + (double)dys);
        }
        o[8] = sm[(long)y * W + x];
      }
    }
    __syncthreads();
    if (tid == 0) base_s += tot;
    __syncthreads();
  }
  if (tid == 0) {
    const int total = base_s;
    count[n] = total > max_cand ? (max_cand | (int)0x80000000) : total;
  }
}

extern "C" int msocr_east_decode(const float* score, const float* geo, int N, int H, int W, float thresh, float scale, int quant,
                                 float* cand_out, int32_t* count_out, int max_cand, void* stream) {
  if (!score || !geo || !cand_out || !count_out || N <= 0 || H <= 0 || W <= 0 || quant <= 0 || max_cand <= 0) return MSOCR_E_ARG;
  if (H % quant || W % quant) return MSOCR_E_ARG;  // the reference would index out of range (utils.py:349-356,369)
  MSOCR_LAUNCH(east_decode_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, score, geo, H, W, thresh, (double)scale, quant,
                     cand_out, count_out, max_cand);
  return LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ fp64 geometry
#define MAXV 20  // lanms.py:34

__device__ double d_polygon_area(const double* poly, int n) {  // lanms.py:7-14
  double area = 0.0;
  for (int i = 0; i < n; i++) {
    const int j = (i + 1) % n;
    area += poly[2 * i] * poly[2 * j + 1] - poly[2 * j] * poly[2 * i + 1];
  }
  return fabs(area) / 2.0;
}

__device__ void d_compute_intersection(const double* p1, const double* p2, const double* A, const double* B, double* out) {  // :17-29
  const double BAx = p2[0] - p1[0], BAy = p2[1] - p1[1];
  const double DCx = B[0] - A[0], DCy = B[1] - A[1];
  const double denom = BAx * DCy - BAy * DCx;
  const double CAx = A[0] - p1[0], CAy = A[1] - p1[1];
  if (denom == 0) {
    out[0] = p1[0];
    out[1] = p1[1];
    return;
  }
  const double t = (CAx * DCy - CAy * DCx) / denom;
  out[0] = p1[0] + t * BAx;
  out[1] = p1[1] + t * BAy;
}

__device__ int d_clip_polygon(const double* subject, int n, const double* A, const double* B, double* out) {  // :32-57
  int count = 0;
  for (int i = 0; i < n; i++) {
    const double* curr = subject + 2 * i;
    const double* prev = subject + 2 * ((i - 1 + n) % n);
    const bool curr_in = (B[0] - A[0]) * (curr[1] - A[1]) - (B[1] - A[1]) * (curr[0] - A[0]) >= 0;
    const bool prev_in = (B[0] - A[0]) * (prev[1] - A[1]) - (B[1] - A[1]) * (prev[0] - A[0]) >= 0;
    if (curr_in) {
      if (!prev_in) {
        d_compute_intersection(prev, curr, A, B, out + 2 * count);
        count++;
      }
      out[2 * count] = curr[0];
      out[2 * count + 1] = curr[1];
      count++;
    } else if (prev_in) {
      d_compute_intersection(prev, curr, A, B, out + 2 * count);
      count++;
    }
  }
  return count;
}

__device__ double d_polygon_iou(const double* poly1, const double* poly2) {  // :60-91 for two quads
  double bufa[2 * MAXV], bufb[2 * MAXV];
  double *cur = bufa, *nxt = bufb;
  for (int k = 0; k < 8; ++k) cur[k] = poly1[k];
  int cnt = 4;
  for (int i = 0; i < 4; i++) {
    cnt = d_clip_polygon(cur, cnt, poly2 + 2 * i, poly2 + 2 * ((i + 1) % 4), nxt);
    double* t = cur;
    cur = nxt;
    nxt = t;
    if (cnt == 0) break;
  }
  double inter_area = 0.0;
  if (cnt > 2) inter_area = d_polygon_area(cur, cnt);
  const double area1 = d_polygon_area(poly1, 4), area2 = d_polygon_area(poly2, 4);
  const double union_area = area1 + area2 - inter_area;
  if (union_area <= 0) return 0.0;
  return inter_area / union_area;
}

__device__ void d_normalize_polygon(const double* ref, const double* poly, double* out) {  // :99-130
  int best_order = 0, best_start = 0;
  double min_d = 1e20;
  for (int start = 0; start < 4; start++) {
    double d = 0.0;
    for (int i = 0; i < 4; i++) {
      const int idx = (start + i) & 3;
      const double dx = ref[2 * i] - poly[2 * idx], dy = ref[2 * i + 1] - poly[2 * idx + 1];
      d += dx * dx + dy * dy;
    }
    if (d < min_d) { min_d = d; best_start = start; best_order = 0; }
  }
  for (int start = 0; start < 4; start++) {
    double d = 0.0;
    for (int i = 0; i < 4; i++) {
      const int idx = (start - i) & 3;
      const double dx = ref[2 * i] - poly[2 * idx], dy = ref[2 * i + 1] - poly[2 * idx + 1];
      d += dx * dx + dy * dy;
    }
    if (d < min_d) { min_d = d; best_start = start; best_order = 1; }
  }
  for (int i = 0; i < 4; i++) {
    const int idx = best_order == 0 ? (best_start + i) & 3 : (best_start - i) & 3;
    out[2 * i] = poly[2 * idx];
    out[2 * i + 1] = poly[2 * idx + 1];
  }
}

// total order used for both sorts: ascending key, NaN last, ties by original index (stable).
// (np.argsort's default sort is unstable: the reference leaves tie order implementation-defined.)
template <typename K>
__device__ __forceinline__ bool key_less(K a, K b) { return a < b || (b != b && a == a); }

// ------------------------------------------------------------------------------------------ LANMS
// workspace layout per page (all 8-byte aligned):
//   order  int32[max_cand]     x0-sorted candidate indices, later score-sorted merged indices
//   mpoly  double[max_cand*8]  merged polygons (phase 1)
//   mscore double[max_cand]
//   supp   int32[max_cand]     suppression flags (phase 2)
static inline int64_t lanms_ws_per_page(int max_cand) {
  return (int64_t)max_cand * (4 + 64 + 8 + 4) + 64;
}
extern "C" int64_t msocr_lanms_workspace_bytes(int N, int max_cand) {
  if (N <= 0 || max_cand <= 0) return 0;
  return (lanms_ws_per_page(max_cand) + 63) / 64 * 64 * N;
}

__global__ __launch_bounds__(1024) void east_lanms_kernel(const float* __restrict__ cand, const int32_t* __restrict__ counts,
                                                           int max_cand, double thr, float* __restrict__ boxes_out,
                                                           int32_t* __restrict__ nbox_out, char* __restrict__ ws, long ws_stride) {
  const int pg = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const float* cb = cand + (long)pg * max_cand * 9;
  const int n = counts[pg] & 0x7fffffff;
  char* w = ws + (long)pg * ws_stride;
  double* mpoly = reinterpret_cast<double*>(w);
  double* mscore = mpoly + (long)max_cand * 8;
  int32_t* order = reinterpret_cast<int32_t*>(mscore + max_cand);
  int32_t* supp = order + max_cand;
  float* ob = boxes_out + (long)pg * max_cand * 9;
  __shared__ int nm_s, nk_s, cur_alive;
  __shared__ double cur_poly[8];

  if (n == 0) {
    if (tid == 0) nbox_out[pg] = 0;
    return;
  }
  // ---- sort by x0 (f32 key), stable: rank = #{j : key_j < key_i or (tie and j < i)}  (lanms.py:166-168)
  for (int i = tid; i < n; i += nthr) {
    const float ki = cb[(long)i * 9];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float kj = cb[(long)j * 9];
      rank += (key_less(kj, ki) || (!key_less(ki, kj) && j < i)) ? 1 : 0;
    }
    order[rank] = i;
  }
  __syncthreads();
  // ---- phase 1: sequential weighted merge with the LAST merged polygon (lanms.py:174-192)
  if (tid == 0) {
    int nm = 0;
    double last[8], lscore = 0.0, lw = 0.0;
    for (int s = 0; s < n; ++s) {
      const float* b = cb + (long)order[s] * 9;
      double poly[8];
      for (int k = 0; k < 8; ++k) poly[k] = (double)b[k];
      const double sc = (double)b[8];
      if (nm > 0 && d_polygon_iou(poly, last) > thr) {
        double al[8];
        d_normalize_polygon(last, poly, al);
        const double tw = lw + sc;
        for (int k = 0; k < 8; ++k) last[k] = (last[k] * lw + al[k] * sc) / tw;
        lw = tw;
        lscore = lscore > sc ? lscore : sc;
        continue;
      }
      if (nm > 0) {
        for (int k = 0; k < 8; ++k) mpoly[(long)(nm - 1) * 8 + k] = last[k];
        mscore[nm - 1] = lscore;
      }
      for (int k = 0; k < 8; ++k) last[k] = poly[k];
      lscore = sc;
      lw = sc;
      nm++;
    }
    for (int k = 0; k < 8; ++k) mpoly[(long)(nm - 1) * 8 + k] = last[k];
    mscore[nm - 1] = lscore;
    nm_s = nm;
  }
  __syncthreads();
  const int nm = nm_s;
  // ---- phase 2: order = argsort(-score) (stable), greedy suppression (lanms.py:133-153)
  for (int i = tid; i < nm; i += nthr) {
    const double ki = -mscore[i];
    int rank = 0;
    for (int j = 0; j < nm; ++j) {
      const double kj = -mscore[j];
      rank += (key_less(kj, ki) || (!key_less(ki, kj) && j < i)) ? 1 : 0;
    }
    order[rank] = i;
    supp[i] = 0;
  }
  if (tid == 0) nk_s = 0;
  __syncthreads();
  for (int i = 0; i < nm; ++i) {
    const int idx = order[i];
    if (tid == 0) {
      cur_alive = supp[idx] == 0;
      if (cur_alive) {
        for (int k = 0; k < 8; ++k) cur_poly[k] = mpoly[(long)idx * 8 + k];
        float* o = ob + (long)nk_s * 9;
        for (int k = 0; k < 8; ++k) o[k] = (float)cur_poly[k];
        o[8] = (float)mscore[idx];
        nk_s++;
      }
    }
    __syncthreads();
    if (cur_alive) {
      double a[8];
      for (int k = 0; k < 8; ++k) a[k] = cur_poly[k];
      for (int j = i + 1 + tid; j < nm; j += nthr) {
        const int idj = order[j];
        if (supp[idj]) continue;
        if (d_polygon_iou(a, mpoly + (long)idj * 8) > thr) supp[idj] = 1;
      }
    }
    __syncthreads();
  }
  if (tid == 0) nbox_out[pg] = nk_s;
}

extern "C" int msocr_east_lanms(const float* cand, const int32_t* counts, int N, int max_cand, double iou_thr, float* boxes_out,
                                int32_t* nbox_out, void* workspace, void* stream) {
  if (!cand || !counts || !boxes_out || !nbox_out || !workspace || N <= 0 || max_cand <= 0) return MSOCR_E_ARG;
  if ((uintptr_t)workspace & 7) return MSOCR_E_ARG;
  const long stride = (lanms_ws_per_page(max_cand) + 63) / 64 * 64;
  MSOCR_LAUNCH(east_lanms_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out, nbox_out,
                     (char*)workspace, stride);
  return LAUNCH_OK();
}
