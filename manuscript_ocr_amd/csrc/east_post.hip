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
#include <stdio.h>
#include <stdlib.h>

#include "msocr.h"

// Clear any stale (sticky) HIP error left by earlier runtime calls of the host process before a launch,
// so that the status read back after it belongs to this launch.
#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH)

// ------------------------------------------------------------------------------------------ decode
// Cells of q x q pixels in row-major order == np.unique(axis=0) order of the quantised (y, x)
// pairs (utils.py:349-356).  A cell is a candidate iff any of its pixels has score > thr
// (strict, f32).  Row = 4 vertices (x*s + dx*s, y*s + dy*s) + score at the cell CENTRE.
// Hardening (round 4; the reference has no defined behaviour here — NaN polygons run into numba / cv2): a candidate whose centre
// score or any of whose eight offsets is NaN, infinite or >= 1e7 in magnitude is DROPPED here, the one funnel every later kernel
// (LANMS, box filters, reading order, crops) is fed from: sorts, bucket indices and float -> int conversions downstream then only
// ever see finite coordinates below 2^24.  Maps a network can produce are unaffected.
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
      if (on) {
        const int y = q > 1 ? cy * q + q / 2 : cy, x = q > 1 ? cx * q + q / 2 : cx;
        const float* g = gm + ((long)y * W + x) * 8;
        bool sane = fabsf(sm[(long)y * W + x]) < 1.0e7f;   // false for NaN
#pragma unroll
        for (int i = 0; i < 8; ++i) sane &= fabsf(g[i]) < 1.0e7f;
        on = sane;
      }
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
      if (!prev_in && count < MAXV) {
        d_compute_intersection(prev, curr, A, B, out + 2 * count);
        count++;
      }
      if (count < MAXV) {
        out[2 * count] = curr[0];
        out[2 * count + 1] = curr[1];
        count++;
      }
    } else if (prev_in && count < MAXV) {
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

// Exact shortcut for far-apart quads: when the clip polygon (poly2) is strictly convex and the two axis-aligned
// bounding boxes are separated by a margin far above rounding error, Sutherland-Hodgman returns an empty
// polygon (every surviving vertex would have to lie within rounding distance of BOTH hulls), so
// inter_area = 0 and the reference's IoU is exactly 0.0 (or 0.0 through its union<=0 guard).  Non-convex or
// self-intersecting clip quads, whose half-plane intersection can be unbounded, always take the full path.
__device__ __forceinline__ bool d_surely_disjoint(const double* p1, const double* p2) {
  double s0 = 0.0;
  for (int i = 0; i < 4; ++i) {
    const double ax = p2[2 * ((i + 1) & 3)] - p2[2 * i], ay = p2[2 * ((i + 1) & 3) + 1] - p2[2 * i + 1];
    const double bx = p2[2 * ((i + 2) & 3)] - p2[2 * ((i + 1) & 3)], by = p2[2 * ((i + 2) & 3) + 1] - p2[2 * ((i + 1) & 3) + 1];
    const double c = ax * by - ay * bx;
    if (!(fabs(c) > 1e-9)) return false;           // degenerate turn (or NaN): not provably convex
    if (i == 0) s0 = c; else if ((c > 0) != (s0 > 0)) return false;
  }
  double l1 = p1[0], r1 = p1[0], t1 = p1[1], b1 = p1[1], l2 = p2[0], r2 = p2[0], t2 = p2[1], b2 = p2[1];
  for (int i = 1; i < 4; ++i) {
    l1 = fmin(l1, p1[2 * i]); r1 = fmax(r1, p1[2 * i]); t1 = fmin(t1, p1[2 * i + 1]); b1 = fmax(b1, p1[2 * i + 1]);
    l2 = fmin(l2, p2[2 * i]); r2 = fmax(r2, p2[2 * i]); t2 = fmin(t2, p2[2 * i + 1]); b2 = fmax(b2, p2[2 * i + 1]);
  }
  const double ext = fmax(fmax(fabs(l1), fabs(r1)), fmax(fmax(fabs(t1), fabs(b1)), fmax(fmax(fabs(l2), fabs(r2)), fmax(fabs(t2), fabs(b2)))));
  const double margin = 1e-6 * (1.0 + ext);
  if (!(ext < 1e12)) return false;                  // inf / NaN coordinates: full path
  return (l1 > r2 + margin) || (l2 > r1 + margin) || (t1 > b2 + margin) || (t2 > b1 + margin);
}
__device__ int g_lanms_shortcut = 1;

// ---- register-resident polygon_iou for the common case ------------------------------------------------------------------
// d_polygon_iou keeps its two clip buffers (2 x 20 vertices, dynamically indexed) in scratch memory, i.e. every vertex of
// every Sutherland-Hodgman stage is a round trip through the memory pipeline: ~40 us per call, and the sequential merge scan
// is a chain of such calls.  Clipping a quad by the four half-planes of a convex quad never holds more than 8 vertices, so
// this variant keeps the polygon in 8 + 8 registers with statically indexed (fully unrolled, predicated) loops and a select
// chain for the one dynamic operation, "append at position count".  It performs the reference's operations in the
// reference's order (lanms.py:17-91) — same values bit for bit — and reports failure the moment a stage would hold a 9th
// vertex (possible only for non-convex / self-intersecting quads); the caller then runs the general version.
struct Poly8 {
  double x[8], y[8];
  int n;
};
__device__ __forceinline__ void p8_push(Poly8& o, double px, double py, bool& over) {
  if (o.n >= 8) { over = true; return; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o.x[k] = (o.n == k) ? px : o.x[k];
    o.y[k] = (o.n == k) ? py : o.y[k];
  }
  ++o.n;
}
__device__ __forceinline__ void p8_clip(const Poly8& s, double Ax, double Ay, double Bx, double By, Poly8& o, bool& over) {
  o.n = 0;
  // prev of vertex 0 is vertex n-1
  double px = s.x[0], py = s.y[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    px = (s.n - 1 == k) ? s.x[k] : px;
    py = (s.n - 1 == k) ? s.y[k] : py;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < s.n) {
      const double cx = s.x[i], cy = s.y[i];
      const bool curr_in = (Bx - Ax) * (cy - Ay) - (By - Ay) * (cx - Ax) >= 0;
      const bool prev_in = (Bx - Ax) * (py - Ay) - (By - Ay) * (px - Ax) >= 0;
      if (curr_in != prev_in) {  // compute_intersection(prev, curr, A, B)  (lanms.py:17-29)
        const double BAx = cx - px, BAy = cy - py;
        const double DCx = Bx - Ax, DCy = By - Ay;
        const double denom = BAx * DCy - BAy * DCx;
        const double CAx = Ax - px, CAy = Ay - py;
        double ix = px, iy = py;
        if (denom != 0) {
          const double t = (CAx * DCy - CAy * DCx) / denom;
          ix = px + t * BAx;
          iy = py + t * BAy;
        }
        p8_push(o, ix, iy, over);
      }
      if (curr_in) p8_push(o, cx, cy, over);
      px = cx;
      py = cy;
    }
  }
}
__device__ __forceinline__ double p8_area(const Poly8& p) {  // polygon_area (lanms.py:7-14)
  double area = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < p.n) {
      const double xj = (i + 1 < p.n && i + 1 < 8) ? p.x[(i + 1) & 7] : p.x[0];
      const double yj = (i + 1 < p.n && i + 1 < 8) ? p.y[(i + 1) & 7] : p.y[0];
      area += p.x[i] * yj - xj * p.y[i];
    }
  }
  return fabs(area) / 2.0;
}
// returns false (result unspecified) when a clip stage needs more than 8 vertices
__device__ __forceinline__ bool d_polygon_iou_fast(const double* poly1, const double* poly2, double* iou) {
  Poly8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a.x[k] = 0.0; a.y[k] = 0.0; b.x[k] = 0.0; b.y[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { a.x[k] = poly1[2 * k]; a.y[k] = poly1[2 * k + 1]; }
  a.n = 4;
  bool over = false;
  // the four clip stages of polygon_intersection (lanms.py:60-77); the live polygon alternates between b and a
  p8_clip(a, poly2[0], poly2[1], poly2[2], poly2[3], b, over);
  int cnt = b.n;
  bool in_a = false;
  if (cnt != 0) {
    p8_clip(b, poly2[2], poly2[3], poly2[4], poly2[5], a, over);
    cnt = a.n;
    in_a = true;
    if (cnt != 0) {
      p8_clip(a, poly2[4], poly2[5], poly2[6], poly2[7], b, over);
      cnt = b.n;
      in_a = false;
      if (cnt != 0) {
        p8_clip(b, poly2[6], poly2[7], poly2[0], poly2[1], a, over);
        cnt = a.n;
        in_a = true;
      }
    }
  }
  if (over) return false;
  double inter_area = 0.0;
  if (cnt > 2) inter_area = in_a ? p8_area(a) : p8_area(b);
  double s1 = 0.0, s2 = 0.0;  // polygon_area of the two quads (lanms.py:7-14)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) & 3;
    s1 += poly1[2 * i] * poly1[2 * j + 1] - poly1[2 * j] * poly1[2 * i + 1];
    s2 += poly2[2 * i] * poly2[2 * j + 1] - poly2[2 * j] * poly2[2 * i + 1];
  }
  const double area1 = fabs(s1) / 2.0, area2 = fabs(s2) / 2.0;
  const double union_area = area1 + area2 - inter_area;
  *iou = union_area <= 0 ? 0.0 : inter_area / union_area;
  return true;
}
__device__ int g_lanms_fastclip = 1;
__device__ __forceinline__ double d_polygon_iou_q(const double* poly1, const double* poly2) {
  if (g_lanms_shortcut && d_surely_disjoint(poly1, poly2)) return 0.0;
  double r;
  if (g_lanms_fastclip && d_polygon_iou_fast(poly1, poly2, &r)) return r;
  return d_polygon_iou(poly1, poly2);
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
// One workgroup (1024 threads) per page.
//
// Phase 1 of the reference (lanms.py:174-192) is a sequential scan: each x0-sorted candidate either merges
// into the LAST merged polygon (weighted mean after vertex-order normalisation) or starts a new one.  It
// is parallelised EXACTLY with a speculative segmented scan:
//   A. every thread scans its own contiguous segment starting from an EMPTY state and records where it
//      starts new polygons (spec_break) and its final open polygon (carry_spec);
//   B. with the predecessor's carry as the true incoming state, a thread re-scans only until it starts a
//      new polygon at a position where the speculative scan ALSO started one: from there both scans hold
//      the identical state fresh(candidate) and see identical inputs, so the speculative tail is the true
//      tail.  Carries are iterated to a fixed point (a run that spans several segments needs one round per
//      segment; typical pages need two rounds);
//   C. the true prefix is replayed once more to emit closed polygons into slots indexed by the polygon's
//      LAST candidate (unique), replacing the speculative emissions of that prefix;
//   D. the flagged slots are compacted in order.
// Every floating-point operation is the reference's, in the reference's order, so results are bit-identical.
//
// workspace layout per page (8-byte aligned blocks):
//   mpoly  f64[max_cand*8], mscore f64[max_cand]   compacted merged polygons (phase 1 output)
//   spoly  f64[max_cand*8], sscore f64[max_cand]   staged slots (by last candidate)
//   carry  f64[3][1024][11]                        spec / current / next carry states (poly, weight, score, valid)
//   order  i32[max_cand], supp i32[max_cand], flag i32[max_cand], sbreak i32[max_cand]
#define LANMS_T 1024
#define CARRY_W 11
// phase 2 across the chip: pages with at most NMS_BITCAP merged polygons get their "IoU > thr" relation as a bit matrix
// computed by many workgroups (lanms_iou_bits_kernel) and a single wave then replays the greedy pass on the bits
#define NMS_BITCAP 8192
#define NMS_BITW (NMS_BITCAP / 32)
static inline int64_t lanms_bits_off(int max_cand) {
  return ((int64_t)max_cand * (2 * (64 + 8) + 4 * 4) + (int64_t)3 * LANMS_T * CARRY_W * 8 + 64 + 63) / 64 * 64;
}
static inline int nms_bitcap(int max_cand) { return ((max_cand < NMS_BITCAP ? max_cand : NMS_BITCAP) + 31) / 32 * 32; }
static inline int64_t lanms_ws_per_page(int max_cand) {
  const int64_t cap = nms_bitcap(max_cand);
  return lanms_bits_off(max_cand) + cap * (cap / 32) * 4 + 64;  // ... + bit matrix [cap][cap/32] u32 + {nm} header
}
extern "C" int64_t msocr_lanms_workspace_bytes(int N, int max_cand) {
  if (N <= 0 || max_cand <= 0) return 0;
  return (lanms_ws_per_page(max_cand) + 63) / 64 * 64 * N;
}

// Order-preserving integer images of the sort keys: ascending float order, -0 == +0, every NaN last; the original
// index in the low bits makes the order strict and the sort stable (np.argsort's default is unstable: ties are
// implementation-defined in the reference).  rank_i = #{j : K_j < K_i} is then ONE unsigned compare per pair.
__device__ __forceinline__ uint32_t sortable_f32(float x) {
  if (x != x) return 0xffffffffu;
  x = x + 0.0f;  // -0 -> +0
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long sortable_f64(double x) {
  if (x != x) return 0xffffffffffffffffull;
  x = x + 0.0;
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

#define RS_TILE 2048
#define RS_Q 8
// x0 sort of every page's candidates, spread over the whole chip: grid (i chunks, j chunks, pages).  A workgroup counts, for its
// 256*RS_Q candidates, how many of ITS RS_TILE candidates j sort before them (keys staged through LDS as 64-bit (key, index)
// words) and adds that partial rank into rank_acc (zeroed by lanms_zero_kernel); the page kernel then scatters order[rank] = i.
// (Round 1 ran one workgroup per i chunk over ALL j: 5 workgroups per 10 k-candidate page, 0.59 ms; the 2-D split is ~n/2048
// times more parallel.)
__global__ __launch_bounds__(256) void lanms_zero_kernel(const int32_t* __restrict__ counts, char* __restrict__ ws, long ws_stride, long acc_off) {
  const int pg = blockIdx.y;
  const int n = counts[pg] & 0x7fffffff;
  int32_t* acc = reinterpret_cast<int32_t*>(ws + (long)pg * ws_stride + acc_off);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc[i] = 0;
}
__global__ __launch_bounds__(256) void lanms_rank_x0_kernel(const float* __restrict__ cand, const int32_t* __restrict__ counts,
                                                             int max_cand, char* __restrict__ ws, long ws_stride, long acc_off) {
  const int pg = blockIdx.z, tid = threadIdx.x;
  const int n = counts[pg] & 0x7fffffff;
  const int i0 = blockIdx.x * 256 * RS_Q, j0 = blockIdx.y * RS_TILE;
  if (i0 >= n || j0 >= n) return;
  const float* cb = cand + (long)pg * max_cand * 9;
  int32_t* acc = reinterpret_cast<int32_t*>(ws + (long)pg * ws_stride + acc_off);
  __shared__ unsigned long long tile[RS_TILE];
  unsigned long long ki[RS_Q];
  int rank[RS_Q];
#pragma unroll
  for (int q = 0; q < RS_Q; ++q) {
    const int i = i0 + q * 256 + tid;
    ki[q] = i < n ? (((unsigned long long)sortable_f32(cb[(long)i * 9]) << 32) | (unsigned)i) : 0ull;
    rank[q] = 0;
  }
  const int jn = min(RS_TILE, n - j0);
  for (int j = tid; j < jn; j += 256) tile[j] = ((unsigned long long)sortable_f32(cb[(long)(j0 + j) * 9]) << 32) | (unsigned)(j0 + j);
  __syncthreads();
  for (int j = 0; j < jn; ++j) {
    const unsigned long long kj = tile[j];
#pragma unroll
    for (int q = 0; q < RS_Q; ++q) rank[q] += kj < ki[q] ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < RS_Q; ++q) {
    const int i = i0 + q * 256 + tid;
    if (i < n && rank[q]) atomicAdd(&acc[i], rank[q]);
  }
}

// in-workgroup stable rank sort of nm f64 keys (descending score = ascending -score), nm is a few thousand at most
__device__ __forceinline__ void block_rank_sort_neg_f64(int n, const double* __restrict__ score, int32_t* __restrict__ order,
                                                        unsigned long long* tile /* LDS [RS_TILE] */) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i0 = 0; i0 < n; i0 += nthr) {
    const int i = i0 + tid;
    const unsigned long long ki = i < n ? sortable_f64(-score[i]) : 0ull;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += RS_TILE) {
      const int jn = min(RS_TILE, n - j0);
      __syncthreads();
      for (int j = tid; j < jn; j += nthr) tile[j] = sortable_f64(-score[j0 + j]);
      __syncthreads();
      for (int j = 0; j < jn; ++j) {
        const unsigned long long kj = tile[j];
        rank += (kj < ki || (kj == ki && j0 + j < i)) ? 1 : 0;
      }
    }
    if (i < n) order[rank] = i;
  }
  __syncthreads();
}

struct MergeState {
  double p[8], w, sc;
  int valid;
};
__device__ __forceinline__ void st_load(MergeState& st, const double* c) {
  for (int k = 0; k < 8; ++k) st.p[k] = c[k];
  st.w = c[8];
  st.sc = c[9];
  st.valid = c[10] != 0.0;
}
__device__ __forceinline__ void st_store(const MergeState& st, double* c) {
  for (int k = 0; k < 8; ++k) c[k] = st.p[k];
  c[8] = st.w;
  c[9] = st.sc;
  c[10] = st.valid ? 1.0 : 0.0;
}
__device__ __forceinline__ bool st_same(const MergeState& st, const double* c) {
  if ((c[10] != 0.0) != (st.valid != 0)) return false;
  if (!st.valid) return true;
  bool same = __double_as_longlong(st.w) == __double_as_longlong(c[8]) && __double_as_longlong(st.sc) == __double_as_longlong(c[9]);
  for (int k = 0; k < 8; ++k) same = same && __double_as_longlong(st.p[k]) == __double_as_longlong(c[k]);
  return same;
}
// one candidate of the reference loop body (lanms.py:174-192); returns true if it started a new polygon
__device__ __forceinline__ bool merge_step(MergeState& st, const float* b, double thr, MergeState* closed) {
  double poly[8];
  for (int k = 0; k < 8; ++k) poly[k] = (double)b[k];
  const double sc = (double)b[8];
  if (st.valid && d_polygon_iou_q(poly, st.p) > thr) {
    double al[8];
    d_normalize_polygon(st.p, poly, al);
    const double tw = st.w + sc;
    for (int k = 0; k < 8; ++k) st.p[k] = (st.p[k] * st.w + al[k] * sc) / tw;
    st.w = tw;
    st.sc = st.sc > sc ? st.sc : sc;
    return false;
  }
  if (closed) *closed = st;
  for (int k = 0; k < 8; ++k) st.p[k] = poly[k];
  st.w = sc;
  st.sc = sc;
  st.valid = 1;
  return true;
}

// TT threads per page (workspace carries are laid out for LANMS_T >= TT); REGNMS: keep the in-kernel register NMS loop
template <int TT, bool REGNMS>
__global__ __launch_bounds__(TT) void east_lanms_kernel(const float* __restrict__ cand, const int32_t* __restrict__ counts,
                                                              int max_cand, double thr, float* __restrict__ boxes_out,
                                                              int32_t* __restrict__ nbox_out, char* __restrict__ ws, long ws_stride, long long* dbg,
                                                              long bits_off, int bitcap) {
  const int pg = blockIdx.x;
  const int tid = threadIdx.x, nthr = TT;
#define DBG_STAMP(k) do { if (dbg && tid == 0) dbg[pg * 8 + (k)] = wall_clock64(); } while (0)
  DBG_STAMP(0);
  const float* cb = cand + (long)pg * max_cand * 9;
  const int n = counts[pg] & 0x7fffffff;
  char* w = ws + (long)pg * ws_stride;
  double* mpoly = reinterpret_cast<double*>(w);
  double* mscore = mpoly + (long)max_cand * 8;
  double* spoly = mscore + max_cand;
  double* sscore = spoly + (long)max_cand * 8;
  double* carry_spec = sscore + max_cand;
  double* carry_cur = carry_spec + LANMS_T * CARRY_W;
  double* carry_nxt = carry_cur + LANMS_T * CARRY_W;
  int32_t* order = reinterpret_cast<int32_t*>(carry_nxt + LANMS_T * CARRY_W);
  int32_t* supp = order + max_cand;
  int32_t* flag = supp + max_cand;
  int32_t* sbreak = flag + max_cand;
  float* ob = boxes_out + (long)pg * max_cand * 9;
  __shared__ int nm_s, nk_s, any_changed;
  __shared__ int scan_s[TT / 64];
  __shared__ unsigned char ch_s[TT];

  int32_t* nm_hdr = reinterpret_cast<int32_t*>(w + bits_off);  // {merged polygons for the bit-matrix path, or -1}
  if (n == 0) {
    if (tid == 0) { nbox_out[pg] = 0; nm_hdr[0] = -1; }
    return;
  }
  // ---- `supp` holds every candidate's rank in the stable argsort by x0 (lanms.py:166-168), accumulated by lanms_rank_x0_kernel
  __shared__ unsigned long long rs_tile[RS_TILE];
  for (int i = tid; i < n; i += nthr) {
    order[supp[i]] = i;
    flag[i] = 0;
    sbreak[i] = 0;
  }
  __threadfence();
  __syncthreads();
  DBG_STAMP(1);
  // ---- phase 1: speculative segmented scan -------------------------------------------------------------
  int S = (n + 7) / 8;  // >= 8 candidates per segment
  if (S > nthr) S = nthr;
  const int L = (n + S - 1) / S;
  S = (n + L - 1) / L;
  const int seg0 = tid * L, seg1 = min(n, seg0 + L);
  const bool active = tid < S;
  // A. speculative scan from EMPTY
  if (active) {
    MergeState st = {};
    for (int s = seg0; s < seg1; ++s) {
      MergeState closed = {};
      if (merge_step(st, cb + (long)order[s] * 9, thr, &closed)) {
        sbreak[s] = 1;
        if (closed.valid) {  // polygon that ended at candidate s-1
          for (int k = 0; k < 8; ++k) spoly[(long)(s - 1) * 8 + k] = closed.p[k];
          sscore[s - 1] = closed.sc;
          flag[s - 1] = 1;
        }
      }
    }
    st_store(st, carry_spec + tid * CARRY_W);
    st_store(st, carry_cur + tid * CARRY_W);
  }
  if (tid == 0) any_changed = 0;
  __syncthreads();
  DBG_STAMP(2);
  // B. fixed point of the carries (thread 0's carry is already true: its incoming state IS empty)
  int in_dirty = active && tid > 0;  // my incoming carry has not been consumed yet
  for (int round = 0; round < S; ++round) {
    int changed = 0;
    if (in_dirty) {
      MergeState st = {};
      st_load(st, carry_cur + (tid - 1) * CARRY_W);
      bool resync = false;
      for (int s = seg0; s < seg1; ++s) {
        if (merge_step(st, cb + (long)order[s] * 9, thr, nullptr) && sbreak[s]) {
          resync = true;
          break;
        }
      }
      if (resync) st_load(st, carry_spec + tid * CARRY_W);
      changed = !st_same(st, carry_cur + tid * CARRY_W);
      if (changed) st_store(st, carry_nxt + tid * CARRY_W);
    }
    ch_s[tid] = (unsigned char)changed;
    if (changed) atomicOr(&any_changed, 1);
    __syncthreads();  // every thread has read its predecessor's carry_cur and published `changed`
    const int any = any_changed;
    if (changed)
      for (int k = 0; k < CARRY_W; ++k) carry_cur[tid * CARRY_W + k] = carry_nxt[tid * CARRY_W + k];
    in_dirty = active && tid > 0 && ch_s[tid - 1];
    __syncthreads();
    if (tid == 0) any_changed = 0;
    __syncthreads();
    if (!any) break;
  }
  DBG_STAMP(3);
  // C. replay the true prefix of every segment, emitting closed polygons into their slots
  if (active) {
    MergeState st = {};
    if (tid > 0) st_load(st, carry_cur + (tid - 1) * CARRY_W);
    bool resync = (tid == 0);  // segment 0's speculative scan is the true scan
    for (int s = seg0; s < seg1 && !resync; ++s) {
      MergeState closed = {};
      const bool brk = merge_step(st, cb + (long)order[s] * 9, thr, &closed);
      if (s > 0) {
        if (brk && closed.valid) {
          for (int k = 0; k < 8; ++k) spoly[(long)(s - 1) * 8 + k] = closed.p[k];
          sscore[s - 1] = closed.sc;
          flag[s - 1] = 1;
        } else {
          flag[s - 1] = 0;  // a speculative emission that the true scan does not make
        }
      }
      if (brk && sbreak[s]) resync = true;
    }
    if (tid == S - 1) {  // the last open polygon of the page ends at candidate n-1
      MergeState fin = st;
      if (resync) st_load(fin, carry_spec + tid * CARRY_W);
      for (int k = 0; k < 8; ++k) spoly[(long)(n - 1) * 8 + k] = fin.p[k];
      sscore[n - 1] = fin.sc;
      flag[n - 1] = 1;
    }
  }
  __syncthreads();
  DBG_STAMP(4);
  // D. ordered compaction of the flagged slots
  {
    const int per = (n + nthr - 1) / nthr;
    const int a0 = tid * per, a1 = min(n, a0 + per);
    int cnt = 0;
    for (int s = a0; s < a1; ++s) cnt += flag[s];
    const int lane = tid & 63, wv = tid >> 6;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o);
      if (lane >= o) inc += v;
    }
    if (lane == 63) scan_s[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wv; ++q) base += scan_s[q];
    int pos = base + inc - cnt;
    for (int s = a0; s < a1; ++s)
      if (flag[s]) {
        for (int k = 0; k < 8; ++k) mpoly[(long)pos * 8 + k] = spoly[(long)s * 8 + k];
        mscore[pos] = sscore[s];
        ++pos;
      }
    if (tid == nthr - 1) nm_s = base + inc;
  }
  __syncthreads();
  const int nm = nm_s;
  DBG_STAMP(5);
  // ---- phase 2: order = argsort(-score) (stable), greedy suppression (lanms.py:133-153)
  block_rank_sort_neg_f64(nm, mscore, order, rs_tile);
  if (bitcap > 0 && nm <= bitcap) {  // greedy suppression continues in lanms_iou_bits_kernel + lanms_greedy_bits_kernel
    if (tid == 0) nm_hdr[0] = nm;
    DBG_STAMP(6);
    if (dbg && tid == 0) dbg[pg * 8 + 7] = ((long long)n << 32) | (unsigned)nm;
    return;
  }
  if (tid == 0) nm_hdr[0] = -1;
  for (int i = tid; i < nm; i += nthr) supp[i] = 0;
  __syncthreads();
  int nk = 0;  // kept count (every thread tracks it; the owner of a kept polygon writes its row)
  constexpr int NQ = 4;
  if (REGNMS && nm <= NQ * TT) {
    // fast path: sorted position j = tid + q*1024 lives in thread tid's registers (polygon + suppressed flag); per
    // iteration the owner broadcasts polygon i through LDS (double-buffered: one barrier per iteration).
    __shared__ double bc_poly[2][8];
    __shared__ int bc_alive[2];
    double mp[NQ][8];
    int ms_idx[NQ];
    bool sup[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int j = tid + q * TT;
      sup[q] = true;
      ms_idx[q] = 0;
      if (j < nm) {
        ms_idx[q] = order[j];
        sup[q] = false;
        for (int k = 0; k < 8; ++k) mp[q][k] = mpoly[(long)ms_idx[q] * 8 + k];
      } else {
        for (int k = 0; k < 8; ++k) mp[q][k] = 0.0;
      }
    }
    for (int i = 0; i < nm; ++i) {
      const int buf = i & 1, own = i % TT, oq = i / TT;
      if (tid == own) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q == oq) {
            bc_alive[buf] = sup[q] ? 0 : 1;
            if (!sup[q]) {
              for (int k = 0; k < 8; ++k) bc_poly[buf][k] = mp[q][k];
              float* o = ob + (long)nk * 9;
              for (int k = 0; k < 8; ++k) o[k] = (float)mp[q][k];
              o[8] = (float)mscore[ms_idx[q]];
            }
          }
      }
      __syncthreads();
      if (!bc_alive[buf]) continue;
      ++nk;
      double a[8];
      for (int k = 0; k < 8; ++k) a[k] = bc_poly[buf][k];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j = tid + q * TT;
        if (j > i && j < nm && !sup[q] && d_polygon_iou_q(a, mp[q]) > thr) sup[q] = true;
      }
    }
  } else {
    for (int i = 0; i < nm; ++i) {
      const int idx = order[i];
      if (supp[idx]) continue;  // uniform: written before the barrier that ended an earlier iteration
      double a[8];
      for (int k = 0; k < 8; ++k) a[k] = mpoly[(long)idx * 8 + k];
      if (tid == 0) {
        float* o = ob + (long)nk * 9;
        for (int k = 0; k < 8; ++k) o[k] = (float)a[k];
        o[8] = (float)mscore[idx];
      }
      ++nk;
      for (int j = i + 1 + tid; j < nm; j += nthr) {
        const int idj = order[j];
        if (supp[idj]) continue;
        if (d_polygon_iou_q(a, mpoly + (long)idj * 8) > thr) supp[idj] = 1;
      }
      __syncthreads();
    }
  }
  if (tid == 0) nk_s = nk;
  if (tid == 0) nbox_out[pg] = nk_s;
  DBG_STAMP(6);
  if (dbg && tid == 0) dbg[pg * 8 + 7] = ((long long)n << 32) | (unsigned)nm;
}

// bit j of bits[i][w] (j = 32w + b > i, sorted positions): polygon_iou(poly_i, poly_j) > thr, the test of lanms.py:147-150 with
// the same argument order.  grid (row blocks of 8, pages), 256 threads: thread = (row, word) pairs in a strided loop.
__global__ __launch_bounds__(256) void lanms_iou_bits_kernel(char* __restrict__ ws, long ws_stride, int max_cand, double thr, long bits_off,
                                                              int bitcap) {
  const int pg = blockIdx.y;
  char* w = ws + (long)pg * ws_stride;
  const int nm = reinterpret_cast<const int32_t*>(w + bits_off)[0];
  if (nm <= 0) return;
  const int W = (nm + 31) >> 5, capw = bitcap >> 5;
  const double* mpoly = reinterpret_cast<const double*>(w);
  const int32_t* order = reinterpret_cast<const int32_t*>(w + ((long)max_cand * (2 * (64 + 8)) + (long)3 * LANMS_T * CARRY_W * 8));
  uint32_t* bits = reinterpret_cast<uint32_t*>(w + bits_off + 64);
  const long total = (long)nm * W;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx / W), wd = (int)(idx - (long)i * W);
    uint32_t v = 0u;
    if (32 * wd + 31 > i) {
      double a[8];
      const double* pa = mpoly + (long)order[i] * 8;
      for (int k = 0; k < 8; ++k) a[k] = pa[k];
      for (int b = 0; b < 32; ++b) {
        const int j = 32 * wd + b;
        if (j > i && j < nm && d_polygon_iou_q(a, mpoly + (long)order[j] * 8) > thr) v |= 1u << b;
      }
    }
    bits[(long)i * capw + wd] = v;
  }
}

// the greedy pass of standard_nms (lanms.py:141-152) on the bit matrix: one wave per page, lane l owns the suppression words
// l, l + 64, ...; a kept polygon ORs its row into them.  Output rows in kept order, like the reference's `keep` list.
__global__ __launch_bounds__(64) void lanms_greedy_bits_kernel(char* __restrict__ ws, long ws_stride, int max_cand, long bits_off, int bitcap,
                                                                float* __restrict__ boxes_out, int32_t* __restrict__ nbox_out) {
  const int pg = blockIdx.x, lane = threadIdx.x;
  char* w = ws + (long)pg * ws_stride;
  const int nm = reinterpret_cast<const int32_t*>(w + bits_off)[0];
  if (nm <= 0) return;  // n == 0 or the in-kernel path: nbox_out is already written
  const int W = (nm + 31) >> 5, capw = bitcap >> 5;
  const double* mpoly = reinterpret_cast<const double*>(w);
  const double* mscore = mpoly + (long)max_cand * 8;
  const int32_t* order = reinterpret_cast<const int32_t*>(w + ((long)max_cand * (2 * (64 + 8)) + (long)3 * LANMS_T * CARRY_W * 8));
  const uint32_t* bits = reinterpret_cast<const uint32_t*>(w + bits_off + 64);
  float* ob = boxes_out + (long)pg * max_cand * 9;
  constexpr int KW = NMS_BITW / 64;
  uint32_t sup[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) sup[k] = 0u;
  int nk = 0;
  for (int i = 0; i < nm; ++i) {
    const int wi = i >> 5, owner = wi & 63, slot = wi >> 6;
    uint32_t mine = 0u;
#pragma unroll
    for (int k = 0; k < KW; ++k) mine = (k == slot) ? sup[k] : mine;
    const uint32_t word = __shfl(mine, owner);
    if ((word >> (i & 31)) & 1u) continue;  // suppressed (wave-uniform)
    const int idx = order[i];
    if (lane < 9) ob[(long)nk * 9 + lane] = lane < 8 ? (float)mpoly[(long)idx * 8 + lane] : (float)mscore[idx];
    ++nk;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int wd = lane + 64 * k;
      if (wd < W) sup[k] |= bits[(long)i * capw + wd];
    }
  }
  if (lane == 0) nbox_out[pg] = nk;
}

extern "C" int msocr_east_lanms(const float* cand, const int32_t* counts, int N, int max_cand, double iou_thr, float* boxes_out,
                                int32_t* nbox_out, void* workspace, void* stream) {
  if (!cand || !counts || !boxes_out || !nbox_out || !workspace || N <= 0 || max_cand <= 0) return MSOCR_E_ARG;
  if ((uintptr_t)workspace & 7) return MSOCR_E_ARG;
  const long stride = (lanms_ws_per_page(max_cand) + 63) / 64 * 64;
  {
    const char* e = getenv("MSOCR_LANMS_SHORTCUT");
    const int v = (e && e[0] == '0') ? 0 : 1;
    static int cur = -1;
    if (v != cur) {
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_lanms_shortcut), &v, sizeof(int)) != hipSuccess) return MSOCR_E_LAUNCH;
      cur = v;
    }
    const char* f = getenv("MSOCR_LANMS_FASTCLIP");  // 0: always the general (scratch-buffer) polygon clip (diagnostic)
    const int vf = (f && f[0] == '0') ? 0 : 1;
    static int curf = -1;
    if (vf != curf) {
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_lanms_fastclip), &vf, sizeof(int)) != hipSuccess) return MSOCR_E_LAUNCH;
      curf = vf;
    }
  }
  {
    // rank accumulator = the `supp` array of the page workspace (order | supp | flag | sbreak, 4 bytes each per candidate)
    const long acc_off = ((long)max_cand * (2 * (64 + 8)) + (long)3 * LANMS_T * CARRY_W * 8) + (long)max_cand * 4;
    const int ichunks = (max_cand + 256 * RS_Q - 1) / (256 * RS_Q), jchunks = (max_cand + RS_TILE - 1) / RS_TILE;
    if (jchunks > 65535 || N > 65535) return MSOCR_E_ARG;
    MSOCR_LAUNCH(lanms_zero_kernel, dim3(min(ichunks * RS_Q, 64), N), dim3(256), 0, (hipStream_t)stream, counts, (char*)workspace, stride, acc_off);
    if (hipGetLastError() != hipSuccess) return MSOCR_E_LAUNCH;
    MSOCR_LAUNCH(lanms_rank_x0_kernel, dim3(ichunks, jchunks, N), dim3(256), 0, (hipStream_t)stream, cand, counts, max_cand, (char*)workspace,
                 stride, acc_off);
    if (hipGetLastError() != hipSuccess) return MSOCR_E_LAUNCH;
  }
  long long* dbg = nullptr;
  const bool want_dbg = getenv("MSOCR_LANMS_DEBUG") != nullptr;  // diagnostic only: synchronises and prints phase times
  if (want_dbg && hipMalloc(&dbg, sizeof(long long) * 8 * N) != hipSuccess) dbg = nullptr;
  const int bits_on = getenv("MSOCR_LANMS_BITS") ? atoi(getenv("MSOCR_LANMS_BITS")) : 1;  // 0: greedy pass inside the page kernel (diagnostic)
  const long bits_off = lanms_bits_off(max_cand);
  const int bitcap = bits_on ? nms_bitcap(max_cand) : 0;
  // page kernel geometry: 768 threads = 3 waves per SIMD = 168 VGPRs per lane, enough for the register-resident polygon clip
  // (at 1024 threads = 128 VGPRs it spills and the carry fix-up rounds get slower).  Measured on a 13 k-candidate page, phase 1 +
  // score sort: 1024 threads with the old scratch-buffer clip 2.6 ms; 1024 / 768 / 512 / 256 threads with the register clip
  // 2.3 / 1.7 / 2.1 / 2.7 ms.  MSOCR_LANMS_T=1024|1025|768|512|256 selects the others (diagnostics).
  const int lt = getenv("MSOCR_LANMS_T") ? atoi(getenv("MSOCR_LANMS_T")) : 768;
  if (lt == 1024)
    MSOCR_LAUNCH((east_lanms_kernel<1024, true>), dim3(N), dim3(1024), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out,
                 nbox_out, (char*)workspace, stride, dbg, bits_off, bitcap);
  else if (lt == 1025)
    MSOCR_LAUNCH((east_lanms_kernel<1024, false>), dim3(N), dim3(1024), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out,
                 nbox_out, (char*)workspace, stride, dbg, bits_off, bitcap);
  else if (lt == 768)
    MSOCR_LAUNCH((east_lanms_kernel<768, false>), dim3(N), dim3(768), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out,
                 nbox_out, (char*)workspace, stride, dbg, bits_off, bitcap);
  else if (lt == 256)
    MSOCR_LAUNCH((east_lanms_kernel<256, false>), dim3(N), dim3(256), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out,
                 nbox_out, (char*)workspace, stride, dbg, bits_off, bitcap);
  else
    MSOCR_LAUNCH((east_lanms_kernel<512, false>), dim3(N), dim3(512), 0, (hipStream_t)stream, cand, counts, max_cand, iou_thr, boxes_out,
                 nbox_out, (char*)workspace, stride, dbg, bits_off, bitcap);
  int rc = LAUNCH_OK();
  if (rc == MSOCR_OK && bitcap > 0) {
    MSOCR_LAUNCH(lanms_iou_bits_kernel, dim3(256, N), dim3(256), 0, (hipStream_t)stream, (char*)workspace, stride, max_cand, iou_thr, bits_off,
                 bitcap);
    rc = LAUNCH_OK();
    if (rc == MSOCR_OK) {
      MSOCR_LAUNCH(lanms_greedy_bits_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, (char*)workspace, stride, max_cand, bits_off, bitcap,
                   boxes_out, nbox_out);
      rc = LAUNCH_OK();
    }
  }
  if (dbg) {
    long long* h = (long long*)malloc(sizeof(long long) * 8 * N);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipMemcpy(h, dbg, sizeof(long long) * 8 * N, hipMemcpyDeviceToHost);
    for (int p = 0; p < N && p < 2; ++p) {
      const long long* t = h + p * 8;
      fprintf(stderr, "[lanms dbg] page %d n=%lld nm=%lld  us: sort %.0f specA %.0f fixB %.0f replayC %.0f compactD %.0f nms %.0f\n", p,
              t[7] >> 32, t[7] & 0xffffffff, (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0, (t[4] - t[3]) / 100.0,
              (t[5] - t[4]) / 100.0, (t[6] - t[5]) / 100.0);
    }
    free(h);
    (void)hipFree(dbg);
  }
  return rc;
}
