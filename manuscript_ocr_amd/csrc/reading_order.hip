// reading_order.hip — the Pipeline glue between detector and recogniser ON THE DEVICE, one workgroup per page:
// final detector boxes -> word AABBs -> reading order -> crop descriptors for msocr_crop_resize_pad.
//
//   Pipeline.predict sort + crop section              _pipeline.py:100-137  (np.array(polygon, int32) truncation, min_text_size)
//   Pipeline._extract_word_image                      _pipeline.py:204-221  (clamped AABB)
//   resolve_intersections                             detectors/_east/utils.py:500-547
//   sort_boxes_reading_order                          detectors/_east/utils.py:550-607
//   sort_boxes_reading_order_with_resolutions         detectors/_east/utils.py:610-644 (dict(zip(shrunk, boxes)): later duplicate wins)
//   first-equal-word re-match                         _pipeline.py:113-121
//   ResizeAndPadA size arithmetic                     recognizers/_trba/data/transforms.py:91-95,114-117 (Python round = rint)
//
// Everything is integer or f64 arithmetic in the reference's order, so the result is bit-identical to the host helper
// msocr_reading_order_host (host_glue.hip) + ops.crop_descriptors, which stay as the fallback for pages this kernel flags
// (ncrop = -1: more boxes than the capacity, more than RO_MAXLINES text lines, or more intersecting pairs than the pair buffer).
//
// resolve_intersections is a SEQUENTIAL sweep over all pairs (i < j): an intersecting pair shrinks both boxes at once, which
// changes what later pairs of the same sweep see.  It is parallelised exactly: boxes only ever shrink during a sweep, so the
// pairs that intersect at the START of a sweep are a superset of the pairs the sequential sweep will find; those candidates are
// found by all threads, ordered lexicographically by a prefix sum, and replayed in order by one thread with the live boxes.
// Compile with -ffp-contract=off (the Makefile does): int(x1 - (x1 - x0) * 0.1) must round like Python.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace {
constexpr int RO_T = 1024;         // threads per page
constexpr int RO_CAP = 16384;      // boxes per page
constexpr int RO_MAXLINES = 4096;  // text lines per page (line state lives in LDS)

struct Box4 { int x0, y0, x1, y1; };

__host__ __device__ inline int ro_cap(int max_cand) { return max_cand < RO_CAP ? max_cand : RO_CAP; }
__host__ __device__ inline long ro_pair_cap(int cap) { return 8L * cap + 4096; }
// per-page workspace in 4-byte words: ob[4c] sb[4c] cnt[c+4] pairs[2P] sorted[c] lineof[c] seq[c] emitted[c] keepf[c]
__host__ __device__ inline long ro_ws_words(int cap) { return (4L * cap + 4L * cap + (cap + 4) + 2 * ro_pair_cap(cap) + 5L * cap + 3) / 4 * 4; }

__device__ __forceinline__ bool ro_hit(const Box4& a, const Box4& b) {  // utils.py:515-523
  return !(a.x1 <= b.x0 || b.x1 <= a.x0 || a.y1 <= b.y0 || b.y1 <= a.y0);
}
__device__ __forceinline__ int ro_shrink(int lo, int hi) {  // int(hi - (hi - lo) * 0.1): f64, truncation toward zero
  return (int)((double)hi - (double)(hi - lo) * 0.1);
}
__device__ __forceinline__ bool ro_same(const Box4& a, const Box4& b) { return a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1; }

// in-place exclusive prefix sum of arr[0..n) by the whole workgroup (contiguous chunk per thread); returns the total
__device__ int ro_block_scan(int* arr, int n, int* wave_tot /* LDS [RO_T/64] */, int* total_s /* LDS */) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + RO_T - 1) / RO_T;
  const int a0 = min(n, tid * per), a1 = min(n, a0 + per);
  int sum = 0;
  for (int k = a0; k < a1; ++k) sum += arr[k];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < wv; ++q) base += wave_tot[q];
  if (tid == RO_T - 1) *total_s = base + inc;
  int run = base + inc - sum;
  for (int k = a0; k < a1; ++k) {
    const int v = arr[k];
    arr[k] = run;
    run += v;
  }
  __threadfence();
  __syncthreads();
  return *total_s;
}

__global__ __launch_bounds__(RO_T) void reading_order_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ nbox,
                                                              int max_cand, int cap, int page_h, int page_w, int min_text,
                                                              int img_h, int img_w, double y_tol_ratio, double x_gap_ratio,
                                                              int page_base, int32_t* __restrict__ order_out,
                                                              int32_t* __restrict__ keep_out, int32_t* __restrict__ desc_out,
                                                              int32_t* __restrict__ ncrop_out, int32_t* __restrict__ ws) {
  const int pg = blockIdx.x, tid = threadIdx.x;
  const int n = nbox[pg];
  if (n < 0 || n > cap) {
    if (tid == 0) ncrop_out[pg] = -1;
    return;
  }
  if (n == 0) {
    if (tid == 0) ncrop_out[pg] = 0;
    return;
  }
  int32_t* w = ws + (long)pg * ro_ws_words(cap);
  Box4* ob = reinterpret_cast<Box4*>(w);
  Box4* sb = ob + cap;
  int* cnt = reinterpret_cast<int*>(sb + cap);
  const long P = ro_pair_cap(cap);
  int* pairs = cnt + cap + 4;
  int* sorted = pairs + 2 * P;
  int* lineof = sorted + cap;
  int* seq = lineof + cap;
  int* emitted = seq + cap;
  int* keepf = emitted + cap;
  int32_t* oo = order_out + (long)pg * max_cand;
  int32_t* ko = keep_out + (long)pg * max_cand;
  int32_t* dout = desc_out + (long)pg * max_cand * 8;

  __shared__ double lsum[RO_MAXLINES];
  __shared__ int lcnt[RO_MAXLINES], lmaxx[RO_MAXLINES], lstart[RO_MAXLINES];
  __shared__ int wave_tot[RO_T / 64];
  __shared__ int total_s, nlines_s, fail_s;
  __shared__ long long hsum_s;

  // ---- word AABBs: np.array(polygon, dtype=np.int32) truncates toward zero, then min / max over the 4 points (_pipeline.py:106-109)
  const float* ib = boxes + (long)pg * max_cand * 9;
  for (int i = tid; i < n; i += RO_T) {
    const float* q = ib + 9 * (long)i;
    int xs[4], ys[4];
    for (int k = 0; k < 4; ++k) { xs[k] = (int)q[2 * k]; ys[k] = (int)q[2 * k + 1]; }
    Box4 b;
    b.x0 = min(min(xs[0], xs[1]), min(xs[2], xs[3])); b.x1 = max(max(xs[0], xs[1]), max(xs[2], xs[3]));
    b.y0 = min(min(ys[0], ys[1]), min(ys[2], ys[3])); b.y1 = max(max(ys[0], ys[1]), max(ys[2], ys[3]));
    ob[i] = b;
    sb[i] = b;
  }
  if (tid == 0) { fail_s = 0; hsum_s = 0; }
  __threadfence();
  __syncthreads();

  // ---- resolve_intersections: up to 50 sweeps (utils.py:507-546)
  for (int sweep = 0; sweep < 50; ++sweep) {
    for (int i = tid; i < n; i += RO_T) {
      const Box4 a = sb[i];
      int c = 0;
      for (int j = i + 1; j < n; ++j) c += ro_hit(a, sb[j]) ? 1 : 0;
      cnt[i] = c;
    }
    __threadfence();
    __syncthreads();
    const int total = ro_block_scan(cnt, n, wave_tot, &total_s);
    if (total == 0) break;
    if (total > P) {
      if (tid == 0) fail_s = 1;
      break;
    }
    for (int i = tid; i < n; i += RO_T) {
      const Box4 a = sb[i];
      int o = cnt[i];
      for (int j = i + 1; j < n; ++j)
        if (ro_hit(a, sb[j])) { pairs[2 * (long)o] = i; pairs[2 * (long)o + 1] = j; ++o; }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {  // the sequential sweep, restricted to the candidate pairs, with the live boxes
      for (int p = 0; p < total; ++p) {
        const int i = pairs[2 * (long)p], j = pairs[2 * (long)p + 1];
        Box4 a = sb[i], b = sb[j];
        if (!ro_hit(a, b)) continue;
        a.x1 = ro_shrink(a.x0, a.x1); a.y1 = ro_shrink(a.y0, a.y1);
        b.x1 = ro_shrink(b.x0, b.x1); b.y1 = ro_shrink(b.y0, b.y1);
        sb[i] = a; sb[j] = b;
      }
    }
    __threadfence();
    __syncthreads();
  }
  __syncthreads();
  if (fail_s) {
    if (tid == 0) ncrop_out[pg] = -1;
    return;
  }

  // ---- sort_boxes_reading_order on the shrunk boxes (utils.py:550-607)
  {
    long long h = 0;
    for (int i = tid; i < n; i += RO_T) h += (long long)(sb[i].y1 - sb[i].y0);
    for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o);
    if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&hsum_s), (unsigned long long)h);
  }
  // stable sort by cy = (y0 + y1) / 2: rank on the integer 2*cy, ties by index
  for (int i = tid; i < n; i += RO_T) {
    const int ki = sb[i].y0 + sb[i].y1;
    int r = 0;
    for (int j = 0; j < n; ++j) {
      const int kj = sb[j].y0 + sb[j].y1;
      r += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
    }
    sorted[r] = i;
    seq[i] = r;
  }
  __threadfence();
  __syncthreads();
  const double avg_h = (double)hsum_s / (double)n;  // np.mean of ints: exact sum, one division
  const double tol = avg_h * y_tol_ratio, gap = avg_h * x_gap_ratio;
  if (tid < 64) {  // one wave walks the cy-sorted boxes; its lanes test 64 lines at a time, the FIRST matching line wins
    const int lane = tid;
    int L = 0;
    bool over = false;
    for (int s = 0; s < n && !over; ++s) {
      const int i = sorted[s];
      const Box4 b = sb[i];
      const double c = (double)(b.y0 + b.y1) / 2.0;
      int home = -1;
      for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        bool ok = false;
        if (l < L) {
          const double line_cy = lsum[l] / (double)lcnt[l];
          ok = fabs(c - line_cy) <= tol && (double)(b.x0 - lmaxx[l]) <= gap;
        }
        const unsigned long long bal = __ballot(ok);
        if (bal) { home = l0 + (int)__ffsll((long long)bal) - 1; break; }
      }
      if (home < 0) {
        if (L >= RO_MAXLINES) { over = true; break; }
        if (lane == 0) { lsum[L] = c; lcnt[L] = 1; lmaxx[L] = b.x1; lineof[i] = L; }
        ++L;
      } else if (lane == 0) {
        lsum[home] += c; lcnt[home] += 1; lmaxx[home] = max(lmaxx[home], b.x1); lineof[i] = home;
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);  // lane 0's LDS stores stay ahead of the next iteration's reads (in-order per wave)
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { nlines_s = L; if (over) fail_s = 1; }
  }
  __threadfence();
  __syncthreads();
  if (fail_s) {
    if (tid == 0) ncrop_out[pg] = -1;
    return;
  }
  const int L = nlines_s;
  // lines sorted by their mean cy (stable: creation order breaks ties), then the first output position of every line
  for (int l = tid; l < L; l += RO_T) {
    const double m = lsum[l] / (double)lcnt[l];
    int start = 0;
    for (int k = 0; k < L; ++k) {
      const double mk = lsum[k] / (double)lcnt[k];
      if (mk < m || (mk == m && k < l)) start += lcnt[k];
    }
    lstart[l] = start;
  }
  __syncthreads();
  // inside a line: stable sort by x0 (insertion order = cy-sorted order breaks ties)
  for (int i = tid; i < n; i += RO_T) {
    const int li = lineof[i], xi = sb[i].x0, si = seq[i];
    int r = 0;
    for (int m = 0; m < n; ++m)
      if (lineof[m] == li) {
        const int xm = sb[m].x0;
        r += (xm < xi || (xm == xi && seq[m] < si)) ? 1 : 0;
      }
    emitted[lstart[li] + r] = i;
  }
  __threadfence();
  __syncthreads();
  // shrunk box -> original (dict: the LAST box with an equal shrunk box, utils.py:639) -> the FIRST word with an equal
  // original box (_pipeline.py:113-121); size filter, clamped crop window, ResizeAndPadA's size arithmetic
  for (int pos = tid; pos < n; pos += RO_T) {
    const int i = emitted[pos];
    const Box4 si = sb[i];
    int back = i;
    for (int j = n - 1; j > i; --j)
      if (ro_same(sb[j], si)) { back = j; break; }
    const Box4 o = ob[back];
    int wi = back;
    for (int j = 0; j < back; ++j)
      if (ro_same(ob[j], o)) { wi = j; break; }
    oo[pos] = wi;
    const Box4 bx = ob[wi];
    int keep = 0;
    int a = 0, b = 0, c = 0, d = 0;
    if ((bx.x1 - bx.x0) >= min_text && (bx.y1 - bx.y0) >= min_text) {
      a = max(0, bx.x0); b = max(0, bx.y0);
      c = min(page_w, bx.x1); d = min(page_h, bx.y1);
      if (c < 0) c = max(page_w + c, 0);  // Python slice semantics of image[y1:y2, x1:x2] with a negative stop
      if (d < 0) d = max(page_h + d, 0);
      keep = (c > a && d > b) ? 1 : 0;
    }
    keepf[pos] = keep;
    ko[pos] = keep;
    if (keep) {  // staged at the word's position, compacted below
      const int h = d - b, wd = c - a;
      const double scale = fmin((double)img_h / (double)max(h, 1), (double)img_w / (double)max(wd, 1));
      const int nw = max(1, (int)rint((double)wd * scale)), nh = max(1, (int)rint((double)h * scale));
      const int yy = max(0, min((img_h - nh) / 2, img_h - nh));  // floor division of a non-negative numerator when nh <= img_h
      int32_t* t = dout + 8 * (long)pos;
      t[0] = page_base + pg; t[1] = a; t[2] = b; t[3] = c; t[4] = d; t[5] = nw; t[6] = nh; t[7] = yy;
    }
  }
  __threadfence();
  __syncthreads();
  const int nk = ro_block_scan(keepf, n, wave_tot, &total_s);  // keepf[pos] = index of the crop among the page's crops
  // compaction through a staging buffer (pairs[] holds >= 16 * cap words and is free again)
  for (int pos = tid; pos < n; pos += RO_T)
    if (ko[pos])
      for (int e = 0; e < 8; ++e) pairs[8 * (long)keepf[pos] + e] = dout[8 * (long)pos + e];
  __threadfence();
  __syncthreads();
  for (long k = tid; k < 8L * nk; k += RO_T) dout[k] = pairs[k];
  if (tid == 0) ncrop_out[pg] = nk;
}
}  // namespace

extern "C" int64_t msocr_reading_order_workspace_bytes(int N, int max_cand) {
  return (N > 0 && max_cand > 0) ? (int64_t)N * ro_ws_words(ro_cap(max_cand)) * 4 : 0;
}

extern "C" int msocr_reading_order_crops(const float* boxes, const int32_t* nbox, int N, int max_cand, int page_h, int page_w,
                                         int min_text_size, int img_h, int img_w, double y_tol_ratio, double x_gap_ratio,
                                         int page_base, int32_t* order_out, int32_t* keep_out, int32_t* desc_out,
                                         int32_t* ncrop_out, void* workspace, void* stream) {
  if (!boxes || !nbox || !order_out || !keep_out || !desc_out || !ncrop_out || !workspace) return MSOCR_E_ARG;
  if (N <= 0 || max_cand <= 0 || page_h <= 0 || page_w <= 0 || img_h <= 0 || img_w <= 0 || page_base < 0) return MSOCR_E_ARG;
  if ((uintptr_t)workspace & 15) return MSOCR_E_ARG;
  MSOCR_LAUNCH(reading_order_kernel, dim3(N), dim3(RO_T), 0, (hipStream_t)stream, boxes, nbox, max_cand, ro_cap(max_cand), page_h,
               page_w, min_text_size, img_h, img_w, y_tol_ratio, x_gap_ratio, page_base, order_out, keep_out, desc_out, ncrop_out,
               (int32_t*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}
