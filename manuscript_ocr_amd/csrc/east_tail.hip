// east_tail.hip — the box filters that follow the NMS in EAST.predict, on the device (one workgroup per page), so that
// only the final boxes cross PCIe:
//   expand_boxes                      detectors/_east/utils.py:384-422
//   EAST._scale_boxes_to_original     detectors/_east/infer.py:134-147
//   EAST._remove_fully_contained_boxes (+ _polygon_area_batch, _is_quad_inside = cv2.pointPolygonTest >= 0)  infer.py:174-214
//   EAST._remove_area_anomalies       infer.py:216-233
//   EAST._convert_to_axis_aligned     infer.py:149-172
// The reference evaluates these with NumPy in f32; every operation here is the same f32 (or, inside the point-in-polygon
// test, f64) operation in the same order, including NumPy's pairwise summation in np.mean / np.std, so the result is
// bit-identical to the host implementation (detectors/_east/post.py), which stays as the path for pages of unequal size.
// The arithmetic lives in __host__ __device__ functions: msocr_east_box_tail_host runs the identical code on the CPU
// (used by the CPU test-suite to pin it against post.py without a GPU).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "msocr.h"

#define MSOCR_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define HD __host__ __device__ __forceinline__

namespace {

struct TailParams {
  float kx_m1, ky_m1;   // f32(1 + expand) - 1.0f, as (k - 1.0) of utils.py:420
  int expand;           // expand_w != 0 || expand_h != 0
  float sx, sy;         // f32(orig_w / target_w), f32(orig_h / target_h)
  int axis_aligned, anomalies, min_count;
  double sigma;
};

// ---- expand_boxes + scale: q[9] -> o[9] -----------------------------------------------------------------------------
HD void expand_scale(const float* q, const TailParams& p, float* o) {
  float x[4], y[4];
  for (int k = 0; k < 4; ++k) { x[k] = q[2 * k]; y[k] = q[2 * k + 1]; }
  float mx[4], my[4];
  if (p.expand) {
    float area = 0.f;  // np.sum over 4 elements: sequential
    for (int k = 0; k < 4; ++k) area = area + (x[k] * y[(k + 1) & 3] - x[(k + 1) & 3] * y[k]);
    const float sign = area < 0.f ? -1.f : 1.f;  // np.sign, 0 -> 1
    for (int k = 0; k < 4; ++k) {
      const int pk = (k + 3) & 3, nk = (k + 1) & 3;
      const float eix = x[k] - x[pk], eiy = y[k] - y[pk];
      const float eox = x[nk] - x[k], eoy = y[nk] - y[k];
      const float l_in = sqrtf(eix * eix + eiy * eiy);
      const float l_out = sqrtf(eox * eox + eoy * eoy);
      const float di = l_in + 1e-6f, dq = l_out + 1e-6f;
      const float nix = (sign * eiy) / di, niy = (sign * (-eix)) / di;
      const float nox = (sign * eoy) / dq, noy = (sign * (-eox)) / dq;
      float bx = nix + nox, by = niy + noy;
      const float nrm = sqrtf(bx * bx + by * by);
      if (nrm > 0.f) { bx = bx / nrm; by = by / nrm; } else { bx = 0.f; by = 0.f; }
      const float reach = fminf(l_in, l_out);
      mx[k] = x[k] + (p.kx_m1 * reach) * bx;
      my[k] = y[k] + (p.ky_m1 * reach) * by;
    }
  } else {
    for (int k = 0; k < 4; ++k) { mx[k] = x[k]; my[k] = y[k]; }
  }
  for (int k = 0; k < 4; ++k) { o[2 * k] = mx[k] * p.sx; o[2 * k + 1] = my[k] * p.sy; }
  o[8] = q[8];
}

HD float quad_area(const float* q) {  // 0.5 * |sum(x_k*y_{k+1} - y_k*x_{k+1})|, f32
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s = s + (q[2 * k] * q[2 * ((k + 1) & 3) + 1] - q[2 * k + 1] * q[2 * ((k + 1) & 3)]);
  return 0.5f * fabsf(s);
}

HD void quad_bbox(const float* q, float* b) {  // x0, x1, y0, y1
  b[0] = fminf(fminf(q[0], q[2]), fminf(q[4], q[6]));
  b[1] = fmaxf(fmaxf(q[0], q[2]), fmaxf(q[4], q[6]));
  b[2] = fminf(fminf(q[1], q[3]), fminf(q[5], q[7]));
  b[3] = fmaxf(fmaxf(q[1], q[3]), fmaxf(q[5], q[7]));
}

// cv2.pointPolygonTest(outer, (px, py), False) >= 0 : OpenCV's even-odd crossing rule, differences in f32, products in f64
HD bool vertex_not_outside(float px, float py, const float* outer) {
  int counter = 0;
  bool on_edge = false;
  for (int i = 0; i < 4; ++i) {
    const int i0 = (i + 3) & 3;
    const float v0x = outer[2 * i0], v0y = outer[2 * i0 + 1], vx = outer[2 * i], vy = outer[2 * i + 1];
    const bool skip = (v0y <= py && vy <= py) || (v0y > py && vy > py) || (v0x < px && vx < px);
    const bool hit = skip && (py == vy) && ((px == vx) || ((py == v0y) && ((v0x <= px && px <= vx) || (vx <= px && px <= v0x))));
    double dist = (double)(py - v0y) * (double)(vx - v0x) - (double)(px - v0x) * (double)(vy - v0y);
    const bool zero = !skip && dist == 0.0;
    const bool live = !on_edge;
    on_edge = on_edge || ((hit || zero) && live);
    if (vy < v0y) dist = -dist;
    counter += (!skip && live && !zero && dist > 0.0) ? 1 : 0;
  }
  return on_edge || (counter & 1);
}

// "all four vertices of quad i are inside or on quad j, and j is not smaller": infer.py:203-211
HD bool quad_contained(const float* qi, const float* bi, float ai, const float* qj, const float* bj, float aj) {
  if (!(bi[0] >= bj[0] && bi[1] <= bj[1] && bi[2] >= bj[2] && bi[3] <= bj[3])) return false;  // a vertex outside j's bbox is outside j
  if (!((aj + 1e-6f) >= ai)) return false;
  for (int k = 0; k < 4; ++k)
    if (!vertex_not_outside(qi[2 * k], qi[2 * k + 1], qj)) return false;
  return true;
}

// NumPy's float32 add.reduce (pairwise summation: 8 interleaved partial sums per block of <= 128 elements)
template <typename F>
HD float np_pairwise_sum(F elem, int lo, int n) {
  // iterative form of the recursion (n/2 rounded down to a multiple of 8): blocks visited left to right with a small stack
  struct Frame { int lo, n; int state; float left; };
  Frame st[24];
  int sp = 0;
  st[0] = Frame{lo, n, 0, 0.f};
  float ret = 0.f;
  while (sp >= 0) {
    Frame& f = st[sp];
    if (f.state == 0) {
      if (f.n < 8) {
        float r = 0.f;
        for (int i = 0; i < f.n; ++i) r = r + elem(f.lo + i);
        ret = r; --sp;
      } else if (f.n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = elem(f.lo + j);
        int i = 8;
        for (; i < f.n - (f.n % 8); i += 8)
          for (int j = 0; j < 8; ++j) r[j] = r[j] + elem(f.lo + i + j);
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < f.n; ++i) res = res + elem(f.lo + i);
        ret = res; --sp;
      } else {
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        f.state = 1;
        st[sp + 1] = Frame{f.lo, n2, 0, 0.f};
        ++sp;
      }
    } else if (f.state == 1) {
      f.left = ret;
      int n2 = f.n / 2;
      n2 -= n2 % 8;
      f.state = 2;
      st[sp + 1] = Frame{f.lo + n2, f.n - n2, 0, 0.f};
      ++sp;
    } else {
      ret = f.left + ret; --sp;
    }
  }
  return ret;
}

// threshold of _remove_area_anomalies: returns false when the filter does not apply (std == 0)
template <typename F>
HD bool anomaly_threshold(F area_of, int n, double sigma, float* thr) {
  const float mean = np_pairwise_sum(area_of, 0, n) / (float)n;
  auto sq = [&](int i) { const float d = area_of(i) - mean; return d * d; };
  const float var = np_pairwise_sum(sq, 0, n) / (float)n;
  const float sd = sqrtf(var);
  if ((double)sd == 0.0) return false;
  *thr = (float)((double)mean + sigma * (double)sd);  // Python float arithmetic, compared against the f32 areas as f32
  return true;
}

HD void axis_aligned(const float* q, float* o) {
  float b[4];
  quad_bbox(q, b);
  o[0] = b[0]; o[1] = b[2]; o[2] = b[1]; o[3] = b[2]; o[4] = b[1]; o[5] = b[3]; o[6] = b[0]; o[7] = b[3];
  o[8] = q[8];
}

TailParams make_params(double expand_w, double expand_h, double scale_x, double scale_y, int axis_al, int anomalies, double sigma,
                       int min_count) {
  TailParams p;
  p.expand = (expand_w != 0.0 || expand_h != 0.0) ? 1 : 0;
  p.kx_m1 = (float)(1.0 + expand_w) - 1.0f;
  p.ky_m1 = (float)(1.0 + expand_h) - 1.0f;
  p.sx = (float)scale_x; p.sy = (float)scale_y;
  p.axis_aligned = axis_al; p.anomalies = anomalies; p.min_count = min_count; p.sigma = sigma;
  return p;
}

}  // namespace

// ---- host twin (same arithmetic, sequential) ------------------------------------------------------------------------
extern "C" int msocr_east_box_tail_host(const float* quads_host, int M, double expand_w, double expand_h, double scale_x,
                                        double scale_y, int axis_aligned_output, int remove_anomalies, double sigma, int min_count,
                                        float* out_host, int32_t* n_out_host) {
  if (M < 0 || (M > 0 && (!quads_host || !out_host)) || !n_out_host) return MSOCR_E_ARG;
  const TailParams p = make_params(expand_w, expand_h, scale_x, scale_y, axis_aligned_output, remove_anomalies, sigma, min_count);
  std::vector<float> Q((size_t)M * 9), area(M), bb((size_t)M * 4);
  for (int i = 0; i < M; ++i) expand_scale(quads_host + 9 * i, p, &Q[9 * (size_t)i]);
  std::vector<char> keep(M, 1);
  if (M > 1) {
    for (int i = 0; i < M; ++i) { area[i] = quad_area(&Q[9 * (size_t)i]); quad_bbox(&Q[9 * (size_t)i], &bb[4 * (size_t)i]); }
    std::vector<int> order(M);
    for (int i = 0; i < M; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return area[a] < area[b]; });
    for (int i : order)
      for (int j = 0; j < M; ++j)
        if (j != i && keep[j] && quad_contained(&Q[9 * (size_t)i], &bb[4 * (size_t)i], area[i], &Q[9 * (size_t)j], &bb[4 * (size_t)j], area[j])) {
          keep[i] = 0;
          break;
        }
  }
  std::vector<int> kept;
  for (int i = 0; i < M; ++i) if (keep[i]) kept.push_back(i);
  if (p.anomalies && (int)kept.size() > p.min_count) {
    auto area_of = [&](int r) { return quad_area(&Q[9 * (size_t)kept[r]]); };
    float thr;
    if (anomaly_threshold(area_of, (int)kept.size(), p.sigma, &thr)) {
      std::vector<int> k2;
      for (size_t r = 0; r < kept.size(); ++r) if (area_of((int)r) <= thr) k2.push_back(kept[r]);
      if (!k2.empty()) kept.swap(k2);
    }
  }
  for (size_t r = 0; r < kept.size(); ++r) {
    const float* q = &Q[9 * (size_t)kept[r]];
    if (p.axis_aligned) axis_aligned(q, out_host + 9 * r);
    else for (int c = 0; c < 9; ++c) out_host[9 * r + c] = q[c];
  }
  *n_out_host = (int32_t)kept.size();
  return MSOCR_OK;
}

// ---- device: one 256-thread workgroup per page ------------------------------------------------------------------------
namespace {
constexpr int TAIL_LDSM = 2048;               // boxes per page whose per-box arrays fit the workgroup's LDS
constexpr int TAIL_CAPM = 16384;              // boxes per page the device path handles at all (more: n_out = -1, host path)
constexpr int TAIL_KW = TAIL_CAPM / 32 / 64;  // keep-mask words per lane of the greedy wave

// per-page workspace: containment matrix [cap][cap/32] u32, then (pages above TAIL_LDSM boxes only) the per-box arrays
__host__ __device__ inline int tail_cap(int max_cand) { return ((max_cand < TAIL_CAPM ? max_cand : TAIL_CAPM) + 31) & ~31; }
__host__ __device__ inline long tail_ws_words(int cap) { return (long)cap * (cap / 32) + (long)cap * 16; }

__global__ __launch_bounds__(256) void east_box_tail_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ nbox,
                                                             int max_cand, int cap, TailParams p, float* __restrict__ out,
                                                             int32_t* __restrict__ n_out, uint32_t* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ uint32_t keepbits[TAIL_CAPM / 32];
  __shared__ int s_n;
  const int pg = blockIdx.x, tid = threadIdx.x;
  const int M = nbox[pg];
  float* ob = out + (long)pg * max_cand * 9;
  if (M > cap || M < 0) {
    if (tid == 0) n_out[pg] = -1;
    return;
  }
  const int capw = cap / 32;
  uint32_t* inside = ws + (long)pg * tail_ws_words(cap);
  // per-box arrays: LDS for ordinary pages, the (L2-resident) workspace for pages above TAIL_LDSM boxes
  float* base = M <= TAIL_LDSM ? lds : reinterpret_cast<float*>(inside + (long)cap * capw);
  const int am = M <= TAIL_LDSM ? TAIL_LDSM : cap;
  float* Q = base;                                  // [am][9]
  float* area = Q + (long)am * 9;                   // [am]
  float* bb = area + am;                            // [am][4]
  int* order = reinterpret_cast<int*>(bb + (long)am * 4);  // [am]
  int* kept = order + am;                           // [am]
  const float* ib = boxes + (long)pg * max_cand * 9;
  const int W = (M + 31) >> 5;
  for (int i = tid; i < M; i += 256) {
    expand_scale(ib + 9 * i, p, Q + 9 * i);
    area[i] = quad_area(Q + 9 * i);
    quad_bbox(Q + 9 * i, bb + 4 * i);
  }
  for (int w = tid; w < TAIL_CAPM / 32; w += 256)
    keepbits[w] = w < W ? (w == W - 1 && (M & 31) ? ((1u << (M & 31)) - 1u) : 0xffffffffu) : 0u;
  __threadfence();
  __syncthreads();
  if (M > 1) {
    // stable ascending order of the areas: rank = number of boxes that sort before box i
    for (int i = tid; i < M; i += 256) {
      const float ai = area[i];
      int r = 0;
      for (int j = 0; j < M; ++j) r += (area[j] < ai || (area[j] == ai && j < i)) ? 1 : 0;
      order[r] = i;
    }
    // containment matrix: bit j of inside[i][w] = quad i lies inside quad 32w + j
    for (int idx = tid; idx < M * W; idx += 256) {
      const int i = idx / W, w = idx - i * W;
      uint32_t bits = 0;
      for (int b = 0; b < 32; ++b) {
        const int j = 32 * w + b;
        if (j < M && j != i && quad_contained(Q + 9 * i, bb + 4 * i, area[i], Q + 9 * j, bb + 4 * j, area[j])) bits |= 1u << b;
      }
      inside[(long)i * capw + w] = bits;
    }
    __threadfence();
    __syncthreads();
    // greedy pass in ascending-area order (infer.py:203-213): one wave, lane l owns words l, l+64, ... of the keep mask
    if (tid < 64) {
      uint32_t mykeep[TAIL_KW];
#pragma unroll
      for (int k = 0; k < TAIL_KW; ++k) mykeep[k] = keepbits[tid + 64 * k];
      for (int r = 0; r < M; ++r) {
        const int i = order[r];
        uint32_t v = 0u;
#pragma unroll
        for (int k = 0; k < TAIL_KW; ++k)
          if (tid + 64 * k < W) v |= inside[(long)i * capw + tid + 64 * k] & mykeep[k];
        if (__any(v != 0u)) {
#pragma unroll
          for (int k = 0; k < TAIL_KW; ++k)
            if (tid + 64 * k == (i >> 5)) mykeep[k] &= ~(1u << (i & 31));
        }
      }
#pragma unroll
      for (int k = 0; k < TAIL_KW; ++k) keepbits[tid + 64 * k] = mykeep[k];
    }
    __syncthreads();
  }
  if (tid == 0) {
    int n1 = 0;
    for (int i = 0; i < M; ++i)
      if (keepbits[i >> 5] >> (i & 31) & 1u) kept[n1++] = i;
    if (p.anomalies && n1 > p.min_count) {
      auto area_of = [&](int r) { return area[kept[r]]; };
      float thr;
      if (anomaly_threshold(area_of, n1, p.sigma, &thr)) {
        int n2 = 0;
        for (int r = 0; r < n1; ++r) n2 += area_of(r) <= thr ? 1 : 0;
        if (n2 > 0) {
          int o = 0;
          for (int r = 0; r < n1; ++r)
            if (area_of(r) <= thr) kept[o++] = kept[r];
          n1 = n2;
        }
      }
    }
    s_n = n1;
    n_out[pg] = n1;
  }
  __threadfence();
  __syncthreads();
  const int n = s_n;
  for (int r = tid; r < n; r += 256) {
    const float* q = Q + 9 * kept[r];
    if (p.axis_aligned) axis_aligned(q, ob + 9 * r);
    else for (int c = 0; c < 9; ++c) ob[9 * r + c] = q[c];
  }
}
}  // namespace

extern "C" int64_t msocr_east_box_tail_workspace_bytes(int N, int max_cand) {
  return (N > 0 && max_cand > 0) ? (int64_t)N * tail_ws_words(tail_cap(max_cand)) * 4 : 0;
}

extern "C" int msocr_east_box_tail(const float* boxes, const int32_t* nbox, int N, int max_cand, double expand_w, double expand_h,
                                   double scale_x, double scale_y, int axis_aligned_output, int remove_anomalies, double sigma,
                                   int min_count, float* out, int32_t* n_out, void* workspace, void* stream) {
  if (!boxes || !nbox || !out || !n_out || !workspace || N <= 0 || max_cand <= 0) return MSOCR_E_ARG;
  const TailParams p = make_params(expand_w, expand_h, scale_x, scale_y, axis_aligned_output, remove_anomalies, sigma, min_count);
  const size_t ldsz = (size_t)TAIL_LDSM * (9 + 1 + 4 + 1 + 1) * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)east_box_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz) != hipSuccess)
      return MSOCR_E_LAUNCH;
    attr = true;
  }
  MSOCR_LAUNCH(east_box_tail_kernel, dim3(N), dim3(256), ldsz, (hipStream_t)stream, boxes, nbox, max_cand, tail_cap(max_cand), p, out,
               n_out, (uint32_t*)workspace);
  return hipGetLastError() == hipSuccess ? MSOCR_OK : MSOCR_E_LAUNCH;
}
