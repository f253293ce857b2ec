// host_glue.hip — HOST-side helper of the Pipeline glue (no device code): reading order of the word boxes of a page.
//
// The reference does this in pure Python per page (an O(n^2) shrink loop of up to 50 sweeps plus a line-grouping loop);
// after the GPU offload that Python became the longest host stage between "boxes arrive" and "recogniser enqueued"
// (4 ms per 480-word page), i.e. device idle time.  Same integer / double arithmetic, literally the reference's loops:
//   detectors/_east/utils.py:500-547 (resolve_intersections), :550-607 (sort_boxes_reading_order),
//   :610-644 (sort_boxes_reading_order_with_resolutions: dict(zip(shrunk, boxes)) — later duplicate wins),
//   _pipeline.py:113-121 (re-match a sorted box to the FIRST word with an equal box).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <vector>

#include "msocr.h"

namespace {
struct Box {
  long long x0, y0, x1, y1;
  bool operator<(const Box& o) const {
    if (x0 != o.x0) return x0 < o.x0;
    if (y0 != o.y0) return y0 < o.y0;
    if (x1 != o.x1) return x1 < o.x1;
    return y1 < o.y1;
  }
};
inline long long shrink(long long lo, long long hi) { return (long long)((double)hi - (double)(hi - lo) * 0.1); }  // int() truncates
}  // namespace

extern "C" int msocr_reading_order_host(const int32_t* boxes_host, int n, double y_tol_ratio, double x_gap_ratio,
                                        int32_t* order_out_host) {
  if (n < 0 || (n > 0 && (!boxes_host || !order_out_host))) return MSOCR_E_ARG;
  if (n == 0) return MSOCR_OK;
  std::vector<Box> orig(n), b(n);
  for (int i = 0; i < n; ++i) orig[i] = b[i] = Box{boxes_host[4 * i], boxes_host[4 * i + 1], boxes_host[4 * i + 2], boxes_host[4 * i + 3]};
  // resolve_intersections: both members of every intersecting pair shrink by 10 % towards their top-left corner
  for (int sweep = 0; sweep < 50; ++sweep) {
    bool dirty = false;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) {
        const Box &p = b[i], &q = b[j];
        if (p.x1 <= q.x0 || q.x1 <= p.x0 || p.y1 <= q.y0 || q.y1 <= p.y0) continue;
        b[i].x1 = shrink(b[i].x0, b[i].x1); b[i].y1 = shrink(b[i].y0, b[i].y1);
        b[j].x1 = shrink(b[j].x0, b[j].x1); b[j].y1 = shrink(b[j].y0, b[j].y1);
        dirty = true;
      }
    if (!dirty) break;
  }
  // dict(zip(shrunk, boxes)): identical shrunk boxes collapse, the later original wins;  first word with an equal box
  std::map<Box, int> back, first;
  for (int i = 0; i < n; ++i) back[b[i]] = i;
  for (int i = n - 1; i >= 0; --i) first[orig[i]] = i;
  // sort_boxes_reading_order on the shrunk boxes
  double hsum = 0.0;
  for (int i = 0; i < n; ++i) hsum += (double)(b[i].y1 - b[i].y0);
  const double avg_h = hsum / n, tol = avg_h * y_tol_ratio, gap = avg_h * x_gap_ratio;
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  auto cy = [&](int i) { return (double)(b[i].y0 + b[i].y1) / 2.0; };
  std::stable_sort(idx.begin(), idx.end(), [&](int u, int v) { return cy(u) < cy(v); });
  std::vector<std::vector<int>> lines;
  std::vector<double> sums;
  std::vector<long long> maxx;
  for (int i : idx) {
    const double c = cy(i);
    int home = -1;
    for (size_t li = 0; li < lines.size(); ++li) {
      const double line_cy = sums[li] / (double)lines[li].size();
      if (fabs(c - line_cy) <= tol && (double)(b[i].x0 - maxx[li]) <= gap) { home = (int)li; break; }
    }
    if (home < 0) {
      lines.push_back({i}); sums.push_back(c); maxx.push_back(b[i].x1);
    } else {
      lines[home].push_back(i); sums[home] += c; maxx[home] = std::max(maxx[home], b[i].x1);
    }
  }
  std::vector<int> lorder(lines.size());
  for (size_t li = 0; li < lines.size(); ++li) lorder[li] = (int)li;
  std::stable_sort(lorder.begin(), lorder.end(),
                   [&](int u, int v) { return sums[u] / (double)lines[u].size() < sums[v] / (double)lines[v].size(); });
  int k = 0;
  for (int li : lorder) {
    std::vector<int>& ln = lines[li];
    std::stable_sort(ln.begin(), ln.end(), [&](int u, int v) { return b[u].x0 < b[v].x0; });
    for (int i : ln) order_out_host[k++] = first[orig[back[b[i]]]];
  }
  return MSOCR_OK;
}
