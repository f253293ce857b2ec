/* oracle/lanms.c — plain-C fp64 restatement of the reference LANMS.
 * TEST INFRASTRUCTURE ONLY (checker + cpu_baseline "port"); never shipped.
 *
 * Follows /root/reference/src/manuscript/detectors/_east/lanms.py line by line:
 *   polygon_area          lanms.py:7-14
 *   compute_intersection  lanms.py:17-29
 *   clip_polygon          lanms.py:32-57
 *   polygon_intersection  lanms.py:60-77
 *   polygon_iou           lanms.py:80-91
 *   should_merge          lanms.py:94-96
 *   normalize_polygon     lanms.py:99-130
 *   standard_nms          lanms.py:133-153
 *   locality_aware_nms    lanms.py:156-207
 * Build with -ffp-contract=off: numba (no fastmath) never fuses a*b-c*d.
 * Sorts are STABLE (key, original index): the reference's np.argsort default is
 * unstable, so ties are implementation-defined there.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXV 20 /* lanms.py:34 out = np.empty((20, 2)) */

double orc_polygon_area(const double *poly, int n) {
    double area = 0.0;
    for (int i = 0; i < n; i++) {
        int j = (i + 1) % n;
        area += poly[2 * i] * poly[2 * j + 1] - poly[2 * j] * poly[2 * i + 1];
    }
    return fabs(area) / 2.0;
}

void orc_compute_intersection(const double *p1, const double *p2, const double *A, const double *B, double *out) {
    double BAx = p2[0] - p1[0];
    double BAy = p2[1] - p1[1];
    double DCx = B[0] - A[0];
    double DCy = B[1] - A[1];
    double denom = BAx * DCy - BAy * DCx;
    double CAx = A[0] - p1[0];
    double CAy = A[1] - p1[1];
    if (denom == 0) {
        out[0] = p1[0];
        out[1] = p1[1];
        return;
    }
    double t = (CAx * DCy - CAy * DCx) / denom;
    out[0] = p1[0] + t * BAx;
    out[1] = p1[1] + t * BAy;
}

int orc_clip_polygon(const double *subject, int n, const double *A, const double *B, double *out) {
    int count = 0;
    for (int i = 0; i < n; i++) {
        const double *curr = subject + 2 * i;
        const double *prev = subject + 2 * ((i - 1 + n) % n);
        int curr_inside = (B[0] - A[0]) * (curr[1] - A[1]) - (B[1] - A[1]) * (curr[0] - A[0]) >= 0;
        int prev_inside = (B[0] - A[0]) * (prev[1] - A[1]) - (B[1] - A[1]) * (prev[0] - A[0]) >= 0;
        if (curr_inside) {
            if (!prev_inside) {
                orc_compute_intersection(prev, curr, A, B, out + 2 * count);
                count++;
            }
            out[2 * count] = curr[0];
            out[2 * count + 1] = curr[1];
            count++;
        } else if (prev_inside) {
            orc_compute_intersection(prev, curr, A, B, out + 2 * count);
            count++;
        }
    }
    return count;
}

int orc_polygon_intersection(const double *poly1, int n, const double *poly2, int m, double *result) {
    double bufa[2 * MAXV], bufb[2 * MAXV];
    double *cur = bufa, *nxt = bufb;
    memcpy(cur, poly1, sizeof(double) * 2 * n);
    int cnt = n;
    for (int i = 0; i < m; i++) {
        const double *A = poly2 + 2 * i;
        const double *B = poly2 + 2 * ((i + 1) % m);
        cnt = orc_clip_polygon(cur, cnt, A, B, nxt);
        double *t = cur;
        cur = nxt;
        nxt = t;
        if (cnt == 0) break;
    }
    memcpy(result, cur, sizeof(double) * 2 * cnt);
    return cnt;
}

double orc_polygon_iou(const double *poly1, int n, const double *poly2, int m) {
    double inter[2 * MAXV];
    int k = orc_polygon_intersection(poly1, n, poly2, m, inter);
    double inter_area = 0.0;
    if (k > 2) inter_area = orc_polygon_area(inter, k);
    double area1 = orc_polygon_area(poly1, n);
    double area2 = orc_polygon_area(poly2, m);
    double union_area = area1 + area2 - inter_area;
    if (union_area <= 0) return 0.0;
    return inter_area / union_area;
}

int orc_should_merge(const double *p1, const double *p2, double thr) { return orc_polygon_iou(p1, 4, p2, 4) > thr; }

void orc_normalize_polygon(const double *ref, const double *poly, double *out) {
    int best_order = 0, best_start = 0;
    double min_d = 1e20;
    for (int start = 0; start < 4; start++) {
        double d = 0.0;
        for (int i = 0; i < 4; i++) {
            double dx = ref[2 * i] - poly[2 * ((start + i) % 4)];
            double dy = ref[2 * i + 1] - poly[2 * ((start + i) % 4) + 1];
            d += dx * dx + dy * dy;
        }
        if (d < min_d) {
            min_d = d;
            best_start = start;
            best_order = 0;
        }
    }
    for (int start = 0; start < 4; start++) {
        double d = 0.0;
        for (int i = 0; i < 4; i++) {
            int idx = ((start - i) % 4 + 4) % 4;
            double dx = ref[2 * i] - poly[2 * idx];
            double dy = ref[2 * i + 1] - poly[2 * idx + 1];
            d += dx * dx + dy * dy;
        }
        if (d < min_d) {
            min_d = d;
            best_start = start;
            best_order = 1;
        }
    }
    for (int i = 0; i < 4; i++) {
        int idx = best_order == 0 ? (best_start + i) % 4 : ((best_start - i) % 4 + 4) % 4;
        out[2 * i] = poly[2 * idx];
        out[2 * i + 1] = poly[2 * idx + 1];
    }
}

/* stable argsort helpers (merge sort on index arrays); total order = ascending key, NaN last
 * (np.argsort semantics), ties by original index */
static int less_f32(float a, float b) { return a < b || (b != b && a == a); }
static int less_f64(double a, double b) { return a < b || (b != b && a == a); }
static void msort_f32(const float *key, int64_t *idx, int64_t *tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    msort_f32(key, idx, tmp, h);
    msort_f32(key, idx + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = less_f32(key[idx[j]], key[idx[i]]) ? idx[j++] : idx[i++];
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, sizeof(int64_t) * n);
}
static void msort_f64_desc(const double *key, int64_t *idx, int64_t *tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    msort_f64_desc(key, idx, tmp, h);
    msort_f64_desc(key, idx + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    /* ascending in -score */
    while (i < h && j < n) tmp[k++] = less_f64(-key[idx[j]], -key[idx[i]]) ? idx[j++] : idx[i++];
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, sizeof(int64_t) * n);
}

/* standard_nms: polys (M,4,2) f64, scores (M) f64 -> keep indices (in kept order). returns count */
int64_t orc_standard_nms(const double *polys, const double *scores, int64_t M, double thr, int64_t *keep) {
    if (M == 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * M);
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * M);
    unsigned char *sup = (unsigned char *)calloc(M, 1);
    for (int64_t i = 0; i < M; i++) order[i] = i;
    msort_f64_desc(scores, order, tmp, M);
    int64_t nk = 0;
    for (int64_t i = 0; i < M; i++) {
        int64_t idx = order[i];
        if (sup[idx]) continue;
        keep[nk++] = idx;
        for (int64_t j = i + 1; j < M; j++) {
            int64_t idj = order[j];
            if (sup[idj]) continue;
            if (orc_should_merge(polys + 8 * idx, polys + 8 * idj, thr)) sup[idj] = 1;
        }
    }
    free(order);
    free(tmp);
    free(sup);
    return nk;
}

/* locality_aware_nms: boxes (N,9) f32 -> out (<=N,9) f32; returns M.
 * n_merged_out (optional) receives the phase-1 count M'. */
int64_t orc_locality_aware_nms(const float *boxes, int64_t N, double thr, float *out, int64_t *n_merged_out) {
    if (N == 0) {
        if (n_merged_out) *n_merged_out = 0;
        return 0;
    }
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * N);
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * N);
    float *key = (float *)malloc(sizeof(float) * N);
    for (int64_t i = 0; i < N; i++) {
        order[i] = i;
        key[i] = boxes[9 * i];
    }
    msort_f32(key, order, tmp, N);
    double *mp = (double *)malloc(sizeof(double) * 8 * N);
    double *ms = (double *)malloc(sizeof(double) * N);
    double *ws = (double *)malloc(sizeof(double) * N);
    int64_t nm = 0;
    for (int64_t s = 0; s < N; s++) {
        const float *b = boxes + 9 * order[s];
        double poly[8], aligned[8];
        for (int k = 0; k < 8; k++) poly[k] = (double)b[k];
        double score = (double)b[8];
        if (nm > 0) {
            double *last = mp + 8 * (nm - 1);
            if (orc_should_merge(poly, last, thr)) {
                orc_normalize_polygon(last, poly, aligned);
                double tw = ws[nm - 1] + score;
                for (int k = 0; k < 8; k++) last[k] = (last[k] * ws[nm - 1] + aligned[k] * score) / tw;
                ws[nm - 1] = tw;
                ms[nm - 1] = ms[nm - 1] > score ? ms[nm - 1] : score; /* max(a,b): a if not b>a */
                continue;
            }
        }
        memcpy(mp + 8 * nm, poly, sizeof(poly));
        ms[nm] = score;
        ws[nm] = score;
        nm++;
    }
    if (n_merged_out) *n_merged_out = nm;
    int64_t *keep = (int64_t *)malloc(sizeof(int64_t) * nm);
    int64_t nk = orc_standard_nms(mp, ms, nm, thr, keep);
    for (int64_t i = 0; i < nk; i++) {
        for (int k = 0; k < 8; k++) out[9 * i + k] = (float)mp[8 * keep[i] + k];
        out[9 * i + 8] = (float)ms[keep[i]];
    }
    free(order); free(tmp); free(key); free(mp); free(ms); free(ws); free(keep);
    return nk;
}
