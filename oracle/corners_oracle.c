/*
 * corners_oracle.c -- CPU restatement of cv::goodFeaturesToTrack as called by
 * FeatureDetector::generateFlowFeatures, Source Files/feature_processing.cpp:87-99
 * (Shi-Tomasi: no mask, blockSize 3, Sobel aperture 3, no Harris).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: restates the published OpenCV 4.x
 * algorithm (modules/imgproc/src/featureselect.cpp, corner.cpp) --
 *   cornerMinEigenVal: Dx, Dy = Sobel(3x3, scale 1 / (4 * 3 * 255), BORDER_REFLECT_101) in
 *   float; cov = (dx dx, dx dy, dy dy); 3x3 unnormalised box filter (BORDER_REFLECT_101);
 *   eig = (a + c) - sqrt((a - c)^2 + b^2) with a = cov0 / 2, b = cov1, c = cov2 / 2;
 *   threshold at max * qualityLevel (THRESH_TOZERO); candidates are interior pixels whose
 *   value is non-zero and equals the 3x3 dilation; sorted by value, descending, ties by
 *   address, descending; greedy selection with a grid of round(minDistance) cells, a corner
 *   being dropped when an accepted one in the 3x3 cells around it is closer than minDistance;
 *   at most maxCorners.
 * One step is fixed here where OpenCV's result depends on its build and thread count: its box
 * filter keeps running row and column sums (add the entering element, subtract the leaving
 * one) restarted per stripe of rows, so the rounding of a sum depends on the stripe partition;
 * this restatement sums the nine products directly, rows first: ((l + m) + r) per row, then
 * ((top + mid) + bottom).  Sobel's smoothing tap order is (s0 + s2) * k1 + s1 * k0 with the
 * scale folded into the smoothing kernel, as OpenCV's scalar small-kernel filters compute it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static inline int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* cornerMinEigenVal(image, eig, 3, 3): eig is [h][w] float. */
void orc_min_eigen_val(const uint8_t* img, int w, int h, size_t stride, float* eig) {
    const double scale_d = 1.0 / (4.0 * 3.0 * 255.0);
    const float k0 = (float)(2.0 * scale_d), k1 = (float)(1.0 * scale_d); /* [1 2 1] * scale */
    float* dx = (float*)malloc(sizeof(float) * (size_t)w * h);
    float* dy = (float*)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * stride;
        const uint8_t* r1 = img + (size_t)y * stride;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * stride;
        for (int x = 0; x < w; ++x) {
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            /* Dx: rows [-1 0 1] (exact), columns [1 2 1] * scale */
            const float d0 = (float)(r0[xr] - r0[xl]);
            const float d1 = (float)(r1[xr] - r1[xl]);
            const float d2 = (float)(r2[xr] - r2[xl]);
            dx[(size_t)y * w + x] = (d0 + d2) * k1 + d1 * k0;
            /* Dy: rows [1 2 1] * scale, columns [-1 0 1] */
            const float s0 = (float)(r0[xl] + r0[xr]) * k1 + (float)r0[x] * k0;
            const float s2 = (float)(r2[xl] + r2[xr]) * k1 + (float)r2[x] * k0;
            dy[(size_t)y * w + x] = s2 - s0;
        }
    }
    for (int y = 0; y < h; ++y) {
        const int ys[3] = {reflect101(y - 1, h), y, reflect101(y + 1, h)};
        for (int x = 0; x < w; ++x) {
            const int xs[3] = {reflect101(x - 1, w), x, reflect101(x + 1, w)};
            float rows[3][3];
            for (int j = 0; j < 3; ++j) {
                float p[3][3];
                for (int i = 0; i < 3; ++i) {
                    const float gx = dx[(size_t)ys[j] * w + xs[i]];
                    const float gy = dy[(size_t)ys[j] * w + xs[i]];
                    p[0][i] = gx * gx;
                    p[1][i] = gx * gy;
                    p[2][i] = gy * gy;
                }
                for (int c = 0; c < 3; ++c) rows[c][j] = (p[c][0] + p[c][1]) + p[c][2];
            }
            const float cxx = (rows[0][0] + rows[0][1]) + rows[0][2];
            const float cxy = (rows[1][0] + rows[1][1]) + rows[1][2];
            const float cyy = (rows[2][0] + rows[2][1]) + rows[2][2];
            const float a = cxx * 0.5f, b = cxy, c = cyy * 0.5f;
            eig[(size_t)y * w + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        }
    }
    free(dx);
    free(dy);
}

typedef struct {
    float val;
    int32_t pos; /* y * w + x: the order of addresses in a continuous image */
} orc_cand;

static int cand_greater(const void* pa, const void* pb) {
    const orc_cand* a = (const orc_cand*)pa;
    const orc_cand* b = (const orc_cand*)pb;
    if (a->val > b->val) return -1;
    if (a->val < b->val) return 1;
    return a->pos > b->pos ? -1 : (a->pos < b->pos ? 1 : 0);
}

/* cv::goodFeaturesToTrack(image, corners, maxCorners, qualityLevel, minDistance).
 * corners has room for max_out points (x, y); returns the number found (not more than max_out;
 * maxCorners <= 0 means no limit, as in OpenCV).  n_candidates, when given, receives the number
 * of local maxima above the threshold. */
int orc_good_features(const uint8_t* img, int w, int h, size_t stride, int max_corners, double quality,
                      double min_distance, float* corners, int max_out, int* n_candidates) {
    if (n_candidates) *n_candidates = 0;
    if (w <= 0 || h <= 0) return 0;
    float* eig = (float*)malloc(sizeof(float) * (size_t)w * h);
    orc_min_eigen_val(img, w, h, stride, eig);
    double max_val = -INFINITY; /* minMaxLoc */
    for (size_t i = 0; i < (size_t)w * h; ++i)
        if (eig[i] > max_val) max_val = eig[i];
    const float thresh = (float)(max_val * quality);
    for (size_t i = 0; i < (size_t)w * h; ++i) eig[i] = eig[i] > thresh ? eig[i] : 0.f; /* THRESH_TOZERO */
    orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * (size_t)w * h);
    size_t total = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float v = eig[(size_t)y * w + x];
            if (v == 0) continue;
            float m = v; /* dilate 3x3 */
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    const float u = eig[(size_t)(y + j) * w + x + i];
                    if (u > m) m = u;
                }
            if (v == m) {
                cand[total].val = v;
                cand[total].pos = y * w + x;
                ++total;
            }
        }
    if (n_candidates) *n_candidates = (int)total;
    qsort(cand, total, sizeof(orc_cand), cand_greater);
    int n = 0;
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        /* per cell: a growing list of accepted points */
        int* head = (int*)malloc(sizeof(int) * (size_t)gw * gh);
        int* next = (int*)malloc(sizeof(int) * (total ? total : 1));
        float* acc = (float*)malloc(sizeof(float) * 2 * (total ? total : 1));
        for (int i = 0; i < gw * gh; ++i) head[i] = -1;
        const double md2 = min_distance * min_distance;
        int n_acc = 0;
        for (size_t i = 0; i < total; ++i) {
            const int y = cand[i].pos / w, x = cand[i].pos - y * w;
            const int xc = x / cell, yc = y / cell;
            int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
            if (x1 < 0) x1 = 0;
            if (y1 < 0) y1 = 0;
            if (x2 > gw - 1) x2 = gw - 1;
            if (y2 > gh - 1) y2 = gh - 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
                        const float ddx = (float)x - acc[2 * j], ddy = (float)y - acc[2 * j + 1];
                        if (ddx * ddx + ddy * ddy < md2) {
                            good = 0;
                            break;
                        }
                    }
            if (good) {
                acc[2 * n_acc] = (float)x;
                acc[2 * n_acc + 1] = (float)y;
                next[n_acc] = head[yc * gw + xc];
                head[yc * gw + xc] = n_acc;
                ++n_acc;
                if (n < max_out) {
                    corners[2 * n] = (float)x;
                    corners[2 * n + 1] = (float)y;
                }
                ++n;
                if ((max_corners > 0 && n == max_corners) || n >= max_out) break;
            }
        }
        free(head);
        free(next);
        free(acc);
    } else {
        for (size_t i = 0; i < total; ++i) {
            const int y = cand[i].pos / w, x = cand[i].pos - y * w;
            if (n < max_out) {
                corners[2 * n] = (float)x;
                corners[2 * n + 1] = (float)y;
            }
            ++n;
            if ((max_corners > 0 && n == max_corners) || n >= max_out) break;
        }
    }
    free(cand);
    free(eig);
    return n < max_out ? n : max_out;
}
