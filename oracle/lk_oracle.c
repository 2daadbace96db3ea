/*
 * lk_oracle.c -- CPU restatement of the pyramidal Lucas-Kanade stage.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h).
 *
 * Reference call site: OptFlow::computeFlow, Source Files/
 * feature_processing.cpp:238-249 (calc at :241, error gate at :245-248); the
 * tracker object is built at :228-236 from the criteria made at
 * Source Files/app_solver.cpp:108.  The arithmetic is OpenCV's
 * SparsePyrLKOpticalFlowImpl::calc -> buildOpticalFlowPyramid -> pyrDown,
 * calcSharrDeriv and LKTrackerInvoker (modules/video/src/lkpyramid.cpp,
 * modules/imgproc/src/pyramids.cpp), restated below for 8-bit single-channel
 * images, flags == 0.
 *
 * OpenCV accumulates the normal-equation sums in float in an order that
 * depends on the SIMD width it was built with.  acc_exact == 0 uses float
 * accumulators in plain row-major order (a build without SIMD); acc_exact != 0
 * uses int64 accumulators, which is the exact value every float order
 * approximates.  All other operations are identical in both modes.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* cv::borderInterpolate with BORDER_REFLECT_101 */
static inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* pyramids.cpp pyrDown_ with FixPtCast<uchar, 8>: separable [1 4 6 4 1],
 * integer sums, (v + 128) >> 8. */
void orc_pyr_down(const uint8_t* src, int w, int h, size_t src_stride,
                  uint8_t* dst, size_t dst_stride) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int* rows = (int*)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int y = 0; y < dh; ++y) {
        for (int k = 0; k < 5; ++k) {
            int sy = reflect101(2 * y + k - 2, h);
            const uint8_t* s = src + (size_t)sy * src_stride;
            int* r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; ++x) {
                int xm2 = reflect101(2 * x - 2, w), xm1 = reflect101(2 * x - 1, w);
                int x0 = reflect101(2 * x, w);
                int xp1 = reflect101(2 * x + 1, w), xp2 = reflect101(2 * x + 2, w);
                r[x] = s[x0] * 6 + (s[xm1] + s[xp1]) * 4 + s[xm2] + s[xp2];
            }
        }
        for (int x = 0; x < dw; ++x) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 +
                    rows[x] + rows[4 * dw + x];
            dst[(size_t)y * dst_stride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* lkpyramid.cpp calcSharrDeriv: smoothing [3 10 3], difference [-1 0 1],
 * unnormalised, reflect-101 at the image edge. */
void orc_scharr_deriv(const uint8_t* src, int w, int h, size_t src_stride,
                      int16_t* dst, size_t dst_stride) {
    int16_t* buf = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(w + 2));
    int16_t* trow0 = buf + 1;
    int16_t* trow1 = buf + (w + 2) + 1;
    for (int y = 0; y < h; ++y) {
        const uint8_t* s0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * src_stride;
        const uint8_t* s1 = src + (size_t)y * src_stride;
        const uint8_t* s2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * src_stride;
        int16_t* d = dst + (size_t)y * dst_stride;
        for (int x = 0; x < w; ++x) {
            trow0[x] = (int16_t)((s0[x] + s2[x]) * 3 + s1[x] * 10);
            trow1[x] = (int16_t)(s2[x] - s0[x]);
        }
        {
            int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
            trow0[-1] = trow0[x0];
            trow0[w] = trow0[x1];
            trow1[-1] = trow1[x0];
            trow1[w] = trow1[x1];
        }
        for (int x = 0; x < w; ++x) {
            d[2 * x] = (int16_t)(trow0[x + 1] - trow0[x - 1]);
            d[2 * x + 1] = (int16_t)((trow1[x + 1] + trow1[x - 1]) * 3 + trow1[x] * 10);
        }
    }
    free(buf);
}

/* lkpyramid.cpp buildOpticalFlowPyramid: the level loop stops after the first
 * level whose successor would not exceed the window in both dimensions. */
int orc_lk_num_levels(int w, int h, int win_w, int win_h, int max_level) {
    int sw = w, sh = h;
    for (int level = 0; level <= max_level; ++level) {
        sw = (sw + 1) / 2;
        sh = (sh + 1) / 2;
        if (sw <= win_w || sh <= win_h) return level;
    }
    return max_level;
}

typedef struct {
    int w, h;
    size_t stride;    /* bytes per padded image row */
    size_t dstride;   /* int16 elements per padded derivative row */
    uint8_t* img_buf; /* padded by (win_w, win_h), reflect-101 */
    uint8_t* img;     /* pixel (0,0) */
    int16_t* der_buf; /* padded by (win_w, win_h), zeros; NULL for the next image */
    int16_t* der;     /* pixel (0,0) */
} lk_level;

static void level_alloc(lk_level* L, int w, int h, int bw, int bh, int with_deriv) {
    L->w = w;
    L->h = h;
    L->stride = (size_t)(w + 2 * bw);
    L->img_buf = (uint8_t*)malloc(L->stride * (size_t)(h + 2 * bh));
    L->img = L->img_buf + (size_t)bh * L->stride + bw;
    L->dstride = 2 * (size_t)(w + 2 * bw);
    if (with_deriv) {
        L->der_buf = (int16_t*)calloc(L->dstride * (size_t)(h + 2 * bh), sizeof(int16_t));
        L->der = L->der_buf + (size_t)bh * L->dstride + 2 * bw;
    } else {
        L->der_buf = NULL;
        L->der = NULL;
    }
}

/* copyMakeBorder(..., BORDER_REFLECT_101) around the level image */
static void level_make_border(lk_level* L, int bw, int bh) {
    for (int y = -bh; y < L->h + bh; ++y) {
        int sy = reflect101(y, L->h);
        uint8_t* drow = L->img + (ptrdiff_t)y * (ptrdiff_t)L->stride;
        const uint8_t* srow = L->img + (ptrdiff_t)sy * (ptrdiff_t)L->stride;
        if (y < 0 || y >= L->h)
            for (int x = 0; x < L->w; ++x) drow[x] = srow[x];
        for (int x = -bw; x < 0; ++x) drow[x] = srow[reflect101(x, L->w)];
        for (int x = L->w; x < L->w + bw; ++x) drow[x] = srow[reflect101(x, L->w)];
    }
}

static void build_pyramid(const uint8_t* img, int w, int h, size_t stride, int bw,
                          int bh, int levels, int with_deriv, lk_level* out) {
    for (int l = 0; l <= levels; ++l) {
        if (l == 0) {
            level_alloc(&out[0], w, h, bw, bh, with_deriv);
            for (int y = 0; y < h; ++y)
                memcpy(out[0].img + (size_t)y * out[0].stride, img + (size_t)y * stride, (size_t)w);
        } else {
            int pw = out[l - 1].w, ph = out[l - 1].h;
            level_alloc(&out[l], (pw + 1) / 2, (ph + 1) / 2, bw, bh, with_deriv);
            orc_pyr_down(out[l - 1].img, pw, ph, out[l - 1].stride, out[l].img, out[l].stride);
        }
        level_make_border(&out[l], bw, bh);
        if (with_deriv)
            orc_scharr_deriv(out[l].img, out[l].w, out[l].h, out[l].stride, out[l].der,
                             out[l].dstride);
    }
}

static void free_pyramid(lk_level* p, int levels) {
    for (int l = 0; l <= levels; ++l) {
        free(p[l].img_buf);
        free(p[l].der_buf);
    }
}

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline void bilinear_weights(float a, float b, int* iw00, int* iw01, int* iw10,
                                    int* iw11) {
    /* cvRound = round half to even (lrint under the default rounding mode) */
    *iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
    *iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
    *iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
    *iw11 = (1 << W_BITS) - *iw00 - *iw01 - *iw10;
}

/* LKTrackerInvoker::operator() for one point on one level */
static int64_t lk_track_point(const lk_level* I, const lk_level* J, int level, int max_level,
                              int win_w, int win_h, int max_count, double epsilon,
                              float min_eig_threshold, const float* prev_pt, float* next_pt,
                              uint8_t* status, float* err, int acc_exact, int16_t* IWin,
                              int16_t* dIWin) {
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half_x = (win_w - 1) * 0.5f, half_y = (win_h - 1) * 0.5f;
    const float lscale = (float)(1. / (1 << level));
    float prev_x = prev_pt[0] * lscale, prev_y = prev_pt[1] * lscale;
    float next_x, next_y;
    int64_t iters = 0;

    if (level == max_level) {
        next_x = prev_x;
        next_y = prev_y;
    } else {
        next_x = next_pt[0] * 2.f;
        next_y = next_pt[1] * 2.f;
    }
    next_pt[0] = next_x;
    next_pt[1] = next_y;

    prev_x -= half_x;
    prev_y -= half_y;
    int ipx = (int)floorf(prev_x), ipy = (int)floorf(prev_y);

    if (ipx < -win_w || ipx >= I->w || ipy < -win_h || ipy >= I->h) {
        if (level == 0) {
            *status = 0;
            *err = 0;
        }
        return 0;
    }

    float a = prev_x - ipx, b = prev_y - ipy;
    int iw00, iw01, iw10, iw11;
    bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);

    const ptrdiff_t stepI = (ptrdiff_t)I->stride, dstep = (ptrdiff_t)I->dstride;
    const ptrdiff_t stepJ = (ptrdiff_t)J->stride;
    float fA11 = 0, fA12 = 0, fA22 = 0;
    int64_t lA11 = 0, lA12 = 0, lA22 = 0;

    for (int y = 0; y < win_h; ++y) {
        const uint8_t* src = I->img + (y + ipy) * stepI + ipx;
        const int16_t* dsrc = I->der + (y + ipy) * dstep + ipx * 2;
        int16_t* Iptr = IWin + y * win_w;
        int16_t* dIptr = dIWin + y * win_w * 2;
        for (int x = 0; x < win_w; ++x, dsrc += 2, dIptr += 2) {
            int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                   src[x + stepI + 1] * iw11,
                               W_BITS - 5);
            int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                                    dsrc[dstep + 2] * iw11,
                                W_BITS);
            int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 +
                                    dsrc[dstep + 3] * iw11,
                                W_BITS);
            Iptr[x] = (int16_t)ival;
            dIptr[0] = (int16_t)ixval;
            dIptr[1] = (int16_t)iyval;
            if (acc_exact) {
                lA11 += (int64_t)ixval * ixval;
                lA12 += (int64_t)ixval * iyval;
                lA22 += (int64_t)iyval * iyval;
            } else {
                fA11 += (float)(ixval * ixval);
                fA12 += (float)(ixval * iyval);
                fA22 += (float)(iyval * iyval);
            }
        }
    }
    if (acc_exact) {
        fA11 = (float)lA11;
        fA12 = (float)lA12;
        fA22 = (float)lA22;
    }

    float A11 = fA11 * FLT_SCALE, A12 = fA12 * FLT_SCALE, A22 = fA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                   (float)(2 * win_w * win_h);

    if (minEig < min_eig_threshold || D < FLT_EPSILON) {
        if (level == 0) *status = 0;
        return 0;
    }
    D = 1.f / D;

    next_x -= half_x;
    next_y -= half_y;
    float prev_dx = 0, prev_dy = 0;

    for (int j = 0; j < max_count; ++j) {
        int inx = (int)floorf(next_x), iny = (int)floorf(next_y);
        if (inx < -win_w || inx >= J->w || iny < -win_h || iny >= J->h) {
            if (level == 0) *status = 0;
            break;
        }
        ++iters;
        a = next_x - inx;
        b = next_y - iny;
        bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);
        float fb1 = 0, fb2 = 0;
        int64_t lb1 = 0, lb2 = 0;
        for (int y = 0; y < win_h; ++y) {
            const uint8_t* Jptr = J->img + (y + iny) * stepJ + inx;
            const int16_t* Iptr = IWin + y * win_w;
            const int16_t* dIptr = dIWin + y * win_w * 2;
            for (int x = 0; x < win_w; ++x, dIptr += 2) {
                int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                       Jptr[x + stepJ + 1] * iw11,
                                   W_BITS - 5) -
                           Iptr[x];
                if (acc_exact) {
                    lb1 += (int64_t)diff * dIptr[0];
                    lb2 += (int64_t)diff * dIptr[1];
                } else {
                    fb1 += (float)(diff * dIptr[0]);
                    fb2 += (float)(diff * dIptr[1]);
                }
            }
        }
        if (acc_exact) {
            fb1 = (float)lb1;
            fb2 = (float)lb2;
        }
        float b1 = fb1 * FLT_SCALE, b2 = fb2 * FLT_SCALE;
        float dx = (float)((A12 * b2 - A22 * b1) * D);
        float dy = (float)((A12 * b1 - A11 * b2) * D);

        next_x += dx;
        next_y += dy;
        next_pt[0] = next_x + half_x;
        next_pt[1] = next_y + half_y;

        if ((double)dx * dx + (double)dy * dy <= epsilon) break;

        if (j > 0 && fabs((double)(dx + prev_dx)) < 0.01 && fabs((double)(dy + prev_dy)) < 0.01) {
            next_pt[0] -= dx * 0.5f;
            next_pt[1] -= dy * 0.5f;
            break;
        }
        prev_dx = dx;
        prev_dy = dy;
    }

    if (*status && level == 0) {
        float ex = next_pt[0] - half_x, ey = next_pt[1] - half_y;
        int inx = (int)floorf(ex), iny = (int)floorf(ey);
        if (inx < -win_w || inx >= J->w || iny < -win_h || iny >= J->h) {
            *status = 0;
            return iters;
        }
        float aa = ex - inx, bb = ey - iny;
        bilinear_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
        float errval = 0.f;
        for (int y = 0; y < win_h; ++y) {
            const uint8_t* Jptr = J->img + (y + iny) * stepJ + inx;
            const int16_t* Iptr = IWin + y * win_w;
            for (int x = 0; x < win_w; ++x) {
                int diff = DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 + Jptr[x + stepJ] * iw10 +
                                       Jptr[x + stepJ + 1] * iw11,
                                   W_BITS - 5) -
                           Iptr[x];
                errval += fabsf((float)diff);
            }
        }
        *err = errval * 1.f / (float)(32 * win_w * win_h);
    }
    return iters;
}

/* Optional per (level, point) iteration counts for studying the distribution: when set, the
 * next orc_pyrlk calls store their counts at buf[level * n + i]. */
static int32_t* g_iter_buf = NULL;
void orc_lk_set_iter_buffer(int32_t* buf) { g_iter_buf = buf; }

void orc_pyrlk(const uint8_t* prev, const uint8_t* curr, int w, int h, size_t stride,
               const float* prev_pts, int n, int win_w, int win_h, int max_level, int crit_type,
               int max_count, double epsilon, double min_eig_threshold, float max_error,
               float* curr_pts, uint8_t* status, float* err, int acc_exact, int threads,
               int64_t* iters_out) {
    if (iters_out) *iters_out = 0;
    if (n <= 0) return;

    /* SparsePyrLKOpticalFlowImpl::calc: criteria clamping */
    if ((crit_type & 1) == 0)
        max_count = 30;
    else
        max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    if ((crit_type & 2) == 0)
        epsilon = 0.01;
    else
        epsilon = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    epsilon *= epsilon;

    int levels = orc_lk_num_levels(w, h, win_w, win_h, max_level);
    lk_level* P = (lk_level*)calloc((size_t)levels + 1, sizeof(lk_level));
    lk_level* N = (lk_level*)calloc((size_t)levels + 1, sizeof(lk_level));
    build_pyramid(prev, w, h, stride, win_w, win_h, levels, 1, P);
    build_pyramid(curr, w, h, stride, win_w, win_h, levels, 0, N);

    for (int i = 0; i < n; ++i) {
        status[i] = 1;
        err[i] = 0.f;
    }

    int nthr = 1;
#ifdef _OPENMP
    nthr = threads <= 0 ? omp_get_max_threads() : threads;
#else
    (void)threads;
#endif
    int64_t total_iters = 0;
    for (int level = levels; level >= 0; --level) {
#pragma omp parallel num_threads(nthr) reduction(+ : total_iters)
        {
            int16_t* IWin = (int16_t*)malloc(sizeof(int16_t) * 3 * (size_t)win_w * (size_t)win_h);
            int16_t* dIWin = IWin + (size_t)win_w * (size_t)win_h;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < n; ++i) {
                const int64_t it = lk_track_point(&P[level], &N[level], level, levels, win_w, win_h,
                                                  max_count, epsilon, (float)min_eig_threshold,
                                                  prev_pts + 2 * i, curr_pts + 2 * i, status + i,
                                                  err + i, acc_exact, IWin, dIWin);
                total_iters += it;
                if (g_iter_buf) g_iter_buf[(size_t)level * n + i] = (int32_t)it;
            }
            free(IWin);
        }
    }
    if (iters_out) *iters_out = total_iters;

    /* feature_processing.cpp:245-248 */
    for (int i = 0; i < n; ++i)
        if (err[i] > max_error) status[i] = 0;

    free_pyramid(P, levels);
    free_pyramid(N, levels);
    free(P);
    free(N);
}
