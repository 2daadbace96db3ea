// Epipolar filter of DescriptorMatcher::findRobustMatches: cv::findFundamentalMat(prevAligPts,
// currAligPts, inliersMask) at Source Files/feature_processing.cpp:187-188, i.e. FM_RANSAC
// with threshold 3, confidence 0.99 and at most 1000 iterations (OpenCV itself switches to
// LMedS below 15 correspondences and to one direct 7-point solve at exactly 7).
//
// OpenCV's loop is sequential: draw 7 correspondences, solve for up to three matrices, count
// inliers, shrink the iteration budget when a better model appears.  Only two parts of it are
// truly sequential, and both are tiny: the random draws (one generator, rejection of repeated
// and collinear picks) and the "best so far / iterations left" bookkeeping.  So, per point set:
//   1. one lane replays the draws and records every subset the loop could ever use;
//   2. one lane per subset solves the 7-point problem;
//   3. one wave per subset scores its models against all correspondences;
//   4. one lane replays the bookkeeping over the recorded scores, in iteration order, and
//      stops where OpenCV's loop would have stopped;
//   5. the mask of the winning model is written.
// Steps 2 and 3 evaluate subsets the sequential loop never reaches; they are cheap (a few
// hundred microseconds for 32 point sets of 6000 correspondences) and independent.
//
// The two-dimensional null space of the 7x9 system is taken by Gauss-Jordan elimination with
// complete pivoting and Gram-Schmidt, not by OpenCV's SVD (whose basis of that plane depends
// on its sweep order and on a random completion); the matrices found along the pencil do not
// depend on the basis beyond rounding.
#include <cfloat>
#include <cmath>

#include "sfm_internal.h"

namespace sfm {

namespace {

enum Regime { kNone = 0, kDirect = 1, kLmeds = 2, kRansac = 3 };

struct FmArgs {
    const float2* pts1;
    const float2* pts2;
    size_t stride;          // points between consecutive problems
    const int32_t* counts;  // per problem, or nullptr
    int count_uniform;
    int n_problems;
    int max_iters;
    double thresh, confidence;
    int32_t* subsets;   // [problem][max_iters][7]
    int32_t* n_drawn;   // [problem]
    double* models;     // [problem][max_iters][3][9]
    int32_t* n_models;  // [problem][max_iters]
    float* scores;      // [problem][max_iters][3]: inlier count (RANSAC) or median error (LMedS)
    double* best;       // [problem][10]: matrix, inlier threshold
    double* F;          // [problem][9] or nullptr
    uint8_t* mask;      // [problem][stride]
    int32_t* found;     // [problem]
};

__device__ __forceinline__ int problem_count(const FmArgs& a, int p) {
    const int c = a.counts ? a.counts[p] : a.count_uniform;
    return c < 0 ? 0 : (c > (int)a.stride ? (int)a.stride : c);
}

__device__ __forceinline__ int regime_of(int count) {
    return count < 7 ? kNone : count == 7 ? kDirect : count < 15 ? kLmeds : kRansac;
}

// cv::RNG (multiply with carry)
struct Rng {
    unsigned long long state;
    __device__ unsigned next() {
        state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// haveCollinearPoints: is the last of the seven points on a line through two earlier ones
__device__ bool have_collinear(const float2 (&m)[7]) {
    const int i = 6;
    for (int j = 0; j < i; j++) {
        const double dx1 = m[j].x - m[i].x;
        const double dy1 = m[j].y - m[i].y;
        for (int k = 0; k < j; k++) {
            const double dx2 = m[k].x - m[i].x;
            const double dy2 = m[k].y - m[i].y;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

// Step 1: the subsets getSubset would hand to the loop, in order.
__global__ void fm_subsets_kernel(FmArgs a) {
    const int p = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    int drawn = 0;
    if (regime == kRansac || regime == kLmeds) {
        const float2* m1 = a.pts1 + (size_t)p * a.stride;
        const float2* m2 = a.pts2 + (size_t)p * a.stride;
        int32_t* out = a.subsets + (size_t)p * a.max_iters * 7;
        const int max_attempts = regime == kRansac ? 10000 : 1000;
        Rng rng{~0ULL};
        for (; drawn < a.max_iters; ++drawn) {
            int idx[7];
            float2 s1[7], s2[7];
            int iters = 0;
            bool complete = false;
            for (; iters < max_attempts; iters++) {
                for (int i = 0; i < 7; ++i) {
                    int pick;
                    for (;;) {
                        pick = rng.uniform(0, count);
                        bool dup = false;
                        for (int k = 0; k < i; ++k) dup |= idx[k] == pick;
                        if (!dup) break;
                    }
                    idx[i] = pick;
                    s1[i] = m1[pick];
                    s2[i] = m2[pick];
                }
                complete = true;
                if (have_collinear(s1) || have_collinear(s2)) continue;
                break;
            }
            if (!(complete && iters < max_attempts)) break;
            for (int k = 0; k < 7; ++k) out[drawn * 7 + k] = idx[k];
        }
    }
    a.n_drawn[p] = drawn;
}

// cv::solveCubic
__device__ int solve_cubic(const double (&c)[4], double (&roots)[3]) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0., x1 = 0., x2 = 0.;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0)
                n = a3 == 0 ? -1 : 0;
            else {
                x0 = -a3 / a2;
                n = 1;
            }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                const double q1 = (-a2 + d) * 0.5;
                const double q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) {
                    x0 = q1 / a1;
                    x1 = a3 / q1;
                } else {
                    x0 = q2 / a1;
                    x1 = a3 / q2;
                }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0;
        a1 *= a0;
        a2 *= a0;
        a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = acos(R / sqrt(Qcubed));
            const double sqrtQ = sqrt(Q);
            const double t0 = -2 * sqrtQ;
            const double t1 = theta * (1. / 3);
            const double t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) {
                x0 = -2 * pow(R, 1. / 3) - a1 / 3;
                x1 = pow(R, 1. / 3) - a1 / 3;
            } else {
                x0 = 2 * pow(-R, 1. / 3) - a1 / 3;
                x1 = -pow(-R, 1. / 3) - a1 / 3;
            }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            d = sqrt(-d);
            double e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0;
    roots[1] = x1;
    roots[2] = x2;
    return n;
}

// Orthonormal basis (f1, f2) of the null space of the 7x9 matrix a (destroyed).
__device__ bool null_space_7x9(double* a, double* f1, double* f2) {
    int col_of[7];
    bool used_col[9] = {false, false, false, false, false, false, false, false, false};
    for (int k = 0; k < 7; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = k; r < 7; ++r)
            for (int c = 0; c < 9; ++c) {
                if (used_col[c]) continue;
                const double v = fabs(a[r * 9 + c]);
                if (v > best) {
                    best = v;
                    pr = r;
                    pc = c;
                }
            }
        if (pr < 0) return false;
        if (pr != k)
            for (int c = 0; c < 9; ++c) {
                const double t = a[k * 9 + c];
                a[k * 9 + c] = a[pr * 9 + c];
                a[pr * 9 + c] = t;
            }
        used_col[pc] = true;
        col_of[k] = pc;
        const double inv = 1.0 / a[k * 9 + pc];
        for (int c = 0; c < 9; ++c) a[k * 9 + c] *= inv;
        for (int r = 0; r < 7; ++r) {
            if (r == k) continue;
            const double m = a[r * 9 + pc];
            if (m == 0) continue;
            for (int c = 0; c < 9; ++c) a[r * 9 + c] -= m * a[k * 9 + c];
        }
    }
    int free_cols[2], nf = 0;
    for (int c = 0; c < 9; ++c)
        if (!used_col[c]) free_cols[nf++] = c;
    for (int v = 0; v < 2; ++v) {
        double* f = v == 0 ? f1 : f2;
        for (int c = 0; c < 9; ++c) f[c] = 0;
        f[free_cols[v]] = 1.0;
        for (int k = 0; k < 7; ++k) f[col_of[k]] = -a[k * 9 + free_cols[v]];
    }
    double n2 = 0;
    for (int c = 0; c < 9; ++c) n2 += f2[c] * f2[c];
    double inv = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; ++c) f2[c] *= inv;
    double dot = 0;
    for (int c = 0; c < 9; ++c) dot += f1[c] * f2[c];
    for (int c = 0; c < 9; ++c) f1[c] -= dot * f2[c];
    n2 = 0;
    for (int c = 0; c < 9; ++c) n2 += f1[c] * f1[c];
    inv = 1.0 / sqrt(n2);
    for (int c = 0; c < 9; ++c) f1[c] *= inv;
    return true;
}

// run7Point
__device__ int seven_point(const float2 (&m1)[7], const float2 (&m2)[7], double* fmatrix) {
    double a[7 * 9], c[4], r[3] = {0, 0, 0};
    double f1[9], f2[9];
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[i].x, y0 = m1[i].y;
        const double x1 = m2[i].x, y1 = m2[i].y;
        a[i * 9 + 0] = x1 * x0;
        a[i * 9 + 1] = x1 * y0;
        a[i * 9 + 2] = x1;
        a[i * 9 + 3] = y1 * x0;
        a[i * 9 + 4] = y1 * y0;
        a[i * 9 + 5] = y1;
        a[i * 9 + 6] = x0;
        a[i * 9 + 7] = y0;
        a[i * 9 + 8] = 1;
    }
    if (!null_space_7x9(a, f1, f2)) return 0;
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];

    double t0 = f2[4] * f2[8] - f2[5] * f2[7];
    double t1 = f2[3] * f2[8] - f2[5] * f2[6];
    double t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;

    const int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; k++, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) {
            mu = 1. / s;
            lambda *= mu;
            fmatrix[8] = 1.;
        } else
            fmatrix[8] = 0.;
        for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

// Step 2: one lane per (problem, iteration).
__global__ void fm_solve_kernel(FmArgs a) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (it >= a.max_iters) return;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    int32_t* nm = a.n_models + (size_t)p * a.max_iters + it;
    const bool direct = regime == kDirect && it == 0;
    if (!direct && it >= a.n_drawn[p]) {
        *nm = 0;
        return;
    }
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    const int32_t* idx = a.subsets + ((size_t)p * a.max_iters + it) * 7;
    float2 s1[7], s2[7];
    for (int k = 0; k < 7; ++k) {
        const int j = direct ? k : idx[k];
        s1[k] = m1[j];
        s2[k] = m2[j];
    }
    double F[27];
    const int n = seven_point(s1, s2, F);
    *nm = n;
    double* out = a.models + ((size_t)p * a.max_iters + it) * 27;
    for (int k = 0; k < n * 9 && k < 27; ++k) out[k] = F[k];
}

// FMEstimatorCallback::computeError for one correspondence
__device__ __forceinline__ float epipolar_error(const double (&F)[9], float2 p1, float2 p2) {
    const double x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// Step 3: one wave per (problem, iteration): inlier counts (RANSAC) or median error (LMedS)
// of its models.
__global__ __launch_bounds__(64) void fm_score_kernel(FmArgs a) {
    const int it = blockIdx.x, p = blockIdx.y, lane = threadIdx.x;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    const int n = a.n_models[(size_t)p * a.max_iters + it];
    if (n <= 0 || n > 3 || (regime != kRansac && regime != kLmeds)) return;
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    const double* src = a.models + ((size_t)p * a.max_iters + it) * 27;
    float* out = a.scores + ((size_t)p * a.max_iters + it) * 3;
    if (regime == kRansac) {
        double F[3][9];
        for (int m = 0; m < 3; ++m)
            for (int k = 0; k < 9; ++k) F[m][k] = m < n ? src[m * 9 + k] : 0.0;
        const float t = (float)(a.thresh * a.thresh);
        int good[3] = {0, 0, 0};
        for (int i = lane; i < count; i += 64) {
            const float2 p1 = m1[i], p2 = m2[i];
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m < n) good[m] += epipolar_error(F[m], p1, p2) <= t;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            int v = good[m];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && m < n) out[m] = (float)v;  // exact: counts are below 2^24
        }
    } else if (lane < n) {
        // std::nth_element(err, err + count / 2, err + count): count is at most 14
        double F[9];
        for (int k = 0; k < 9; ++k) F[k] = src[lane * 9 + k];
        float e[14];
        for (int i = 0; i < 14; ++i) e[i] = i < count ? epipolar_error(F, m1[i], m2[i]) : FLT_MAX;
        for (int i = 1; i < 14; ++i) {
            const float v = e[i];
            int j = i - 1;
            while (j >= 0 && e[j] > v) {
                e[j + 1] = e[j];
                --j;
            }
            e[j + 1] = v;
        }
        out[lane] = e[count / 2];
    }
}

// RANSACUpdateNumIters
__device__ int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0 ? p : 0;
    p = p < 1 ? p : 1;
    ep = ep > 0 ? ep : 0;
    ep = ep < 1 ? ep : 1;
    double num = 1 - p > DBL_MIN ? 1 - p : DBL_MIN;
    double denom = 1 - pow(1 - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// Step 4: the loop's bookkeeping, replayed in iteration order.
__global__ void fm_select_kernel(FmArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.n_problems) return;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    const int32_t* nm = a.n_models + (size_t)p * a.max_iters;
    const float* sc = a.scores + (size_t)p * a.max_iters * 3;
    const double* models = a.models + (size_t)p * a.max_iters * 27;
    double* best = a.best + (size_t)p * 10;
    int best_it = -1, best_m = 0, found = 0;
    double inlier_thresh = a.thresh;
    if (regime == kDirect) {
        found = nm[0] > 0;
        best_it = found ? 0 : -1;
    } else if (regime == kRansac) {
        const int drawn = a.n_drawn[p];
        int niters = a.max_iters > 1 ? a.max_iters : 1, max_good = 0;
        for (int it = 0; it < niters && it < drawn; ++it) {
            const int n = nm[it];
            if (n <= 0) continue;
            for (int m = 0; m < n && m < 3; ++m) {
                const int good = (int)sc[it * 3 + m];
                if (good > (max_good > 6 ? max_good : 6)) {
                    best_it = it;
                    best_m = m;
                    max_good = good;
                    niters = update_num_iters(a.confidence, (double)(count - good) / count, 7, niters);
                }
            }
        }
        found = max_good > 0;
    } else if (regime == kLmeds) {
        const int drawn = a.n_drawn[p];
        int niters = update_num_iters(a.confidence, 0.45, 7, a.max_iters);
        niters = niters > 3 ? niters : 3;
        double min_median = DBL_MAX;
        for (int it = 0; it < niters && it < drawn && it < a.max_iters; ++it) {
            const int n = nm[it];
            if (n <= 0) continue;
            for (int m = 0; m < n && m < 3; ++m) {
                const double median = sc[it * 3 + m];
                if (median < min_median) {
                    min_median = median;
                    best_it = it;
                    best_m = m;
                }
            }
        }
        if (best_it >= 0) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 7)) * sqrt(min_median);
            inlier_thresh = sigma > 0.001 ? sigma : 0.001;
            found = 1;  // provisional: the mask kernel requires at least 7 inliers
        }
    }
    for (int k = 0; k < 9; ++k) best[k] = best_it >= 0 ? models[(size_t)best_it * 27 + best_m * 9 + k] : 0.0;
    best[9] = inlier_thresh;
    a.found[p] = found;
    if (a.F)
        for (int k = 0; k < 9; ++k) a.F[(size_t)p * 9 + k] = best[k];
}

// Step 5: the mask of the chosen matrix (one workgroup per problem).
__global__ __launch_bounds__(256) void fm_mask_kernel(FmArgs a) {
    __shared__ int total;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    uint8_t* mask = a.mask + (size_t)p * a.stride;
    const int found = a.found[p];
    if (tid == 0) total = 0;
    __syncthreads();
    if (regime == kNone || (regime != kDirect && !found)) {
        // OpenCV leaves the caller's mask alone here; the reference hands in zeros
        for (int i = tid; i < count; i += blockDim.x) mask[i] = 0;
        return;
    }
    if (regime == kDirect) {
        for (int i = tid; i < count; i += blockDim.x) mask[i] = 1;
        return;
    }
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    double F[9];
    for (int k = 0; k < 9; ++k) F[k] = a.best[(size_t)p * 10 + k];
    const double th = a.best[(size_t)p * 10 + 9];
    const float t = (float)(th * th);
    int good = 0;
    for (int i = tid; i < count; i += blockDim.x) {
        const int f = epipolar_error(F, m1[i], m2[i]) <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    if (regime == kLmeds) {
        atomicAdd(&total, good);
        __syncthreads();
        if (tid == 0) a.found[p] = total >= 7;
    }
}

// Aligned point lists of every pair: pts[p][i] = key point of matches[p][i]
// (Source Files/feature_processing.cpp:164-165).
__global__ void fm_gather_kernel(const sfm_dmatch* __restrict__ matches, const int32_t* __restrict__ counts,
                                 int cap, const uint8_t* __restrict__ keypts, size_t frame_stride,
                                 size_t keypt_stride, const int32_t* __restrict__ pairs,
                                 float2* __restrict__ pts1, float2* __restrict__ pts2) {
    const int p = blockIdx.y;
    const int n = min(counts[p], cap);
    const uint8_t* k1 = keypts + (size_t)pairs[2 * p] * frame_stride;
    const uint8_t* k2 = keypts + (size_t)pairs[2 * p + 1] * frame_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const sfm_dmatch m = matches[(size_t)p * cap + i];
        pts1[(size_t)p * cap + i] = *reinterpret_cast<const float2*>(k1 + (size_t)m.queryIdx * keypt_stride);
        pts2[(size_t)p * cap + i] = *reinterpret_cast<const float2*>(k2 + (size_t)m.trainIdx * keypt_stride);
    }
}

// Keeps the matches whose mask entry is set, in order (Source Files/
// feature_processing.cpp:193-204 with useEpipolarFilter == true, then :223).  One workgroup
// per pair; in place: an element is only ever moved to a lower index, and a whole chunk is
// read before any of it is written.
__global__ __launch_bounds__(1024) void fm_compact_kernel(sfm_dmatch* __restrict__ matches,
                                                          int32_t* __restrict__ counts, int cap,
                                                          const uint8_t* __restrict__ mask) {
    __shared__ int wave_count[16];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(counts[p], cap);
    sfm_dmatch* mine = matches + (size_t)p * cap;
    const uint8_t* keep = mask + (size_t)p * cap;
    int running = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const bool ok = i < n && keep[i] != 0;
        sfm_dmatch m{};
        if (ok) m = mine[i];
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        __syncthreads();  // also: every read of this chunk is done
        int offset = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wave_count[w];
            offset += w < wave ? c : 0;
            total += c;
        }
        if (ok) mine[running + offset + prefix] = m;
        running += total;
        __syncthreads();
    }
    if (tid == 0) counts[p] = running;
}

struct FmBuffers {
    int32_t* subsets;
    int32_t* n_drawn;
    double* models;
    int32_t* n_models;
    float* scores;
    double* best;
};

FmBuffers carve_fm(Carver& c, int n_problems, int max_iters) {
    FmBuffers b;
    const size_t pi = (size_t)n_problems * max_iters;
    b.models = c.take<double>(pi * 27);
    b.best = c.take<double>((size_t)n_problems * 10);
    b.subsets = c.take<int32_t>(pi * 7);
    b.n_models = c.take<int32_t>(pi);
    b.scores = c.take<float>(pi * 3);
    b.n_drawn = c.take<int32_t>((size_t)n_problems);
    return b;
}

// The five steps for the point sets described by `a`.
int run_fm(sfm_ctx* ctx, const FmArgs& a) {
    const int n_problems = a.n_problems, max_iters = a.max_iters;
    {
        KernelTimer timer(ctx, KID_FM_SUBSETS);
        hipLaunchKernelGGL(fm_subsets_kernel, dim3(n_problems), dim3(64), 0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_FM_SOLVE);
        hipLaunchKernelGGL(fm_solve_kernel, dim3((max_iters + 63) / 64, n_problems), dim3(64), 0,
                           ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_FM_SCORE);
        hipLaunchKernelGGL(fm_score_kernel, dim3(max_iters, n_problems), dim3(64), 0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_FM_SELECT);
        hipLaunchKernelGGL(fm_select_kernel, dim3((n_problems + 63) / 64), dim3(64), 0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(fm_mask_kernel, dim3(n_problems), dim3(256), 0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    return SFM_OK;
}

}  // namespace

}  // namespace sfm

using namespace sfm;

extern "C" {

size_t sfm_fundamental_workspace(int n_problems, int max_iters) {
    if (n_problems <= 0 || max_iters <= 0) return 0;
    Carver c(nullptr);
    carve_fm(c, n_problems, max_iters);
    return c.total();
}

int sfm_fundamental_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                        const int32_t* d_counts, int count_uniform, int n_problems, double thresh,
                        double confidence, int max_iters, double* d_F, uint8_t* d_mask,
                        int32_t* d_found) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_problems < 0 || max_iters < 1 || max_iters > 100000)
        return set_error(ctx, SFM_E_INVALID, "bad problem or iteration count");
    if (n_problems == 0) return SFM_OK;
    if (!d_pts1 || !d_pts2 || !d_mask || !d_found) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (problem_stride == 0 || problem_stride > 0x7fffffff || (!d_counts && (count_uniform < 0 || (size_t)count_uniform > problem_stride)))
        return set_error(ctx, SFM_E_INVALID, "bad point count or stride");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // findFundamentalMat's own defaulting of out-of-range parameters
    if (thresh <= 0) thresh = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_fundamental_workspace(n_problems, max_iters), &ws));
    Carver c(ws);
    const FmBuffers b = carve_fm(c, n_problems, max_iters);
    FmArgs a{};
    a.pts1 = reinterpret_cast<const float2*>(d_pts1);
    a.pts2 = reinterpret_cast<const float2*>(d_pts2);
    a.stride = problem_stride;
    a.counts = d_counts;
    a.count_uniform = count_uniform;
    a.n_problems = n_problems;
    a.max_iters = max_iters;
    a.thresh = thresh;
    a.confidence = confidence;
    a.subsets = b.subsets;
    a.n_drawn = b.n_drawn;
    a.models = b.models;
    a.n_models = b.n_models;
    a.scores = b.scores;
    a.best = b.best;
    a.F = d_F;
    a.mask = d_mask;
    a.found = d_found;
    return run_fm(ctx, a);
}

int sfm_fundamental(sfm_ctx* ctx, const float* pts1, const float* pts2, int n, double thresh,
                    double confidence, int max_iters, double* F, uint8_t* mask, int* found) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (found) *found = 0;
    if (n == 0) return SFM_OK;
    if (!pts1 || !pts2 || !mask) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Carver sz(nullptr);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>(2 * (size_t)n);
    sz.take<double>(9);
    sz.take<uint8_t>((size_t)n);
    sz.take<int32_t>(4);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    float* d1 = c.take<float>(2 * (size_t)n);
    float* d2 = c.take<float>(2 * (size_t)n);
    double* dF = c.take<double>(9);
    uint8_t* dm = c.take<uint8_t>((size_t)n);
    int32_t* df = c.take<int32_t>(4);
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d2, pts2, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    SFM_RETURN_IF(sfm_fundamental_dev(ctx, d1, d2, (size_t)n, nullptr, n, 1, thresh, confidence, max_iters,
                                      dF, dm, df));
    int32_t h_found = 0;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(mask, dm, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (F) SFM_HIP_CHECK(ctx, hipMemcpyAsync(F, dF, sizeof(double) * 9, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&h_found, df, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (found) *found = h_found;
    return SFM_OK;
}

size_t sfm_epipolar_filter_pairs_workspace(int n_pairs, int cap, int max_iters) {
    if (n_pairs <= 0 || cap <= 0 || max_iters <= 0) return 0;
    Carver c(nullptr);
    c.take<float2>((size_t)n_pairs * cap);
    c.take<float2>((size_t)n_pairs * cap);
    c.take<uint8_t>((size_t)n_pairs * cap);
    c.take<int32_t>((size_t)n_pairs);
    return c.total() + sfm_fundamental_workspace(n_pairs, max_iters);
}

int sfm_epipolar_filter_pairs_dev(sfm_ctx* ctx, const void* d_keypts, size_t frame_stride,
                                  size_t keypt_stride, const int32_t* d_pairs, int n_pairs,
                                  sfm_dmatch* d_matches, int cap, int32_t* d_match_counts, double thresh,
                                  double confidence, int max_iters, uint8_t* d_mask) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_pairs < 0 || cap < 0 || max_iters < 1 || max_iters > 100000)
        return set_error(ctx, SFM_E_INVALID, "bad pair, capacity or iteration count");
    if (n_pairs == 0 || cap == 0) return SFM_OK;
    if (!d_keypts || !d_pairs || !d_matches || !d_match_counts)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (keypt_stride < 8 || (keypt_stride & 3) || (frame_stride & 3))
        return set_error(ctx, SFM_E_INVALID, "key point strides must be multiples of 4, at least 8");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (thresh <= 0) thresh = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    // one workspace: the point lists and the mask in front, the RANSAC buffers behind them
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_epipolar_filter_pairs_workspace(n_pairs, cap, max_iters), &ws));
    Carver c(ws);
    float2* pts1 = c.take<float2>((size_t)n_pairs * cap);
    float2* pts2 = c.take<float2>((size_t)n_pairs * cap);
    uint8_t* mask = c.take<uint8_t>((size_t)n_pairs * cap);
    int32_t* found = c.take<int32_t>((size_t)n_pairs);
    const FmBuffers b = carve_fm(c, n_pairs, max_iters);
    {
        KernelTimer timer(ctx, KID_FM_SELECT);
        hipLaunchKernelGGL(fm_gather_kernel, dim3((cap + 255) / 256, n_pairs), dim3(256), 0, ctx->stream,
                           d_matches, d_match_counts, cap, static_cast<const uint8_t*>(d_keypts), frame_stride,
                           keypt_stride, d_pairs, pts1, pts2);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    FmArgs a{};
    a.pts1 = pts1;
    a.pts2 = pts2;
    a.stride = (size_t)cap;
    a.counts = d_match_counts;
    a.n_problems = n_pairs;
    a.max_iters = max_iters;
    a.thresh = thresh;
    a.confidence = confidence;
    a.subsets = b.subsets;
    a.n_drawn = b.n_drawn;
    a.models = b.models;
    a.n_models = b.n_models;
    a.scores = b.scores;
    a.best = b.best;
    a.F = nullptr;
    a.mask = mask;
    a.found = found;
    SFM_RETURN_IF(run_fm(ctx, a));
    if (d_mask)
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(d_mask, mask, (size_t)n_pairs * cap, hipMemcpyDeviceToDevice,
                                          ctx->stream));
    KernelTimer timer(ctx, KID_FM_SELECT);
    hipLaunchKernelGGL(fm_compact_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, d_matches,
                       d_match_counts, cap, mask);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // extern "C"
