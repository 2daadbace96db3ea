// Relative pose from tracked points, first half: cv::findEssentialMat(prevPts, currPts, cameraK,
// method, prob, threshold, mask) of Tracking::findCameraPose, Source Files/tracking.cpp:94
// (five-point minimal solver inside OpenCV's RANSAC or LMedS point set registrator, at most
// 1000 iterations).
//
// The sequential loop is run in rounds of iterations [0, 64), [64, 256), [256, 1000], exactly as
// the epipolar filter's (fundamental_ransac.hip): one wave per point set replays the loop's
// bookkeeping over the scores of the round before and continues the random draws; the solves and
// the scoring of a round's live iterations run in parallel; a point set whose loop has ended
// costs later rounds nothing.  Per iteration:
//   em_solve_kernel   one lane per 5-point sample (work arrays in LDS): null space of the 5x9 epipolar system
//                     (Gauss-Jordan with complete pivoting + Gram-Schmidt), the ten cubic
//                     constraints det E = 0, E E^T E - tr(E E^T) E / 2 = 0 by polynomial
//                     products, elimination in Nister's monomial order, the determinant of the
//                     3x3 polynomial matrix B(z) (degree 10), its roots by Durand-Kerner from
//                     OpenCV's start values (stopped when the corrections fall below rounding
//                     instead of after a fixed 300 sweeps), (x, y) from the null vector of B(z);
//   em_score_kernel   Sampson distance of every correspondence to each of the (up to ten)
//                     matrices, inliers counted with integer atomics.
#include <cfloat>
#include <utility>
#include <cmath>

#include "sfm_internal.h"

namespace sfm {

namespace {

enum Method { kRansac = 0, kLmeds = 1 };
constexpr int kMaxModels = 10;
constexpr int kRawPerBatch = 512;
constexpr int kScoreSlice = 1024;

struct EmState {
    unsigned long long rng;
    double min_median;
    int32_t n_drawn, niters, max_good, best_it, best_m, processed, done, pad;
};

struct EmArgs {
    const float2* pix1;     // pixel coordinates, problem p at p * stride
    const float2* pix2;
    size_t stride;
    const int32_t* counts;  // per problem, or nullptr
    int count_uniform;
    int n_problems;
    int max_iters;
    int lo, hi;
    int method;
    double fx, fy, cx, cy;
    double thresh, confidence;  // thresh already divided by (fx + fy) / 2
    double2* m1;            // [problem][stride] normalised points
    double2* m2;
    EmState* state;
    int32_t* subsets;       // [problem][max_iters][5]
    double* models;         // [problem][max_iters][10][9]
    int32_t* n_models;      // [problem][max_iters]
    int32_t* scores;        // [problem][max_iters][10]: inlier counts, or bits of the float median
    double* E;              // [problem][9]
    uint8_t* mask;          // [problem][stride]
    int32_t* found;         // [problem]
};

__device__ __forceinline__ int problem_count(const EmArgs& a, int p) {
    const int c = a.counts ? a.counts[p] : a.count_uniform;
    return c < 0 ? 0 : (c > (int)a.stride ? (int)a.stride : c);
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

__global__ void em_prep_kernel(EmArgs a) {
    const int p = blockIdx.y;
    const int n = problem_count(a, p);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t k = (size_t)p * a.stride + i;
        const float2 u = a.pix1[k], v = a.pix2[k];
        a.m1[k] = make_double2(((double)u.x - a.cx) / a.fx, ((double)u.y - a.cy) / a.fy);
        a.m2[k] = make_double2(((double)v.x - a.cx) / a.fx, ((double)v.y - a.cy) / a.fy);
    }
}

// ---- the five-point solver -------------------------------------------------------------------

// Monomials of degree <= 3 in (x, y, z), Nister's elimination order:
// x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
struct Mono {
    int8_t e[20][3];
    int8_t lin[4];          // indices of x, y, z, 1
    int8_t times[20][4];    // monomial * (x, y, z, 1), -1 above degree 3
};
constexpr Mono make_mono() {
    Mono m{};
    const int8_t e[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                             {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                             {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
    for (int i = 0; i < 20; ++i)
        for (int k = 0; k < 3; ++k) m.e[i][k] = e[i][k];
    m.lin[0] = 12;
    m.lin[1] = 15;
    m.lin[2] = 18;
    m.lin[3] = 19;
    for (int i = 0; i < 20; ++i)
        for (int v = 0; v < 4; ++v) {
            int t[3] = {e[i][0], e[i][1], e[i][2]};
            if (v < 3) t[v] += 1;
            m.times[i][v] = -1;
            for (int j = 0; j < 20; ++j)
                if (e[j][0] == t[0] && e[j][1] == t[1] && e[j][2] == t[2]) m.times[i][v] = (int8_t)j;
        }
    return m;
}
constexpr Mono kM = make_mono();  // read at compile time only: every loop over monomials is unrolled
constexpr int mono_degree(int i) { return kM.e[i][0] + kM.e[i][1] + kM.e[i][2]; }
// the six terms of a 3 x 3 determinant: column of rows 0, 1, 2 and the sign
constexpr int kPerm[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {0, 2, 1, -1}, {1, 0, 2, -1}, {2, 1, 0, -1}};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose counter is a
// compile-time constant inside the body
template <class F, int... I>
__device__ __forceinline__ void unroll_seq(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void unroll(F&& f) {
    unroll_seq(f, std::make_integer_sequence<int, N>{});
}

// A workgroup is one wave and solves four samples, one per row of 16 lanes.  The solver's work
// arrays live in LDS, one column per sample: element i of a sample's array sits kSolveRows
// doubles after element i - 1 (scratch memory made every one of the solver's few thousand
// dependent array accesses an L2 round trip).
constexpr int kSolveRows = 4;  // samples per workgroup
struct Lds {
    double* p;
    __device__ __forceinline__ double& operator[](int i) const { return p[i * kSolveRows]; }
    __device__ __forceinline__ Lds at(int offset) const { return {p + offset * kSolveRows}; }
};
// doubles per sample
constexpr int kSolveQ = 45, kSolveEe = 36, kSolveLam = 180, kSolveA = 200, kSolveB = 39;
constexpr int kSolveDoubles = kSolveQ + kSolveEe + kSolveLam + kSolveA + kSolveB;

// out += scale * a * l on registers: l a linear polynomial given by its (x, y, z, 1)
// coefficients, a a polynomial of total degree <= DEG (its other coefficients are zero by
// construction and are skipped); NEG: scale = -1.  Terms are added monomial by monomial of a,
// then coefficient by coefficient of l.
template <int DEG, bool NEG>
__device__ __forceinline__ void mul_linear_acc_reg(const double (&a)[20], const double (&l)[4], double (&out)[20]) {
    unroll<20>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if constexpr (mono_degree(P) <= DEG) {
            const double ap = NEG ? -a[P] : a[P];
            unroll<4>([&](auto vc) {
                constexpr int V = decltype(vc)::value;
                constexpr int T = kM.times[P][V];
                if constexpr (T >= 0) out[T] += ap * l[V];
            });
        }
    });
}

struct Cplx {
    double re, im;
};
__device__ __forceinline__ Cplx c_mul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// cv::solvePoly (Durand-Kerner), generic degree: the rare case of a vanishing leading
// coefficient.  coeffs lowest degree first.
__device__ void solve_poly_generic(const double* coeffs, int n, Cplx* roots) {
    Cplx p = {1, 0};
    const Cplx r = {1, 1};
    for (int i = 0; i < n; i++) {
        roots[i] = p;
        p = c_mul(p, r);
    }
    for (int iter = 0; iter < 300; iter++) {
        double max_rel = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            Cplx num = {coeffs[n], 0}, denom = {coeffs[n], 0};
            for (int j = 0; j < n; j++) {
                num = c_mul(num, p);
                num.re += coeffs[n - j - 1];
                if (j != i) {
                    const Cplx d = {p.re - roots[j].re, p.im - roots[j].im};
                    if (d.re != 0 || d.im != 0) denom = c_mul(denom, d);
                }
            }
            const double t = 1. / (denom.re * denom.re + denom.im * denom.im);
            const Cplx q = {(num.re * denom.re + num.im * denom.im) * t, (-num.re * denom.im + num.im * denom.re) * t};
            roots[i] = {p.re - q.re, p.im - q.im};
            const double rel = sqrt(q.re * q.re + q.im * q.im) / (1.0 + sqrt(p.re * p.re + p.im * p.im));
            max_rel = rel > max_rel ? rel : max_rel;
        }
        if (!(max_rel > 4e-16)) break;
    }
}

// Value of lane I of the caller's row of 16 lanes, in every lane of the row.  All 64 lanes
// must execute it (a lane that is switched off is read as zero).
template <int I>
__device__ __forceinline__ double row_bcast(double x) {
    constexpr int kPattern = (I << 5) | 0x10;  // lane = (lane & 0x10) | I within each half wave
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), kPattern);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), kPattern);
    return __hiloint2double(hi, lo);
}
// max over the row, in every lane of the row
__device__ __forceinline__ double row_max(double x) {
#define SFM_ROW_MAX_STEP(XOR)                                                                     \
    {                                                                                             \
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), ((XOR) << 10) | 0x1F);      \
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), ((XOR) << 10) | 0x1F);      \
        const double o = __hiloint2double(hi, lo);                                                \
        x = o > x ? o : x;                                                                        \
    }
    SFM_ROW_MAX_STEP(1)
    SFM_ROW_MAX_STEP(2)
    SFM_ROW_MAX_STEP(4)
    SFM_ROW_MAX_STEP(8)
#undef SFM_ROW_MAX_STEP
    return x;
}

// cv::solvePoly for degree 10 with a leading coefficient that does not vanish, one polynomial
// per row of 16 lanes: lane `sub` < 10 of a row owns root `sub` (the other lanes run along).
// solvePoly updates the roots one after the other inside a sweep, each from the roots already
// updated; that order is kept: the ten Horner evaluations of a sweep (each needs only its own
// root) run side by side, and the product of differences of root i takes the factors of roots
// 0 .. i-1 one by one as these are finished, then, alone, the factors of the roots after it.
// Every root sees the same operations in the same order as in the sequential loop.
// OpenCV sweeps until a sweep changes nothing at all (in practice all 300 times); here the
// sweeps stop once the corrections are rounding noise (below 1e-11 relative and no longer
// shrinking).  The size of a sweep's corrections is measured as max |q|^2 / (1 + |p|^2) over
// the roots p and their corrections q -- squares and a reciprocal estimate instead of two square
// roots and a division per root: the quantity only steers when to stop.
// live: the row has a polynomial (row-uniform).  All 64 lanes must call.
__device__ void solve_poly10_rows(const double (&c)[11], int sub, bool live, Cplx& root) {
    {
        Cplx p = {1, 0};
        const Cplx r = {1, 1};
        root = {0, 0};
#pragma unroll
        for (int i = 0; i < 10; i++) {
            if (i == sub) root = p;
            p = c_mul(p, r);
        }
    }
    bool active = live;
    double prev_r2 = DBL_MAX;
    for (int iter = 0; iter < 300; iter++) {
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        const Cplx p = root;
        Cplx num = {c[10], 0};
#pragma unroll
        for (int j = 0; j < 10; j++) {
            num = c_mul(num, p);
            num.re += c[9 - j];
        }
        Cplx old[10];
        unroll<10>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            old[J] = {row_bcast<J>(p.re), row_bcast<J>(p.im)};
        });
        Cplx fresh[10];         // the roots of this sweep as they are finished
        Cplx denom = {c[10], 0};  // lane sub: c10 times the factors of the finished roots so far
        Cplx q = {0, 0};
        unroll<10>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            // what lane I needs now: the factors of the roots after it (every lane computes on
            // its own values; only lane I's result is used)
            Cplx dn = denom;
#pragma unroll
            for (int j = I + 1; j < 10; j++) dn = c_mul(dn, Cplx{p.re - old[j].re, p.im - old[j].im});
            // solvePoly leaves a difference that is exactly zero out of the product.  With such
            // a factor the plain product is exactly zero (or not a number, after an overflow):
            // only then are the factors tested one by one.
            const bool redo = sub == I && active && !(fabs(dn.re) + fabs(dn.im) > 0);
            if (__builtin_amdgcn_ballot_w64(redo) != 0) {
                Cplx d2 = {c[10], 0};
#pragma unroll
                for (int j = 0; j < 10; j++) {
                    if (j == I) continue;
                    const Cplx rj = j < I ? fresh[j] : old[j];
                    const Cplx d = {p.re - rj.re, p.im - rj.im};
                    if (d.re != 0 || d.im != 0) d2 = c_mul(d2, d);
                }
                if (redo) dn = d2;
            }
            const double t = 1. / (dn.re * dn.re + dn.im * dn.im);
            const Cplx qi = {(num.re * dn.re + num.im * dn.im) * t, (-num.re * dn.im + num.im * dn.re) * t};
            const Cplx ri = {p.re - qi.re, p.im - qi.im};
            if (sub == I) {
                q = qi;
                if (active) root = ri;
            }
            // the row learns root I; the lanes after it take its factor (the lanes before it no
            // longer need their product)
            fresh[I] = {row_bcast<I>(ri.re), row_bcast<I>(ri.im)};
            denom = c_mul(denom, Cplx{p.re - fresh[I].re, p.im - fresh[I].im});
        });
        double r2 = (q.re * q.re + q.im * q.im) * __builtin_amdgcn_rcp(1.0 + (p.re * p.re + p.im * p.im));
        r2 = sub < 10 && r2 > 0 ? r2 : 0.0;  // also drops a correction that is not a number, as a running maximum does
        const double max_r2 = row_max(r2);
        // converged: the corrections are at the rounding floor of the Horner evaluation
        // (they stop shrinking once they are tiny) -- or vanish altogether
        if (!(max_r2 > 1.6e-31) || (max_r2 < 1e-22 && max_r2 > 0.0625 * prev_r2)) active = false;
        prev_r2 = max_r2;
    }
}

// Where the front half of the solver leaves its results for the row (the Lambda area is free
// by then): the determinant polynomial, or the ten roots when its leading coefficient vanished.
constexpr int kSolveCz = kSolveQ + kSolveEe, kSolveRz = kSolveCz + 11;

// EMEstimatorCallback::runKernel for five normalised correspondences, first half, run by one
// lane per sample: everything up to the degree-10 polynomial in z, whose coefficients go to
// LDS (lowest degree first).  Returns 0: no solution, 1: polynomial stored, 2: the leading
// coefficient vanished and the roots themselves are stored (generic root finder).
// `work` is this sample's column of the workgroup's LDS.
__device__ int five_point_front(const double2* s1, const double2* s2, Lds work) {
    const Lds q = work, ee = q.at(kSolveQ), lam = ee.at(kSolveEe), a = lam.at(kSolveLam), b = a.at(kSolveA);
    for (int i = 0; i < 5; ++i) {
        const double x1 = s1[i].x, y1 = s1[i].y, x2 = s2[i].x, y2 = s2[i].y;
        const Lds r = q.at(9 * i);
        r[0] = x2 * x1;
        r[1] = x2 * y1;
        r[2] = x2;
        r[3] = y2 * x1;
        r[4] = y2 * y1;
        r[5] = y2;
        r[6] = x1;
        r[7] = y1;
        r[8] = 1.0;
    }
    // null space: Gauss-Jordan with complete pivoting
    int col_of[5];
    unsigned used = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int pr = -1, pc = -1;
        double best = 0;
        for (int r = k; r < 5; ++r)
            for (int c = 0; c < 9; ++c) {
                if ((used >> c) & 1u) continue;
                const double v = fabs(q[r * 9 + c]);
                if (v > best) best = v, pr = r, pc = c;
            }
        if (pr < 0) return 0;
        if (pr != k)
            for (int c = 0; c < 9; ++c) {
                const double t = q[k * 9 + c];
                q[k * 9 + c] = q[pr * 9 + c];
                q[pr * 9 + c] = t;
            }
        used |= 1u << pc;
        col_of[k] = pc;
        const double inv = 1.0 / q[k * 9 + pc];
        for (int c = 0; c < 9; ++c) q[k * 9 + c] *= inv;
        for (int r = 0; r < 5; ++r) {
            if (r == k) continue;
            const double m = q[r * 9 + pc];
            if (m == 0) continue;
            for (int c = 0; c < 9; ++c) q[r * 9 + c] -= m * q[k * 9 + c];
        }
    }
    // ee[v * 9 + k]: E = x ee[0] + y ee[1] + z ee[2] + ee[3]
    {
        int nf = 0;
        for (int c = 0; c < 9 && nf < 4; ++c) {
            if ((used >> c) & 1u) continue;
            const Lds f = ee.at(9 * nf++);
            for (int j = 0; j < 9; ++j) f[j] = 0;
            f[c] = 1.0;
#pragma unroll
            for (int k = 0; k < 5; ++k) f[col_of[k]] = -q[k * 9 + c];
        }
        for (int v = 0; v < 4; ++v) {  // modified Gram-Schmidt
            for (int u = 0; u < v; ++u) {
                double dot = 0;
                for (int j = 0; j < 9; ++j) dot += ee[v * 9 + j] * ee[u * 9 + j];
                for (int j = 0; j < 9; ++j) ee[v * 9 + j] -= dot * ee[u * 9 + j];
            }
            double n2 = 0;
            for (int j = 0; j < 9; ++j) n2 += ee[v * 9 + j] * ee[v * 9 + j];
            if (!(n2 > 0)) return 0;
            const double inv = 1.0 / sqrt(n2);
            for (int j = 0; j < 9; ++j) ee[v * 9 + j] *= inv;
        }
    }
    // From here to the root finder everything is straight-line code on registers with every
    // monomial index a compile-time constant; sums are formed in the order of the loops that
    // define them (terms whose factor is zero by construction are left out: they add nothing).
    // entries of E as linear polynomials: lin[k][v] = (x, y, z, 1) coefficient of entry k
    double lin[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) lin[k][v] = ee[v * 9 + k];
    // Lambda = E E^T - tr(E E^T) / 2 I: nine quadratics, kept in LDS (lam[20 * entry + monomial])
    unroll<9>([&](auto rc) {
        constexpr int R = decltype(rc)::value / 3, C = decltype(rc)::value % 3;
        double o[20] = {};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double l1[20] = {};
#pragma unroll
            for (int v = 0; v < 4; ++v) l1[kM.lin[v]] = lin[3 * R + k][v];
            mul_linear_acc_reg<1, false>(l1, lin[3 * C + k], o);
        }
        unroll<20>([&](auto mc) {
            constexpr int M = decltype(mc)::value;
            if constexpr (mono_degree(M) <= 2) lam[20 * (3 * R + C) + M] = o[M];
        });
    });
    unroll<20>([&](auto mc) {
        constexpr int M = decltype(mc)::value;
        if constexpr (mono_degree(M) <= 2) {
            const double half = 0.5 * (lam[M] + lam[80 + M] + lam[160 + M]);
            lam[M] -= half;
            lam[80 + M] -= half;
            lam[160 + M] -= half;
        }
    });
    // ten cubic equations, rows of a: det E, then the nine entries of Lambda E
    {
        double row[20] = {};
        unroll<6>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            double l1[20] = {}, t20[20] = {};
#pragma unroll
            for (int v = 0; v < 4; ++v) l1[kM.lin[v]] = lin[kPerm[P][0]][v];
            mul_linear_acc_reg<1, false>(l1, lin[3 + kPerm[P][1]], t20);
            mul_linear_acc_reg<2, (kPerm[P][3] < 0)>(t20, lin[6 + kPerm[P][2]], row);
        });
#pragma unroll
        for (int m = 0; m < 20; ++m) a[m] = row[m];
    }
    unroll<9>([&](auto rc) {
        constexpr int R = decltype(rc)::value / 3, C = decltype(rc)::value % 3;
        double row[20] = {};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double lq[20] = {};
            unroll<20>([&](auto mc) {
                constexpr int M = decltype(mc)::value;
                if constexpr (mono_degree(M) <= 2) lq[M] = lam[20 * (3 * R + k) + M];
            });
            mul_linear_acc_reg<2, false>(lq, lin[3 * k + C], row);
        }
#pragma unroll
        for (int m = 0; m < 20; ++m) a[20 * (1 + 3 * R + C) + m] = row[m];
    });
    // A[:, :10]^-1 A[:, 10:] by Gauss-Jordan with partial pivoting.  Columns left of the pivot
    // are already zero in the pivot row and the pivot column is not read again, so step k
    // touches columns k+1 .. 19 only; a row with a zero multiplier is updated by zeros.
    bool singular = false;
    unroll<10>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        int pr = K;
        double best = fabs(a[K * 20 + K]);
#pragma unroll
        for (int r = K + 1; r < 10; ++r) {
            const double v = fabs(a[r * 20 + K]);
            if (v > best) best = v, pr = r;
        }
        singular |= best == 0;
        double pk[20], rk[20];
#pragma unroll
        for (int c = K; c < 20; ++c) {
            pk[c] = a[pr * 20 + c];
            rk[c] = a[K * 20 + c];
        }
        const double inv = 1.0 / pk[K];
#pragma unroll
        for (int c = K + 1; c < 20; ++c) {
            a[pr * 20 + c] = rk[c];       // the swap (a no-op when pr == K: overwritten next)
            pk[c] *= inv;
        }
        a[pr * 20 + K] = rk[K];
#pragma unroll
        for (int c = K + 1; c < 20; ++c) a[K * 20 + c] = pk[c];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == K) continue;
            const double m = a[r * 20 + K];
#pragma unroll
            for (int c = K + 1; c < 20; ++c) a[r * 20 + c] -= m * pk[c];
        }
    });
    if (singular) return 0;
    // B(z): bm[i][k], columns x z^3, x z^2, x z, x | y z^3, y z^2, y z, y | z^4, z^3, z^2, z, 1
    double bm[3][13];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double r1[10], r2[10], row1[13] = {}, row2[13] = {};
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            r1[k] = a[(4 + 2 * i) * 20 + 10 + k];
            r2[k] = a[(5 + 2 * i) * 20 + 10 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            row1[1 + k] = r1[k];
            row1[5 + k] = r1[3 + k];
            row2[k] = r2[k];
            row2[4 + k] = r2[3 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            row1[9 + k] = r1[6 + k];
            row2[8 + k] = r2[6 + k];
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            bm[i][k] = row1[k] - row2[k];
            b[i * 13 + k] = bm[i][k];
        }
    }
    double c[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) c[k] = 0;
    {
        // pz[i][j][k]: entry (i, j) of B as a polynomial in z, lowest degree first
        double pz[3][3][5];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 5; ++k) pz[i][0][k] = pz[i][1][k] = pz[i][2][k] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pz[i][0][k] = bm[i][3 - k];
                pz[i][1][k] = bm[i][7 - k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) pz[i][2][k] = bm[i][12 - k];
        }
        unroll<6>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            double t[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) t[u] = 0;
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) t[u + v] += pz[0][kPerm[P][0]][u] * pz[1][kPerm[P][1]][v];
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v)
                    if (u + v <= 10) c[u + v] += kPerm[P][3] * t[u] * pz[2][kPerm[P][2]][v];
        });
    }
    int n = 10;
    for (; n > 1; n--)
        if (fabs(c[n]) > DBL_EPSILON) break;
    if (n < 10) {
        Cplx roots[10];
        solve_poly_generic(c, n, roots);
        for (; n < 10; n++) roots[n] = roots[n - 1];
        for (int i = 0; i < 10; ++i) {
            work[kSolveRz + 2 * i] = roots[i].re;
            work[kSolveRz + 2 * i + 1] = roots[i].im;
        }
        return 2;
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) work[kSolveCz + k] = c[k];
    return 1;
}

// Second half, one lane per root z (real part z1, imaginary part z_im): (x, y) from the null
// vector of B(z), the matrix, its normalisation.  Returns whether the root gives a matrix.
__device__ bool five_point_model(double z1, double z_im, Lds work, double (&e9)[9]) {
    const Lds ee = work.at(kSolveQ), b = work.at(kSolveQ + kSolveEe + kSolveLam + kSolveA);
    bool ok = !(fabs(z_im) > 1e-10);
    const double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    double bz[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const Lds br = b.at(13 * j);
        bz[j][0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
        bz[j][1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
        bz[j][2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
    }
    double best[3] = {0, 0, 0}, best_n2 = -1;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const double* r0 = bz[u];
        const double* r1 = bz[(u + 1) % 3];
        const double v0 = r0[1] * r1[2] - r0[2] * r1[1], v1 = r0[2] * r1[0] - r0[0] * r1[2],
                     v2 = r0[0] * r1[1] - r0[1] * r1[0];
        const double n2 = v0 * v0 + v1 * v1 + v2 * v2;
        if (n2 > best_n2) best_n2 = n2, best[0] = v0, best[1] = v1, best[2] = v2;
    }
    ok = ok && best_n2 > 0;
    ok = ok && !(fabs(best[2] * (1.0 / sqrt(best_n2))) < 1e-10);
    const double x = best[0] / best[2], y = best[1] / best[2];
    double n2 = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e9[k] = ee[k] * x + ee[9 + k] * y + ee[18 + k] * z1 + ee[27 + k];
        n2 += e9[k] * e9[k];
    }
    const double inv = 1.0 / sqrt(n2);
#pragma unroll
    for (int k = 0; k < 9; ++k) e9[k] *= inv;
    return ok;
}

// ---- rounds ----------------------------------------------------------------------------------

__device__ EmState initial_state(const EmArgs& a, int count) {
    EmState st;
    st.rng = ~0ULL;
    st.min_median = DBL_MAX;
    st.n_drawn = 0;
    st.niters = a.max_iters > 1 ? a.max_iters : 1;
    if (a.method == kLmeds) {
        const int n = update_num_iters(a.confidence, 0.45, 5, a.max_iters);
        st.niters = n > 3 ? n : 3;
        if (st.niters > a.max_iters) st.niters = a.max_iters;
    }
    st.max_good = 0;
    st.best_it = -1;
    st.best_m = 0;
    st.processed = 0;
    st.done = count <= 5;  // fewer than five: nothing; exactly five: one direct solve
    st.pad = 0;
    return st;
}

// The 5-point samples up to iteration a.hi: cv::RNG advanced by one lane, the reduction to
// indices and the redrawing of repeated picks for 64 samples in parallel (see the epipolar
// filter for the placement argument).  No degeneracy test: every sample is used.
__device__ void draw_round(const EmArgs& a, int p, int count, EmState& st, int lane, unsigned long long* raw,
                           int32_t* pick_of) {
    if (st.done) return;
    int32_t* out = a.subsets + (size_t)p * a.max_iters * 5;
    const int target = min(min(a.hi, st.niters), a.max_iters);
    while (st.n_drawn < target) {
        if (lane == 0) {
            unsigned long long x = st.rng;
            for (int i = 0; i < kRawPerBatch; ++i) {
                x = (unsigned long long)(unsigned)x * 4164903690ULL + (unsigned)(x >> 32);
                raw[i] = x;
            }
        }
        __syncthreads();
        for (int i = lane; i < kRawPerBatch; i += 64) pick_of[i] = (int32_t)((unsigned)raw[i] % (unsigned)count);
        __syncthreads();
        int idx[5];
        int shift = 0, extra = 0;
        bool complete = true;
        for (;;) {
            const int pos = 5 * lane + shift;
            extra = 0;
            complete = true;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int pick = 0;
                for (;;) {
                    const int q = pos + i + extra;
                    if (q >= kRawPerBatch) {
                        complete = false;
                        break;
                    }
                    pick = pick_of[q];
                    bool dup = false;
#pragma unroll
                    for (int k = 0; k < i; ++k) dup |= idx[k] == pick;
                    if (!dup) break;
                    ++extra;
                }
                idx[i] = pick;
            }
            if (!complete) extra = 0;
            int incl = extra;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const int placed = incl - extra;
            const bool moved = placed != shift;
            shift = placed;
            if (!__any(moved)) break;
        }
        const unsigned long long incomplete = __ballot(!complete);
        const int n_att = incomplete ? __ffsll((long long)incomplete) - 1 : 64;
        if (n_att == 0) {  // five distinct picks not found in 512 draws: cannot happen with count >= 6
            st.done = 1;
            break;
        }
        const int slot = st.n_drawn + lane;
        if (lane < n_att && slot < a.max_iters) {
#pragma unroll
            for (int k = 0; k < 5; ++k) out[slot * 5 + k] = idx[k];
        }
        st.n_drawn = min(st.n_drawn + n_att, a.max_iters);
        const int used = __shfl(5 * lane + shift + 5 + extra, n_att - 1);
        st.rng = raw[used - 1];
        __syncthreads();
    }
}

// The loop's bookkeeping over the scored iterations [st.processed, hi), in order: the lanes
// find the (iteration, model) pairs that beat everything before them; only those are applied.
__device__ void select_round(const EmArgs& a, int p, int count, EmState& st, int lane, int hi) {
    if (st.done) return;
    const int32_t* nm = a.n_models + (size_t)p * a.max_iters;
    const int32_t* sc = a.scores + (size_t)p * a.max_iters * kMaxModels;
    const int end = min(hi, st.n_drawn);
    const bool ransac = a.method == kRansac;
    for (int base = st.processed; base < end && base < st.niters; base += 64) {
        const int it = base + lane;
        int n = 0;
        double merit = -DBL_MAX;  // larger is better
        if (it < end) {
            n = nm[it];
            n = n < 0 ? 0 : (n > kMaxModels ? kMaxModels : n);
            for (int m = 0; m < n; ++m) {
                const int32_t s = sc[it * kMaxModels + m];
                const double v = ransac ? (double)s : -(double)__int_as_float(s);
                merit = v > merit ? v : merit;
            }
        }
        double before = merit;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double v = __shfl_up(before, o);
            if (lane >= o) before = v > before ? v : before;
        }
        before = __shfl_up(before, 1);
        const double carried = ransac ? (double)(st.max_good > 4 ? st.max_good : 4) : -st.min_median;
        if (lane == 0 || !(before > carried)) before = carried;
        unsigned long long events = __ballot(n > 0 && merit > before);
        bool cut = false;
        while (events) {
            const int l = __ffsll((long long)events) - 1;
            events &= events - 1;
            const int it_l = base + l;
            if (it_l >= st.niters) {
                cut = true;
                break;
            }
            const int n_l = min(max(nm[it_l], 0), kMaxModels);
            for (int m = 0; m < n_l; ++m) {
                const int32_t s = sc[it_l * kMaxModels + m];
                if (ransac) {
                    if (s > (st.max_good > 4 ? st.max_good : 4)) {
                        st.best_it = it_l;
                        st.best_m = m;
                        st.max_good = s;
                        st.niters = update_num_iters(a.confidence, (double)(count - s) / count, 5, st.niters);
                    }
                } else {
                    const double median = __int_as_float(s);
                    if (median < st.min_median) {
                        st.min_median = median;
                        st.best_it = it_l;
                        st.best_m = m;
                    }
                }
            }
        }
        if (cut) break;
    }
    st.processed = min(end, st.niters);
    st.done = st.processed >= st.niters || hi >= a.max_iters;
}

__global__ __launch_bounds__(64) void em_advance_kernel(EmArgs a) {
    __shared__ unsigned long long raw[kRawPerBatch];
    __shared__ int32_t pick_of[kRawPerBatch];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int count = problem_count(a, p);
    EmState st;
    if (a.lo == 0) {
        st = initial_state(a, count);
    } else {
        st = a.state[p];
        if (st.done) return;
        select_round(a, p, count, st, lane, a.lo);
    }
    draw_round(a, p, count, st, lane, raw, pick_of);
    if (lane == 0) a.state[p] = st;
}

__device__ __forceinline__ int live_end(const EmArgs& a, int p, int count) {
    if (count == 5) return a.lo == 0 ? 1 : a.lo;
    const EmState* st = a.state + p;
    if (count < 5 || st->done) return a.lo;
    return max(a.lo, min(a.hi, min(st->niters, st->n_drawn)));
}

// EMEstimatorCallback::computeError (Sampson distance) for one correspondence
__device__ __forceinline__ float sampson(const double (&E)[9], double2 p1, double2 p2) {
    const double ex0 = E[0] * p1.x + E[1] * p1.y + E[2] * 1., ex1 = E[3] * p1.x + E[4] * p1.y + E[5] * 1.,
                 ex2 = E[6] * p1.x + E[7] * p1.y + E[8] * 1.;
    const double et0 = E[0] * p2.x + E[3] * p2.y + E[6] * 1., et1 = E[1] * p2.x + E[4] * p2.y + E[7] * 1.;
    const double x2tex1 = p2.x * ex0 + p2.y * ex1 + 1. * ex2;
    const double a = ex0 * ex0, b = ex1 * ex1, c = et0 * et0, d = et1 * et1;
    return (float)(x2tex1 * x2tex1 / (a + b + c + d));
}

// The 5-point solves of a round's live iterations, four per workgroup of one wave: a sample
// has a row of 16 lanes.  Lane 0 of the row runs the front half (a chain of dependent work on
// one sample); the ten roots are found by ten lanes together; then every lane turns its root
// into a matrix and the row writes the matrices of the real roots in root order.
__global__ __launch_bounds__(64) void em_solve_kernel(EmArgs a) {
    __shared__ double s_work[kSolveDoubles * kSolveRows];
    const int p = blockIdx.y;
    const int count = problem_count(a, p);
    const int end = live_end(a, p, count);
    const double2* m1 = a.m1 + (size_t)p * a.stride;
    const double2* m2 = a.m2 + (size_t)p * a.stride;
    const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const Lds work = {s_work + row};
    for (int first = a.lo + blockIdx.x * kSolveRows; first < end; first += gridDim.x * kSolveRows) {
        const int it = first + row;
        const bool live = it < end;
        const size_t slot = (size_t)p * a.max_iters + (live ? it : first);
        if (sub == 0) {
            int flag = 0;
            if (live) {
                double2 s1[5], s2[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int j = count == 5 ? k : a.subsets[slot * 5 + k];
                    s1[k] = m1[j];
                    s2[k] = m2[j];
                }
                flag = five_point_front(s1, s2, work);
            }
            work[kSolveCz + 31] = (double)flag;
        }
        __syncthreads();
        const int flag = (int)work[kSolveCz + 31];
        double c[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) c[k] = flag == 1 ? work[kSolveCz + k] : (k == 10 ? 1.0 : 0.0);
        Cplx root;
        solve_poly10_rows(c, sub, flag == 1, root);
        if (flag == 2 && sub < 10) root = {work[kSolveRz + 2 * sub], work[kSolveRz + 2 * sub + 1]};
        double e9[9];
        const bool ok = five_point_model(root.re, root.im, work, e9) && flag != 0 && sub < 10;
        const unsigned row_ok = (unsigned)(__builtin_amdgcn_ballot_w64(ok) >> (16 * row)) & 0xFFFFu;
        if (ok) {
            double* out = a.models + slot * kMaxModels * 9 + 9 * __popc(row_ok & ((1u << sub) - 1u));
#pragma unroll
            for (int k = 0; k < 9; ++k) out[k] = e9[k];
        }
        if (live && sub == 0) a.n_models[slot] = __popc(row_ok);
        if (live && sub < kMaxModels) a.scores[slot * kMaxModels + sub] = 0;
        __syncthreads();  // the next sample's front half overwrites the work arrays
    }
}

// Median error of every model of the round's live iterations (LMedS): one wave per model.
// std::nth_element(err, err + count / 2, err + count) is the element of rank count / 2; its
// bits are built from the top: the largest value v with at most count / 2 errors below it.
// (Errors are non-negative floats, so their bit patterns order as the values do.)
__global__ __launch_bounds__(64) void em_median_kernel(EmArgs a) {
    const int p = blockIdx.y, lane = threadIdx.x;
    const int count = problem_count(a, p);
    if (a.method != kLmeds || count <= 5) return;
    const int end = live_end(a, p, count);
    const double2* m1 = a.m1 + (size_t)p * a.stride;
    const double2* m2 = a.m2 + (size_t)p * a.stride;
    for (int job = blockIdx.x; job < (end - a.lo) * kMaxModels; job += gridDim.x) {
        const int it = a.lo + job / kMaxModels, m = job % kMaxModels;
        const size_t slot = (size_t)p * a.max_iters + it;
        if (m >= a.n_models[slot]) continue;
        double E[9];
        for (int k = 0; k < 9; ++k) E[k] = a.models[(slot * kMaxModels + m) * 9 + k];
        const int want = count / 2;
        unsigned result = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = result | (1u << bit);
            int below = 0;
            for (int i = lane; i < count; i += 64) below += __float_as_uint(sampson(E, m1[i], m2[i])) < cand;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
            if (below <= want) result = cand;
        }
        if (lane == 0) a.scores[slot * kMaxModels + m] = (int32_t)result;
    }
}

// Inlier counts (RANSAC) of the models of each live iteration: one wave per slice of the
// correspondences, integer atomics.
__global__ __launch_bounds__(256) void em_score_kernel(EmArgs a) {
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = blockIdx.z * 4 + (threadIdx.x >> 6);
    const int count = problem_count(a, p);
    if (a.method != kRansac || count <= 5 || wave * kScoreSlice >= count) return;
    const int it_end = live_end(a, p, count);
    const double2* m1 = a.m1 + (size_t)p * a.stride;
    const double2* m2 = a.m2 + (size_t)p * a.stride;
    const float t = (float)(a.thresh * a.thresh);
    for (int it = a.lo + blockIdx.x; it < it_end; it += gridDim.x) {
        const size_t slot = (size_t)p * a.max_iters + it;
        const int n = min(a.n_models[slot], kMaxModels);
        for (int m = 0; m < n; ++m) {
            double E[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) E[k] = a.models[(slot * kMaxModels + m) * 9 + k];
            int good = 0;
            for (int base = wave * kScoreSlice; base < count; base += gridDim.z * 4 * kScoreSlice) {
                const int end = min(base + kScoreSlice, count);
                for (int i = base + lane; i < end; i += 64) good += sampson(E, m1[i], m2[i]) <= t;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) good += __shfl_xor(good, o);
            if (lane == 0 && good) atomicAdd(a.scores + slot * kMaxModels + m, good);
        }
    }
}

// The last bookkeeping, the outcome and the mask.  One workgroup per point set.
// found[p]: number of 3x3 matrices in E (1; several for exactly five points; 0 none).
__global__ __launch_bounds__(256) void em_finish_kernel(EmArgs a) {
    __shared__ int s_found;
    __shared__ double s_best[10];
    __shared__ int s_good;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int count = problem_count(a, p);
    uint8_t* mask = a.mask + (size_t)p * a.stride;
    if (tid < 64) {
        EmState st = a.state[p];
        if (count > 5) select_round(a, p, count, st, lane, a.hi);
        if (lane == 0) {
            const double* models = a.models + (size_t)p * a.max_iters * kMaxModels * 9;
            int found = 0, best_it = st.best_it;
            double thresh = a.thresh;
            if (count == 5) {
                found = max(a.n_models[(size_t)p * a.max_iters], 0);
                // every solution, stacked, as OpenCV returns them for a minimal set
                for (int k = 0; k < found * 9; ++k) a.E[(size_t)p * kMaxModels * 9 + k] = models[k];
                best_it = -1;
            } else if (count > 5 && a.method == kRansac) {
                found = st.max_good > 0;
            } else if (count > 5 && best_it >= 0) {
                const double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 5)) * sqrt(st.min_median);
                thresh = sigma > 0.001 ? sigma : 0.001;
                found = 1;  // provisional: at least 5 inliers are required below
            }
            for (int k = 0; k < 9; ++k)
                s_best[k] = best_it >= 0 ? models[((size_t)best_it * kMaxModels + st.best_m) * 9 + k] : 0.0;
            s_best[9] = thresh;
            s_found = found;
            s_good = 0;
            if (count != 5)
                for (int k = 0; k < 9; ++k) a.E[(size_t)p * kMaxModels * 9 + k] = s_best[k];
        }
    }
    __syncthreads();
    const int found = s_found;
    if (count == 5) {
        for (int i = tid; i < count; i += blockDim.x) mask[i] = 1;
        if (tid == 0) a.found[p] = found;
        return;
    }
    if (count < 5 || !found) {  // OpenCV leaves the caller's mask alone; the reference hands in an empty one
        for (int i = tid; i < count; i += blockDim.x) mask[i] = 0;
        if (tid == 0) a.found[p] = 0;
        return;
    }
    const double2* m1 = a.m1 + (size_t)p * a.stride;
    const double2* m2 = a.m2 + (size_t)p * a.stride;
    double E[9];
    for (int k = 0; k < 9; ++k) E[k] = s_best[k];
    const float t = (float)(s_best[9] * s_best[9]);
    int good = 0;
    for (int i = tid; i < count; i += blockDim.x) {
        const int f = sampson(E, m1[i], m2[i]) <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    if (a.method == kLmeds) {
        atomicAdd(&s_good, good);
        __syncthreads();
        if (tid == 0) a.found[p] = s_good >= 5;
    } else if (tid == 0) {
        a.found[p] = 1;
    }
}

struct EmBuffers {
    double2 *m1, *m2;
    EmState* state;
    int32_t* subsets;
    double* models;
    int32_t* n_models;
    int32_t* scores;
};

EmBuffers carve_em(Carver& c, int n_problems, size_t stride, int max_iters) {
    EmBuffers b;
    const size_t pi = (size_t)n_problems * max_iters;
    b.m1 = c.take<double2>((size_t)n_problems * stride);
    b.m2 = c.take<double2>((size_t)n_problems * stride);
    b.models = c.take<double>(pi * kMaxModels * 9);
    b.state = c.take<EmState>((size_t)n_problems);
    b.subsets = c.take<int32_t>(pi * 5);
    b.n_models = c.take<int32_t>(pi);
    b.scores = c.take<int32_t>(pi * kMaxModels);
    return b;
}

}  // namespace

}  // namespace sfm

using namespace sfm;

extern "C" {

size_t sfm_find_essential_workspace(int n_problems, size_t problem_stride, int max_iters) {
    if (n_problems <= 0 || max_iters <= 0 || problem_stride == 0) return 0;
    Carver c(nullptr);
    carve_em(c, n_problems, problem_stride, max_iters);
    return c.total();
}

int sfm_find_essential_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                           const int32_t* d_counts, int count_uniform, int n_problems, const double* K, int method,
                           double prob, double threshold, int max_iters, double* d_E, uint8_t* d_mask,
                           int32_t* d_found) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_problems < 0 || max_iters < 1 || max_iters > 100000)
        return set_error(ctx, SFM_E_INVALID, "bad problem or iteration count");
    if (n_problems == 0) return SFM_OK;
    if (!d_pts1 || !d_pts2 || !K || !d_E || !d_mask || !d_found)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (method != kRansac && method != kLmeds) return set_error(ctx, SFM_E_INVALID, "method must be 0 (RANSAC) or 1 (LMEDS)");
    if (problem_stride == 0 || problem_stride > 0x7fffffff ||
        (!d_counts && (count_uniform < 0 || (size_t)count_uniform > problem_stride)))
        return set_error(ctx, SFM_E_INVALID, "bad point count or stride");
    if (!(K[0] != 0) || !(K[4] != 0)) return set_error(ctx, SFM_E_INVALID, "camera matrix without focal lengths");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_find_essential_workspace(n_problems, problem_stride, max_iters), &ws));
    Carver c(ws);
    const EmBuffers b = carve_em(c, n_problems, problem_stride, max_iters);
    EmArgs a{};
    a.pix1 = reinterpret_cast<const float2*>(d_pts1);
    a.pix2 = reinterpret_cast<const float2*>(d_pts2);
    a.stride = problem_stride;
    a.counts = d_counts;
    a.count_uniform = count_uniform;
    a.n_problems = n_problems;
    a.max_iters = max_iters;
    a.method = method;
    a.fx = K[0];
    a.fy = K[4];
    a.cx = K[2];
    a.cy = K[5];
    a.thresh = threshold / ((a.fx + a.fy) / 2);
    a.confidence = prob;
    a.m1 = b.m1;
    a.m2 = b.m2;
    a.state = b.state;
    a.subsets = b.subsets;
    a.models = b.models;
    a.n_models = b.n_models;
    a.scores = b.scores;
    a.E = d_E;
    a.mask = d_mask;
    a.found = d_found;
    hipStream_t s = ctx->stream;
    {
        KernelTimer timer(ctx, KID_EM_SOLVE);
        hipLaunchKernelGGL(em_prep_kernel, dim3((unsigned)std::min<size_t>((problem_stride + 255) / 256, 1024), n_problems),
                           dim3(256), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    const size_t slices = (problem_stride + kScoreSlice - 1) / kScoreSlice;
    const unsigned score_z = (unsigned)std::min<size_t>((slices + 3) / 4, 1024);
    for (int lo = 0, hi = std::min(64, max_iters);; lo = hi, hi = std::min(4 * hi, max_iters)) {
        a.lo = lo;
        a.hi = hi;
        {
            KernelTimer timer(ctx, KID_EM_ADVANCE);
            hipLaunchKernelGGL(em_advance_kernel, dim3(n_problems), dim3(64), 0, s, a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        {
            KernelTimer timer(ctx, KID_EM_SOLVE);
            hipLaunchKernelGGL(em_solve_kernel, dim3(std::min((hi - lo + kSolveRows - 1) / kSolveRows, 64), n_problems),
                               dim3(64), 0, s, a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        {
            KernelTimer timer(ctx, KID_EM_SCORE);
            if (method == kRansac)
                hipLaunchKernelGGL(em_score_kernel, dim3(std::min(hi - lo, 128), n_problems, score_z), dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL(em_median_kernel, dim3(std::min((hi - lo) * kMaxModels, 1024), n_problems), dim3(64), 0, s,
                                   a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        if (hi >= max_iters) break;
    }
    KernelTimer timer(ctx, KID_EM_ADVANCE);
    hipLaunchKernelGGL(em_finish_kernel, dim3(n_problems), dim3(256), 0, s, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int sfm_find_essential(sfm_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K, int method,
                       double prob, double threshold, int max_iters, double* E, uint8_t* mask, int* n_matrices) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_matrices) *n_matrices = 0;
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    if (n == 0) return SFM_OK;
    if (!pts1 || !pts2 || !K || !E || !mask) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Carver sz(nullptr);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>(2 * (size_t)n);
    sz.take<double>(kMaxModels * 9);
    sz.take<uint8_t>((size_t)n);
    sz.take<int32_t>(4);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    float* d1 = c.take<float>(2 * (size_t)n);
    float* d2 = c.take<float>(2 * (size_t)n);
    double* dE = c.take<double>(kMaxModels * 9);
    uint8_t* dm = c.take<uint8_t>((size_t)n);
    int32_t* df = c.take<int32_t>(4);
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d2, pts2, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    SFM_RETURN_IF(sfm_find_essential_dev(ctx, d1, d2, (size_t)n, nullptr, n, 1, K, method, prob, threshold, max_iters, dE,
                                         dm, df));
    int32_t h_found = 0;
    double h_E[kMaxModels * 9];
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(mask, dm, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(h_E, dE, sizeof(h_E), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&h_found, df, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 9 * std::max(h_found, 1) && k < kMaxModels * 9; ++k) E[k] = h_E[k];
    if (n_matrices) *n_matrices = h_found;
    return SFM_OK;
}

size_t sfm_find_camera_pose_workspace(int n_problems, size_t problem_stride) {
    return std::max(sfm_find_essential_workspace(n_problems, problem_stride, 1000),
                    sfm_recover_pose_workspace(n_problems, problem_stride));
}

int sfm_find_camera_pose_dev(sfm_ctx* ctx, const float* d_pts1, const float* d_pts2, size_t problem_stride,
                             const int32_t* d_counts, int count_uniform, int n_problems, const double* K, int method,
                             double prob, double threshold, double* d_E, double* d_R, double* d_t, uint8_t* d_mask,
                             int32_t* d_found, int32_t* d_inliers) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    // both stages carve the same workspace from its start; reserving the larger one first
    // keeps the second call from reallocating it under the first one's kernels
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_find_camera_pose_workspace(n_problems, problem_stride), &ws));
    SFM_RETURN_IF(sfm_find_essential_dev(ctx, d_pts1, d_pts2, problem_stride, d_counts, count_uniform, n_problems, K,
                                         method, prob, threshold, 1000, d_E, d_mask, d_found));
    return sfm_recover_pose_dev(ctx, d_E, (size_t)kMaxModels * 9, d_found, d_pts1, d_pts2, problem_stride, d_counts,
                                count_uniform, n_problems, K, 50.0, 1e-3, d_R, d_t, d_mask, d_inliers);
}

}  // extern "C"
