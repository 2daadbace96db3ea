// Epipolar filter of DescriptorMatcher::findRobustMatches: cv::findFundamentalMat(prevAligPts,
// currAligPts, inliersMask) at Source Files/feature_processing.cpp:187-188, i.e. FM_RANSAC
// with threshold 3, confidence 0.99 and at most 1000 iterations (OpenCV itself switches to
// LMedS below 15 correspondences and to one direct 7-point solve at exactly 7).
//
// OpenCV's loop is sequential: draw 7 correspondences, solve for up to three matrices, count
// inliers, shrink the iteration budget when a better model appears.  Only two parts of it are
// truly sequential, and both are tiny: the random draws (one generator, rejection of repeated
// and collinear picks) and the "best so far / iterations left" bookkeeping.  The loop is
// therefore run in rounds of iterations [0, 64), [64, 256), [256, 1024), ...; per round and
// point set:
//   1. one wave continues the draws and records the subsets of the round;
//   2. one wave per subset solves the 7-point problem (one matrix element per lane);
//   3. waves score the models against slices of the correspondences;
//   4. one lane replays the bookkeeping over the recorded scores, in iteration order, and
//      marks the point set done where OpenCV's loop would have stopped;
// and at the end the mask of the winning model is written.  A point set that is done costs
// the later rounds nothing but empty waves, so the work follows the iteration count the
// sequential loop would have reached (tens, for the usual inlier ratios), not its maximum.
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

// What the loop carries from one round to the next.
struct FmState {
    unsigned long long rng;
    double min_median;
    int32_t n_drawn;     // subsets recorded so far
    int32_t reject_run;  // rejected attempts since the last accepted subset
    int32_t exhausted;   // getSubset gave up: no further subsets exist
    int32_t niters;      // the loop's current iteration budget
    int32_t max_good;
    int32_t best_it, best_m;
    int32_t processed;   // iterations the bookkeeping has replayed
    int32_t done;
    int32_t pad;
};

struct FmArgs {
    const float2* pts1;
    const float2* pts2;
    size_t stride;          // points between consecutive problems
    const int32_t* counts;  // per problem, or nullptr
    int count_uniform;
    int n_problems;
    int max_iters;
    int lo, hi;             // iterations of this round
    double thresh, confidence;
    FmState* state;     // [problem]
    int32_t* subsets;   // [problem][max_iters][7]
    double* models;     // [problem][max_iters][3][9]
    int32_t* n_models;  // [problem][max_iters]
    int32_t* scores;    // [problem][max_iters][3]: inlier count (RANSAC) or the bits of the
                        // float median error (LMedS)
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

// haveCollinearPoints: is the last of the seven points on a line through two earlier ones
__device__ bool have_collinear(const float2 (&m)[7]) {
    const int i = 6;
    bool hit = false;
#pragma unroll
    for (int j = 0; j < i; j++) {
        const double dx1 = m[j].x - m[i].x;
        const double dy1 = m[j].y - m[i].y;
#pragma unroll
        for (int k = 0; k < j; k++) {
            const double dx2 = m[k].x - m[i].x;
            const double dy2 = m[k].y - m[i].y;
            hit |= fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
        }
    }
    return hit;
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

constexpr int kRawPerBatch = 512;  // random numbers per batch of 64 attempts (7 each + repeats)

// The state of a point set before the first iteration.
__device__ FmState initial_state(const FmArgs& a, int regime) {
    FmState st;
    st.rng = ~0ULL;
    st.min_median = DBL_MAX;
    st.n_drawn = 0;
    st.reject_run = 0;
    st.exhausted = 0;
    st.niters = a.max_iters > 1 ? a.max_iters : 1;
    if (regime == kLmeds) {
        const int n = update_num_iters(a.confidence, 0.45, 7, a.max_iters);
        st.niters = n > 3 ? n : 3;
        if (st.niters > a.max_iters) st.niters = a.max_iters;
    }
    st.max_good = 0;
    st.best_it = -1;
    st.best_m = 0;
    st.processed = 0;
    st.done = !(regime == kRansac || regime == kLmeds);
    st.pad = 0;
    return st;
}

// Step 1: the subsets getSubset would hand to the loop up to iteration a.hi, in order.  By one
// wave.  cv::RNG (multiply with carry, seeded with ~0) is advanced by one lane; the reduction
// of its outputs to indices, the rejection of repeated picks and the collinearity test of 64
// attempts run in parallel.  An attempt's place in the random stream depends on the repeats
// in the attempts before it: the lanes assume none, measure their own, and re-place
// themselves until nothing moves (at once when no pick repeats, the usual case).
__device__ void draw_round(const FmArgs& a, int p, int count, int regime, FmState& st, int lane,
                           unsigned long long* raw, int32_t* pick_of) {
    if (!st.done) {
        const float2* m1 = a.pts1 + (size_t)p * a.stride;
        const float2* m2 = a.pts2 + (size_t)p * a.stride;
        int32_t* out = a.subsets + (size_t)p * a.max_iters * 7;
        const int max_attempts = regime == kRansac ? 10000 : 1000;
        const int target = min(min(a.hi, st.niters), a.max_iters);
        while (st.n_drawn < target && !st.exhausted) {
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
            int idx[7];
            int shift = 0, extra = 0;
            bool complete = true;
            for (;;) {
                const int pos = 7 * lane + shift;
                extra = 0;
                complete = true;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
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
                int incl = extra;  // inclusive prefix sum over the lanes
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
            // attempts whose numbers all lie inside this batch: the lanes before the first
            // incomplete one (a later lane's place depends on every earlier one)
            const unsigned long long incomplete = __ballot(!complete);
            const int n_att = incomplete ? __ffsll((long long)incomplete) - 1 : 64;
            if (n_att == 0) {  // seven distinct picks not found in 512 draws: cannot happen with count >= 8
                st.exhausted = 1;
                break;
            }
            bool ok = false;
            if (lane < n_att) {
                float2 s1[7], s2[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    s1[i] = m1[idx[i]];
                    s2[i] = m2[idx[i]];
                }
                ok = !(have_collinear(s1) || have_collinear(s2));
            }
            const unsigned long long accepted = __ballot(ok);
            const int first = accepted ? __ffsll((long long)accepted) - 1 : n_att;
            if (st.reject_run + first >= max_attempts) {
                st.exhausted = 1;  // max_attempts rejections in a row: getSubset fails, the loop ends
            } else if (accepted) {
                const int slot = st.n_drawn + __popcll(accepted & ((1ULL << lane) - 1ULL));
                if (ok && slot < a.max_iters) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) out[slot * 7 + k] = idx[k];
                }
                st.n_drawn = min(st.n_drawn + (int)__popcll(accepted), a.max_iters);
                st.reject_run = n_att - 1 - (63 - __clzll((long long)accepted));
            } else {
                st.reject_run += n_att;
            }
            const int used = __shfl(7 * lane + shift + 7 + extra, n_att - 1);
            st.rng = raw[used - 1];
            __syncthreads();
        }
    }
}

// Step 4: the loop's bookkeeping over the scored iterations [st.processed, hi), replayed in
// order by one wave.  The lanes read 64 iterations at a time and find the ones that beat
// everything before them (a running maximum of the inlier count, or minimum of the median);
// only those change the state, and they are then applied one by one, so that the shrinking
// iteration budget cuts the replay off exactly where it cuts the sequential loop off.
__device__ void select_round(const FmArgs& a, int p, int count, int regime, FmState& st, int lane, int hi) {
    if (st.done) return;
    const int32_t* nm = a.n_models + (size_t)p * a.max_iters;
    const int32_t* sc = a.scores + (size_t)p * a.max_iters * 3;
    const int end = min(hi, st.n_drawn);
    const bool ransac = regime == kRansac;
    for (int base = st.processed; base < end && base < st.niters; base += 64) {
        const int it = base + lane;
        int n = 0;
        int32_t s[3] = {0, 0, 0};
        if (it < end) {
            n = nm[it];
            n = n < 0 ? 0 : (n > 3 ? 3 : n);
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (m < n) s[m] = sc[it * 3 + m];
        }
        // per-iteration figure of merit, larger is better
        double merit = -DBL_MAX;
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (m < n) {
                const double v = ransac ? (double)s[m] : -(double)__int_as_float(s[m]);
                merit = v > merit ? v : merit;
            }
        double before = merit;  // best of the lanes before this one
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double v = __shfl_up(before, o);
            if (lane >= o) before = v > before ? v : before;
        }
        before = __shfl_up(before, 1);
        const double carried = ransac ? (double)(st.max_good > 6 ? st.max_good : 6) : -st.min_median;
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
            const int n_l = __builtin_amdgcn_readlane(n, l);
            const int32_t s_l[3] = {__builtin_amdgcn_readlane(s[0], l), __builtin_amdgcn_readlane(s[1], l),
                                    __builtin_amdgcn_readlane(s[2], l)};
            for (int m = 0; m < n_l; ++m) {
                if (ransac) {
                    const int good = s_l[m];
                    if (good > (st.max_good > 6 ? st.max_good : 6)) {
                        st.best_it = it_l;
                        st.best_m = m;
                        st.max_good = good;
                        st.niters = update_num_iters(a.confidence, (double)(count - good) / count, 7, st.niters);
                    }
                } else {
                    const double median = __int_as_float(s_l[m]);
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
    st.done = st.processed >= st.niters || (st.exhausted && st.processed >= st.n_drawn) || hi >= a.max_iters;
}

// Between two rounds, one wave per point set: the bookkeeping of the round just scored
// ([.., a.lo)), then the draws of the next one ([a.lo, a.hi)).
__global__ __launch_bounds__(64) void fm_advance_kernel(FmArgs a) {
    __shared__ unsigned long long raw[kRawPerBatch];
    __shared__ int32_t pick_of[kRawPerBatch];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    FmState st;
    if (a.lo == 0) {
        st = initial_state(a, regime);
    } else {
        st = a.state[p];
        if (st.done) return;
        select_round(a, p, count, regime, st, lane, a.lo);
    }
    draw_round(a, p, count, regime, st, lane, raw, pick_of);
    if (lane == 0) a.state[p] = st;
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

// x of lane `src`, which every lane names alike.
__device__ __forceinline__ double from_lane(double x, int src) {
    src = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// Orthonormal basis (f1, f2) of the null space of a 7x9 matrix held one element per lane
// (lane 9 r + c, lane 63 idle): Gauss-Jordan elimination with complete pivoting.  The pivot
// is the first largest magnitude in row-major order, as a sequential scan would pick it.
__device__ bool null_space_7x9(double a, int lane, double (&f1)[9], double (&f2)[9]) {
    const bool valid = lane < 63;
    const int r = lane / 9, c = lane - 9 * r;
    unsigned used_cols = 0;
    int col_of[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double mag = fabs(a);
        double v = valid && r >= k && !((used_cols >> c) & 1u) && mag > 0 ? mag : -1.0;
        int who = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(v, o);
            const int ow = __shfl_xor(who, o);
            if (ov > v || (ov == v && ow < who)) {
                v = ov;
                who = ow;
            }
        }
        if (!(v > 0)) return false;
        who = __builtin_amdgcn_readfirstlane(who);
        const int pr = who / 9, pc = who - 9 * pr;
        int src = lane;
        if (valid && r == k) src = pr * 9 + c;
        else if (valid && r == pr) src = k * 9 + c;
        a = __shfl(a, src);
        used_cols |= 1u << pc;
        col_of[k] = pc;
        const double inv = 1.0 / from_lane(a, k * 9 + pc);
        if (valid && r == k) a *= inv;
        const double m = __shfl(a, valid ? r * 9 + pc : lane);
        const double row_k = __shfl(a, k * 9 + c);
        if (valid && r != k && m != 0) a -= m * row_k;
    }
    const unsigned free_mask = ~used_cols & 0x1ffu;
    const int fc0 = __ffs((int)free_mask) - 1;
    const int fc1 = __ffs((int)(free_mask & (free_mask - 1))) - 1;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        f1[j] = j == fc0 ? 1.0 : 0.0;
        f2[j] = j == fc1 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double t0 = -from_lane(a, k * 9 + fc0);
        const double t1 = -from_lane(a, k * 9 + fc1);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            f1[j] = col_of[k] == j ? t0 : f1[j];
            f2[j] = col_of[k] == j ? t1 : f2[j];
        }
    }
    double n2 = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) n2 += f2[j] * f2[j];
    double inv = 1.0 / sqrt(n2);
#pragma unroll
    for (int j = 0; j < 9; ++j) f2[j] *= inv;
    double dot = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) dot += f1[j] * f2[j];
#pragma unroll
    for (int j = 0; j < 9; ++j) f1[j] -= dot * f2[j];
    n2 = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) n2 += f1[j] * f1[j];
    inv = 1.0 / sqrt(n2);
#pragma unroll
    for (int j = 0; j < 9; ++j) f1[j] *= inv;
    return true;
}

// run7Point, by one wave: lane 9 r + c holds the coefficient of matrix entry c in the
// equation of correspondence r; `mine1`, `mine2` are that correspondence.  Everything after
// the elimination is the same arithmetic in every lane; lane 0 writes the matrices found to
// `fmatrix` and to `copy`.
__device__ int seven_point(float2 mine1, float2 mine2, int lane, double* fmatrix, double* copy) {
    double c[4], r[3] = {0, 0, 0};
    double f1[9], f2[9];
    {
        const double x0 = mine1.x, y0 = mine1.y;
        const double x1 = mine2.x, y1 = mine2.y;
        const int col = lane % 9;
        const double left = col < 3 ? x1 : col < 6 ? y1 : 1.0;
        const int sub = col % 3;
        const double right = sub == 0 ? x0 : sub == 1 ? y0 : 1.0;
        // x1 x0, x1 y0, x1, y1 x0, y1 y0, y1, x0, y0, 1
        const double coeff = col == 2 || col == 5 ? left : col >= 6 ? right : left * right;
        if (!null_space_7x9(coeff, lane, f1, f2)) return 0;
    }
#pragma unroll
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
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k >= n) break;
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        double last;
        if (fabs(s) > DBL_EPSILON) {
            mu = 1. / s;
            lambda *= mu;
            last = 1.;
        } else
            last = 0.;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) copy[9 * k + i] = fmatrix[9 * k + i] = f1[i] * lambda + f2[i] * mu;
            copy[9 * k + 8] = fmatrix[9 * k + 8] = last;
        }
    }
    return n;
}

// FMEstimatorCallback::computeError for one correspondence
__device__ __forceinline__ float epipolar_error(const double (&F)[9], double x1, double y1, double x2, double y2) {
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

__device__ __forceinline__ float epipolar_error(const double (&F)[9], float2 p1, float2 p2) {
    return epipolar_error(F, (double)p1.x, (double)p1.y, (double)p2.x, (double)p2.y);
}

// End of the iterations of the round [a.lo, a.hi) that the loop of point set p can still
// reach (a.lo when there are none).
__device__ __forceinline__ int live_end(const FmArgs& a, int p, int regime) {
    if (regime == kDirect) return a.lo == 0 ? 1 : a.lo;
    const FmState* st = a.state + p;
    if ((regime != kRansac && regime != kLmeds) || st->done) return a.lo;
    return max(a.lo, min(a.hi, min(st->niters, st->n_drawn)));
}

// Step 2: one wave per (point set, live iteration of the round); for point sets of 8 to 14
// correspondences also their scores (step 3), the median error of each matrix.
__global__ __launch_bounds__(256) void fm_solve_kernel(FmArgs a) {
    __shared__ float s_err[4][3][14];
    __shared__ double s_models[4][27];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.y;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    const int end = live_end(a, p, regime);
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    const int row = min(lane / 9, 6);
    for (int it = a.lo + blockIdx.x * 4 + (threadIdx.x >> 6); it < end; it += gridDim.x * 4) {
        const size_t slot = (size_t)p * a.max_iters + it;
        if (lane < 3) a.scores[slot * 3 + lane] = 0;
        const int j = regime == kDirect ? row : a.subsets[slot * 7 + row];
        double* copy = s_models[threadIdx.x >> 6];
        const int n = seven_point(m1[j], m2[j], lane, a.models + slot * 27, copy);
        if (lane == 0) a.n_models[slot] = n;
        if (regime == kLmeds) {
            __threadfence_block();  // lane 0's matrices in LDS, for the lanes of this wave
            __builtin_amdgcn_wave_barrier();
        }
        if (regime == kLmeds && lane < n) {
            // least median of squares, one lane per matrix: std::nth_element(err, err +
            // count / 2, err + count) of at most 14 errors
            double mine[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) mine[k] = copy[9 * lane + k];
            float* e = s_err[threadIdx.x >> 6][lane];  // rare regime: kept out of the registers
#pragma unroll 1
            for (int i = 0; i < count; ++i) {
                const float v = epipolar_error(mine, m1[i], m2[i]);
                int q = i - 1;
#pragma unroll 1
                while (q >= 0 && e[q] > v) {
                    e[q + 1] = e[q];
                    --q;
                }
                e[q + 1] = v;
            }
            a.scores[slot * 3 + lane] = __float_as_int(e[count / 2]);
        }
    }
}

// (float)epipolar_error(F, p1, p2) <= t, decided without the two divisions wherever the
// outcome is not within rounding of the threshold.  Errors below the midpoint between t and the
// next float above it convert to a float <= t.  With q the squared norm of an epipolar line
// and d the point's residual, the error term is fl(fl(d d) fl(1 / q)), within 2^-51 of d d / q;
// comparing d d with q times the midpoint, at a margin of 2^-48 either way (t_lo, t_hi),
// decides all but a sliver, and there (and for any non-finite value) the reference expression
// itself is evaluated.
__device__ __forceinline__ bool epipolar_inlier(const double (&F)[9], double x1, double y1, double x2, double y2,
                                                double t_lo, double t_hi, float t) {
    const double a2 = F[0] * x1 + F[1] * y1 + F[2];
    const double b2 = F[3] * x1 + F[4] * y1 + F[5];
    const double c2 = F[6] * x1 + F[7] * y1 + F[8];
    const double q2 = a2 * a2 + b2 * b2;
    const double d2 = x2 * a2 + y2 * b2 + c2;
    const double a1 = F[0] * x2 + F[3] * y2 + F[6];
    const double b1 = F[1] * x2 + F[4] * y2 + F[7];
    const double c1 = F[2] * x2 + F[5] * y2 + F[8];
    const double q1 = a1 * a1 + b1 * b1;
    const double d1 = x1 * a1 + y1 * b1 + c1;
    const double dd1 = d1 * d1, dd2 = d2 * d2;
    const bool in1 = dd1 < t_lo * q1, in2 = dd2 < t_lo * q2;
    const bool out1 = dd1 > t_hi * q1, out2 = dd2 > t_hi * q2;
    const bool decided = (in1 | out1) & (in2 | out2);
    bool inlier = in1 & in2;
    if (__builtin_expect(!decided, 0)) inlier = epipolar_error(F, x1, y1, x2, y2) <= t;
    return inlier;
}

// The midpoint between t and the next float above it.
__device__ __forceinline__ double float_upper_midpoint(float t) {
    const float next = __int_as_float(__float_as_int(t) + 1);  // t is positive and finite
    return 0.5 * ((double)t + (double)next);
}

constexpr int kScoreSlice = 1024;  // correspondences per wave and pass of the scoring kernel

// Step 3 (RANSAC): inlier counts of the models of each live iteration of the round; one wave
// per slice of the correspondences, summed with integer atomics.  Bound by the rate of fp64
// instructions: about 48 per correspondence and model, none of them fused (the results have
// to match a CPU evaluation that rounds after every operation).
__global__ __launch_bounds__(256) void fm_score_kernel(FmArgs a) {
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = blockIdx.z * 4 + (threadIdx.x >> 6);
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    if (regime != kRansac || wave * kScoreSlice >= count) return;
    const int it_end = live_end(a, p, regime);
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    const float t = (float)(a.thresh * a.thresh);
    const double t_lo = float_upper_midpoint(t) * (1.0 - 0x1p-48);
    const double t_hi = float_upper_midpoint(t) * (1.0 + 0x1p-48);
    for (int it = a.lo + blockIdx.x; it < it_end; it += gridDim.x) {
        const size_t slot = (size_t)p * a.max_iters + it;
        const int n = a.n_models[slot];
        if (n <= 0 || n > 3) continue;
        const double* src = a.models + slot * 27;
        int32_t* out = a.scores + slot * 3;
        double F[3][9];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int k = 0; k < 9; ++k) F[m][k] = m < n ? src[m * 9 + k] : 0.0;
        int good[3] = {0, 0, 0};
        for (int base = wave * kScoreSlice; base < count; base += gridDim.z * 4 * kScoreSlice) {
            const int end = min(base + kScoreSlice, count);
            for (int i = base + lane; i < end; i += 64) {
                const float2 p1 = m1[i], p2 = m2[i];
                const double x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    if (m < n) good[m] += epipolar_inlier(F[m], x1, y1, x2, y2, t_lo, t_hi, t);
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            int v = good[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && m < n && v) atomicAdd(out + m, v);
        }
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

// Step 5, one workgroup per point set: the bookkeeping of the last round, the outcome, the
// mask of the chosen matrix and, when `matches` is given, the matches whose mask entry is set,
// kept in order (Source Files/feature_processing.cpp:193-204 with useEpipolarFilter == true,
// then :223).  The compaction is in place: an element only ever moves to a lower index, and
// a whole chunk is read before any of it is written.
__global__ __launch_bounds__(1024) void fm_finish_kernel(FmArgs a, sfm_dmatch* __restrict__ matches,
                                                         int32_t* match_counts) {
    __shared__ int wave_count[16];
    __shared__ int s_found;
    __shared__ double s_best[10];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int count = problem_count(a, p);
    const int regime = regime_of(count);
    uint8_t* mask = a.mask + (size_t)p * a.stride;
    if (wave == 0) {
        FmState st = a.state[p];
        select_round(a, p, count, regime, st, lane, a.hi);
        if (lane == 0) {
            const double* models = a.models + (size_t)p * a.max_iters * 27;
            int best_it = st.best_it, found = 0;
            double inlier_thresh = a.thresh;
            if (regime == kDirect) {
                found = a.n_models[(size_t)p * a.max_iters] > 0;
                best_it = found ? 0 : -1;
            } else if (regime == kRansac) {
                found = st.max_good > 0;
            } else if (regime == kLmeds && best_it >= 0) {
                const double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 7)) * sqrt(st.min_median);
                inlier_thresh = sigma > 0.001 ? sigma : 0.001;
                found = 1;  // provisional: at least 7 inliers are required below
            }
            for (int k = 0; k < 9; ++k)
                s_best[k] = best_it >= 0 ? models[(size_t)best_it * 27 + st.best_m * 9 + k] : 0.0;
            s_best[9] = inlier_thresh;
            s_found = found;
            if (a.F)
                for (int k = 0; k < 9; ++k) a.F[(size_t)p * 9 + k] = s_best[k];
        }
    }
    __syncthreads();
    const int found = s_found;
    // OpenCV leaves the caller's mask alone when nothing is found; the reference hands in zeros
    const int fixed = regime == kNone || (regime != kDirect && !found) ? 0 : regime == kDirect ? 1 : -1;
    const float2* m1 = a.pts1 + (size_t)p * a.stride;
    const float2* m2 = a.pts2 + (size_t)p * a.stride;
    double F[9];
    for (int k = 0; k < 9; ++k) F[k] = s_best[k];
    const double th = s_best[9];
    const float t = (float)(th * th);
    sfm_dmatch* mine = matches ? matches + (size_t)p * a.stride : nullptr;
    int running = 0;
    for (int i0 = 0; i0 < count; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        if (i < count) {
            ok = fixed >= 0 ? fixed != 0 : epipolar_error(F, m1[i], m2[i]) <= t;
            mask[i] = (uint8_t)ok;
        }
        sfm_dmatch m{};
        if (ok && mine) m = mine[i];
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
        if (ok && mine) mine[running + offset + prefix] = m;
        running += total;
        __syncthreads();
    }
    if (tid == 0) {
        a.found[p] = regime == kLmeds && found ? running >= 7 : found;
        if (match_counts) match_counts[p] = running;
    }
}

struct FmBuffers {
    FmState* state;
    int32_t* subsets;
    double* models;
    int32_t* n_models;
    int32_t* scores;
    double* best;
};

FmBuffers carve_fm(Carver& c, int n_problems, int max_iters) {
    FmBuffers b;
    const size_t pi = (size_t)n_problems * max_iters;
    b.models = c.take<double>(pi * 27);
    b.best = c.take<double>((size_t)n_problems * 10);
    b.state = c.take<FmState>((size_t)n_problems);
    b.subsets = c.take<int32_t>(pi * 7);
    b.n_models = c.take<int32_t>(pi);
    b.scores = c.take<int32_t>(pi * 3);
    return b;
}

void bind_fm(FmArgs& a, const FmBuffers& b) {
    a.state = b.state;
    a.subsets = b.subsets;
    a.models = b.models;
    a.n_models = b.n_models;
    a.scores = b.scores;
    a.best = b.best;
}

// The rounds for the point sets described by `args`, then the masks (and, for match lists
// whose capacity is the point stride, their compaction).
int run_fm(sfm_ctx* ctx, const FmArgs& args, sfm_dmatch* matches = nullptr, int32_t* match_counts = nullptr) {
    FmArgs a = args;
    const int n_problems = a.n_problems, max_iters = a.max_iters;
    const size_t slices = (a.stride + kScoreSlice - 1) / kScoreSlice;
    const unsigned score_z = (unsigned)std::min<size_t>((slices + 3) / 4, 1024);
    for (int lo = 0, hi = std::min(64, max_iters);; lo = hi, hi = std::min(4 * hi, max_iters)) {
        a.lo = lo;
        a.hi = hi;
        {
            KernelTimer timer(ctx, KID_FM_SUBSETS);
            hipLaunchKernelGGL(fm_advance_kernel, dim3(n_problems), dim3(64), 0, ctx->stream, a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        {
            KernelTimer timer(ctx, KID_FM_SOLVE);
            hipLaunchKernelGGL(fm_solve_kernel, dim3(std::min((hi - lo + 3) / 4, 64), n_problems), dim3(256), 0, ctx->stream, a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        {
            KernelTimer timer(ctx, KID_FM_SCORE);
            hipLaunchKernelGGL(fm_score_kernel, dim3(std::min(hi - lo, 128), n_problems, score_z), dim3(256), 0, ctx->stream, a);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        if (hi >= max_iters) break;
    }
    KernelTimer timer(ctx, KID_FM_SELECT);
    hipLaunchKernelGGL(fm_finish_kernel, dim3(n_problems), dim3(1024), 0, ctx->stream, a, matches, match_counts);
    SFM_HIP_CHECK(ctx, hipGetLastError());
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
    bind_fm(a, b);
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
    bind_fm(a, b);
    a.F = nullptr;
    a.mask = d_mask ? d_mask : mask;
    a.found = found;
    SFM_RETURN_IF(run_fm(ctx, a, d_matches, d_match_counts));
    return SFM_OK;
}

}  // extern "C"
