// Brute-force L2 2-nearest-neighbour search for float descriptors on the matrix cores of gfx950.
//
// Replaces cv::BFMatcher(NORM_L2 / NORM_L2SQR)::knnMatch(q, t, out, 2) as called
// from DescriptorMatcher::ratioMaches (Source Files/feature_processing.cpp:121)
// for the "BRUTEFORCE" and "BRUTEFORCE_SL2" matcher types (:111, :115).
//
// Passes.
//   0. Packing (l2_pack_kernel, l2_narrow_kernel): rows zero-padded to K floats, ||row||^2, and
//      narrow copies of the rows for pass 1.
//   1. Candidate pass (MFMA).  The ranking key ||q - t||^2 - ||q||^2 = ||t||^2 - 2 q.t is formed as
//      a dense contraction: the train tile is the A operand (rows = train), the query tile the
//      B operand (columns = queries), so each lane's accumulator registers hold 16 train
//      candidates of ONE query and the running best-two list of that query lives in the lane's
//      registers with no cross-lane traffic.  The queries of a wave stay in VGPRs for the whole
//      kernel; train tiles are copied global -> LDS by global_load_lds (XOR-swizzled through the
//      source address so that the ds_read_b128 fragment reads are conflict-free); the
//      accumulators start at the norm term, so the accumulator is the ranking key when the
//      k-loop ends.  Every (lane half, train split) keeps its own best two, which are written
//      out as candidates.  By default the contraction runs on the fp16 matrix cores, sixteen
//      times the rate, over fp16 copies of the rows (l2_candidates_narrow_kernel); SFM_L2_MFMA
//      selects a three-product bfloat16 split (bf16) or v_mfma_f32_32x32x2_f32 on the floats
//      themselves (f32, l2_candidates_kernel).
//      The key only ranks: its rounding is covered by the proof of pass 2.
//   2. Re-score pass.  For the four candidates of a query with the smallest ranking
//      keys -- and the next four when the fifth key could still reach the second place -- the
//      squared distance is recomputed as the sequential float sum
//      d += (a-b)*(a-b)  (the scalar order of OpenCV's normL2Sqr), the square root
//      is taken for NORM_L2, and the best two are chosen by (distance, train index).
//      Every train row that was NOT re-scored has a ranking key at least T = the smaller
//      of the smallest candidate key not re-scored and the third-smallest key of any candidate
//      list (the candidate pass keeps that third key too).  With a bound E on the
//      difference between a row's sequential distance and its MFMA key + ||q||^2, the
//      result is proven exact when T + ||q||^2 - E exceeds the re-scored second-best
//      squared distance; a query for which it does not is appended to a list.
//      (The narrow passes keep the best two of every stretch of 256 train rows, and the
//      third key is the third-smallest among those: it bounds every row but the ones of a
//      stretch that gave a list both its entries, which rank behind the second of them.  When
//      that second is too near for the proof, the stretch is listed for pass 3.)
//   3. Exhaustive pass over the listed queries (normally none): one workgroup per query
//      recomputes the sequential distance to EVERY train row and takes the best two by
//      (distance, train index) -- and, for a listed stretch, to the rows of the stretch, joined
//      to the result of pass 2.  Results are therefore always those of the sequential
//      sum with the lower-index tie-break, whatever the rounding of the MFMA form.
#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "sfm_internal.h"
#include <type_traits>

namespace sfm {

int launch_ratio_cross_knn(sfm_ctx* ctx, const int32_t* d_knn_idx, const float* d_knn_dist,
                           int rows_max, const int32_t* d_counts, int rows_uniform,
                           const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                           float* d_best_dist, sfm_dmatch* d_matches, int cap,
                           int32_t* d_match_counts);

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kTileRows = 64;   // train rows per LDS tile (two 32-row MFMA tiles)
constexpr int kMaxSplits = 16;  // 4 candidates per split and query, 64 per query at most
// The narrow candidate kernel joins the running lists once per kMergeTiles tiles: a key carries
// its position among the 32 kMergeTiles accumulators of that stretch in its low mantissa bits.
// (Eight tiles and eight tag bits measured no faster than four and seven.)
constexpr int kMergeTiles = 4;
constexpr int kNarrowTagBits = 7;
static_assert((32 * kMergeTiles) == (1 << kNarrowTagBits), "tag = tiles folded since, position in the tile");

// l2_pack_kernel and l2_narrow_kernel leave two maxima per workgroup, slots[b] and
// slots[workgroups + b]: over its rows in the group of its first row, and over its rows in later
// groups (a workgroup across a frame boundary; none when the groups are whole multiples of 32 rows) -- plain stores; an atomic maximum per workgroup on a
// per-group address costs more than the packing itself (1250 workgroups, all resident at once:
// 5 to 12 us per launch).  The largest value of group g is the maximum of the first slots of
// the workgroups that start in g and of the rows of g that sit in the workgroup before them,
// which are read from the per-row array (fewer than 32).  NaN values are skipped: their rows
// never rank.
struct FoldSet {
    const float* slots;     // [2][workgroups]
    const float* per_row;   // [rows]
    const int32_t* counts;  // rows in use per group, or null: the others do not count
    float* group;           // [groups]: the result
    int rows, rows_per_group, groups;
};
struct FoldArgs {
    FoldSet set[2];   // groups of set 0, then of set 1
};

struct L2Args {
    const float* q_base;  // packed rows of K floats
    const float* t_base;
    const float* t_norms;  // ||t||^2 per packed train row, same frame layout (rows_cap per frame)
    const float* t_acc0;   // start values of the narrow passes: -||t||^2 / 2 (fp16: times s^2)
    const float* narrow_scale;  // fp16 pass: s^2, 1 / s^2
    size_t frame_floats;   // floats per frame in q_base / t_base
    int rows_cap;          // rows per frame in the norm array and the candidate array
    const int32_t* counts;
    int nq_uniform, nt_uniform;
    const int32_t* pairs;
    int n_groups;
    int qtiles, splits, blocks_per_group;
    int32_t* cand;  // [problem][split][rows_cap][4]
    float* cand_key;  // the candidates' ranking keys ||t||^2 - 2 q.t, same layout
    float* cand_third;  // [problem][split][rows_cap][2]: third-smallest key of each candidate list (narrow passes: among the best two of every stretch, see RescoreArgs::stretch_rows)
    // fp16 pass: the last fold_groups workgroups of the launch do not rank; they fold what
    // l2_pack_kernel and l2_narrow_kernel left per workgroup into the groups' largest norms and
    // residual norms, which the re-score pass reads (a launch of their own would cost more
    // than they do)
    FoldArgs fold[2];
    int fold_groups;
};

__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gptr,
        (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

// mask ? a : b per lane, as one v_cndmask_b32
__device__ __forceinline__ float select_f(unsigned long long mask, float a, float b) {
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ int select_i(unsigned long long mask, int a, int b) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// ---- largest norm / residual norm of every group of rows (a frame; the queries; the train rows)
// (four waves per workgroup; sixteen, for a quarter of the slots, take half as long again: 14 against 9 us)
constexpr int kPackBlock = 256;
constexpr int kPackRowsPerWave = 8;
constexpr int kPackRowsPerBlock = (kPackBlock / 64) * kPackRowsPerWave;

// Maxima of two values over the workgroup (up to 16 waves), in every thread.
__device__ __forceinline__ void block_max2(float& a, float& b) {
    __shared__ float s_max[16][2];
    for (int o = 32; o > 0; o >>= 1) {
        a = fmaxf(a, __shfl_xor(a, o));
        b = fmaxf(b, __shfl_xor(b, o));
    }
    if ((threadIdx.x & 63) == 0) {
        s_max[threadIdx.x >> 6][0] = a;
        s_max[threadIdx.x >> 6][1] = b;
    }
    __syncthreads();
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        a = fmaxf(a, s_max[w][0]);
        b = fmaxf(b, s_max[w][1]);
    }
    __syncthreads();   // the array may be written again
}

// (the whole workgroup calls this)
__device__ __forceinline__ void fold_group(const FoldArgs& f, int g_all) {
    const bool second = g_all >= f.set[0].groups;
    const FoldSet& p = f.set[second ? 1 : 0];
    const int g = second ? g_all - f.set[0].groups : g_all;
    const long blocks = ((long)p.rows + kPackRowsPerBlock - 1) / kPackRowsPerBlock;
    const long first_row = (long)g * p.rows_per_group;
    const long end_row = min((long)p.rows, first_row + p.rows_per_group);
    // workgroup b starts in the group of row 32 b
    const long b0 = (first_row + kPackRowsPerBlock - 1) / kPackRowsPerBlock;
    const long b1 = min(blocks, (first_row + p.rows_per_group + kPackRowsPerBlock - 1) / kPackRowsPerBlock);
    float v = 0.f, unused = 0.f;
    for (long b = b0 + threadIdx.x; b < b1; b += blockDim.x) v = fmaxf(v, p.slots[b]);
    const long head_end = min(end_row, b0 * kPackRowsPerBlock);
    const long used = p.counts ? p.counts[g] : p.rows_per_group;
    for (long r = first_row + threadIdx.x; r < head_end; r += blockDim.x)
        if (r - first_row < used) v = fmaxf(v, p.per_row[r]);
    block_max2(v, unused);
    if (threadIdx.x == 0) p.group[g] = v;
}

template <int K, int NT>
__global__ __launch_bounds__(kBlock) void l2_candidates_kernel(L2Args a) {
    constexpr int kChunksPerRow = K / 4;             // 16-byte chunks
    constexpr int kTileFloats = kTileRows * K;
    constexpr int kQPerWave = 32 * NT;
    constexpr int kQPerBlock = 4 * kQPerWave;
    // one array: [2 tiles][kTileFloats] then [2][64] norms
    __shared__ __attribute__((aligned(16))) float lds[2 * kTileFloats + 2 * kTileRows];
    float* norm_lds = lds + 2 * kTileFloats;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    const int g = blockIdx.x / a.blocks_per_group, inner = blockIdx.x % a.blocks_per_group;
    const int per_dir = a.qtiles * a.splits;
    const int dir = inner / per_dir;
    const int qtile = (inner % per_dir) / a.splits;
    const int split = inner % a.splits;
    int qf = 0, tf = 0, problem = 0;
    if (a.pairs) {
        const int fa = a.pairs[2 * g], fb = a.pairs[2 * g + 1];
        qf = dir ? fb : fa;
        tf = dir ? fa : fb;
        problem = 2 * g + dir;
    }
    const int nq = a.counts ? a.counts[qf] : a.nq_uniform;
    const int nt = a.counts ? a.counts[tf] : a.nt_uniform;
    const int q0 = qtile * kQPerBlock;
    if (q0 >= nq) return;
    const float* qrows = a.q_base + (size_t)qf * a.frame_floats;
    const float* trows = a.t_base + (size_t)tf * a.frame_floats;
    const float* tnorm = a.t_norms + (size_t)tf * a.rows_cap;

    // B operand: this lane's query columns, scaled by -2, resident for the whole kernel.
    // Element [n][m][s] is dimension 8m + 4*lh + s of query q0 + wave*kQPerWave + 32n + li.
    float qreg[NT][K / 8][4];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qi = q0 + wave * kQPerWave + 32 * n + li;
        const floatx4* src = reinterpret_cast<const floatx4*>(qrows + (size_t)(qi < nq ? qi : 0) * K);
#pragma unroll
        for (int m = 0; m < K / 8; ++m) {
            floatx4 v = src[2 * m + lh];
            if (qi >= nq) v = floatx4{0.f, 0.f, 0.f, 0.f};
            qreg[n][m][0] = -2.f * v.x;
            qreg[n][m][1] = -2.f * v.y;
            qreg[n][m][2] = -2.f * v.z;
            qreg[n][m][3] = -2.f * v.w;
        }
    }
    float best0[NT], best1[NT], third[NT];
    int bidx0[NT], bidx1[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        best0[n] = best1[n] = FLT_MAX;
        third[n] = INFINITY;
        bidx0[n] = bidx1[n] = -1;
    }

    const int chunk = (nt + a.splits - 1) / a.splits;
    const int t_begin = split * chunk;
    const int t_end = min(nt, t_begin + chunk);
    const int ntiles = t_end > t_begin ? (t_end - t_begin + kTileRows - 1) / kTileRows : 0;

    // global -> LDS copy of one tile: LDS position p (16-byte chunks, linear) holds chunk
    // (p % cpr) ^ (row & 15) of row p / cpr; each wave instruction fills 64 positions.
    auto stage = [&](int k, int buf) {
        const int tbase = t_begin + k * kTileRows;
        float* dst = lds + buf * kTileFloats;
#pragma unroll
        for (int it = 0; it < kChunksPerRow / 4; ++it) {
            const int instr = wave * (kChunksPerRow / 4) + it;  // kChunksPerRow instructions per tile
            const int p = instr * 64 + lane;
            const int row = p / kChunksPerRow, cpos = p % kChunksPerRow;
            const int c = cpos ^ (row & 15);
            int trow = tbase + row;
            trow = trow < nt ? trow : nt - 1;  // clamped rows are masked by an infinite norm
            glds16(trows + (size_t)trow * K + c * 4, dst + instr * 256);
        }
        if (tid < kTileRows) {
            const int trow = tbase + tid;
            norm_lds[buf * kTileRows + tid] = trow < t_end ? tnorm[trow] : INFINITY;
        }
    };

    if (ntiles > 0) stage(0, 0);
    __syncthreads();

    // Two accumulator sets.  While the matrix pipe accumulates tile k into one set, the
    // vector ALU folds the finished tile k-1 (the other set) into the best-two lists, a
    // slice of it after every group of MFMAs, so the fold hides in the MFMA issue shadow.
    constexpr int kMSteps = K / 8;
    constexpr int kGroups = 2 * NT * 4;  // groups of 4 accumulator registers per set
    floatx16 accA[2][NT], accB[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[mt][n][r] = accB[mt][n][r] = INFINITY;

    // Keeps the two smallest keys per lane and query tile.  Candidates arrive in ascending
    // train index and comparisons are strict, so equal keys keep the lower index first; a
    // key that displaces the best pushes the old best (not the old second) down.
    auto fold_key = [&](const floatx16 (&acc)[2][NT], int tbase, int kq) {
        const int gi = kq / 4, r = kq % 4;
        const int mt = gi / (NT * 4), n = (gi / 4) % NT, gq = gi % 4;
        const float key = acc[mt][n][4 * gq + r];
        const int ti = tbase + 32 * mt + 8 * gq + 4 * lh + r;
        // selects written as v_cndmask on lane masks: from C++ selects the compiler builds
        // exec-masked moves behind branches, which stall a wave that has MFMAs to issue
        const unsigned long long c0 = __builtin_amdgcn_ballot_w64(key < best0[n]);
        const unsigned long long c1 = __builtin_amdgcn_ballot_w64(key < best1[n]);
        asm("v_med3_f32 %0, %0, %1, %2" : "+v"(third[n]) : "v"(key), "v"(best1[n]));
        const float t = select_f(c1, key, best1[n]);
        const int tix = select_i(c1, ti, bidx1[n]);
        best1[n] = select_f(c0, best0[n], t);
        bidx1[n] = select_i(c0, bidx0[n], tix);
        best0[n] = select_f(c0, key, best0[n]);
        bidx0[n] = select_i(c0, ti, bidx0[n]);
    };
    auto fold_group = [&](const floatx16 (&acc)[2][NT], int tbase, int gi) {
        const int mt = gi / (NT * 4), n = (gi / 4) % NT, gq = gi % 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float key = acc[mt][n][4 * gq + r];
            const int ti = tbase + 32 * mt + 8 * gq + 4 * lh + r;
            const bool c0 = key < best0[n], c1 = key < best1[n];
            // the key that does not stay in the list, max(key, old second), may be the new third
            asm("v_med3_f32 %0, %0, %1, %2" : "+v"(third[n]) : "v"(key), "v"(best1[n]));
            const float t = c1 ? key : best1[n];
            const int tix = c1 ? ti : bidx1[n];
            best1[n] = c0 ? best0[n] : t;
            bidx1[n] = c0 ? bidx0[n] : tix;
            best0[n] = c0 ? key : best0[n];
            bidx0[n] = c0 ? ti : bidx0[n];
        }
    };

    auto tile_step = [&](floatx16 (&accC)[2][NT], const floatx16 (&accF)[2][NT], int k) {
        const int buf = k & 1;
        if (k + 1 < ntiles) stage(k + 1, buf ^ 1);
        const float* tile = lds + buf * kTileFloats;
        const float* nrm = norm_lds + buf * kTileRows;
        const int tbase_prev = t_begin + (k - 1) * kTileRows;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // accumulator register r of this lane is train row 32mt + 8(r>>2) + 4lh + (r&3)
            floatx16 init;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(nrm + 32 * mt + 8 * gq + 4 * lh);
                init[4 * gq + 0] = v.x;
                init[4 * gq + 1] = v.y;
                init[4 * gq + 2] = v.z;
                init[4 * gq + 3] = v.w;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) accC[mt][n] = init;
        }
        auto load_av = [&](int m, floatx4 (&av)[2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = 32 * mt + li;
                const int cpos = (2 * m + lh) ^ (row & 15);
                av[mt] = *reinterpret_cast<const floatx4*>(tile + row * K + cpos * 4);
            }
        };
        floatx4 av[2], av_next[2];
        load_av(0, av);
        // One key of the finished tile is folded after every kMfmaPerKey-th MFMA, and the
        // order is pinned: left to itself the compiler issues the MFMAs of an m step back to
        // back and the fold after them, and a wave stuck at the head of that MFMA queue
        // issues nothing else, so the fold then runs with the matrix pipe idle.
        constexpr int kMfmaPerKey = (kMSteps * 4 * 2 * NT) / (kGroups * 4);
        static_assert(kMfmaPerKey >= 1 && (kMSteps * 4 * 2 * NT) % (kGroups * 4) == 0, "whole MFMAs per key");
#pragma unroll
        for (int m = 0; m < kMSteps; ++m) {
            if (m + 1 < kMSteps) load_av(m + 1, av_next);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        accC[mt][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][s], qreg[n][m][s],
                                                                           accC[mt][n], 0, 0, 0);
                        const int unit = ((m * 4 + s) * 2 + mt) * NT + n;
                        if (unit % kMfmaPerKey == 0) fold_key(accF, tbase_prev, unit / kMfmaPerKey);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            av[0] = av_next[0];
            av[1] = av_next[1];
        }
        __syncthreads();
    };

    for (int k = 0; k < ntiles; k += 2) {
        tile_step(accA, accB, k);
        if (k + 1 < ntiles) tile_step(accB, accA, k + 1);
    }
    {
        const int tbase_last = t_begin + (ntiles - 1) * kTileRows;
        if (ntiles & 1) {
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi) fold_group(accA, tbase_last, gi);
        } else if (ntiles > 0) {
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi) fold_group(accB, tbase_last, gi);
        }
    }

    const size_t cbase = (((size_t)problem * a.splits + split) * a.rows_cap) * 4;
    int32_t* cand = a.cand + cbase;
    float* ckey = a.cand_key + cbase;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qi = q0 + wave * kQPerWave + 32 * n + li;
        if (qi < nq) {
            cand[(size_t)qi * 4 + 2 * lh] = bidx0[n];
            cand[(size_t)qi * 4 + 2 * lh + 1] = bidx1[n];
            ckey[(size_t)qi * 4 + 2 * lh] = best0[n];
            ckey[(size_t)qi * 4 + 2 * lh + 1] = best1[n];
            a.cand_third[cbase / 2 + (size_t)qi * 2 + lh] = third[n];
        }
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
template <bool F16> struct NarrowFragment { typedef bf16x8 type; };
template <> struct NarrowFragment<true> { typedef halfx8 type; };

__device__ __forceinline__ floatx16 narrow_mfma(bf16x8 a, bf16x8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 narrow_mfma(halfx8 a, halfx8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// The candidate pass on the bfloat16 matrix cores (sixteen times the rate of the fp32 ones).
// Every float x of the packed rows is split into xh + xl, two bfloat16 values that carry its
// upper sixteen significant bits (l2_pack_kernel), and q.t is formed as qh.th + qh.tl + ql.th
// -- three v_mfma_f32_32x32x16_bf16 contractions with exact products and fp32 accumulation.
// What is left out, ql.tl and the bits below xl, is below 4 * 2^-16 (||q||^2 + ||t||^2); the
// re-score pass widens its error bound by that much, so the final result is still the exact
// sequential one.  Layout and staging are those of the fp32 kernel: a split row is 2 K bfloat16
// = K floats wide, upper halves first.  A wave owns one 32-query column block (NT = 1) and two
// waves share a SIMD: one wave issues a vector instruction every six cycles at best, two
// interleave, and the fold -- a tag and five instructions per four keys -- runs underneath the
// matrix instructions of the next tile.
//
// F16 (the default): the narrow rows are single fp16 copies x * s of the floats, s a power of
// two common to all rows that brings the largest below 2^15 (l2_narrow_kernel), and q.t is one
// v_mfma_f32_32x32x16_f16 contraction -- a third of the matrix work on rows half as wide.  What
// the rounding to eleven bits leaves out is bounded per query by the norms of the rows'
// residuals x - fp16(x s) / s, which l2_narrow_kernel measures (Cauchy-Schwarz; zero for
// integer-valued descriptors up to 2048), and the re-score pass adds that bound to E.  The
// accumulators and the running lists then hold keys times s^2; the lists are scaled back when
// they are written out.  With a third of the MFMAs the fragment reads set the pace (a wave's
// ds_read_b128 keeps the LDS pipe 16 cycles: 64 bytes a cycle per CU), so a wave owns two
// column blocks (NT = 2: a fragment feeds two MFMAs) in workgroups of four waves, two of them
// resident per CU.
template <int K, int NT, int WAVES, bool F16>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(F16 ? 2 : 1)))
void l2_candidates_narrow_kernel(L2Args a) {
    typedef typename NarrowFragment<F16>::type frag;
    constexpr int kRowFloats = F16 ? K / 2 : K;      // a narrow row in floats
    constexpr int kChunksPerRow = kRowFloats / 4;    // 16-byte chunks of a narrow row
    constexpr int kSwizzle = kChunksPerRow >= 16 ? 15 : kChunksPerRow - 1;
    constexpr int kTerms = F16 ? 1 : 3;              // contractions per step
    constexpr int kTileFloats = kTileRows * kRowFloats;
    constexpr int kGroup = 1;   // tiles staged (and waited for) together; four measured no faster
    constexpr int kQPerWave = 32 * NT;
    constexpr int kQPerBlock = WAVES * kQPerWave;
    static_assert(kChunksPerRow % WAVES == 0, "the waves share the staging of a tile evenly");
    constexpr int kSteps = K / 16;                   // contraction steps of 16 dimensions
    __shared__ __attribute__((aligned(16))) float lds[2 * kGroup * (kTileFloats + kTileRows)];
    float* norm_lds = lds + 2 * kGroup * kTileFloats;
    static_assert(kGroup * kTileRows <= 64 * WAVES, "a thread per row stages the start values");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    if (F16 && (int)blockIdx.x >= (int)gridDim.x - a.fold_groups) {   // (see L2Args)
        const int g = (int)blockIdx.x - ((int)gridDim.x - a.fold_groups);
        fold_group(a.fold[0], g);
        fold_group(a.fold[1], g);
        return;
    }
    const int g = blockIdx.x / a.blocks_per_group, inner = blockIdx.x % a.blocks_per_group;
    const int per_dir = a.qtiles * a.splits;
    const int dir = inner / per_dir;
    const int qtile = (inner % per_dir) / a.splits;
    const int split = inner % a.splits;
    int qf = 0, tf = 0, problem = 0;
    if (a.pairs) {
        const int fa = a.pairs[2 * g], fb = a.pairs[2 * g + 1];
        qf = dir ? fb : fa;
        tf = dir ? fa : fb;
        problem = 2 * g + dir;
    }
    const int nq = a.counts ? a.counts[qf] : a.nq_uniform;
    const int nt = a.counts ? a.counts[tf] : a.nt_uniform;
    const int q0 = qtile * kQPerBlock;
    if (q0 >= nq) return;
    const float* qrows = a.q_base + (size_t)qf * a.frame_floats;
    const float* trows = a.t_base + (size_t)tf * a.frame_floats;
    const float* tacc0 = a.t_acc0 + (size_t)tf * a.rows_cap;   // -||t||^2 / 2 (times s^2)

    // B operand: this lane's query columns, upper and lower halves, resident for the whole
    // kernel.  Element [n][s] holds dimensions 16 s + 8 lh .. + 7 of query
    // q0 + wave * kQPerWave + 32 n + li.
    frag qh[NT][kSteps], ql[NT][kSteps];   // (ql: bfloat16 split only)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qi = q0 + wave * kQPerWave + 32 * n + li;
        const frag* src = reinterpret_cast<const frag*>(qrows + (size_t)(qi < nq ? qi : 0) * kRowFloats);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            frag h = src[2 * s + lh], l = F16 ? h : src[K / 8 + 2 * s + lh];
            if (qi >= nq) {
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = l[e] = 0;
            }
            qh[n][s] = h;
            ql[n][s] = l;
        }
    }
    // The matrix pipe accumulates q.t - ||t||^2 / 2, the ranking key ||t||^2 - 2 q.t times -1/2
    // (an exact scaling, so the key is the one the fp32 kernel forms): the largest accumulators
    // are the nearest rows, and neither operand needs a factor.
    // Running best two keys of each query column with their train rows, and the third-smallest
    // key seen.  kNoRow = 2^127, huge, finite and with clear low mantissa bits, stands for "no
    // row": a key keeps a position tag in those bits (below), which an infinity cannot carry.
    const float kNoRow = __uint_as_float(0x7F000000u);
    const float kNoRowAcc = -0.5f * kNoRow;
    float best0[NT], best1[NT], third[NT];
    int bidx0[NT], bidx1[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        best0[n] = best1[n] = kNoRow;
        third[n] = INFINITY;
        bidx0[n] = bidx1[n] = -1;
    }

    const int chunk = (nt + a.splits - 1) / a.splits;
    const int t_begin = split * chunk;
    const int t_end = min(nt, t_begin + chunk);
    const int ntiles = t_end > t_begin ? (t_end - t_begin + kTileRows - 1) / kTileRows : 0;

    // group g of kGroup tiles into buffer buf.  A lane's chunk sits at the same place of every
    // tile: while the tile lies inside the set its address is a wave-uniform base, advanced by
    // the scalar unit, plus a byte offset that costs one instruction -- staging instruction i
    // covers rows i R .. i R + R - 1 (R = 64 / chunks per row), so the row of a lane is
    // i R + lane / chunks, its swizzle term (i R & kSwizzle) | lane / chunks, and its offset
    // lane_off ^ 16 (i R & kSwizzle) beyond the instruction's first row.  Only a tile that
    // reaches past the last row forms its addresses per lane (rows past the end read the last
    // row and are masked by a start value of kNoRowAcc).
    constexpr int kStageInstrs = kGroup * kChunksPerRow / WAVES;
    constexpr int kRowsPerInstr = 64 / kChunksPerRow;
    static_assert(kGroup == 1 && 64 % kChunksPerRow == 0 && kRowsPerInstr - 1 <= kSwizzle, "the offset formula");
    const uint32_t lane_off = (uint32_t)((((lane % kChunksPerRow) ^ (lane / kChunksPerRow)) * 16) |
                                         ((lane / kChunksPerRow) * (kRowFloats * 4)));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // (in a scalar register)
    auto stage = [&](int g, int buf) {
        const int tbase = t_begin + g * kGroup * kTileRows;
        float* dst = lds + buf * kGroup * kTileFloats;
        if (tbase + kGroup * kTileRows <= nt) {
            // (formed here, from one register: offsets kept across the loop would be spilled)
            uint32_t lo = lane_off;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int it = 0; it < kStageInstrs; ++it) {
                const int instr = wave_u * kStageInstrs + it;
                const char* base = reinterpret_cast<const char*>(
                    trows + ((size_t)tbase + (size_t)(instr * kRowsPerInstr)) * kRowFloats);
                const uint32_t off = lo ^ (uint32_t)(((instr * kRowsPerInstr) & kSwizzle) * 16);
                glds16(base + off, dst + instr * 256);
            }
        } else {
#pragma unroll
            for (int it = 0; it < kStageInstrs; ++it) {
                const int instr = wave_u * kStageInstrs + it;
                const int p = instr * 64 + lane;
                const int row = p / kChunksPerRow, cpos = p % kChunksPerRow;
                const int c = cpos ^ (row & kSwizzle);
                int trow = tbase + row;
                trow = trow < nt ? trow : nt - 1;
                glds16(trows + (size_t)trow * kRowFloats + c * 4, dst + instr * 256);
            }
        }
        if (tid < kGroup * kTileRows) {
            const int trow = tbase + tid;
            // q.t changes nothing in kNoRowAcc (its ulp is 2^103): a masked row's key is never
            // below a best
            norm_lds[buf * kGroup * kTileRows + tid] = trow < t_end ? tacc0[trow] : kNoRowAcc;
        }
    };

    if (ntiles > 0) stage(0, 0);
    __syncthreads();

    floatx16 accA[2][NT], accB[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[mt][n][r] = accB[mt][n][r] = kNoRowAcc;

    // The fold of a finished tile.  A lane holds 32 accumulators of each of its query columns;
    // the kNarrowTagBits low mantissa bits of one are replaced by a tag (a change below
    // 2^(kNarrowTagBits - 23) of its size, which the error bound of the re-score pass covers):
    // its position in the tile in the five low bits (an inline constant of the instruction),
    // zero above.  The two largest of a stretch of kMergeTiles tiles fall out of medians and
    // maxima of three with no train index carried along (fold_pair): before the next tile of the
    // stretch is folded, 32 is added to the bit patterns of the two (age_tops: the upper tag
    // bits count the tiles folded since, their order stays), so the tagged values of a stretch
    // are all different.  The two enter the running lists once per stretch, as keys, with the
    // train row their tags name.  (Keeping the third largest as well, which makes cand_third
    // a bound on every row outside the lists, costs three instructions per key instead of
    // five per four: 121 against 107 us on 20 000 x 20 000 x 128.)
    constexpr unsigned kTagMask = (1u << kNarrowTagBits) - 1u;
    float top0[NT], top1[NT];
    auto tagged = [&](const floatx16 (&acc)[2][NT], int kq) {
        const int gi = kq / 4, r = kq % 4;
        const int mt = gi / (NT * 4), n = (gi / 4) % NT, gq = gi % 4;
        const unsigned tag = 16 * mt + 4 * gq + r;
        return __uint_as_float((__float_as_uint(acc[mt][n][4 * gq + r]) & ~kTagMask) | tag);
    };
    // Keys 2 p and 2 p + 1 (one query column).  With t0 >= t1 the largest two so far, the
    // second largest of {t0, t1, x, y} is max(t1, median(t0, x, y)); pairs come in twos (the
    // four keys of one register group), and the two updates of t1 are one maximum of three:
    // five instructions per four keys.  That maximum is issued with the next group's first
    // pair, two instructions behind the median it reads (the compiler puts a wait state
    // between an asm statement and a reader right behind it).
    float mid_a[NT], mid_b[NT];
    auto fold_pair = [&](const floatx16 (&acc)[2][NT], int p, int last) {
        const int n = (p / 8) % NT;
        const float x = tagged(acc, 2 * p), y = tagged(acc, 2 * p + 1);
        if ((p & 1) == 0) {
            if (p > 0) {
                const int m = ((p - 1) / 8) % NT;
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top1[m]) : "v"(mid_a[m]), "v"(mid_b[m]));
            }
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid_a[n]) : "v"(top0[n]), "v"(x), "v"(y));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top0[n]) : "v"(x), "v"(y));
        } else {
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid_b[n]) : "v"(top0[n]), "v"(x), "v"(y));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top0[n]) : "v"(x), "v"(y));
            if (p == last)
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(top1[n]) : "v"(mid_a[n]), "v"(mid_b[n]));
        }
    };
    // (after the first tile of a stretch the two are finite: masked rows rank as kNoRowAcc)
    auto age_tops = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            top0[n] = __uint_as_float(__float_as_uint(top0[n]) + 32u);
            top1[n] = __uint_as_float(__float_as_uint(top1[n]) + 32u);
        }
    };
    // which = 0, 1: the largest, second-largest accumulator of the stretch whose last folded
    // tile starts at train row tbase enters the lists of column n as a key (strict
    // comparisons: of equal keys the earlier stretch stays ahead), and moves the third-smallest
    // key seen.
    auto merge = [&](int n, int which, int tbase) {
        const float v = which == 0 ? top0[n] : top1[n];
        const float key = -2.f * v;   // exact
        asm("v_med3_f32 %0, %0, %1, %2" : "+v"(third[n]) : "v"(key), "v"(best1[n]));
        const unsigned tag = __float_as_uint(v);
        const int ti = tbase + 4 * lh + (int)(((tag & 28u) << 1) | (tag & 3u)) - (int)((tag & (kTagMask & ~31u)) << 1);
        const unsigned long long c0 = __builtin_amdgcn_ballot_w64(key < best0[n]);
        const unsigned long long c1 = __builtin_amdgcn_ballot_w64(key < best1[n]);
        const float t = select_f(c1, key, best1[n]);
        const int tix = select_i(c1, ti, bidx1[n]);
        best1[n] = select_f(c0, best0[n], t);
        bidx1[n] = select_i(c0, bidx0[n], tix);
        best0[n] = select_f(c0, key, best0[n]);
        bidx0[n] = select_i(c0, ti, bidx0[n]);
    };
    auto reset_tops = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n) top0[n] = top1[n] = -INFINITY;
    };

    constexpr int kPairs = NT * 16;                // pairs of keys of a finished tile per lane
    constexpr int kMfmas = kSteps * 2 * NT * kTerms;   // MFMAs of a tile per wave
    static_assert(kMfmas > 2 * NT, "the fold needs room beside the merges");
    // tile k, the P-th of its stretch: its MFMAs, and underneath them the fold of tile k - 1;
    // the first tile of a stretch carries the merges of the stretch before it in its last
    // 3 NT units
    auto tile_step = [&](floatx16 (&accC)[2][NT], const floatx16 (&accF)[2][NT], int k, auto phase) {
        constexpr int P = decltype(phase)::value;
        constexpr int PF = (P + kMergeTiles - 1) % kMergeTiles;   // place of the folded tile
        constexpr int kFoldUnits = P == 0 ? kMfmas - 2 * NT : kMfmas;
        const int group = k / kGroup, sub = k % kGroup, buf = group & 1;
        if (sub == 0 && (group + 1) * kGroup < ntiles) stage(group + 1, buf ^ 1);
        const float* tile = lds + (buf * kGroup + sub) * kTileFloats;
        const float* nrm = norm_lds + (buf * kGroup + sub) * kTileRows;
        const int tbase_prev = t_begin + (k - 1) * kTileRows;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            floatx16 init;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(nrm + 32 * mt + 8 * gq + 4 * lh);
                init[4 * gq + 0] = v.x;
                init[4 * gq + 1] = v.y;
                init[4 * gq + 2] = v.z;
                init[4 * gq + 3] = v.w;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) accC[mt][n] = init;
        }
        // fragments of step s: chunk 2 s + lh of the upper halves, the same chunk of the lower
        // halves K / 8 chunks further on
        auto load_a = [&](int s, frag (&ah)[2], frag (&al)[2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = 32 * mt + li;
                const int ch = (2 * s + lh) ^ (row & kSwizzle);
                ah[mt] = *reinterpret_cast<const frag*>(tile + row * kRowFloats + ch * 4);
                if (!F16) {
                    const int cl = (K / 8 + 2 * s + lh) ^ (row & kSwizzle);
                    al[mt] = *reinterpret_cast<const frag*>(tile + row * kRowFloats + cl * 4);
                }
            }
        };
        frag ah[2], al[2], ah_next[2], al_next[2];
        load_a(0, ah, al);
        // the fold of the finished tile is dealt over the MFMAs of this one, in a pinned order
        // (see the fp32 kernel)
        if (PF == 0) reset_tops();
        else age_tops();
        int unit = 0;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            if (s + 1 < kSteps) load_a(s + 1, ah_next, al_next);
            // the three products of one accumulator are 2 NT MFMAs apart: none waits for the
            // one before it, and the fold issues underneath them
#pragma unroll
            for (int term = 0; term < kTerms; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const frag av = term == 2 ? al[mt] : ah[mt];
                        const frag bv = term == 1 ? ql[n][s] : qh[n][s];
                        accC[mt][n] = narrow_mfma(av, bv, accC[mt][n]);
                        if (unit < kFoldUnits) {
#pragma unroll
                            for (int p = (unit * kPairs) / kFoldUnits; p < ((unit + 1) * kPairs) / kFoldUnits; ++p)
                                fold_pair(accF, p, kPairs - 1);
                        } else {
                            merge((unit - kFoldUnits) / 2, (unit - kFoldUnits) % 2, tbase_prev);
                        }
                        ++unit;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                ah[mt] = ah_next[mt];
                al[mt] = al_next[mt];
            }
        }
        if (sub == kGroup - 1) __syncthreads();   // (past the last tile nothing is read)
    };

    reset_tops();   // (the first tile folds the start values: nothing enters)
    // the tiles of a stretch, each with its place as a constant; the accumulator sets alternate
    auto stretch = [&](int k, auto... place) {
        ((k + decltype(place)::value < ntiles
              ? (decltype(place)::value & 1 ? tile_step(accB, accA, k + decltype(place)::value, place)
                                            : tile_step(accA, accB, k + decltype(place)::value, place),
                 0)
              : 0),
         ...);
    };
    for (int k = 0; k < ntiles; k += kMergeTiles) {
        static_assert(kMergeTiles == 4, "one constant per tile of the stretch");
        stretch(k, std::integral_constant<int, 0>(), std::integral_constant<int, 1>(),
                std::integral_constant<int, 2>(), std::integral_constant<int, 3>());
    }
    if (ntiles > 0) {
        // the last tile, and the stretch it ends
        const int tbase_last = t_begin + (ntiles - 1) * kTileRows;
        if ((ntiles - 1) % kMergeTiles == 0) reset_tops();
        else age_tops();
        if (ntiles & 1) {
#pragma unroll
            for (int p = 0; p < kPairs; ++p) fold_pair(accA, p, kPairs - 1);
        } else {
#pragma unroll
            for (int p = 0; p < kPairs; ++p) fold_pair(accB, p, kPairs - 1);
        }
#pragma unroll
        for (int u = 0; u < 2 * NT; ++u) merge(u / 2, u % 2, tbase_last);
    }

    const float unscale = F16 ? a.narrow_scale[1] : 1.f;   // 1 / s^2
    const size_t cbase = (((size_t)problem * a.splits + split) * a.rows_cap) * 4;
    int32_t* cand = a.cand + cbase;
    float* ckey = a.cand_key + cbase;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qi = q0 + wave * kQPerWave + 32 * n + li;
        if (qi < nq) {
            cand[(size_t)qi * 4 + 2 * lh] = bidx0[n];
            cand[(size_t)qi * 4 + 2 * lh + 1] = bidx1[n];
            // (a power of two; an entry without a row may leave the float range: it is not read)
            ckey[(size_t)qi * 4 + 2 * lh] = best0[n] * unscale;
            ckey[(size_t)qi * 4 + 2 * lh + 1] = best1[n] * unscale;
            a.cand_third[cbase / 2 + (size_t)qi * 2 + lh] = third[n] * unscale;
        }
    }
}

// Sum over the wave, the same value in every lane: quad permutes and row mirrors inside each row
// of 16 lanes (no LDS traffic), then the four row sums through scalar registers.
template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = dpp_add_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_f32<0x141>(v);   // row_half_mirror
    v = dpp_add_f32<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Group (frame) and position in it of the rows row0 .. row0 + 7 of a wave (row0 wave-uniform;
// rows past the end repeat the last row): one division for the first row, steps from there.
__device__ __forceinline__ void walk_rows(int row0, int rows, int rows_per_group, int (&group)[kPackRowsPerWave],
                                          int (&local)[kPackRowsPerWave]) {
    const int first = min(row0, rows - 1);
    int g = first / rows_per_group, l = first - g * rows_per_group;
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) {
        group[r] = g;
        local[r] = l;
        if (row0 + r + 1 < rows && ++l == rows_per_group) {
            l = 0;
            ++g;
        }
    }
}

// Up to two sets of float rows (queries and train rows, or the frames of a sequence) are brought
// into the layout of the passes in one launch: zero-padded rows of K floats, ||row||^2, for the
// bfloat16 pass the rows split into 2 K bfloat16 -- the K upper halves xh = bf16(x), then the K
// lower halves xl = bf16(x - xh), both rounded to nearest even (x - xh is exact in fp32) -- and
// the largest norms a workgroup saw (two slots per workgroup, see FoldSet; rows beyond a
// group's count and NaN norms are left out).
struct PackSet {
    const uint8_t* src;
    int rows, rows_per_group;          // row r lies in group r / rows_per_group
    size_t group_stride, row_stride;   // bytes
    float* dst;                        // [rows][K], or null (the rows are used in place)
    __bf16* split;                     // [rows][2 K], or null
    float* norms;                      // [rows]
    float* acc0;                       // [rows]: -||row||^2 / 2, or null
    const int32_t* counts;             // rows in use per group, or null: the rest does not enter the maxima
    float* slots;                      // [2][workgroups of the set]
};
struct PackArgs {
    PackSet set[2];
    int blocks0;   // workgroups of set 0; the rest serve set 1
    int dim, K;
    int32_t* clear[4];   // counters of later kernels that this launch zeroes (or null)
};
// ALIGNED: every row starts on an 8-byte boundary and dim is even; SPLIT: the bfloat16 halves are written.
template <bool ALIGNED, bool SPLIT>
__global__ __launch_bounds__(kPackBlock) void l2_pack_kernel(PackArgs a) {
    const bool second = (int)blockIdx.x >= a.blocks0;
    const PackSet& p = a.set[second ? 1 : 0];
    const int block = second ? blockIdx.x - a.blocks0 : blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = (block * (kPackBlock / 64) + wave) * kPackRowsPerWave;
    const int block_group = min(block * kPackRowsPerBlock, p.rows - 1) / p.rows_per_group;
    int group[kPackRowsPerWave], local[kPackRowsPerWave];
    walk_rows(row0, p.rows, p.rows_per_group, group, local);
    const int K = a.K;
    if (blockIdx.x == 0 && threadIdx.x < 4 && a.clear[threadIdx.x]) *a.clear[threadIdx.x] = 0;
    // A lane takes element pairs lane and lane + 64 of a row (K <= 256: at most 128 pairs; the
    // second pair only exists for K = 256).  All the loads of the wave's rows go out before the
    // first value is used: a lane past the row's end reads the row's last elements instead and
    // replaces them by zeros, so no load sits behind a branch of its own.
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const int dim = a.dim;
    const bool second_pair = K > 128;   // wave-uniform
    floatx2 v[kPackRowsPerWave][2];
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) {
        const float* s = reinterpret_cast<const float*>(p.src + (size_t)group[r] * p.group_stride +
                                                        (size_t)local[r] * p.row_stride);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[r][i] = floatx2{0.f, 0.f};
            if (i == 1 && !second_pair) continue;
            const int j = 2 * (lane + 64 * i);
            if (ALIGNED) {
                const floatx2 x = *reinterpret_cast<const floatx2*>(s + max(min(j, dim - 2), 0));
                v[r][i] = j < dim ? x : floatx2{0.f, 0.f};
            } else {
                const float x0 = s[max(min(j, dim - 1), 0)], x1 = s[max(min(j + 1, dim - 1), 0)];
                v[r][i].x = j < dim ? x0 : 0.f;
                v[r][i].y = j + 1 < dim ? x1 : 0.f;
            }
        }
    }
    float acc[kPackRowsPerWave];
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) acc[r] = 0.f;
    // (the copy is left out when the caller's rows already are the packed ones: the choice is
    // made once, outside the loops)
    auto convert = [&](auto copy) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 2 * (lane + 64 * i);
            if (j < K) {
#pragma unroll
                for (int r = 0; r < kPackRowsPerWave; ++r) {
                    const int row = row0 + r;
                    if (row < p.rows) {   // wave-uniform
                        const floatx2 x = v[r][i];
                        if (decltype(copy)::value) *reinterpret_cast<floatx2*>(p.dst + (size_t)row * K + j) = x;
                        if (SPLIT) {
                            bf16x2 h, l;
                            h.x = (__bf16)x.x;
                            h.y = (__bf16)x.y;
                            l.x = (__bf16)(x.x - (float)h.x);
                            l.y = (__bf16)(x.y - (float)h.y);
                            *reinterpret_cast<bf16x2*>(p.split + (size_t)row * 2 * K + j) = h;
                            *reinterpret_cast<bf16x2*>(p.split + (size_t)row * 2 * K + K + j) = l;
                        }
                        acc[r] += x.x * x.x;
                        acc[r] += x.y * x.y;
                    }
                }
            }
        }
    };
    if (p.dst)
        convert(std::true_type{});
    else
        convert(std::false_type{});
    float largest = 0.f, largest_later = 0.f;
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) {
        const int row = row0 + r;
        if (row >= p.rows) break;   // wave-uniform
        const float norm = wave_sum_f32(acc[r]);
        if (lane == 0) {
            p.norms[row] = norm;
            if (p.acc0) p.acc0[row] = -0.5f * norm;
        }
        // rows past a frame's count are whatever the caller's buffer holds: they are converted
        // like the others (nothing reads them) but must not set the scale or the error bound
        const bool in_use = !p.counts || local[r] < p.counts[group[r]];
        if (group[r] == block_group)
            largest = in_use && norm > largest ? norm : largest;
        else
            largest_later = in_use && norm > largest_later ? norm : largest_later;
    }
    block_max2(largest, largest_later);
    if (threadIdx.x == 0) {
        p.slots[block] = largest;
        p.slots[(p.rows + kPackRowsPerBlock - 1) / kPackRowsPerBlock + block] = largest_later;
    }
}

// The groups' largest norms by a launch of its own: the passes without l2_narrow_kernel, which
// otherwise does this on the side.
__global__ __launch_bounds__(kBlock) void l2_group_fold_kernel(FoldArgs f) { fold_group(f, blockIdx.x); }

// The largest norm of all: only for more slots than every workgroup of l2_narrow_kernel
// would read by itself.
struct SlotsMaxArgs {
    const float* slots[2];
    long n_slots[2];
    float* out;
};
__global__ __launch_bounds__(1024) void l2_slots_max_kernel(SlotsMaxArgs a) {
    float m = 0.f, unused = 0.f;
    for (int i = 0; i < 2; ++i)
        for (long k = threadIdx.x; k < a.n_slots[i]; k += 1024) m = fmaxf(m, a.slots[i][k]);
    block_max2(m, unused);
    if (threadIdx.x == 0) *a.out = m;
}

// Packed rows -> single fp16 copies for the fp16 candidate pass, in one launch for up to two
// sets.  s = 2^shift is common to all rows: the largest norm times s stays below 2^15, so no
// element leaves the fp16 range, and products of scaled elements are exact in the matrix
// pipe's fp32.  Per row: the fp16 row fp16(x s) (round to nearest even); the norm of what the
// rounding left out, ||x - fp16(x s) / s|| (the differences are exact in fp32; the sum is
// rounded up by a part in a thousand, and by the most that a pipe reading fp16 denormals as
// zero could drop); the accumulator start value -||x||^2 s^2 / 2.  Per workgroup the largest
// residual norms (two slots, see FoldSet).  scale_out: s^2 and 1 / s^2.  The largest norm of all
// rows, which sets s, is read from the slots l2_pack_kernel left (norm_slots; every workgroup
// goes through all of them) or, when there are too many for that, from *global_max.
struct NarrowSet {
    const float* packed;   // [rows][K]
    const float* norms;    // [rows]
    int rows;
    _Float16* half;        // [rows][K]
    float* res;            // [rows]
    float* acc0;           // [rows], or null
    int rows_per_group;    // as in PackSet
    const int32_t* counts;
    float* slots;          // [2][workgroups of the set]
};
struct NarrowArgs {
    NarrowSet set[2];
    int blocks0;
    int K;
    const float* norm_slots[2];
    int n_norm_slots[2];       // 0: *global_max holds the largest norm
    const float* global_max;
    float* scale_out;
};
constexpr long kDirectSlots = 8192;   // slots a workgroup of l2_narrow_kernel reads by itself at most

__global__ __launch_bounds__(kPackBlock) void l2_narrow_kernel(NarrowArgs a) {
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
    const bool second = (int)blockIdx.x >= a.blocks0;
    const NarrowSet& p = a.set[second ? 1 : 0];
    const int block = second ? blockIdx.x - a.blocks0 : blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int row0 = (block * (kPackBlock / 64) + wave) * kPackRowsPerWave;
    const int block_group = min(block * kPackRowsPerBlock, p.rows - 1) / p.rows_per_group;
    int group[kPackRowsPerWave], local[kPackRowsPerWave];
    walk_rows(row0, p.rows, p.rows_per_group, group, local);
    const int K = a.K;
    // m: the largest ||row||^2 of all (its loads go first; the rows' loads follow before it is used)
    float m = 0.f, unused = 0.f;
    const bool direct = a.n_norm_slots[0] + a.n_norm_slots[1] > 0;
    if (direct) {
        // eight loads per thread and set go out together (a lane past the end reads the last slot
        // again); only more than 2048 slots of a set take the loop behind them
        constexpr int kAtOnce = 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = a.n_norm_slots[i];
            if (n <= 0) continue;   // workgroup-uniform
            float t[kAtOnce];
#pragma unroll
            for (int u = 0; u < kAtOnce; ++u) t[u] = a.norm_slots[i][min((int)threadIdx.x + u * kPackBlock, n - 1)];
#pragma unroll
            for (int u = 0; u < kAtOnce; ++u) m = fmaxf(m, t[u]);
            for (int k = threadIdx.x + kAtOnce * kPackBlock; k < n; k += kPackBlock) m = fmaxf(m, a.norm_slots[i][k]);
        }
    } else {
        m = *a.global_max;
    }
    const bool second_pair = K > 128;   // wave-uniform: pairs 64 .. 127 of a row exist
    floatx2 v[kPackRowsPerWave][2];
    float norm[kPackRowsPerWave];
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) {
        const int row = min(row0 + r, p.rows - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[r][i] = floatx2{0.f, 0.f};
            if (i == 1 && !second_pair) continue;
            // (a lane past the row reads its last pair instead: no load behind a branch of its own)
            const int j = min(2 * (lane + 64 * i), K - 2);
            v[r][i] = *reinterpret_cast<const floatx2*>(p.packed + (size_t)row * K + j);
        }
        norm[r] = p.norms[row];
    }
    // m: the largest ||row||^2 of all groups; sqrt(m) < 2^e, so |x| s < 2^15 for every element x
    // of every row
    if (direct) block_max2(m, unused);
    int shift = 0;
    if (m > 0.f && m < INFINITY) {
        // m = f 2^ex with 0.5 <= f < 1 (frexp); a denormal m counts as 2^-126, which the cap
        // below covers as well
        const int ex = (int)((__float_as_uint(m) >> 23) & 255u) - 126;
        shift = 15 - ((ex + 1) >> 1);
        shift = shift > 60 ? 60 : shift;   // s^2 stays a float; tiny rows are scaled less
    }
    // powers of two between 2^-120 and 2^120, written as their bit patterns
    auto pow2 = [](int e) { return __uint_as_float((uint32_t)(e + 127) << 23); };
    const float s = pow2(shift), inv_s = pow2(-shift);
    const float s2 = pow2(2 * shift);
    const float denormal_part = sqrtf((float)K) * pow2(-shift - 14);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.scale_out[0] = s2;
        a.scale_out[1] = pow2(-2 * shift);
    }
    float acc[kPackRowsPerWave];
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 2 * (lane + 64 * i);
        if (j < K) {
#pragma unroll
            for (int r = 0; r < kPackRowsPerWave; ++r) {
                const int row = row0 + r;
                if (row < p.rows) {   // wave-uniform
                    const floatx2 x = v[r][i];
                    halfx2 h;
                    h.x = (_Float16)(x.x * s);
                    h.y = (_Float16)(x.y * s);
                    *reinterpret_cast<halfx2*>(p.half + (size_t)row * K + j) = h;
                    const float d0 = x.x - (float)h.x * inv_s, d1 = x.y - (float)h.y * inv_s;
                    acc[r] += d0 * d0;
                    acc[r] += d1 * d1;
                }
            }
        }
    }
    float largest = 0.f, largest_later = 0.f;
#pragma unroll
    for (int r = 0; r < kPackRowsPerWave; ++r) {
        const int row = row0 + r;
        if (row >= p.rows) break;   // wave-uniform
        // (+ what a matrix pipe that reads fp16 denormals as zero would drop: up to 2^-14 / s
        // in each of the K elements)
        const float res = (sqrtf(wave_sum_f32(acc[r])) + denormal_part) * 1.001f;
        if (lane == 0) {
            p.res[row] = res;
            if (p.acc0) p.acc0[row] = -0.5f * norm[r] * s2;
        }
        const bool in_use = !p.counts || local[r] < p.counts[group[r]];
        if (group[r] == block_group)
            largest = in_use && res > largest ? res : largest;
        else
            largest_later = in_use && res > largest_later ? res : largest_later;
    }
    block_max2(largest, largest_later);
    if (threadIdx.x == 0) {
        p.slots[block] = largest;
        p.slots[(p.rows + kPackRowsPerBlock - 1) / kPackRowsPerBlock + block] = largest_later;
    }
}

struct RescoreArgs {
    const float* q_base;
    const float* t_base;
    size_t frame_floats;
    int rows_cap;
    const int32_t* counts;
    int nq_uniform, nt_uniform;
    const int32_t* pairs;
    int splits;
    int K, dim;
    int squared;
    const int32_t* cand;
    const float* cand_key;
    const float* cand_third;
    // Narrow candidate passes: train rows per stretch (kMergeTiles tiles).  A list then holds the
    // best two of every stretch of its rows, and cand_third the third-smallest key among those
    // pairs: the other rows of a stretch rank behind its second, which is behind that key unless
    // the stretch gave the list both its entries -- l2_rescore_kernel scores such a stretch row
    // by row when its second is too near.  0: cand_third bounds every other row of the list.
    int stretch_rows;
    const float* q_norms;  // ||q||^2 per packed row, rows_cap per frame
    const float* t_norm_max;  // per train frame: the largest ||t||^2 of its rows in use
    double key_error;  // bound on |ranking key + ||q||^2 - sequential distance| / (||q||^2 + max ||t||^2)
    const float* q_res;      // fp16 pass: residual norm of every query row (rows_cap per frame), or null
    const float* t_res_max;  // fp16 pass, per train frame: the largest residual norm of its rows in use
    int force_exhaustive;
    int count_stretches;   // the statistics count the stretch entries too (a measuring aid)
    int32_t* knn_idx;  // [problem][rows_cap][2]
    float* knn_dist;
    int32_t* todo_count;  // queries whose result is not proven: count, then (problem, query) pairs
    int32_t* todo;
    // Queries proven but for one stretch of one list (see stretch_rows): count, and entries of
    // eight numbers -- problem, query, first row and end of the stretch, query frame, train
    // frame, lane half of the list -- which say all that l2_exhaustive_kernel needs to start loading (a chain of
    // look-ups per entry would cost it more than the sums).  When the array is full the query
    // goes to the exhaustive pass proper.
    int32_t* stretch_count;
    int32_t* stretch_items;
    int stretch_cap;
    int32_t* exhaustive_done;  // workgroups of the exhaustive pass that have finished (zero between launches)
    int32_t* stat_exhaustive;  // per-context total of such queries since the entry point began
    unsigned long long* partial;  // exhaustive pass: best two of every (entry, slice)
};

// min(v, v of the lane the DPP control names); a lane the control leaves out keeps its own
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < v ? o : v;
}

// Smallest v of the wave, in every lane.  Data-parallel-primitive moves in the vector ALU: a
// dozen of these per query through ds_bpermute kept the LDS pipe of a CU busy longer than the
// rest of the re-score kernel takes.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_min_u64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_min_u64<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_min_u64<0x140, 0xf>(v);   // row_mirror: every lane of a row of 16 holds the row's
    v = dpp_min_u64<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_min_u64<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave's
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 63);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// Smallest v of lanes 0 .. 3, in every lane.
__device__ __forceinline__ unsigned long long quad0_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_min_u64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 0);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), 0);
    return ((unsigned long long)hi << 32) | lo;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
    const unsigned o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
// (The minimum with its operand read through the data-parallel-primitive path, one instruction
// a step in place of copy, wait, move and minimum, measured no faster here.)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {   // same moves as wave_min_u64
    v = dpp_min_u32<0xB1, 0xf>(v);
    v = dpp_min_u32<0x4E, 0xf>(v);
    v = dpp_min_u32<0x141, 0xf>(v);
    v = dpp_min_u32<0x140, 0xf>(v);
    v = dpp_min_u32<0x142, 0xa>(v);
    v = dpp_min_u32<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int kRescored = 8;  // candidates per query whose distance is recomputed
constexpr int kStretchCap = 1 << 16;    // entries of RescoreArgs::stretch_items

// One wave per (problem, query): lane c holds candidate c of the query (four per train
// split).  Up to kRescored candidates with the smallest ranking keys are re-scored, in two
// rounds of four -- the second only when the fifth key is not out of reach of the second place
// after the first (two queries of 20 000 on the bench sets) --; the rest cannot be among the
// two nearest unless more than kRescored - 2 candidates lie within the rounding of the MFMA
// form of the second-best distance.  (Re-scoring all of them gathered 64 train rows per query:
// 655 MB for 20 000 queries, most of this kernel's time.)
//
// The squared differences of a re-scored row are formed by the whole wave -- lane l takes
// elements KPL l .. KPL l + KPL - 1 of the row (one coalesced load), the loads of all the rows
// in flight together -- and parked in LDS; the lane that owns the candidate then adds them up
// in element order.  KPL = K / 64.
template <int KPL>
__global__ __launch_bounds__(kBlock) void l2_rescore_kernel(RescoreArgs a) {
    constexpr int K = 64 * KPL;
    constexpr int kPitch = K + 4;   // the owners' 16-byte reads fall in different banks
    __shared__ __attribute__((aligned(16))) float s_terms[kBlock / 64][kRescored / 2 * kPitch];
    typedef float rowslice __attribute__((ext_vector_type(KPL)));
    const int lane = threadIdx.x & 63;
    const int problem = blockIdx.y;
    const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int qf = 0, tf = 0;
    if (a.pairs) {
        const int gp = problem >> 1, dir = problem & 1;
        const int fa = a.pairs[2 * gp], fb = a.pairs[2 * gp + 1];
        qf = dir ? fb : fa;
        tf = dir ? fa : fb;
    }
    const int nq = a.counts ? a.counts[qf] : a.nq_uniform;
    if (q >= nq) return;  // wave-uniform
    const float* qrow = a.q_base + (size_t)qf * a.frame_floats + (size_t)q * K;
    const float* trows = a.t_base + (size_t)tf * a.frame_floats;
    const int ncand = a.splits * 4;
    // A candidate's ranking key in orderable form, its six low bits replaced by the lane that
    // holds it: all different, so the smallest kRescored are well defined, and within 2^-17 of
    // the key (the error bound of the proof allows for it).
    int ti = -1;
    unsigned pre = 0xFFFFFFFFu, held_key = 0xFFFFFFFFu;
    if (lane < ncand) {
        const int split = lane >> 2, slot = lane & 3;
        const size_t o = ((((size_t)problem * a.splits + split) * a.rows_cap) + q) * 4 + slot;
        ti = a.cand[o];
        if (ti >= 0) {
            const unsigned u = __float_as_uint(a.cand_key[o]);
            held_key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            pre = (held_key & ~63u) | (unsigned)lane;
        }
    }
    const int held = ti;   // (the list entry this lane holds, re-scored or not)
    // The kRescored candidates with the smallest keys, in order (a key names its lane); fifth
    // and ninth: the smallest key behind the first round's rows and behind both rounds' rows.
    constexpr int kRound = kRescored / 2;
    int row_of[kRound];
    unsigned rest = pre;
    auto rank_round = [&]() {   // the next kRound smallest keys' rows; returns the key behind them
#pragma unroll
        for (int r = 0; r < kRound; ++r) {
            const unsigned m = wave_min_u32(rest);   // (0xFFFFFFFF: none left)
            row_of[r] = __builtin_amdgcn_readlane(ti, (int)(m & 63u));
            if (m == 0xFFFFFFFFu) row_of[r] = -1;
            if (rest == m) rest = 0xFFFFFFFFu;
        }
        return wave_min_u32(rest);
    };
    const unsigned fifth = rank_round();
    // Lane 2 split + half looks at the list of that split and lane half: the key cand_third
    // bounds, and -- narrow passes -- whether both entries came from one stretch of the list's
    // rows (same_stretch: the stretch's first row; second_key: the key the stretch's other
    // rows cannot be below).
    unsigned lists = 0xFFFFFFFFu;
    int same_stretch = -1;
    unsigned second_key = 0xFFFFFFFFu;
    const int nt_rows = a.counts ? a.counts[tf] : a.nt_uniform;
    const int split_rows = (nt_rows + a.splits - 1) / a.splits;
    if (lane < 2 * a.splits) {
        const int split = lane >> 1, half = lane & 1;
        const size_t o = (((size_t)problem * a.splits + split) * a.rows_cap) + q;
        const float k3 = a.cand_third[o * 2 + half];
        const unsigned u = __float_as_uint(k3);
        lists = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    }
    if (a.stretch_rows > 0) {   // the list's entries sit in lanes 2 l and 2 l + 1
        const int i0 = __shfl(held, (2 * lane) & 63), i1 = __shfl(held, (2 * lane + 1) & 63);
        const unsigned key1 = __shfl(held_key, (2 * lane + 1) & 63);
        const int first = (lane >> 1) * split_rows;
        if (lane < 2 * a.splits && i0 >= 0 && i1 >= 0 &&
            ((i0 - first) ^ (i1 - first)) < a.stretch_rows) {   // (a power of two; both at or above first)
            same_stretch = first + ((i0 - first) & ~(a.stretch_rows - 1));
            second_key = key1;
        }
    }
    lists = wave_min_u32(lists);
    // Proof that no other row can reach the second place: its sequential squared distance is
    // at least its ranking key + ||q||^2 - E >= T + ||q||^2 - E.  E bounds the rounding of the
    // norms (pairwise sums), of the K accumulation steps of the matrix pipe (each at most one
    // rounding of a partial sum <= ||t||^2 + 2 sum |q_j t_j| <= ||q||^2 + 2 ||t||^2) and of the
    // sequential sum itself (relative (K + 3) eps of a value <= 2 (||q||^2 + ||t||^2)); the
    // factor (4 K + 64) eps is more than twice their total.  The bfloat16 candidate pass has
    // three times the accumulation steps and leaves out 4 * 2^-16 of the norms (see there).
    const double qn = (double)a.q_norms[(size_t)qf * a.rows_cap + q];
    const double tmax = (double)a.t_norm_max[tf];
    double E = a.key_error * (qn + tmax);
    if (a.q_res) {
        // fp16 pass: |q.t - q'.t'| <= |q - q'| |t| + |q| |t - t'| + |q - q'| |t - t'| for
        // the rounded rows q', t' (Cauchy-Schwarz), twice that in the key
        const double rq = (double)a.q_res[(size_t)qf * a.rows_cap + q], rt = (double)a.t_res_max[tf];
        // (single-precision roots, raised past their rounding: E only has to be large enough)
        const double root_t = (double)sqrtf((float)tmax) * 1.000001, root_q = (double)sqrtf((float)qn) * 1.000001;
        E += 2.002 * (rq * root_t + root_q * rt + rq * rt);
    }
    // no row whose ranking key is at least the key of orderable form ord reaches the second
    // place, held by second.  NORM_L2 ranks by the rounded square root, which maps neighbouring
    // sums to one value: a row ties or wins only if its sum is below the square of the next float
    auto out_of_reach = [&](unsigned ord, unsigned long long second) {
        const unsigned u = ord ^ ((ord >> 31) ? 0x80000000u : 0xFFFFFFFFu);
        const double T = (double)__uint_as_float(u);
        const float d1 = __uint_as_float((unsigned)(second >> 32));
        const double d1n = (double)nextafterf(d1, INFINITY);
        return T + qn - E > (a.squared ? (double)d1 : d1n * d1n);
    };
    // One round: the kRound rows of row_of re-scored, the (distance, train index) key of row c in lane c.
    float* terms = s_terms[threadIdx.x >> 6];
    const rowslice qv = *reinterpret_cast<const rowslice*>(qrow + KPL * lane);
    auto score = [&]() {
        rowslice tv[kRound];
#pragma unroll
        for (int c = 0; c < kRound; ++c)   // an absent row reads row 0 and is ignored
            tv[c] = *reinterpret_cast<const rowslice*>(
                trows + (size_t)(row_of[c] < 0 ? 0 : row_of[c]) * K + KPL * lane);
        __builtin_amdgcn_wave_barrier();   // (the sums of the round before have read their terms)
#pragma unroll
        for (int c = 0; c < kRound; ++c) {
            const rowslice v = qv - tv[c];
            *reinterpret_cast<rowslice*>(terms + c * kPitch + KPL * lane) = v * v;
        }
        __builtin_amdgcn_wave_barrier();   // LDS serves a wave in order: the sums below see the terms
        int mine = -1;
#pragma unroll
        for (int c = 0; c < kRound; ++c) mine = lane == c ? row_of[c] : mine;
        unsigned long long key = ~0ull;   // (distance, train index)
        if (mine >= 0) {
            // d += (a-b)*(a-b) in element order, the scalar order of OpenCV's normL2Sqr (rows are
            // zero-padded to K floats and (0 - 0)^2 adds an exact zero)
            const floatx4* t4 = reinterpret_cast<const floatx4*>(terms + lane * kPitch);
            float d = 0.f;
#pragma unroll 8
            for (int j = 0; j < K / 4; ++j) {
                const floatx4 v = t4[j];
                d += v.x;
                d += v.y;
                d += v.z;
                d += v.w;
            }
            if (!a.squared) d = sqrtf(d);  // batchDistL2_32f
            key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)mine;
        }
        return key;
    };
    // The first round takes the four smallest keys; the second only runs when the fifth could
    // still reach the second place (its rows cost as much to fetch as everything else here).
    static_assert(kRound == 4, "the keys of a round sit in the first quad");
    unsigned long long key = score();
    unsigned long long k0 = quad0_min_u64(key);
    unsigned long long k1 = quad0_min_u64(key == k0 ? ~0ull : key);
    // T: no train row outside the re-scored set has a smaller ranking key (orderable form)
    unsigned others = lists < fifth ? lists : fifth;
    if (fifth != 0xFFFFFFFFu && (k1 == ~0ull || !out_of_reach(others, k1))) {   // wave-uniform
        const unsigned ninth = rank_round();
        key = score();
        key = lane == kRound ? k0 : lane == kRound + 1 ? k1 : key;   // (lanes past the round's rows)
        k0 = wave_min_u64(key);
        k1 = wave_min_u64(key == k0 ? ~0ull : key);
        others = lists < ninth ? lists : ninth;
    }
    bool proven = true;
    const bool listed = others != 0xFFFFFFFFu || __builtin_amdgcn_ballot_w64(same_stretch >= 0) != 0;
    if (listed) {   // there are train rows outside the re-scored set
        if (k1 == ~0ull) {
            proven = false;
        } else {
            // A stretch that gave a list both its entries: its other rows rank behind the second
            // of them, which settles them unless that one is (nearly) the second nearest of
            // all.  Then the stretch is left to l2_exhaustive_kernel, which scores its rows one
            // by one and joins them to the result written below -- every other row is out of
            // reach of the second place as it stands, and the second-best distance only falls.
            // Two such stretches (rare) send the query to the exhaustive pass proper.
            const unsigned long long pending =
                __builtin_amdgcn_ballot_w64(same_stretch >= 0 && !out_of_reach(second_key, k1));
            proven = others == 0xFFFFFFFFu || out_of_reach(others, k1);
            if (proven && pending) {
                const int holder = __builtin_ctzll(pending);
                const int first = __builtin_amdgcn_readlane(same_stretch, holder);
                int slot = 0;
                if (!(pending & (pending - 1)) && !a.force_exhaustive) {
                    if (lane == 0) slot = atomicAdd(a.stretch_count, 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                }
                if ((pending & (pending - 1)) || slot >= a.stretch_cap) {
                    proven = false;   // (a count above the capacity is read as the capacity)
                } else if (lane == 0 && !a.force_exhaustive) {
                    typedef int intx4 __attribute__((ext_vector_type(4)));
                    intx4* item = reinterpret_cast<intx4*>(a.stretch_items + 8 * (size_t)slot);
                    item[0] = intx4{problem, q, first, min(nt_rows, ((holder >> 1) + 1) * split_rows)};
                    item[1] = intx4{qf, tf, holder & 1, 0};
                    if (a.count_stretches) atomicAdd(a.stat_exhaustive, 1);
                }
            }
        }
    }
    if (a.force_exhaustive) proven = false;
    if (lane == 0 && !proven) {
        const int slot = atomicAdd(a.todo_count, 1);
        atomicAdd(a.stat_exhaustive, 1);
        a.todo[2 * (size_t)slot] = problem;
        a.todo[2 * (size_t)slot + 1] = q;
    }
    if (lane == 0) {
        const size_t o = ((size_t)problem * a.rows_cap + q) * 2;
        a.knn_idx[o] = k0 == ~0ull ? -1 : (int32_t)(k0 & 0xFFFFFFFFu);
        a.knn_idx[o + 1] = k1 == ~0ull ? -1 : (int32_t)(k1 & 0xFFFFFFFFu);
        a.knn_dist[o] = k0 == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k0 >> 32));
        a.knn_dist[o + 1] = k1 == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k1 >> 32));
    }
}

// The exhaustive pass takes the listed queries in groups of kExQueries whose rows sit in LDS:
// a train row is read once per group, not once per query.
constexpr int kExQueries = 8;    // queries per group
constexpr int kExRows = 4;       // train rows a thread works on at a time

// Slices of the train rows per group: a short list is spread over the whole launch (one
// workgroup alone takes 0.4 ms for 20 000 rows of 128), a long one gets a workgroup per group.
// With few groups the slices shrink to a row per thread (rows_cap bounds every train set), which
// spares the loads and sums of the three other rows a thread would otherwise carry: four listed
// queries of 20 000 x 20 000 x 128 cost 117 us with 64 slices of 313 rows.
__device__ __forceinline__ int exhaustive_slices(int groups, int grid, int rows_cap) {
    return groups >= grid ? 1 : min(max(64, (rows_cap + kBlock - 1) / kBlock), grid / groups);
}

// The listed queries against every train row: one workgroup per (group of entries, slice of the
// train rows) (grid-stride).  A thread forms the sequential distances of kExRows train rows to
// the group's queries, element by element in the order of OpenCV's normL2Sqr, and keeps the
// best two per query by (distance, train index) -- it meets its rows in ascending order, so a
// strict comparison leaves the lower index in front; the workgroup's best two go to
// partial[entry][slice][2], and the workgroup that finishes last joins the slices into
// knn_idx / knn_dist.  With an empty list (the usual case) a workgroup reads the count and
// leaves.  K = 64 KPL.
template <int KPL>
__global__ __launch_bounds__(kBlock) void l2_exhaustive_kernel(RescoreArgs a) {
    constexpr int K = 64 * KPL;
    __shared__ __attribute__((aligned(16))) float s_q[kExQueries][K];
    __shared__ unsigned long long s_red[kBlock / 64][kExQueries][2];
    typedef int intx4 __attribute__((ext_vector_type(4)));
    // (a workgroup's first entry is read beside the counts, not behind them: whatever the
    // array holds there, it is only used below the count)
    const intx4* first_entry = reinterpret_cast<const intx4*>(a.stretch_items + 8 * (size_t)(blockIdx.x % a.stretch_cap));
    intx4 e0 = first_entry[0], e1 = first_entry[1];
    const int count = *a.todo_count;
    const int stretches = min(*a.stretch_count, a.stretch_cap);
    if (count == 0 && stretches == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Stretches of single lists (see RescoreArgs): a workgroup per entry, a thread per row that
    // the list's lanes saw of the stretch, distances in
    // the element order of the re-score pass, joined to the two that pass wrote.
    for (int item = blockIdx.x; item < stretches; item += gridDim.x) {
        if (item != (int)blockIdx.x) {
            e0 = reinterpret_cast<const intx4*>(a.stretch_items + 8 * (size_t)item)[0];
            e1 = reinterpret_cast<const intx4*>(a.stretch_items + 8 * (size_t)item)[1];
        }
        const int problem = e0.x, q = e0.y, first = e0.z, end = e0.w, qf = e1.x, tf = e1.y, half = e1.z;
        const float* trows = a.t_base + (size_t)tf * a.frame_floats;
        const float* qrow = a.q_base + (size_t)qf * a.frame_floats + (size_t)q * K;
        // the result so far (thread 0 joins the stretch to it at the end)
        const size_t o = ((size_t)problem * a.rows_cap + q) * 2;
        int was_idx[2] = {-1, -1};
        float was_dist[2] = {0.f, 0.f};
        if (tid == 0) {
            was_idx[0] = a.knn_idx[o];
            was_idx[1] = a.knn_idx[o + 1];
            was_dist[0] = a.knn_dist[o];
            was_dist[1] = a.knn_dist[o + 1];
        }
        __syncthreads();   // s_q, s_red of the entry before are done with
        for (int i = tid; i < K / 4; i += kBlock)
            *reinterpret_cast<floatx4*>(&s_q[0][4 * i]) = *reinterpret_cast<const floatx4*>(qrow + 4 * i);
        __syncthreads();
        unsigned long long key = ~0ull;
        // the rows the list's lanes saw: every other group of four (half the memory instructions
        // of the whole stretch, and it is their number that this costs)
        for (int r0 = 0; r0 < a.stretch_rows / 2; r0 += kBlock) {   // (one round: half a stretch is below kBlock rows)
            const int i = r0 + tid;
            const int row = first + 8 * (i >> 2) + 4 * half + (i & 3);
            if (i < a.stretch_rows / 2 && row < end) {
                const floatx4* t4 = reinterpret_cast<const floatx4*>(trows + (size_t)row * K);
                float d = 0.f;
                // (every load of up to 32 chunks goes out before the first sum: a thread's row
                // is its own, and what this costs is the latency of its loads)
                constexpr int kBatch = K / 4 < 32 ? K / 4 : 32;
#pragma unroll 1
                for (int j0 = 0; j0 < K / 4; j0 += kBatch) {
                    floatx4 tv[kBatch];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) tv[j] = t4[j0 + j];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) {
                        const floatx4 qq = *reinterpret_cast<const floatx4*>(&s_q[0][4 * (j0 + j)]);
                        const floatx4 u = qq - tv[j];
                        const floatx4 v = u * u;
                        d += v.x;
                        d += v.y;
                        d += v.z;
                        d += v.w;
                    }
                }
                if (!a.squared) d = sqrtf(d);
                const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)row;
                key = k < key ? k : key;   // (one round, see above; more would need the best two here)
            }
        }
        const unsigned long long w0 = wave_min_u64(key);
        const unsigned long long w1 = wave_min_u64(key == w0 ? ~0ull : key);
        if (lane == 0) {
            s_red[wave][0][0] = w0;
            s_red[wave][0][1] = w1;
        }
        __syncthreads();
        if (tid == 0) {
            // best two of the re-score pass's result and the stretch; a row met twice carries
            // the same key twice
            unsigned long long k0 = ~0ull, k1 = ~0ull;
            auto take = [&](unsigned long long k) {
                if (k == k0 || k == k1) return;
                if (k < k0) {
                    k1 = k0;
                    k0 = k;
                } else if (k < k1) {
                    k1 = k;
                }
            };
            for (int r = 0; r < 2; ++r)
                if (was_idx[r] >= 0)
                    take(((unsigned long long)__float_as_uint(was_dist[r]) << 32) | (unsigned)was_idx[r]);
            for (int w = 0; w < kBlock / 64; ++w) {
                take(s_red[w][0][0]);
                take(s_red[w][0][1]);
            }
            a.knn_idx[o] = k0 == ~0ull ? -1 : (int32_t)(k0 & 0xFFFFFFFFu);
            a.knn_idx[o + 1] = k1 == ~0ull ? -1 : (int32_t)(k1 & 0xFFFFFFFFu);
            a.knn_dist[o] = k0 == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k0 >> 32));
            a.knn_dist[o + 1] = k1 == ~0ull ? FLT_MAX : __uint_as_float((unsigned)(k1 >> 32));
        }
    }
    if (count == 0) return;
    const int groups = (count + kExQueries - 1) / kExQueries;
    const int slices = exhaustive_slices(groups, gridDim.x, a.rows_cap);
    for (int item = blockIdx.x; item < groups * slices; item += gridDim.x) {
        const int e0 = (item / slices) * kExQueries, slice = item % slices;
        const int ne = min(kExQueries, count - e0);
        // entries of one problem share the train rows: take them run by run
        for (int r0 = 0; r0 < ne;) {
            const int problem = a.todo[2 * (size_t)(e0 + r0)];
            int r1 = r0 + 1;
            while (r1 < ne && a.todo[2 * (size_t)(e0 + r1)] == problem) ++r1;
            const int nrun = r1 - r0;
            int qf = 0, tf = 0;
            if (a.pairs) {
                const int gp = problem >> 1, dir = problem & 1;
                const int fa = a.pairs[2 * gp], fb = a.pairs[2 * gp + 1];
                qf = dir ? fb : fa;
                tf = dir ? fa : fb;
            }
            const int nt = a.counts ? a.counts[tf] : a.nt_uniform;
            const int per_slice = (nt + slices - 1) / slices;
            const int t_begin = slice * per_slice, t_end = min(nt, t_begin + per_slice);
            const float* trows = a.t_base + (size_t)tf * a.frame_floats;
            __syncthreads();   // the rows of the run before are done with
            for (int i = tid; i < kExQueries * (K / 4); i += kBlock) {
                const int qq = i / (K / 4), j4 = i % (K / 4);
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (qq < nrun) {
                    const int q = a.todo[2 * (size_t)(e0 + r0 + qq) + 1];
                    v = *reinterpret_cast<const floatx4*>(a.q_base + (size_t)qf * a.frame_floats + (size_t)q * K +
                                                          4 * j4);
                }
                *reinterpret_cast<floatx4*>(&s_q[qq][4 * j4]) = v;
            }
            __syncthreads();
            float bd0[kExQueries], bd1[kExQueries];
            int bi0[kExQueries], bi1[kExQueries];
#pragma unroll
            for (int qq = 0; qq < kExQueries; ++qq) {
                bd0[qq] = bd1[qq] = INFINITY;
                bi0[qq] = bi1[qq] = -1;
            }
            auto scan = [&](auto rows_per_thread) {
            constexpr int R = decltype(rows_per_thread)::value;
            for (int t0 = t_begin + tid; t0 < t_end; t0 += R * kBlock) {
                float d[R][kExQueries];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int qq = 0; qq < kExQueries; ++qq) d[r][qq] = 0.f;
                const floatx4* trow[R];
#pragma unroll
                for (int r = 0; r < R; ++r)   // a row past the slice re-reads the last one
                    trow[r] = reinterpret_cast<const floatx4*>(trows + (size_t)min(t0 + r * kBlock, t_end - 1) * K);
                constexpr int kAhead = R == 1 ? 8 : 2;   // chunks of a row whose loads go out together
                static_assert((K / 4) % kAhead == 0, "whole batches");
#pragma unroll 1
                for (int j0 = 0; j0 < K / 4; j0 += kAhead) {
                    floatx4 tv[kAhead][R];
#pragma unroll
                    for (int u = 0; u < kAhead; ++u)
#pragma unroll
                        for (int r = 0; r < R; ++r) tv[u][r] = trow[r][j0 + u];
#pragma unroll
                    for (int u = 0; u < kAhead; ++u)
#pragma unroll
                        for (int qq = 0; qq < kExQueries; ++qq) {
                            const floatx4 qv = *reinterpret_cast<const floatx4*>(&s_q[qq][4 * (j0 + u)]);
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                const float v0 = qv.x - tv[u][r].x, v1 = qv.y - tv[u][r].y, v2 = qv.z - tv[u][r].z,
                                            v3 = qv.w - tv[u][r].w;
                                d[r][qq] += v0 * v0;
                                d[r][qq] += v1 * v1;
                                d[r][qq] += v2 * v2;
                                d[r][qq] += v3 * v3;
                            }
                        }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int t = t0 + r * kBlock;
                    if (t < t_end) {
#pragma unroll
                        for (int qq = 0; qq < kExQueries; ++qq) {
                            const float dd = a.squared ? d[r][qq] : sqrtf(d[r][qq]);
                            if (dd < bd0[qq]) {
                                bd1[qq] = bd0[qq];
                                bi1[qq] = bi0[qq];
                                bd0[qq] = dd;
                                bi0[qq] = t;
                            } else if (dd < bd1[qq]) {
                                bd1[qq] = dd;
                                bi1[qq] = t;
                            }
                        }
                    }
                }
            }
            };
            if (t_end - t_begin <= kBlock)   // (wave-uniform)
                scan(std::integral_constant<int, 1>());
            else
                scan(std::integral_constant<int, kExRows>());
            // best two of the workgroup per query: the smallest key, then the smallest of what is
            // left (keys are distinct: they hold the index)
#pragma unroll
            for (int qq = 0; qq < kExQueries; ++qq) {
                const unsigned long long k0 =
                    bi0[qq] < 0 ? ~0ull : ((unsigned long long)__float_as_uint(bd0[qq]) << 32) | (unsigned)bi0[qq];
                const unsigned long long k1 =
                    bi1[qq] < 0 ? ~0ull : ((unsigned long long)__float_as_uint(bd1[qq]) << 32) | (unsigned)bi1[qq];
                const unsigned long long w0 = wave_min_u64(k0);
                const unsigned long long w1 = wave_min_u64(k0 == w0 ? k1 : k0);
                if (lane == 0) {
                    s_red[wave][qq][0] = w0;
                    s_red[wave][qq][1] = w1;
                }
            }
            __syncthreads();
            if (tid < nrun) {
                unsigned long long k0 = ~0ull, k1 = ~0ull;
                for (int w = 0; w < kBlock / 64; ++w)
                    for (int r = 0; r < 2; ++r) {
                        const unsigned long long key = s_red[w][tid][r];
                        if (key < k0) {
                            k1 = k0;
                            k0 = key;
                        } else if (key < k1) {
                            k1 = key;
                        }
                    }
                const size_t o = 2 * ((size_t)(e0 + r0 + tid) * slices + slice);
                a.partial[o] = k0;
                a.partial[o + 1] = k1;
            }
            r0 = r1;
        }
    }
    // The slices are joined by whichever workgroup finishes last: every workgroup publishes its
    // stores and counts itself; the one that counts to the size of the launch sees them all.
    // (Only the workgroups that had an item count, and one thread of each publishes: a release
    // by every thread of a launch of 1024 workgroups cost 100 us for four listed queries.)
    const int working = min((int)gridDim.x, groups * slices);
    if ((int)blockIdx.x >= working) return;
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = atomicAdd(a.exhaustive_done, 1) == working - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid == 0) *a.exhaustive_done = 0;   // for the next launch
    auto write_entry = [&](int e, unsigned long long k0, unsigned long long k1) {
        const int problem = a.todo[2 * (size_t)e], q = a.todo[2 * (size_t)e + 1];
        const size_t o = ((size_t)problem * a.rows_cap + q) * 2;
        const float d0 = __uint_as_float((unsigned)(k0 >> 32)), d1 = __uint_as_float((unsigned)(k1 >> 32));
        a.knn_idx[o] = k0 == ~0ull ? -1 : (int32_t)(k0 & 0xFFFFFFFFu);
        a.knn_idx[o + 1] = k1 == ~0ull ? -1 : (int32_t)(k1 & 0xFFFFFFFFu);
        a.knn_dist[o] = k0 == ~0ull ? FLT_MAX : d0;
        a.knn_dist[o + 1] = k1 == ~0ull ? FLT_MAX : d1;
    };
    if (slices > 4) {
        // a short list in many slices: a wave per entry, a lane per few slices (one thread per
        // entry would read its 2 x slices values one behind the other)
        for (int e = wave; e < count; e += kBlock / 64) {
            unsigned long long k0 = ~0ull, k1 = ~0ull;
            for (int i = lane; i < 2 * slices; i += 64) {
                const unsigned long long key = a.partial[2 * (size_t)e * slices + i];
                if (key < k0) {
                    k1 = k0;
                    k0 = key;
                } else if (key < k1) {
                    k1 = key;
                }
            }
            const unsigned long long w0 = wave_min_u64(k0);
            const unsigned long long w1 = wave_min_u64(k0 == w0 ? k1 : k0);
            if (lane == 0) write_entry(e, w0, w1);
        }
        return;
    }
    for (int e = tid; e < count; e += kBlock) {
        unsigned long long k0 = ~0ull, k1 = ~0ull;
        for (int i = 0; i < 2 * slices; ++i) {
            const unsigned long long key = a.partial[2 * (size_t)e * slices + i];
            if (key < k0) {
                k1 = k0;
                k0 = key;
            } else if (key < k1) {
                k1 = key;
            }
        }
        write_entry(e, k0, k1);
    }
}

// Which matrix cores rank the candidates.  SFM_L2_MFMA=f32 / bf16 select the other passes (a
// measuring and testing aid, read per call); the default is the fp16 pass.
enum L2Mode { kL2F32, kL2Bf16, kL2F16 };
L2Mode l2_mode() {
    const char* e = getenv("SFM_L2_MFMA");
    if (!e) return kL2F16;
    if (e[0] == 'f' && e[1] == '3') return kL2F32;
    if (e[0] == 'b') return kL2Bf16;
    return kL2F16;
}

struct L2Plan {
    int K, NT, q_per_block, qtiles, splits;
};

int make_l2_plan(const sfm_ctx* ctx, int dim, int nq_max, int nt_max, int n_problems, L2Plan* p) {
    if (dim <= 64) {
        p->K = 64;
        p->NT = 2;
    } else if (dim <= 128) {
        p->K = 128;
        p->NT = 2;
    } else if (dim <= 256) {
        p->K = 256;
        p->NT = 1;
    } else {
        return SFM_E_UNSUPPORTED;
    }
    p->q_per_block = 4 * 32 * p->NT;
    p->qtiles = (nq_max + p->q_per_block - 1) / p->q_per_block;
    // The launch runs in rounds of resident workgroups (one per CU, two of the fp16 pass): pick
    // the split count (<= kMaxSplits, >= 128 train rows each) that wastes the least in the last
    // round of num_cus and in the partial last tile; ties go to fewer splits.  (Counting two
    // workgroups per CU for the fp16 pass picks 12 splits for 20 000 rows instead of 16: no
    // faster, and with fewer candidate lists the odd query fails the proof.)
    const long base = (long)p->qtiles * n_problems;
    int splits = 1;
    double best_eff = -1.0;
    for (int sp = 1; sp <= kMaxSplits; ++sp) {
        const int chunk = (nt_max + sp - 1) / sp;
        if (sp > 1 && chunk < 2 * kTileRows) break;
        const long blocks = base * sp;
        const long rounds = (blocks + ctx->num_cus - 1) / ctx->num_cus;
        const int tiles = (chunk + kTileRows - 1) / kTileRows;
        const double eff = (double)blocks / (double)(rounds * ctx->num_cus) * (double)chunk /
                           (double)(tiles * kTileRows);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            splits = sp;
        }
    }
    p->splits = splits;
    return SFM_OK;
}

template <int K, int NT>
int launch_candidates(sfm_ctx* ctx, const L2Args& a, long grid, L2Mode mode) {
    KernelTimer timer(ctx, KID_L2_CANDIDATES);
    // narrow passes: the same queries per workgroup, one 32-query column block per wave
    if (mode == kL2F16)   // 64 queries per wave: half the fragment reads, two workgroups per CU
        hipLaunchKernelGGL((l2_candidates_narrow_kernel<K, NT, 4, true>), dim3((unsigned)grid),
                           dim3(64 * 4), 0, ctx->stream, a);
    else if (mode == kL2Bf16)
        hipLaunchKernelGGL((l2_candidates_narrow_kernel<K, 1, 4 * NT, false>), dim3((unsigned)grid),
                           dim3(64 * 4 * NT), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((l2_candidates_kernel<K, NT>), dim3((unsigned)grid), dim3(kBlock), 0,
                           ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

// Scratch of one candidate / re-score / exhaustive sequence.
struct L2Scratch {
    int32_t* cand;       // [problems][splits][rows_cap][4]
    float* cand_key;     // same layout
    float* cand_third;   // [problems][splits][rows_cap][2]
    float* scalars;      // see below
    int groups;          // groups of rows (frames; queries and train rows) with maxima of their own
    int32_t* todo;       // [problems][rows_cap][2]
    int32_t* stretch_items;   // [kStretchCap][8]
    unsigned long long* partial;  // [max(problems x rows_cap, kExQueries x kExhaustiveGrid)][2]
    int32_t* knn_idx;    // [problems][rows_cap][2]
    float* knn_dist;
};
constexpr int kExhaustiveGrid = 1024;
// scalars: [count of listed queries][finished workgroups of the exhaustive pass][largest norm
// of all rows][largest norm per group ...][largest residual per group ...][s^2][1 / s^2].  The
// two counters are cleared by l2_pack_kernel.
enum { kScalarTodoCount = 0, kScalarStretchCount, kScalarExhaustiveDone, kScalarGlobalMax, kScalarGroups };
inline float* scratch_group_max(const L2Scratch& s) { return s.scalars + kScalarGroups; }
inline float* scratch_group_res(const L2Scratch& s) { return s.scalars + kScalarGroups + s.groups; }
inline float* scratch_scale(const L2Scratch& s) { return s.scalars + kScalarGroups + 2 * s.groups; }
void l2_scratch_carve(Carver& c, const L2Plan& plan, int rows_cap, size_t n_problems, int groups, L2Scratch* s) {
    s->cand = c.take<int32_t>(n_problems * plan.splits * rows_cap * 4);
    s->cand_key = c.take<float>(n_problems * plan.splits * rows_cap * 4);
    s->cand_third = c.take<float>(n_problems * plan.splits * rows_cap * 2);
    s->groups = groups;
    s->scalars = c.take<float>(kScalarGroups + 2 * (size_t)groups + 2);
    s->todo = c.take<int32_t>(n_problems * rows_cap * 2);
    s->stretch_items = c.take<int32_t>((size_t)kStretchCap * 8);
    s->partial = c.take<unsigned long long>(
        std::max<size_t>(n_problems * rows_cap, (size_t)kExQueries * kExhaustiveGrid) * 2);
    s->knn_idx = c.take<int32_t>(n_problems * rows_cap * 2);
    s->knn_dist = c.take<float>(n_problems * rows_cap * 2);
}
static_assert(sizeof(float) == sizeof(int32_t), "the query count lives among the float scalars");

// SFM_L2_EXHAUSTIVE=1 sends every query through the exhaustive pass (a testing aid, read per call).
bool l2_force_exhaustive() {
    const char* e = getenv("SFM_L2_EXHAUSTIVE");
    return e && e[0] == '1';
}

// The rows of one matching job in the layouts of the passes (rows_cap rows per frame).
struct L2Rows {
    float* packed;   // [rows][K] zero-padded floats
    float* narrow;   // [rows][K] bfloat16 pairs (K floats) or [rows][K] fp16 (K / 2 floats)
    float* norms;    // ||row||^2
    float* acc0;     // start value of the narrow passes' accumulators
    float* res;      // fp16 pass: residual norm
    float* norm_slots;   // per packing workgroup: two largest norms, two largest residual norms
    float* res_slots;    // (see FoldSet)
};
void l2_rows_carve(Carver& c, size_t rows, int K, L2Rows* r) {
    const size_t pack_blocks = rows / kPackRowsPerBlock + 1;
    r->norm_slots = c.take<float>(2 * pack_blocks);
    r->res_slots = c.take<float>(2 * pack_blocks);
    r->packed = c.take<float>(rows * K);
    r->narrow = c.take<float>(rows * K);
    r->norms = c.take<float>(rows);
    r->acc0 = c.take<float>(rows);
    r->res = c.take<float>(rows);
}

// Packing of up to two sets of rows for `mode` (a set of zero rows is skipped): one launch, and
// a second one for the fp16 rows, whose scale depends on the largest norm of all.  The groups'
// largest norms go to the scalars (groups of set 0, then of set 1; the train rows are set 1
// unless set0_is_train), the counters of the later passes are cleared.  fp16 pass: fold[0] and
// fold[1] say how the groups' largest norms and residual norms are formed from what the two
// launches left; the candidate pass does that on the side (l2_knn_packed).
struct L2Source {
    const void* src;
    int rows, rows_per_group;
    size_t group_stride, row_stride;
    const int32_t* counts;   // rows in use per group (device), or null
};
int l2_pack(sfm_ctx* ctx, L2Mode mode, const L2Source& src0, L2Rows* rows0, const L2Source& src1,
            L2Rows* rows1, bool set0_is_train, int dim, int K, const L2Scratch& sc, FoldArgs (&fold)[2]) {
    fold[0] = fold[1] = FoldArgs{};
    const L2Source* src[2] = {&src0, &src1};
    L2Rows* rows[2] = {rows0, rows1};
    // Rows that already are what the packed copy would be -- K floats wide, back to back,
    // 16-byte aligned -- are read in place by every pass: no copy is written.
    for (int i = 0; i < 2; ++i) {
        const L2Source& sr = *src[i];
        const bool one_group = sr.rows <= sr.rows_per_group;
        if (sr.rows > 0 && dim == K && sr.row_stride == (size_t)K * 4 &&
            (one_group || sr.group_stride == (size_t)sr.rows_per_group * K * 4) &&
            (reinterpret_cast<uintptr_t>(sr.src) & 15) == 0)
            rows[i]->packed = nullptr;
    }
    int set_blocks[2], set_groups[2], rpg[2];
    for (int i = 0; i < 2; ++i) {
        rpg[i] = src[i]->rows_per_group > 0 ? src[i]->rows_per_group : 1;
        set_blocks[i] = (src[i]->rows + kPackRowsPerBlock - 1) / kPackRowsPerBlock;
        set_groups[i] = src[i]->rows > 0 ? (src[i]->rows + rpg[i] - 1) / rpg[i] : 0;
    }
    const int blocks = set_blocks[0] + set_blocks[1];
    int32_t* counters = reinterpret_cast<int32_t*>(sc.scalars);
    if (blocks <= 0) {   // nothing to pack: only the counters are reset
        SFM_HIP_CHECK(ctx, hipMemsetAsync(ctx->stats, 0, sizeof(int32_t), ctx->stream));
        SFM_HIP_CHECK(ctx, hipMemsetAsync(counters, 0, kScalarGlobalMax * sizeof(int32_t), ctx->stream));
        return SFM_OK;
    }
    if (set_groups[0] + set_groups[1] > sc.groups) return set_error(ctx, SFM_E_INVALID, "more row groups than planned");
    KernelTimer timer(ctx, KID_L2_PACK);
    PackArgs a{};
    NarrowArgs n{};
    FoldArgs norm_fold{};
    for (int i = 0; i < 2; ++i) {
        const bool train = i == 1 || set0_is_train;
        const int group0 = i ? set_groups[0] : 0;
        a.set[i] = PackSet{static_cast<const uint8_t*>(src[i]->src), src[i]->rows, rpg[i], src[i]->group_stride,
                           src[i]->row_stride, rows[i]->packed,   // (null: no copy)
                           mode == kL2Bf16 ? reinterpret_cast<__bf16*>(rows[i]->narrow) : nullptr,
                           rows[i]->norms, mode == kL2Bf16 ? rows[i]->acc0 : nullptr, src[i]->counts,
                           rows[i]->norm_slots};
        if (!rows[i]->packed) rows[i]->packed = const_cast<float*>(static_cast<const float*>(src[i]->src));
        n.set[i] = NarrowSet{rows[i]->packed, rows[i]->norms, src[i]->rows,
                             reinterpret_cast<_Float16*>(rows[i]->narrow), rows[i]->res,
                             train ? rows[i]->acc0 : nullptr, rpg[i], src[i]->counts, rows[i]->res_slots};
        norm_fold.set[i] = FoldSet{rows[i]->norm_slots, rows[i]->norms, src[i]->counts,
                                   scratch_group_max(sc) + group0, src[i]->rows, rpg[i], set_groups[i]};
        if (mode == kL2F16)
            fold[1].set[i] = FoldSet{rows[i]->res_slots, rows[i]->res, src[i]->counts,
                                     scratch_group_res(sc) + group0, src[i]->rows, rpg[i], set_groups[i]};
    }
    a.blocks0 = n.blocks0 = set_blocks[0];
    a.clear[0] = counters + kScalarTodoCount;
    a.clear[1] = counters + kScalarExhaustiveDone;
    a.clear[2] = ctx->stats;
    a.clear[3] = counters + kScalarStretchCount;
    a.dim = dim;
    a.K = n.K = K;
    bool aligned = dim % 2 == 0;   // rows of an even number of floats on 8-byte boundaries
    for (int i = 0; i < 2; ++i)
        if (src[i]->rows > 0 && ((reinterpret_cast<uintptr_t>(src[i]->src) | src[i]->row_stride | src[i]->group_stride) & 7))
            aligned = false;
    const bool split = mode == kL2Bf16;
    if (aligned && !split)
        hipLaunchKernelGGL((l2_pack_kernel<true, false>), dim3(blocks), dim3(kPackBlock), 0, ctx->stream, a);
    else if (aligned)
        hipLaunchKernelGGL((l2_pack_kernel<true, true>), dim3(blocks), dim3(kPackBlock), 0, ctx->stream, a);
    else if (!split)
        hipLaunchKernelGGL((l2_pack_kernel<false, false>), dim3(blocks), dim3(kPackBlock), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((l2_pack_kernel<false, true>), dim3(blocks), dim3(kPackBlock), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    if (mode != kL2F16) {
        hipLaunchKernelGGL(l2_group_fold_kernel, dim3(set_groups[0] + set_groups[1]), dim3(kBlock), 0, ctx->stream,
                           norm_fold);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        return SFM_OK;
    }
    fold[0] = norm_fold;
    n.scale_out = scratch_scale(sc);
    n.global_max = sc.scalars + kScalarGlobalMax;
    if (2L * blocks <= kDirectSlots) {
        for (int i = 0; i < 2; ++i) {
            // (the second half of a set's slots holds zeros unless a workgroup straddles two groups)
            const bool straddle = set_groups[i] > 1 && rpg[i] % kPackRowsPerBlock != 0;
            n.norm_slots[i] = rows[i]->norm_slots;
            n.n_norm_slots[i] = (straddle ? 2 : 1) * set_blocks[i];
        }
    } else {
        SlotsMaxArgs m{};
        for (int i = 0; i < 2; ++i) {
            m.slots[i] = rows[i]->norm_slots;
            m.n_slots[i] = 2L * set_blocks[i];
        }
        m.out = sc.scalars + kScalarGlobalMax;
        hipLaunchKernelGGL(l2_slots_max_kernel, dim3(1), dim3(1024), 0, ctx->stream, m);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(l2_narrow_kernel, dim3(blocks), dim3(kPackBlock), 0, ctx->stream, n);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

// Candidate, re-score and exhaustive passes over packed rows; writes s.knn_idx / s.knn_dist
// [problem][rows_cap][2].  frame_rows: rows between two frames of q / t (0 for one problem).
// The query count among the scalars is cleared here unless the caller has done so.
int l2_knn_packed(sfm_ctx* ctx, const L2Plan& plan, L2Mode mode, int dim, int squared, const L2Rows& q,
                  const L2Rows& t, size_t frame_rows, int rows_cap, const int32_t* d_counts, int nq_uniform,
                  int nt_uniform, const int32_t* d_pairs, int n_problems, const L2Scratch& s,
                  int t_group0, bool clear_todo, const FoldArgs* fold) {
    int32_t* todo_count = reinterpret_cast<int32_t*>(s.scalars + kScalarTodoCount);
    const int narrow_floats = mode == kL2F16 ? plan.K / 2 : plan.K;   // floats of a narrow row
    L2Args a{};
    a.q_base = mode == kL2F32 ? q.packed : q.narrow;
    a.t_base = mode == kL2F32 ? t.packed : t.narrow;
    a.t_norms = t.norms;
    a.t_acc0 = t.acc0;
    a.narrow_scale = scratch_scale(s);
    a.frame_floats = frame_rows * (mode == kL2F32 ? plan.K : narrow_floats);
    a.rows_cap = rows_cap;
    a.counts = d_counts;
    a.nq_uniform = nq_uniform;
    a.nt_uniform = nt_uniform;
    a.pairs = d_pairs;
    a.n_groups = d_pairs ? n_problems / 2 : 1;
    a.qtiles = plan.qtiles;
    a.splits = plan.splits;
    a.blocks_per_group = (d_pairs ? 2 : 1) * plan.qtiles * plan.splits;
    a.cand = s.cand;
    a.cand_key = s.cand_key;
    a.cand_third = s.cand_third;
    long grid = (long)a.n_groups * a.blocks_per_group;
    if (grid <= 0) return SFM_OK;
    if (fold && mode == kL2F16) {   // the groups' largest norms and residual norms, on the side (see L2Args)
        a.fold[0] = fold[0];
        a.fold[1] = fold[1];
        a.fold_groups = fold[0].set[0].groups + fold[0].set[1].groups;
        grid += a.fold_groups;
    }
    if (grid > 0x7fffffffL) return set_error(ctx, SFM_E_INVALID, "knn launch too large");
    static_assert(kScalarStretchCount == kScalarTodoCount + 1, "the two counts are cleared together");
    if (clear_todo) SFM_HIP_CHECK(ctx, hipMemsetAsync(todo_count, 0, 2 * sizeof(int32_t), ctx->stream));
    if (plan.K == 64)
        SFM_RETURN_IF((launch_candidates<64, 2>(ctx, a, grid, mode)));
    else if (plan.K == 128)
        SFM_RETURN_IF((launch_candidates<128, 2>(ctx, a, grid, mode)));
    else
        SFM_RETURN_IF((launch_candidates<256, 1>(ctx, a, grid, mode)));

    RescoreArgs r{};
    r.q_base = q.packed;
    r.t_base = t.packed;
    r.frame_floats = frame_rows * plan.K;
    r.rows_cap = rows_cap;
    r.counts = d_counts;
    r.nq_uniform = nq_uniform;
    r.nt_uniform = nt_uniform;
    r.pairs = d_pairs;
    r.splits = plan.splits;
    r.K = plan.K;
    r.dim = dim;
    r.squared = squared;
    r.cand = s.cand;
    r.cand_key = s.cand_key;
    r.cand_third = s.cand_third;
    r.stretch_rows = mode == kL2F32 ? 0 : kMergeTiles * kTileRows;
    r.q_norms = q.norms;
    r.t_norm_max = scratch_group_max(s) + t_group0;   // indexed by the train frame
    r.q_res = mode == kL2F16 ? q.res : nullptr;
    r.t_res_max = scratch_group_res(s) + t_group0;
    // Per pass: roundings of the norms, of the accumulation steps (fp32: K, fp16: K exact
    // products, bfloat16: 3 K) and of the sequential sum, counted four per product.  bfloat16:
    // 4 * 2^-16 of the norms left out by the split.  Narrow passes: the position tag in a key's
    // kNarrowTagBits low bits (below 2^(kNarrowTagBits - 23) of a key that is at most twice the
    // norms: 2^-15 for seven bits, counted as 2^-14).  Every pass: the lane number in the six low
    // bits of a key when the re-score pass ranks the candidates (2^-16 of the norms).  The fp16
    // pass adds its measured rounding of the rows per query (l2_rescore_kernel).
    const double eps = (double)FLT_EPSILON, tag = 1.0 / 65536.0;
    const double narrow_tag = (double)(1 << (kNarrowTagBits - 5)) * tag;
    r.key_error = mode == kL2F32   ? (4.0 * plan.K + 64.0) * eps + tag
                  : mode == kL2F16 ? (4.0 * plan.K + 64.0) * eps + tag + narrow_tag
                                   : (12.0 * plan.K + 64.0) * eps + 5.0 * tag + narrow_tag;
    r.force_exhaustive = l2_force_exhaustive() ? 1 : 0;
    {   // SFM_L2_COUNT_STRETCHES=1: sfm_l2_exhaustive_count counts the stretch entries too
        const char* e = getenv("SFM_L2_COUNT_STRETCHES");
        r.count_stretches = e && e[0] == '1' ? 1 : 0;
    }
    r.knn_idx = s.knn_idx;
    r.knn_dist = s.knn_dist;
    r.todo_count = todo_count;
    r.todo = s.todo;
    r.stretch_count = reinterpret_cast<int32_t*>(s.scalars + kScalarStretchCount);
    r.stretch_items = s.stretch_items;
    r.stretch_cap = kStretchCap;
    r.exhaustive_done = reinterpret_cast<int32_t*>(s.scalars + kScalarExhaustiveDone);
    r.stat_exhaustive = ctx->stats;
    r.partial = s.partial;
    const int nq_max = d_counts ? rows_cap : nq_uniform;
    const dim3 rgrid((nq_max + kBlock / 64 - 1) / (kBlock / 64), n_problems);
    {
        KernelTimer timer(ctx, KID_L2_RESCORE);
        if (plan.K == 64)
            hipLaunchKernelGGL(l2_rescore_kernel<1>, rgrid, dim3(kBlock), 0, ctx->stream, r);
        else if (plan.K == 128)
            hipLaunchKernelGGL(l2_rescore_kernel<2>, rgrid, dim3(kBlock), 0, ctx->stream, r);
        else
            hipLaunchKernelGGL(l2_rescore_kernel<4>, rgrid, dim3(kBlock), 0, ctx->stream, r);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    // the listed queries (normally none: the workgroups read a zero count and leave)
    KernelTimer timer(ctx, KID_L2_EXHAUSTIVE);
    if (plan.K == 64)
        hipLaunchKernelGGL(l2_exhaustive_kernel<1>, dim3(kExhaustiveGrid), dim3(kBlock), 0, ctx->stream, r);
    else if (plan.K == 128)
        hipLaunchKernelGGL(l2_exhaustive_kernel<2>, dim3(kExhaustiveGrid), dim3(kBlock), 0, ctx->stream, r);
    else
        hipLaunchKernelGGL(l2_exhaustive_kernel<4>, dim3(kExhaustiveGrid), dim3(kBlock), 0, ctx->stream, r);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace

int l2_knn2_dev(sfm_ctx* ctx, int squared, const float* d_q, int nq, size_t q_stride,
                const float* d_t, int nt, size_t t_stride, int dim, int32_t* d_idx, float* d_dist) {
    if (q_stride < (size_t)dim * 4 || (nt > 0 && t_stride < (size_t)dim * 4))
        return set_error(ctx, SFM_E_INVALID, "row stride smaller than the row");
    if ((q_stride & 3) || (t_stride & 3) || (reinterpret_cast<uintptr_t>(d_q) & 3) ||
        (reinterpret_cast<uintptr_t>(d_t) & 3))
        return set_error(ctx, SFM_E_INVALID, "float rows must be 4-byte aligned");
    L2Plan plan;
    if (make_l2_plan(ctx, dim, nq, nt, 1, &plan) != SFM_OK)
        return set_error(ctx, SFM_E_UNSUPPORTED, "float descriptors above 256 dimensions unsupported");
    const int K = plan.K;
    const int rows_cap = nq > nt ? nq : nt;
    L2Rows q, t;
    L2Scratch sc;
    Carver sz(nullptr);
    l2_rows_carve(sz, (size_t)(nq > 0 ? nq : 1), K, &q);
    l2_rows_carve(sz, (size_t)(nt > 0 ? nt : 1), K, &t);
    l2_scratch_carve(sz, plan, rows_cap, 1, 2, &sc);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
    Carver c(ws);
    l2_rows_carve(c, (size_t)(nq > 0 ? nq : 1), K, &q);
    l2_rows_carve(c, (size_t)(nt > 0 ? nt : 1), K, &t);
    l2_scratch_carve(c, plan, rows_cap, 1, 2, &sc);
    // results go straight to the caller's arrays: one problem, so [q][2] is their layout
    sc.knn_idx = d_idx;
    sc.knn_dist = d_dist;
    const L2Mode mode = l2_mode();
    FoldArgs fold[2];
    SFM_RETURN_IF(l2_pack(ctx, mode, L2Source{d_q, nq, nq, 0, q_stride, nullptr}, &q,
                          L2Source{d_t, nt, nt, 0, t_stride, nullptr}, &t,
                          false, dim, K, sc, fold));
    // (groups: the queries, then the train rows)
    SFM_RETURN_IF(l2_knn_packed(ctx, plan, mode, dim, squared, q, t, 0, rows_cap, nullptr, nq, nt, nullptr, 1, sc,
                                nq > 0 ? 1 : 0, false, fold));
    return SFM_OK;
}

namespace {
struct L2PairBuffers {
    L2Rows rows;
    L2Scratch sc;
    int32_t* best;
    float* best_dist;
};
void l2_pairs_carve(Carver& c, const L2Plan& plan, int n_frames, int rows, int n_pairs,
                    L2PairBuffers* b) {
    l2_rows_carve(c, (size_t)n_frames * rows, plan.K, &b->rows);
    const size_t nprob = 2 * (size_t)n_pairs;
    l2_scratch_carve(c, plan, rows, nprob, n_frames, &b->sc);
    b->best = c.take<int32_t>(nprob * rows);
    b->best_dist = c.take<float>(nprob * rows);
}
}  // namespace

size_t l2_match_pairs_workspace(int n_frames, int rows_per_frame, int dim, int n_pairs) {
    sfm_ctx fake;
    L2Plan plan;
    if (make_l2_plan(&fake, dim, rows_per_frame, rows_per_frame, 2 * n_pairs, &plan) != SFM_OK)
        return 0;
    Carver c(nullptr);
    L2PairBuffers b;
    l2_pairs_carve(c, plan, n_frames, rows_per_frame, n_pairs, &b);
    return c.total();
}

// Every frame is repacked (zero-padded to K floats, norms computed, narrowed) on each call.
int l2_match_frames_dev(sfm_ctx* ctx, int squared, const float* d_desc, int n_frames,
                        size_t frame_stride, int rows_per_frame, const int32_t* d_counts, int dim,
                        const int32_t* d_pairs, int n_pairs, float ratio, sfm_dmatch* d_matches,
                        int cap, int32_t* d_match_counts, int pair_chunk) {
    if (frame_stride < (size_t)rows_per_frame * dim * 4)
        return set_error(ctx, SFM_E_INVALID, "frame stride smaller than a frame");
    if ((frame_stride & 3) || (reinterpret_cast<uintptr_t>(d_desc) & 3))
        return set_error(ctx, SFM_E_INVALID, "float rows must be 4-byte aligned");
    const int chunk = n_pairs < pair_chunk ? n_pairs : pair_chunk;
    L2Plan plan;
    if (make_l2_plan(ctx, dim, rows_per_frame, rows_per_frame, 2 * chunk, &plan) != SFM_OK)
        return set_error(ctx, SFM_E_UNSUPPORTED, "float descriptors above 256 dimensions unsupported");
    Carver sz(nullptr);
    L2PairBuffers b;
    l2_pairs_carve(sz, plan, n_frames, rows_per_frame, chunk, &b);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
    Carver c(ws);
    l2_pairs_carve(c, plan, n_frames, rows_per_frame, chunk, &b);
    const L2Mode mode = l2_mode();
    // every frame in one launch, each serving as query and as train set; rows beyond a frame's
    // count are packed too (they are never read)
    L2Rows spare = b.rows;
    FoldArgs fold[2];
    SFM_RETURN_IF(l2_pack(ctx, mode, L2Source{d_desc, n_frames * rows_per_frame, rows_per_frame, frame_stride,
                                              (size_t)dim * 4, d_counts},
                          &b.rows, L2Source{nullptr, 0, 1, 0, 0, nullptr}, &spare, true, dim, plan.K, b.sc,
                          fold));
    for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int np = chunk < n_pairs - p0 ? chunk : n_pairs - p0;
        const int32_t* pairs = d_pairs + 2 * (size_t)p0;
        SFM_RETURN_IF(l2_knn_packed(ctx, plan, mode, dim, squared, b.rows, b.rows, (size_t)rows_per_frame,
                                    rows_per_frame, d_counts, rows_per_frame, rows_per_frame, pairs, 2 * np,
                                    b.sc, 0, p0 > 0, p0 == 0 ? fold : nullptr));
        SFM_RETURN_IF(launch_ratio_cross_knn(ctx, b.sc.knn_idx, b.sc.knn_dist, rows_per_frame, d_counts,
                                             rows_per_frame, pairs, np, ratio, b.best, b.best_dist,
                                             d_matches + (size_t)p0 * cap, cap,
                                             d_match_counts + p0));
    }
    return SFM_OK;
}

}  // namespace sfm
