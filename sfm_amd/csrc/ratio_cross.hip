// Ratio test, cross-check and ordered compaction of the 2-NN results.
//
// Replaces the host loops of DescriptorMatcher::ratioMaches (Source Files/
// feature_processing.cpp:126-129) and the cross-check of findRobustMatches
// (:159-176).  A first kernel merges the per-split keys of both directions and
// applies  d0 < ratio * d1  in float exactly as the reference does; a second one
// (one 1024-lane workgroup per frame pair) walks the backward list in ascending
// curr index (the reference's output order) and keeps the pairs whose forward
// match points back, compacting them with a ballot / prefix-popcount scan.
#include "sfm_internal.h"

namespace sfm {

namespace {

constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kIdxBits = 18;
constexpr int kThreads = 1024;

struct CrossArgs {
    // Hamming: partial packed keys per direction (0: prev -> curr, 1: curr -> prev), pair p at
    // keys_dir[d] + p * pair_stride_dir[d], part k at + k * rows_max; keys_dir[0] == nullptr ->
    // use knn_idx / knn_dist
    const uint2* keys_dir[2];
    size_t pair_stride_dir[2];
    int parts_dir[2];
    int col_block_rows;       // > 0: direction 1 has one part per col_block_rows rows of prev
    const int32_t* knn_idx;   // [2*n_pairs][rows_max][2]
    const float* knn_dist;
    const int32_t* knn_idx_b;   // set: one pair whose backward search is its own array [n_curr][2]
    const float* knn_dist_b;    //      (knn_idx / knn_dist then hold the forward search [n_prev][2])
    int rows_max;
    const int32_t* counts;
    int n_prev_uniform;
    int n_curr_uniform;
    const int32_t* pairs;
    float ratio;
    int direct;         // 1: one part per direction -- the cross-check reads the keys itself
    int32_t* best;      // [2*n_pairs][rows_max]: ratio-passing neighbour or -1
    float* best_dist;   // [2*n_pairs][rows_max]
    sfm_dmatch* matches;
    int cap;
    int32_t* match_counts;
};

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Phase 1: merge the per-split keys (or read the re-scored neighbours) and apply the
// ratio test.  One lane per query row; grid.y = problem (two per pair).
__global__ __launch_bounds__(256) void merge_ratio_kernel(CrossArgs a) {
    const int problem = blockIdx.y, pair = problem >> 1, dir = problem & 1;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_prev = a.counts ? a.counts[a.pairs[2 * pair]] : a.n_prev_uniform;
    const int n_curr = a.counts ? a.counts[a.pairs[2 * pair + 1]] : a.n_curr_uniform;
    const int nq = dir ? n_curr : n_prev;
    const int nt = dir ? n_prev : n_curr;
    if (q >= nq) return;
    const uint32_t mask = (1u << kIdxBits) - 1;
    float d0, d1;
    int32_t i0, i1;
    if (a.keys_dir[0]) {
        const uint2* keys = a.keys_dir[dir] + (size_t)pair * a.pair_stride_dir[dir];
        const int parts = dir && a.col_block_rows > 0 ? (nt + a.col_block_rows - 1) / a.col_block_rows
                                                      : a.parts_dir[dir];
        uint32_t m1 = kNoKey, m2 = kNoKey;
        for (int s = 0; s < parts; ++s) {
            const uint2 k = keys[(size_t)s * a.rows_max + q];
            m2 = med3_u32(m1, m2, k.x);
            m1 = min(m1, k.x);
            m2 = med3_u32(m1, m2, k.y);
            m1 = min(m1, k.y);
        }
        d0 = (float)(m1 >> kIdxBits);
        d1 = (float)(m2 >> kIdxBits);
        i0 = m1 == kNoKey ? -1 : (int32_t)(m1 & mask);
        i1 = m2 == kNoKey ? -1 : (int32_t)(m2 & mask);
    } else {
        const bool own = a.knn_idx_b != nullptr;
        const int32_t* ki = own && dir ? a.knn_idx_b : a.knn_idx;
        const float* kd = own && dir ? a.knn_dist_b : a.knn_dist;
        const size_t o = own ? (size_t)q * 2 : ((size_t)problem * a.rows_max + q) * 2;
        i0 = ki[o];
        i1 = ki[o + 1];
        d0 = kd[o];
        d1 = kd[o + 1];
    }
    // feature_processing.cpp:127, float * float, strict <
    const bool pass = nt >= 2 && i1 >= 0 && d0 < a.ratio * d1;
    a.best[(size_t)problem * a.rows_max + q] = pass ? i0 : -1;
    a.best_dist[(size_t)problem * a.rows_max + q] = d0;
}

// Phase 2: walk the backward list in ascending curr index and keep the pairs whose
// forward match points back, compacted in order.  One workgroup per pair.
__global__ __launch_bounds__(kThreads) void ratio_cross_kernel(CrossArgs a) {
    __shared__ int wave_count[kThreads / 64];
    const int pair = blockIdx.x;
    const int tid = threadIdx.x;
    const int n_prev = a.counts ? a.counts[a.pairs[2 * pair]] : a.n_prev_uniform;
    const int n_curr = a.counts ? a.counts[a.pairs[2 * pair + 1]] : a.n_curr_uniform;

    const int32_t* fwd = a.best + (2 * (size_t)pair) * a.rows_max;
    const int32_t* bwd = fwd + a.rows_max;
    const float* fwd_dist = a.best_dist + (2 * (size_t)pair) * a.rows_max;
    // direct mode: the single part of each direction, (key of the nearest, key of the second)
    const uint2* kf = a.direct ? a.keys_dir[0] + (size_t)pair * a.pair_stride_dir[0] : nullptr;
    const uint2* kb = a.direct ? a.keys_dir[1] + (size_t)pair * a.pair_stride_dir[1] : nullptr;
    const uint32_t mask = (1u << kIdxBits) - 1;
    // feature_processing.cpp:127 on a key pair: float * float, strict <; nt >= 2 rows
    auto passes = [&](uint2 k, int nt) {
        return nt >= 2 && k.y != kNoKey &&
               (float)(k.x >> kIdxBits) < a.ratio * (float)(k.y >> kIdxBits);
    };
    const int lane = tid & 63, wave = tid >> 6;
    int running = 0;
    for (int c0 = 0; c0 < n_curr; c0 += kThreads) {
        const int c = c0 + tid;
        int p = -1;
        bool ok = false;
        float dist = 0.f;
        if (c < n_curr) {
            if (a.direct) {
                const uint2 b = kb[c];
                if (passes(b, n_prev)) {
                    p = (int)(b.x & mask);
                    if (p < n_prev) {
                        const uint2 f = kf[p];
                        ok = passes(f, n_curr) && (int)(f.x & mask) == c;
                        dist = (float)(f.x >> kIdxBits);
                    }
                }
            } else {
                p = bwd[c];
                ok = p >= 0 && p < n_prev && fwd[p] == c;
                if (ok) dist = fwd_dist[p];
            }
        }
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        __syncthreads();
        int offset = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int cnt = wave_count[w];
            offset += w < wave ? cnt : 0;
            total += cnt;
        }
        if (ok) {
            const int pos = running + offset + prefix;
            if (pos < a.cap) {
                sfm_dmatch m;
                m.queryIdx = p;
                m.trainIdx = c;
                m.imgIdx = 0;
                m.distance = dist;
                a.matches[(size_t)pair * a.cap + pos] = m;
            }
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) a.match_counts[pair] = running;
}

// Single-direction ratio filter with ordered compaction (ratioMaches alone).
__global__ __launch_bounds__(kThreads) void ratio_compact_kernel(const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ dist,
                                                                 int nq, int nt, float ratio,
                                                                 sfm_dmatch* __restrict__ out,
                                                                 int32_t* __restrict__ out_count) {
    __shared__ int wave_count[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int running = 0;
    for (int q0 = 0; q0 < nq; q0 += kThreads) {
        const int q = q0 + tid;
        bool ok = false;
        float d0 = 0.f;
        int t0 = -1;
        if (q < nq && nt >= 2) {
            d0 = dist[2 * q];
            t0 = idx[2 * q];
            ok = d0 < ratio * dist[2 * q + 1];
        }
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        __syncthreads();
        int offset = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int cnt = wave_count[w];
            offset += w < wave ? cnt : 0;
            total += cnt;
        }
        if (ok) {
            sfm_dmatch m;
            m.queryIdx = q;
            m.trainIdx = t0;
            m.imgIdx = 0;
            m.distance = d0;
            out[running + offset + prefix] = m;
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) *out_count = running;
}

}  // namespace

int launch_ratio_cross(sfm_ctx* ctx, const KnnPlan& plan, const uint32_t* d_keys,
                       const int32_t* d_counts, int rows_uniform_prev, int rows_uniform_curr,
                       const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                       float* d_best_dist, sfm_dmatch* d_matches, int cap,
                       int32_t* d_match_counts) {
    if (n_pairs <= 0) return SFM_OK;
    CrossArgs a{};
    a.keys_dir[0] = reinterpret_cast<const uint2*>(d_keys);
    a.keys_dir[1] = a.keys_dir[0] + (size_t)plan.splits * plan.rows_max;
    a.pair_stride_dir[0] = a.pair_stride_dir[1] = 2 * (size_t)plan.splits * plan.rows_max;
    a.parts_dir[0] = a.parts_dir[1] = plan.splits;
    a.rows_max = plan.rows_max;
    a.counts = d_counts;
    a.n_prev_uniform = rows_uniform_prev;
    a.n_curr_uniform = rows_uniform_curr;
    a.pairs = d_pairs;
    a.ratio = ratio;
    a.best = d_best;
    a.best_dist = d_best_dist;
    a.matches = d_matches;
    a.cap = cap;
    a.match_counts = d_match_counts;
    a.direct = plan.splits == 1;  // nothing to merge: the cross-check reads the keys itself
    if (!a.direct) {
        KernelTimer timer(ctx, KID_MERGE_RATIO);
        hipLaunchKernelGGL(merge_ratio_kernel, dim3((a.rows_max + 255) / 256, 2 * n_pairs), dim3(256),
                           0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    KernelTimer timer(ctx, KID_RATIO_CROSS);
    hipLaunchKernelGGL(ratio_cross_kernel, dim3(n_pairs), dim3(kThreads), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_ratio_cross_pairs(sfm_ctx* ctx, const PairPlan& plan, const uint32_t* d_keys_row,
                             const uint32_t* d_keys_col, const int32_t* d_counts, int rows_uniform,
                             const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                             float* d_best_dist, sfm_dmatch* d_matches, int cap,
                             int32_t* d_match_counts) {
    if (n_pairs <= 0) return SFM_OK;
    CrossArgs a{};
    a.keys_dir[0] = reinterpret_cast<const uint2*>(d_keys_row);
    a.keys_dir[1] = reinterpret_cast<const uint2*>(d_keys_col);
    a.pair_stride_dir[0] = (size_t)plan.splits * plan.rows_max;
    a.pair_stride_dir[1] = (size_t)plan.qblocks * plan.rows_max;
    a.parts_dir[0] = plan.splits;
    a.parts_dir[1] = plan.qblocks;
    a.col_block_rows = hamming_pair_block_rows(plan.row_words);
    a.rows_max = plan.rows_max;
    a.counts = d_counts;
    a.n_prev_uniform = rows_uniform;
    a.n_curr_uniform = rows_uniform;
    a.pairs = d_pairs;
    a.ratio = ratio;
    a.best = d_best;
    a.best_dist = d_best_dist;
    a.matches = d_matches;
    a.cap = cap;
    a.match_counts = d_match_counts;
    {
        KernelTimer timer(ctx, KID_MERGE_RATIO);
        hipLaunchKernelGGL(merge_ratio_kernel, dim3((a.rows_max + 255) / 256, 2 * n_pairs), dim3(256),
                           0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    KernelTimer timer(ctx, KID_RATIO_CROSS);
    hipLaunchKernelGGL(ratio_cross_kernel, dim3(n_pairs), dim3(kThreads), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_ratio_cross_knn(sfm_ctx* ctx, const int32_t* d_knn_idx, const float* d_knn_dist,
                           int rows_max, const int32_t* d_counts, int rows_uniform,
                           const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                           float* d_best_dist, sfm_dmatch* d_matches, int cap,
                           int32_t* d_match_counts) {
    if (n_pairs <= 0) return SFM_OK;
    CrossArgs a{};
    a.knn_idx = d_knn_idx;
    a.knn_dist = d_knn_dist;
    a.rows_max = rows_max;
    a.counts = d_counts;
    a.n_prev_uniform = rows_uniform;
    a.n_curr_uniform = rows_uniform;
    a.pairs = d_pairs;
    a.ratio = ratio;
    a.best = d_best;
    a.best_dist = d_best_dist;
    a.matches = d_matches;
    a.cap = cap;
    a.match_counts = d_match_counts;
    {
        KernelTimer timer(ctx, KID_MERGE_RATIO);
        hipLaunchKernelGGL(merge_ratio_kernel, dim3((a.rows_max + 255) / 256, 2 * n_pairs), dim3(256),
                           0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    KernelTimer timer(ctx, KID_RATIO_CROSS);
    hipLaunchKernelGGL(ratio_cross_kernel, dim3(n_pairs), dim3(kThreads), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

namespace {

// Offsets of one chunk of pairs in the compact output: exclusive scan of min(count, cap) over
// the chunk's n pairs, continuing from offsets[0] (which the first chunk sets to 0).  One
// workgroup; a chunk holds at most kThreads pairs.
__global__ __launch_bounds__(kThreads) void compact_offsets_kernel(const int32_t* __restrict__ counts, int n,
                                                                   int cap, int first_chunk,
                                                                   long long* __restrict__ offsets) {
    __shared__ long long wave_sum[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = tid < n ? counts[tid] : 0;
    c = c < 0 ? 0 : (c > cap ? cap : c);
    // inclusive scan within the wave
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    long long base = first_chunk ? 0 : offsets[0];
    for (int w = 0; w < wave; ++w) base += wave_sum[w];
    __syncthreads();
    if (tid == 0 && first_chunk) offsets[0] = 0;
    if (tid < n) offsets[tid + 1] = base + incl;
}

// Pair p's first min(count, cap) matches as 6-byte records at out + offsets[p]; records that
// would land at or beyond out_cap are dropped.  grid = (pairs of the chunk, slices of a pair).
__global__ __launch_bounds__(256) void compact_write_kernel(const sfm_dmatch* __restrict__ matches,
                                                            const int32_t* __restrict__ counts, int cap,
                                                            const long long* __restrict__ offsets,
                                                            sfm_match16* __restrict__ out, long long out_cap) {
    const int pair = blockIdx.x;
    int c = counts[pair];
    c = c < 0 ? 0 : (c > cap ? cap : c);
    const long long base = offsets[pair];
    const sfm_dmatch* m = matches + (size_t)pair * cap;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < c; i += gridDim.y * blockDim.x) {
        const long long o = base + i;
        if (o >= out_cap) break;
        sfm_match16 r;
        r.queryIdx = (uint16_t)m[i].queryIdx;
        r.trainIdx = (uint16_t)m[i].trainIdx;
        r.distance = (uint16_t)m[i].distance;
        out[o] = r;
    }
}

// Segment s: len[s] elements of `words` 16-bit words from src + src_off[s] to dst + dst_off[s]
// (offsets in elements).  One workgroup walks segments blockIdx.x, + gridDim.x, ...
__global__ __launch_bounds__(256) void gather_segments_kernel(const uint16_t* __restrict__ src,
                                                              const long long* __restrict__ src_off,
                                                              const long long* __restrict__ dst_off,
                                                              const int32_t* __restrict__ len, int n_segments,
                                                              int words, uint16_t* __restrict__ dst) {
    for (int s = blockIdx.x; s < n_segments; s += gridDim.x) {
        const long long n = (long long)(len[s] < 0 ? 0 : len[s]) * words;
        const uint16_t* a = src + src_off[s] * words;
        uint16_t* b = dst + dst_off[s] * words;
        for (long long i = threadIdx.x; i < n; i += blockDim.x) b[i] = a[i];
    }
}

}  // namespace

int launch_compact_matches(sfm_ctx* ctx, const sfm_dmatch* d_matches, const int32_t* d_counts, int n_pairs,
                           int cap, bool first_chunk, long long* d_offsets, sfm_match16* d_out,
                           long long out_cap) {
    if (n_pairs <= 0) return SFM_OK;
    if (n_pairs > kThreads) return set_error(ctx, SFM_E_INVALID, "compaction chunk above %d pairs", kThreads);
    KernelTimer timer(ctx, KID_COMPACT_MATCHES);
    hipLaunchKernelGGL(compact_offsets_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, d_counts, n_pairs, cap,
                       first_chunk ? 1 : 0, d_offsets);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    const int slices = n_pairs >= 512 ? 1 : (n_pairs >= 64 ? 4 : 16);
    hipLaunchKernelGGL(compact_write_kernel, dim3(n_pairs, slices), dim3(256), 0, ctx->stream, d_matches,
                       d_counts, cap, d_offsets, d_out, out_cap);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

// One pair from two finished 2-NN searches (forward [n_prev][2], backward [n_curr][2]).
// d_best / d_best_dist: scratch of 2 * max(n_prev, n_curr) entries each.
int launch_ratio_cross_two(sfm_ctx* ctx, const int32_t* d_fwd_idx, const float* d_fwd_dist, int n_prev,
                           const int32_t* d_bwd_idx, const float* d_bwd_dist, int n_curr, float ratio,
                           int32_t* d_best, float* d_best_dist, sfm_dmatch* d_matches, int cap,
                           int32_t* d_match_count) {
    CrossArgs a{};
    a.knn_idx = d_fwd_idx;
    a.knn_dist = d_fwd_dist;
    a.knn_idx_b = d_bwd_idx;
    a.knn_dist_b = d_bwd_dist;
    a.rows_max = n_prev > n_curr ? n_prev : n_curr;
    a.n_prev_uniform = n_prev;
    a.n_curr_uniform = n_curr;
    a.ratio = ratio;
    a.best = d_best;
    a.best_dist = d_best_dist;
    a.matches = d_matches;
    a.cap = cap;
    a.match_counts = d_match_count;
    {
        KernelTimer timer(ctx, KID_MERGE_RATIO);
        hipLaunchKernelGGL(merge_ratio_kernel, dim3((a.rows_max + 255) / 256, 2), dim3(256), 0, ctx->stream, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    KernelTimer timer(ctx, KID_RATIO_CROSS);
    hipLaunchKernelGGL(ratio_cross_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_gather_segments(sfm_ctx* ctx, const void* d_src, const long long* d_src_off,
                           const long long* d_dst_off, const int32_t* d_len, int n_segments, int words,
                           void* d_dst) {
    if (n_segments <= 0) return SFM_OK;
    KernelTimer timer(ctx, KID_COMPACT_MATCHES);
    const int blocks = n_segments < 65536 ? n_segments : 65536;
    hipLaunchKernelGGL(gather_segments_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                       static_cast<const uint16_t*>(d_src), d_src_off, d_dst_off, d_len, n_segments, words,
                       static_cast<uint16_t*>(d_dst));
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_ratio_compact(sfm_ctx* ctx, const int32_t* d_idx, const float* d_dist, int nq, int nt,
                         float ratio, sfm_dmatch* d_out, int32_t* d_count) {
    KernelTimer timer(ctx, KID_RATIO_COMPACT);
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(kThreads), 0, ctx->stream, d_idx,
                       d_dist, nq, nt, ratio, d_out, d_count);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace sfm
