// Both directions of a frame pair's Hamming 2-NN search from ONE pass over the
// distance matrix.
//
// DescriptorMatcher::findRobustMatches (Source Files/feature_processing.cpp:148-149)
// runs knnMatch(prev, curr) and knnMatch(curr, prev).  Both read the same matrix
// D[i][j] = popcount(prev_i ^ curr_j), once by rows and once by columns, and the xor /
// popcount work is 80 % of a direction's instructions.  This kernel evaluates every D[i][j]
// once and feeds it to two top-2 trackers: the row one (best curr rows of prev_i, held in
// VGPRs by the lane that owns prev_i) and the column one (best prev rows of curr_j).
//
// A lane owns Q rows of prev for its whole life and, per step, R rows of curr.  The 64
// lanes of a wave hold 64 different groups of R curr rows; after every step each lane moves
// on to the next group (lane l handles group (l + s) mod 64 at step s), so after 64 steps
// all 64 * Q prev rows of the wave have met all 64 * R curr rows of the batch.  The curr
// rows come from an LDS tile shared by the four waves of the workgroup; the column
// trackers of a group live in LDS too (per wave) and are handed from lane to lane through
// it: a wave's LDS operations execute in order, so no barrier is needed for that.  After a
// batch the four waves' column trackers are merged and written out as one partial result
// per (workgroup, curr row); the row trackers are written once per workgroup.  The keys are
// (distance << 18 | index), whose minimum does not depend on the visiting order, so the
// result is the one the single-direction kernel gives (hamming_knn2.hip).
#include <type_traits>

#include "sfm_internal.h"

namespace sfm {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

struct PairArgs {
    const uint32_t* base;      // packed descriptor frames
    size_t frame_words;
    const int32_t* counts;     // per-frame row counts or nullptr
    int rows_uniform;
    const int32_t* pairs;      // [n_pairs][2]
    int n_pairs;
    int qblocks;               // workgroups along prev (kBlock * Q rows each)
    int splits;                // workgroups along curr
    int batches_per_split;
    int rows_max;
    int grouped;               // 1: block -> pair mapping keeps 8 pairs per XCD label
    uint2* keys_row;           // [pair][split][rows_max]
    uint2* keys_col;           // [pair][qblock][rows_max]
};

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// popcount(x) + acc in one instruction.  Written as asm because the compiler otherwise
// re-associates the eight-term sum into v_bcnt(x, 0) + v_add3 trees: four more
// instructions per distance.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// (d << 18) | idx in one instruction (the compiler shares the shift between the row key
// and the column key and then needs three).  18 = the index bits of the keys the merge
// kernels read.
__device__ __forceinline__ uint32_t make_key(uint32_t d, uint32_t idx) {
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 18, %2" : "=v"(r) : "v"(d), "v"(idx));
    return r;
}

__device__ __forceinline__ void top2(uint32_t& m1, uint32_t& m2, uint32_t key) {
    m2 = med3_u32(m1, m2, key);
    m1 = min(m1, key);
}

template <int W, int Q, int R>
__global__ __launch_bounds__(kBlock) void hamming_pair_kernel(PairArgs a) {
    static_assert(R == 4, "the column trackers are kept as one uint4 per group");
    constexpr int kVecPerRow = W / 4;
    constexpr int kBatch = 64 * R;                      // curr rows per batch
    constexpr int kGroupVec = R * kVecPerRow + 1;       // 16-byte units per group, one of padding:
                                                        // 16 consecutive groups then start in 16 different
                                                        // bank quads, so the rotating reads do not collide
    constexpr int kTileVec = 64 * kGroupVec;
    constexpr int kLoadsPerThread = kBatch * kVecPerRow / kBlock;
    static_assert(kBatch * kVecPerRow % kBlock == 0, "tile load must divide evenly");
    __shared__ uint4 tile[2][kTileVec];
    __shared__ uint4 col1[kWaves][64], col2[kWaves][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // With many pairs, groups of 8 pairs share an XCD label (blockIdx % 8) so that a pair's
    // two frames stay in one XCD's L2; a few pairs are spread over all XCDs instead.
    const int per_pair = a.qblocks * a.splits;
    int pair, inner;
    if (a.grouped) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pair = (slot / per_pair) * 8 + xcd;
        inner = slot % per_pair;
    } else {
        pair = blockIdx.x / per_pair;
        inner = blockIdx.x % per_pair;
    }
    if (pair >= a.n_pairs) return;
    const int qb = inner / a.splits, split = inner % a.splits;
    const int fa = a.pairs[2 * pair], fb = a.pairs[2 * pair + 1];
    const int n_prev = a.counts ? a.counts[fa] : a.rows_uniform;
    const int n_curr = a.counts ? a.counts[fb] : a.rows_uniform;
    const int q_block0 = qb * (kBlock * Q);
    if (q_block0 >= n_prev) return;  // the merge reads only the partials of existing blocks
    const uint32_t* prev = a.base + (size_t)fa * a.frame_words;
    const uint4* curr = reinterpret_cast<const uint4*>(a.base + (size_t)fb * a.frame_words);

    // this lane's Q rows of prev
    uint32_t qr[Q][W], qidx[Q], qinv[Q];
    const int q_wave0 = q_block0 + wave * (64 * Q);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int qi = q_wave0 + j * 64 + lane;
        qidx[j] = (uint32_t)qi;
        qinv[j] = qi < n_prev ? 0u : kNoKey;
        if (qi < n_prev) {
            const uint4* src = reinterpret_cast<const uint4*>(prev + (size_t)qi * W);
#pragma unroll
            for (int v = 0; v < kVecPerRow; ++v) {
                const uint4 x = src[v];
                qr[j][4 * v + 0] = x.x;
                qr[j][4 * v + 1] = x.y;
                qr[j][4 * v + 2] = x.z;
                qr[j][4 * v + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) qr[j][w] = 0;
        }
    }
    uint32_t r1[Q], r2[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) r1[j] = r2[j] = kNoKey;

    const int n_batches = (n_curr + kBatch - 1) / kBatch;
    const int b_begin = split * a.batches_per_split;
    const int b_end = min(n_batches, b_begin + a.batches_per_split);
    const size_t vec_end = (size_t)n_curr * kVecPerRow;
    const bool prev_full = q_block0 + kBlock * Q <= n_prev;

    uint4 pre[kLoadsPerThread];
    auto fetch = [&](int b) {
        const size_t base = (size_t)b * kBatch * kVecPerRow;
#pragma unroll
        for (int k = 0; k < kLoadsPerThread; ++k) {
            const size_t i = base + tid + k * kBlock;
            pre[k] = i < vec_end ? curr[i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto stash = [&](uint4* dst) {
#pragma unroll
        for (int k = 0; k < kLoadsPerThread; ++k) {
            const int v = tid + k * kBlock;  // 16-byte unit of the dense batch
            dst[(v / (R * kVecPerRow)) * kGroupVec + v % (R * kVecPerRow)] = pre[k];
        }
    };
    if (b_begin < b_end) {
        fetch(b_begin);
        stash(tile[0]);
    }
    __syncthreads();

    for (int b = b_begin; b < b_end; ++b) {
        if (b + 1 < b_end) fetch(b + 1);
        const uint4* cur = tile[(b - b_begin) & 1];
        const int t0 = b * kBatch;
        const bool full = prev_full && t0 + kBatch <= n_curr;
        col1[wave][lane] = make_uint4(kNoKey, kNoKey, kNoKey, kNoKey);
        col2[wave][lane] = make_uint4(kNoKey, kNoKey, kNoKey, kNoKey);

        auto steps = [&](auto masked_tag) {
            constexpr bool kMasked = decltype(masked_tag)::value;
            for (int s = 0; s < 64; ++s) {
                const int g = (lane + s) & 63;
                const uint4* grp = cur + g * kGroupVec;
                const uint4 c1v = col1[wave][g], c2v = col2[wave][g];
                uint32_t c1[R] = {c1v.x, c1v.y, c1v.z, c1v.w};
                uint32_t c2[R] = {c2v.x, c2v.y, c2v.z, c2v.w};
                const uint32_t tg = (uint32_t)(t0 + g * R);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t tr[W];
#pragma unroll
                    for (int v = 0; v < kVecPerRow; ++v) {
                        const uint4 x = grp[r * kVecPerRow + v];
                        tr[4 * v + 0] = x.x;
                        tr[4 * v + 1] = x.y;
                        tr[4 * v + 2] = x.z;
                        tr[4 * v + 3] = x.w;
                    }
                    const uint32_t tidx = tg + r;
                    const uint32_t tinv = kMasked ? ((int)tidx < n_curr ? 0u : kNoKey) : 0u;
#pragma unroll
                    for (int j = 0; j < Q; ++j) {
                        uint32_t dsum = __builtin_popcount(qr[j][0] ^ tr[0]);
#pragma unroll
                        for (int w = 1; w < W; ++w) dsum = bcnt_acc(qr[j][w] ^ tr[w], dsum);
                        uint32_t krow = make_key(dsum, tidx);
                        uint32_t kcol = make_key(dsum, qidx[j]);
                        if (kMasked) {
                            krow |= tinv;      // a curr row past the end never becomes a neighbour
                            kcol |= qinv[j];   // nor does a prev row past the end
                        }
                        top2(r1[j], r2[j], krow);
                        top2(c1[r], c2[r], kcol);
                    }
                }
                col1[wave][g] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
                col2[wave][g] = make_uint4(c2[0], c2[1], c2[2], c2[3]);
            }
        };
        if (full)
            steps(std::false_type{});
        else
            steps(std::true_type{});

        if (b + 1 < b_end) stash(tile[(b + 1 - b_begin) & 1]);
        __syncthreads();
        // merge the four waves' column trackers: wave w takes row w of every group
        {
            const uint32_t* p1 = reinterpret_cast<const uint32_t*>(&col1[0][0]);
            const uint32_t* p2 = reinterpret_cast<const uint32_t*>(&col2[0][0]);
            uint32_t m1 = kNoKey, m2 = kNoKey;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                top2(m1, m2, p1[(w * 64 + lane) * 4 + wave]);
                top2(m1, m2, p2[(w * 64 + lane) * 4 + wave]);
            }
            const int t = t0 + lane * R + wave;
            if (t < n_curr)
                a.keys_col[((size_t)pair * a.qblocks + qb) * a.rows_max + t] = make_uint2(m1, m2);
        }
        __syncthreads();
    }

    uint2* out = a.keys_row + ((size_t)pair * a.splits + split) * (size_t)a.rows_max;
#pragma unroll
    for (int j = 0; j < Q; ++j)
        if ((int)qidx[j] < n_prev) out[qidx[j]] = make_uint2(r1[j], r2[j]);
}

}  // namespace

// Rows of prev one workgroup owns for a descriptor of `row_words` words, or 0 when this
// kernel has no instantiation for that width.
int hamming_pair_block_rows(int row_words) {
    switch (row_words) {
        case 4:
        case 8: return kBlock * 4;  // Q = 4 rows of prev per lane
        case 16: return kBlock * 2; // Q = 2: 64-byte rows take twice the registers
        default: return 0;
    }
}

PairPlan make_pair_plan(const sfm_ctx* ctx, int row_words, int rows_max, int n_pairs) {
    PairPlan p{};
    p.row_words = row_words;
    p.rows_max = rows_max;
    const int block_rows = hamming_pair_block_rows(row_words);
    p.qblocks = (rows_max + block_rows - 1) / block_rows;
    const int batch = 64 * 4;
    const int n_batches = (rows_max + batch - 1) / batch;
    // about 16 workgroups per CU, at least one batch per split
    const long want = 16L * ctx->num_cus;
    const long blocks = (long)p.qblocks * n_pairs;
    int splits = (int)((want + blocks - 1) / blocks);
    if (splits > n_batches) splits = n_batches;
    if (splits < 1) splits = 1;
    p.batches_per_split = (n_batches + splits - 1) / splits;
    p.splits = (n_batches + p.batches_per_split - 1) / p.batches_per_split;
    return p;
}

int launch_hamming_pairs(sfm_ctx* ctx, const PairPlan& plan, const void* packed, size_t frame_stride,
                         int rows_uniform, const int32_t* d_counts, const int32_t* d_pairs,
                         int n_pairs, uint32_t* d_keys_row, uint32_t* d_keys_col) {
    if (n_pairs <= 0) return SFM_OK;
    PairArgs a{};
    a.base = static_cast<const uint32_t*>(packed);
    a.frame_words = frame_stride / 4;
    a.counts = d_counts;
    a.rows_uniform = rows_uniform;
    a.pairs = d_pairs;
    a.n_pairs = n_pairs;
    a.qblocks = plan.qblocks;
    a.splits = plan.splits;
    a.batches_per_split = plan.batches_per_split;
    a.rows_max = plan.rows_max;
    a.keys_row = reinterpret_cast<uint2*>(d_keys_row);
    a.keys_col = reinterpret_cast<uint2*>(d_keys_col);
    a.grouped = n_pairs >= 64 ? 1 : 0;
    const long pairs_padded = a.grouped ? (long)((n_pairs + 7) / 8) * 8 : n_pairs;
    const long grid = pairs_padded * plan.qblocks * plan.splits;
    if (grid > 0x7fffffffL) return set_error(ctx, SFM_E_INVALID, "pair launch too large");
    KernelTimer timer(ctx, KID_HAMMING_PAIR);
    switch (plan.row_words) {
        case 4:
            hipLaunchKernelGGL((hamming_pair_kernel<4, 4, 4>), dim3((unsigned)grid), dim3(kBlock), 0,
                               ctx->stream, a);
            break;
        case 8:
            hipLaunchKernelGGL((hamming_pair_kernel<8, 4, 4>), dim3((unsigned)grid), dim3(kBlock), 0,
                               ctx->stream, a);
            break;
        case 16:
            hipLaunchKernelGGL((hamming_pair_kernel<16, 2, 4>), dim3((unsigned)grid), dim3(kBlock), 0,
                               ctx->stream, a);
            break;
        default:
            return set_error(ctx, SFM_E_UNSUPPORTED, "no pair kernel for %d-word rows", plan.row_words);
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace sfm
