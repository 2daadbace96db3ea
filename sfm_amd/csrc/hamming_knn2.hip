// Brute-force Hamming 2-nearest-neighbour search for gfx950.
//
// Replaces cv::BFMatcher(NORM_HAMMING)::knnMatch(q, t, out, 2) as called from
// DescriptorMatcher::ratioMaches (Source Files/feature_processing.cpp:121).
//
// Work split.  A workgroup of 256 lanes owns 256*Q query rows, Q rows per lane,
// kept in VGPRs for the whole kernel.  Train rows are streamed through LDS in
// 8 KB tiles (double buffered, one barrier per tile); every lane reads the same
// train row at the same time, so the LDS reads are broadcasts.  Per (query,
// train) pair the lane issues W xor + W popcount-accumulate instructions
// (v_bcnt_u32_b32 adds into its accumulator), one v_lshl_or to build the key
// (distance << 18 | train index) and two instructions to keep the two smallest
// keys (v_med3_u32 + v_min_u32).  Packing the train index below the distance
// makes "equal distance -> lower train index first" the natural order of the
// keys, which is the order OpenCV's batchDistance produces.
//
// The train dimension can be split over workgroups (to fill the chip when there
// are few query tiles); each split writes its two best keys per query and a
// second kernel merges them.
#include <cstdlib>

#include "sfm_internal.h"

namespace sfm {

namespace {

constexpr int kBlock = 256;
constexpr int kTileBytes = 8192;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kIdxBits = 18;

struct KnnArgs {
    const uint32_t* q_base;
    const uint32_t* t_base;
    size_t q_frame_words;
    size_t t_frame_words;
    const int32_t* counts_q;
    const int32_t* counts_t;
    int nq_uniform;
    int nt_uniform;
    const int32_t* pairs;
    int n_groups;          // pairs (two problems each) or 1
    int qtiles;
    int splits;
    int rows_max;
    int blocks_per_group;  // dirs * qtiles * splits
    int grouped;           // 1: groups of 8 share an XCD label (blockIdx % 8)
    uint32_t* keys;
};

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int W, int Q>
__global__ __launch_bounds__(kBlock) void hamming_knn2_kernel(KnnArgs a) {
    constexpr int kTileRows = kTileBytes / (W * 4);
    constexpr int kVecPerRow = W / 4;
    constexpr int kVecPerTile = kTileBytes / 16;  // 512
    __shared__ uint4 tile[2][kVecPerTile];

    const int tid = threadIdx.x;
    int g, inner;
    if (a.grouped) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = (slot / a.blocks_per_group) * 8 + xcd;
        inner = slot % a.blocks_per_group;
    } else {
        g = blockIdx.x / a.blocks_per_group;
        inner = blockIdx.x % a.blocks_per_group;
    }
    if (g >= a.n_groups) return;
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
    const int nq = a.counts_q ? a.counts_q[qf] : a.nq_uniform;
    const int nt = a.counts_t ? a.counts_t[tf] : a.nt_uniform;
    const int q0 = qtile * (kBlock * Q);
    if (q0 >= nq) return;

    const uint32_t* qrows = a.q_base + (size_t)qf * a.q_frame_words;
    const uint4* trows = reinterpret_cast<const uint4*>(a.t_base + (size_t)tf * a.t_frame_words);

    // this lane's Q query rows
    uint32_t qr[Q][W];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int qi = q0 + j * kBlock + tid;
        if (qi < nq) {
            const uint4* src = reinterpret_cast<const uint4*>(qrows + (size_t)qi * W);
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
    uint32_t m1[Q], m2[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) m1[j] = m2[j] = kNoKey;

    const int chunk = (nt + a.splits - 1) / a.splits;
    const int t_begin = split * chunk;
    const int t_end = min(nt, t_begin + chunk);
    const int ntiles = t_end > t_begin ? (t_end - t_begin + kTileRows - 1) / kTileRows : 0;
    const size_t vec_end = (size_t)t_end * kVecPerRow;

    uint4 r0, r1;
    auto fetch = [&](int k) {
        const size_t base = (size_t)(t_begin + k * kTileRows) * kVecPerRow;
        const size_t i0 = base + tid, i1 = base + tid + kBlock;
        r0 = i0 < vec_end ? trows[i0] : make_uint4(0, 0, 0, 0);
        r1 = i1 < vec_end ? trows[i1] : make_uint4(0, 0, 0, 0);
    };
    if (ntiles > 0) {
        fetch(0);
        tile[0][tid] = r0;
        tile[0][tid + kBlock] = r1;
    }
    __syncthreads();

    for (int k = 0; k < ntiles; ++k) {
        if (k + 1 < ntiles) fetch(k + 1);
        const int tbase = t_begin + k * kTileRows;
        const int rows = min(kTileRows, t_end - tbase);
        const uint4* cur = tile[k & 1];
#pragma unroll 2
        for (int r = 0; r < rows; ++r) {
            uint32_t tr[W];
#pragma unroll
            for (int v = 0; v < kVecPerRow; ++v) {
                const uint4 x = cur[r * kVecPerRow + v];
                tr[4 * v + 0] = x.x;
                tr[4 * v + 1] = x.y;
                tr[4 * v + 2] = x.z;
                tr[4 * v + 3] = x.w;
            }
            const uint32_t tidx = (uint32_t)(tbase + r);
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                uint32_t dsum = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) dsum += __builtin_popcount(qr[j][w] ^ tr[w]);
                const uint32_t key = (dsum << kIdxBits) | tidx;
                m2[j] = med3_u32(m1[j], m2[j], key);
                m1[j] = min(m1[j], key);
            }
        }
        if (k + 1 < ntiles) {
            uint4* nxt = tile[(k + 1) & 1];
            nxt[tid] = r0;
            nxt[tid + kBlock] = r1;
        }
        __syncthreads();
    }

    uint2* out = reinterpret_cast<uint2*>(a.keys) +
                 ((size_t)problem * a.splits + split) * (size_t)a.rows_max;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int qi = q0 + j * kBlock + tid;
        if (qi < nq) out[qi] = make_uint2(m1[j], m2[j]);
    }
}

__global__ void pack_rows_kernel(const uint8_t* __restrict__ src, int n, size_t src_stride,
                                 int d_bytes, uint32_t* __restrict__ dst, int row_words) {
    const size_t total = (size_t)n * row_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / row_words;
        const int w = (int)(i % row_words);
        const uint8_t* s = src + row * src_stride;
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int o = 4 * w + b;
            if (o < d_bytes) v |= (uint32_t)s[o] << (8 * b);
        }
        dst[i] = v;
    }
}

__global__ void knn_finalize_kernel(const uint2* __restrict__ keys, int splits, int rows_max,
                                    int nq, int32_t* __restrict__ idx, float* __restrict__ dist) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t m1 = kNoKey, m2 = kNoKey;
    for (int s = 0; s < splits; ++s) {
        const uint2 k = keys[(size_t)s * rows_max + q];
        m2 = med3_u32(m1, m2, k.x);
        m1 = min(m1, k.x);
        m2 = med3_u32(m1, m2, k.y);
        m1 = min(m1, k.y);
    }
    const uint32_t mask = (1u << kIdxBits) - 1;
    idx[2 * q] = m1 == kNoKey ? -1 : (int32_t)(m1 & mask);
    idx[2 * q + 1] = m2 == kNoKey ? -1 : (int32_t)(m2 & mask);
    dist[2 * q] = m1 == kNoKey ? 3.402823466e+38f : (float)(m1 >> kIdxBits);
    dist[2 * q + 1] = m2 == kNoKey ? 3.402823466e+38f : (float)(m2 >> kIdxBits);
}

template <int W, int Q>
int launch_t(sfm_ctx* ctx, const KnnArgs& a, int grid) {
    KernelTimer timer(ctx, KID_HAMMING_KNN2);
    hipLaunchKernelGGL((hamming_knn2_kernel<W, Q>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace

// Padded words per row and query rows per lane for a Hamming descriptor of d bytes.
static void hamming_shape(int d, int* row_words, int* q_per_lane) {
    const int w = (d + 3) / 4;
    if (w <= 4) {
        *row_words = 4;
        *q_per_lane = 4;
    } else if (w <= 8) {
        *row_words = 8;
        *q_per_lane = 4;
    } else if (w <= 16) {
        *row_words = 16;
        *q_per_lane = 2;
    } else if (w <= 32) {
        *row_words = 32;
        *q_per_lane = 1;
    } else {
        *row_words = 64;
        *q_per_lane = 1;
    }
}

KnnPlan make_knn_plan(const sfm_ctx* ctx, int norm, int d, int nq_max, int nt_max, int n_problems) {
    KnnPlan p{};
    p.norm = norm;
    p.d = d;
    p.rows_max = nq_max;
    int qpl = 1;
    hamming_shape(d, &p.row_words, &qpl);
    p.q_per_block = kBlock * qpl;
    const int qtiles = (nq_max + p.q_per_block - 1) / p.q_per_block;
    // Split the train rows until the launch has about 32 workgroups per CU (about six are
    // resident at a time; many short workgroups keep every SIMD fed until the end of the
    // launch, measured 5 % faster than 8 per CU), but keep at least 64 train rows per split.
    const int tile_rows = kTileBytes / (p.row_words * 4);
    static const int per_cu = [] {
        const char* e = getenv("SFM_KNN_BLOCKS_PER_CU");  // tuning knob
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 32;
    }();
    const long want = (long)per_cu * ctx->num_cus;
    long blocks = (long)qtiles * n_problems;
    if (blocks < 1) blocks = 1;  // an empty query set or no problems: any plan will do
    int splits = (int)((want + blocks - 1) / blocks);
    // at least 64 train rows per split
    (void)tile_rows;
    const int max_splits = (nt_max + 63) / 64;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.splits = splits;
    return p;
}

int pack_rows(sfm_ctx* ctx, const void* d_src, int n, size_t src_stride, int d_bytes, void* d_dst,
              int row_words) {
    if (n <= 0) return SFM_OK;
    const size_t total = (size_t)n * row_words;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    KernelTimer timer(ctx, KID_PACK_ROWS);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream,
                       static_cast<const uint8_t*>(d_src), n, src_stride, d_bytes,
                       static_cast<uint32_t*>(d_dst), row_words);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_hamming_knn2(sfm_ctx* ctx, const KnnPlan& plan, const KnnBatch& b, uint32_t* d_keys) {
    KnnArgs a{};
    a.q_base = static_cast<const uint32_t*>(b.q_base);
    a.t_base = static_cast<const uint32_t*>(b.t_base);
    a.q_frame_words = b.q_frame_stride / 4;
    a.t_frame_words = b.t_frame_stride / 4;
    a.counts_q = b.d_counts_q;
    a.counts_t = b.d_counts_t;
    a.nq_uniform = b.nq_uniform;
    a.nt_uniform = b.nt_uniform;
    a.pairs = b.d_pairs;
    a.n_groups = b.d_pairs ? b.n_problems / 2 : 1;
    a.qtiles = (plan.rows_max + plan.q_per_block - 1) / plan.q_per_block;
    a.splits = plan.splits;
    a.rows_max = plan.rows_max;
    const int dirs = b.d_pairs ? 2 : 1;
    a.blocks_per_group = dirs * a.qtiles * a.splits;
    a.grouped = a.n_groups >= 64 ? 1 : 0;
    a.keys = d_keys;
    const long groups_padded = a.grouped ? (long)((a.n_groups + 7) / 8) * 8 : a.n_groups;
    const long grid = groups_padded * a.blocks_per_group;
    if (grid <= 0) return SFM_OK;
    if (grid > 0x7fffffffL) return set_error(ctx, SFM_E_INVALID, "knn launch too large");
    switch (plan.row_words) {
        case 4: return launch_t<4, 4>(ctx, a, (int)grid);
        case 8: return launch_t<8, 4>(ctx, a, (int)grid);
        case 16: return launch_t<16, 2>(ctx, a, (int)grid);
        case 32: return launch_t<32, 1>(ctx, a, (int)grid);
        case 64: return launch_t<64, 1>(ctx, a, (int)grid);
    }
    return set_error(ctx, SFM_E_UNSUPPORTED, "unsupported descriptor width %d words", plan.row_words);
}

int launch_knn_finalize(sfm_ctx* ctx, const KnnPlan& plan, const uint32_t* d_keys, int nq, int nt,
                        int32_t* d_idx, float* d_dist) {
    (void)nt;
    if (nq <= 0) return SFM_OK;
    KernelTimer timer(ctx, KID_KNN_FINALIZE);
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint2*>(d_keys), plan.splits, plan.rows_max, nq,
                       d_idx, d_dist);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace sfm
