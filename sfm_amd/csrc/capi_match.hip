// extern "C" entry points of the descriptor-matching stage (see include/sfm_hip.h).
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

#include "sfm_internal.h"

namespace sfm {
int launch_ratio_cross_two(sfm_ctx* ctx, const int32_t* d_fwd_idx, const float* d_fwd_dist, int n_prev,
                           const int32_t* d_bwd_idx, const float* d_bwd_dist, int n_curr, float ratio,
                           int32_t* d_best, float* d_best_dist, sfm_dmatch* d_matches, int cap,
                           int32_t* d_match_count);
int launch_ratio_compact(sfm_ctx* ctx, const int32_t* d_idx, const float* d_dist, int nq, int nt,
                         float ratio, sfm_dmatch* d_out, int32_t* d_count);
int l2_knn2_dev(sfm_ctx* ctx, int squared, const float* d_q, int nq, size_t q_stride,
                const float* d_t, int nt, size_t t_stride, int dim, int32_t* d_idx, float* d_dist);
int l2_match_frames_dev(sfm_ctx* ctx, int squared, const float* d_desc, int n_frames,
                        size_t frame_stride, int rows_per_frame, const int32_t* d_counts, int dim,
                        const int32_t* d_pairs, int n_pairs, float ratio, sfm_dmatch* d_matches,
                        int cap, int32_t* d_match_counts, int pair_chunk);
size_t l2_match_pairs_workspace(int n_frames, int rows_per_frame, int dim, int n_pairs);
}  // namespace sfm

using namespace sfm;

namespace {

constexpr int kMaxTrainRows = 1 << 18;
constexpr int kMaxHammingBytes = 256;
// Pairs handled per launch sequence: bounds the key workspace (2 problems per
// pair, 8 bytes per query row and split).
constexpr int kPairChunk = 1024;
constexpr int kL2PairChunk = 256;  // float candidates take 16 bytes per query row and split

int check_common(sfm_ctx* ctx, int norm, int n_a, int n_b, int d) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (norm != SFM_NORM_HAMMING && norm != SFM_NORM_L2 && norm != SFM_NORM_L2SQR)
        return set_error(ctx, SFM_E_INVALID, "unknown norm %d", norm);
    if (n_a < 0 || n_b < 0) return set_error(ctx, SFM_E_INVALID, "negative row count");
    if (n_a >= kMaxTrainRows || n_b >= kMaxTrainRows)
        return set_error(ctx, SFM_E_INVALID, "row count must be below 2^18");
    if (d <= 0) return set_error(ctx, SFM_E_INVALID, "descriptor length must be positive");
    if (norm == SFM_NORM_HAMMING && d > kMaxHammingBytes)
        return set_error(ctx, SFM_E_UNSUPPORTED, "Hamming descriptors above %d bytes unsupported",
                         kMaxHammingBytes);
    return SFM_OK;
}

bool is_packed(const void* p, size_t stride, int d, int row_words) {
    return d == row_words * 4 && stride == (size_t)d && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// Which kernel computes the two 2-NN searches of every pair.
enum HammingPath {
    kPathKnn2 = 0,   // hamming_knn2_kernel, once per direction: any width, finely split
    kPathPair = 1,   // hamming_pair_kernel: both directions from one pass of xor / popcount
    kPathMfma = 2    // hamming_mfma_kernel: signed-byte contraction on the matrix cores
};
// SFM_PAIR_KERNEL=0/1/2 forces a path wherever the row width allows it (a debugging and
// measurement aid).  Unset: the matrix kernel for 16-, 32- and 64-byte descriptors with at
// least 128 rows per frame -- it was the fastest of the three from one 500 x 500 pair to 32
// pairs of 8192 x 8192 (tools/hamming_paths.py); otherwise the pair kernel when there are
// enough of its coarse units (one wave: 256 rows of prev against 256 of curr) to occupy the
// chip, else the finely split single-direction kernel.
HammingPath hamming_path(const sfm_ctx* ctx, int row_words, int rows_max, int n_pairs) {
    const char* e = getenv("SFM_PAIR_KERNEL");  // read per call: tests switch it in-process
    const int mode = e && *e ? atoi(e) : -1;
    const bool pair_ok = hamming_pair_block_rows(row_words) != 0;
    const bool mfma_ok = hamming_mfma_q_per_block(row_words) != 0 && rows_max >= 128;
    if (mode == 0) return kPathKnn2;
    if (mode == 1) return pair_ok ? kPathPair : kPathKnn2;
    if (mode == 2) return mfma_ok ? kPathMfma : kPathKnn2;
    const long tiles = (rows_max + 255) / 256;
    if (mfma_ok) return kPathMfma;
    if (pair_ok && (long)n_pairs * tiles * tiles >= 3L * ctx->num_cus) return kPathPair;
    return kPathKnn2;
}

// Hamming robust matching over frames already packed to plan.row_words words per
// row (and, for the matrix path, their signed-byte form `expanded`).  Workspace is carved
// from `c`.
int hamming_pairs_packed(sfm_ctx* ctx, const KnnPlan& plan, HammingPath path, const void* packed,
                         size_t frame_stride, const void* expanded, int rows_uniform,
                         const int32_t* d_counts, const int32_t* d_pairs, int n_pairs, float ratio,
                         sfm_dmatch* d_matches, int cap, int32_t* d_match_counts, Carver& c) {
    const size_t nprob = 2 * (size_t)n_pairs;
    if (path == kPathMfma) {
        const MfmaPlan mp = make_mfma_plan(ctx, plan.row_words, plan.rows_max, (int)nprob);
        uint32_t* keys = c.take<uint32_t>(nprob * mp.splits * mp.rows_max * 2);
        int32_t* best = c.take<int32_t>(nprob * plan.rows_max);
        float* best_dist = c.take<float>(nprob * plan.rows_max);
        SFM_RETURN_IF(launch_hamming_mfma(ctx, mp, expanded, expanded, rows_uniform, rows_uniform,
                                          d_counts, d_pairs, n_pairs, keys));
        KnnPlan merged = plan;
        merged.splits = mp.splits;
        return launch_ratio_cross(ctx, merged, keys, d_counts, rows_uniform, rows_uniform, d_pairs,
                                  n_pairs, ratio, best, best_dist, d_matches, cap, d_match_counts);
    }
    if (path == kPathPair) {
        const PairPlan pp = make_pair_plan(ctx, plan.row_words, plan.rows_max, n_pairs);
        uint32_t* keys_row = c.take<uint32_t>((size_t)n_pairs * pp.splits * pp.rows_max * 2);
        uint32_t* keys_col = c.take<uint32_t>((size_t)n_pairs * pp.qblocks * pp.rows_max * 2);
        int32_t* best = c.take<int32_t>(nprob * plan.rows_max);
        float* best_dist = c.take<float>(nprob * plan.rows_max);
        SFM_RETURN_IF(launch_hamming_pairs(ctx, pp, packed, frame_stride, rows_uniform, d_counts,
                                           d_pairs, n_pairs, keys_row, keys_col));
        return launch_ratio_cross_pairs(ctx, pp, keys_row, keys_col, d_counts, rows_uniform, d_pairs,
                                        n_pairs, ratio, best, best_dist, d_matches, cap,
                                        d_match_counts);
    }
    uint32_t* keys = c.take<uint32_t>(nprob * plan.splits * plan.rows_max * 2);
    int32_t* best = c.take<int32_t>(nprob * plan.rows_max);
    float* best_dist = c.take<float>(nprob * plan.rows_max);
    KnnBatch b{};
    b.q_base = b.t_base = packed;
    b.q_frame_stride = b.t_frame_stride = frame_stride;
    b.d_counts_q = b.d_counts_t = d_counts;
    b.nq_uniform = b.nt_uniform = rows_uniform;
    b.d_pairs = d_pairs;
    b.n_problems = (int)nprob;
    SFM_RETURN_IF(launch_hamming_knn2(ctx, plan, b, keys));
    return launch_ratio_cross(ctx, plan, keys, d_counts, rows_uniform, rows_uniform, d_pairs,
                              n_pairs, ratio, best, best_dist, d_matches, cap, d_match_counts);
}

size_t hamming_pairs_bytes(const sfm_ctx* ctx, const KnnPlan& plan, HammingPath path, int n_pairs) {
    Carver c(nullptr);
    const size_t nprob = 2 * (size_t)n_pairs;
    if (path == kPathMfma) {
        const MfmaPlan mp = make_mfma_plan(ctx, plan.row_words, plan.rows_max, (int)nprob);
        c.take<uint32_t>(nprob * mp.splits * mp.rows_max * 2);
    } else if (path == kPathPair) {
        const PairPlan pp = make_pair_plan(ctx, plan.row_words, plan.rows_max, n_pairs);
        c.take<uint32_t>((size_t)n_pairs * pp.splits * pp.rows_max * 2);
        c.take<uint32_t>((size_t)n_pairs * pp.qblocks * pp.rows_max * 2);
    } else {
        c.take<uint32_t>(nprob * plan.splits * plan.rows_max * 2);
    }
    c.take<int32_t>(nprob * plan.rows_max);
    c.take<float>(nprob * plan.rows_max);
    return c.total();
}

}  // namespace

extern "C" {

int sfm_knn2_dev(sfm_ctx* ctx, int norm, const void* d_q, int nq, size_t q_stride, const void* d_t,
                 int nt, size_t t_stride, int d, int32_t* d_idx, float* d_dist) {
    SFM_RETURN_IF(check_common(ctx, norm, nq, nt, d));
    if (nq == 0) return SFM_OK;
    if (!d_q || !d_idx || !d_dist || (nt > 0 && !d_t))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (norm != SFM_NORM_HAMMING)
        return l2_knn2_dev(ctx, norm == SFM_NORM_L2SQR, static_cast<const float*>(d_q), nq,
                           q_stride, static_cast<const float*>(d_t), nt, t_stride, d, d_idx,
                           d_dist);
    if (q_stride < (size_t)d || (nt > 0 && t_stride < (size_t)d))
        return set_error(ctx, SFM_E_INVALID, "row stride smaller than the row");

    KnnPlan plan = make_knn_plan(ctx, norm, d, nq, nt, 1);
    const size_t row_bytes = (size_t)plan.row_words * 4;
    const bool q_ok = is_packed(d_q, q_stride, d, plan.row_words);
    const bool t_ok = nt == 0 || is_packed(d_t, t_stride, d, plan.row_words);
    // the matrix kernel (see hamming_path) when both sets are large enough for its tiles
    const bool mfma = nt >= 128 && hamming_path(ctx, plan.row_words, nq, 1) == kPathMfma;
    const MfmaPlan mp = mfma ? make_mfma_plan(ctx, plan.row_words, std::max(nq, nt), 1) : MfmaPlan{};
    if (mfma) {
        plan.splits = mp.splits;  // the key layout the matrix kernel writes
        plan.rows_max = mp.rows_max;
    }
    Carver sz(nullptr);
    if (!q_ok) sz.take<uint8_t>((size_t)nq * row_bytes);
    if (!t_ok) sz.take<uint8_t>((size_t)nt * row_bytes);
    if (mfma) {
        sz.take<uint8_t>(hamming_mfma_frame_bytes(plan.row_words, nq));
        sz.take<uint8_t>(hamming_mfma_frame_bytes(plan.row_words, nt));
    }
    sz.take<uint32_t>((size_t)plan.splits * plan.rows_max * 2);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
    Carver c(ws);
    const void* qp = d_q;
    const void* tp = d_t;
    if (!q_ok) {
        void* p = c.take<uint8_t>((size_t)nq * row_bytes);
        SFM_RETURN_IF(pack_rows(ctx, d_q, nq, q_stride, d, p, plan.row_words));
        qp = p;
    }
    if (!t_ok) {
        void* p = c.take<uint8_t>((size_t)nt * row_bytes);
        SFM_RETURN_IF(pack_rows(ctx, d_t, nt, t_stride, d, p, plan.row_words));
        tp = p;
    }
    if (mfma) {
        void* qe = c.take<uint8_t>(hamming_mfma_frame_bytes(plan.row_words, nq));
        void* te = c.take<uint8_t>(hamming_mfma_frame_bytes(plan.row_words, nt));
        uint32_t* keys = c.take<uint32_t>((size_t)plan.splits * plan.rows_max * 2);
        SFM_RETURN_IF(launch_hamming_expand(ctx, qp, (size_t)nq * row_bytes, plan.row_words, nq, 1, qe));
        SFM_RETURN_IF(launch_hamming_expand(ctx, tp, (size_t)nt * row_bytes, plan.row_words, nt, 1, te));
        SFM_RETURN_IF(launch_hamming_mfma(ctx, mp, qe, te, nq, nt, nullptr, nullptr, 0, keys));
        return launch_knn_finalize(ctx, plan, keys, nq, nt, d_idx, d_dist);
    }
    uint32_t* keys = c.take<uint32_t>((size_t)plan.splits * plan.rows_max * 2);
    KnnBatch b{};
    b.q_base = qp;
    b.t_base = tp;
    b.nq_uniform = nq;
    b.nt_uniform = nt;
    b.n_problems = 1;
    SFM_RETURN_IF(launch_hamming_knn2(ctx, plan, b, keys));
    return launch_knn_finalize(ctx, plan, keys, nq, nt, d_idx, d_dist);
}

int sfm_knn2(sfm_ctx* ctx, int norm, const void* q, int nq, size_t q_stride, const void* t, int nt,
             size_t t_stride, int d, int32_t* idx, float* dist) {
    SFM_RETURN_IF(check_common(ctx, norm, nq, nt, d));
    if (nq == 0) return SFM_OK;
    if (!q || !idx || !dist || (nt > 0 && !t))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t elem = norm == SFM_NORM_HAMMING ? 1 : sizeof(float);
    const size_t row = (size_t)d * elem;
    if (q_stride < row || (nt > 0 && t_stride < row))
        return set_error(ctx, SFM_E_INVALID, "row stride smaller than the row");
    // rows are staged densely (stride == row bytes)
    Carver sz(nullptr);
    sz.take<uint8_t>((size_t)nq * row);
    sz.take<uint8_t>((size_t)nt * row);
    sz.take<int32_t>((size_t)nq * 2);
    sz.take<float>((size_t)nq * 2);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    uint8_t* dq = c.take<uint8_t>((size_t)nq * row);
    uint8_t* dt = c.take<uint8_t>((size_t)nt * row);
    int32_t* di = c.take<int32_t>((size_t)nq * 2);
    float* dd = c.take<float>((size_t)nq * 2);
    SFM_RETURN_IF(upload_rows(ctx, dq, q, q_stride, row, (size_t)nq));
    if (nt > 0)
        SFM_RETURN_IF(upload_rows(ctx, dt, t, t_stride, row, (size_t)nt));
    SFM_RETURN_IF(sfm_knn2_dev(ctx, norm, dq, nq, row, dt, nt, row, d, di, dd));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(idx, di, sizeof(int32_t) * 2 * nq, hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dist, dd, sizeof(float) * 2 * nq, hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pinned_used = 0;
    return SFM_OK;
}

int sfm_ratio_cross_dev(sfm_ctx* ctx, const int32_t* d_fwd_idx, const float* d_fwd_dist, int n_prev,
                        const int32_t* d_bwd_idx, const float* d_bwd_dist, int n_curr, float ratio,
                        sfm_dmatch* d_matches, int cap, int32_t* d_match_count) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_prev < 0 || n_curr < 0 || cap < 0) return set_error(ctx, SFM_E_INVALID, "negative count");
    if (!d_match_count || (cap > 0 && !d_matches)) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if ((n_prev > 0 && (!d_fwd_idx || !d_fwd_dist)) || (n_curr > 0 && (!d_bwd_idx || !d_bwd_dist)))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (n_prev >= kMaxTrainRows || n_curr >= kMaxTrainRows)
        return set_error(ctx, SFM_E_INVALID, "a frame must have fewer than 2^18 rows");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n_prev == 0 || n_curr == 0) {   // feature_processing.cpp:184: nothing to cross-check
        SFM_HIP_CHECK(ctx, hipMemsetAsync(d_match_count, 0, sizeof(int32_t), ctx->stream));
        return SFM_OK;
    }
    const size_t rows = (size_t)std::max(n_prev, n_curr);
    Carver sz(nullptr);
    sz.take<int32_t>(2 * rows);
    sz.take<float>(2 * rows);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
    Carver c(ws);
    int32_t* best = c.take<int32_t>(2 * rows);
    float* best_dist = c.take<float>(2 * rows);
    return launch_ratio_cross_two(ctx, d_fwd_idx, d_fwd_dist, n_prev, d_bwd_idx, d_bwd_dist, n_curr, ratio, best,
                                  best_dist, d_matches, cap, d_match_count);
}

int sfm_ratio_matches(sfm_ctx* ctx, int norm, const void* q, int nq, size_t q_stride, const void* t,
                      int nt, size_t t_stride, int d, float ratio, sfm_dmatch* out,
                      int* out_count) {
    SFM_RETURN_IF(check_common(ctx, norm, nq, nt, d));
    if (!out_count) return set_error(ctx, SFM_E_INVALID, "out_count is NULL");
    *out_count = 0;
    if (nq == 0 || nt < 2) return SFM_OK;
    if (!q || !t || !out) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t elem = norm == SFM_NORM_HAMMING ? 1 : sizeof(float);
    const size_t row = (size_t)d * elem;
    if (q_stride < row || t_stride < row)
        return set_error(ctx, SFM_E_INVALID, "row stride smaller than the row");
    Carver sz(nullptr);
    sz.take<uint8_t>((size_t)nq * row);
    sz.take<uint8_t>((size_t)nt * row);
    sz.take<int32_t>((size_t)nq * 2);
    sz.take<float>((size_t)nq * 2);
    sz.take<sfm_dmatch>((size_t)nq);
    sz.take<int32_t>(4);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    uint8_t* dq = c.take<uint8_t>((size_t)nq * row);
    uint8_t* dt = c.take<uint8_t>((size_t)nt * row);
    int32_t* di = c.take<int32_t>((size_t)nq * 2);
    float* dd = c.take<float>((size_t)nq * 2);
    sfm_dmatch* dm = c.take<sfm_dmatch>((size_t)nq);
    int32_t* dc = c.take<int32_t>(4);
    SFM_RETURN_IF(upload_rows(ctx, dq, q, q_stride, row, (size_t)nq));
    SFM_RETURN_IF(upload_rows(ctx, dt, t, t_stride, row, (size_t)nt));
    SFM_RETURN_IF(sfm_knn2_dev(ctx, norm, dq, nq, row, dt, nt, row, d, di, dd));
    SFM_RETURN_IF(launch_ratio_compact(ctx, di, dd, nq, nt, ratio, dm, dc));
    int32_t count = 0;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&count, dc, sizeof(int32_t), hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pinned_used = 0;
    if (count > 0)
        SFM_HIP_CHECK(ctx, hipMemcpy(out, dm, sizeof(sfm_dmatch) * (size_t)count,
                                     hipMemcpyDeviceToHost));
    *out_count = count;
    return SFM_OK;
}

size_t sfm_match_pairs_workspace(int norm, int n_frames, int rows_per_frame, int d, int n_pairs) {
    if (n_frames <= 0 || rows_per_frame <= 0 || d <= 0 || n_pairs <= 0) return 0;
    const int chunk = std::min(n_pairs, kPairChunk);
    if (norm != SFM_NORM_HAMMING)
        return l2_match_pairs_workspace(n_frames, rows_per_frame, d, std::min(chunk, kL2PairChunk));
    sfm_ctx fake;
    KnnPlan plan = make_knn_plan(&fake, norm, d, rows_per_frame, rows_per_frame, 2 * chunk);
    const bool packed = d == plan.row_words * 4;
    const HammingPath path = hamming_path(&fake, plan.row_words, plan.rows_max, chunk);
    return hamming_pairs_bytes(&fake, plan, path, chunk) +
           (packed ? 0 : align_up((size_t)n_frames * rows_per_frame * plan.row_words * 4, 256)) +
           (path == kPathMfma
                ? align_up((size_t)n_frames * hamming_mfma_frame_bytes(plan.row_words, rows_per_frame), 256)
                : 0);
}

int sfm_match_pairs_dev(sfm_ctx* ctx, int norm, const void* d_desc, int n_frames,
                        size_t frame_stride, int rows_per_frame, const int32_t* d_counts, int d,
                        const int32_t* d_pairs, int n_pairs, float ratio, sfm_dmatch* d_matches,
                        int cap, int32_t* d_match_counts) {
    SFM_RETURN_IF(check_common(ctx, norm, rows_per_frame, rows_per_frame, d));
    if (n_pairs < 0 || n_frames < 0 || cap < 0)
        return set_error(ctx, SFM_E_INVALID, "negative count");
    if (n_pairs == 0) return SFM_OK;
    if (!d_match_counts || ((rows_per_frame > 0 && cap > 0) && (!d_desc || !d_pairs || !d_matches)))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (rows_per_frame < 2) {  // no ratio test can pass: every pair has no match
        SFM_HIP_CHECK(ctx, hipMemsetAsync(d_match_counts, 0, sizeof(int32_t) * (size_t)n_pairs, ctx->stream));
        return SFM_OK;
    }
    if (!d_desc || !d_pairs || (cap > 0 && !d_matches))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (norm != SFM_NORM_HAMMING) {
        // SFM_L2_PAIR_CHUNK=<pairs>: a testing aid (read per call) that cuts the pair list into
        // smaller pieces than the workspace needs, to run the many-pieces path on few pairs
        int chunk = kL2PairChunk;
        if (const char* e = getenv("SFM_L2_PAIR_CHUNK")) chunk = std::max(1, std::min(atoi(e), kL2PairChunk));
        return l2_match_frames_dev(ctx, norm == SFM_NORM_L2SQR, static_cast<const float*>(d_desc),
                                   n_frames, frame_stride, rows_per_frame, d_counts, d, d_pairs,
                                   n_pairs, ratio, d_matches, cap, d_match_counts, chunk);
    }
    if (frame_stride < (size_t)rows_per_frame * d)
        return set_error(ctx, SFM_E_INVALID, "frame stride smaller than a frame");

    const int chunk = std::min(n_pairs, kPairChunk);
    KnnPlan plan = make_knn_plan(ctx, norm, d, rows_per_frame, rows_per_frame, 2 * chunk);
    const size_t row_bytes = (size_t)plan.row_words * 4;
    const bool packed = d == plan.row_words * 4 && (frame_stride & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(d_desc) & 15) == 0;
    const HammingPath path = hamming_path(ctx, plan.row_words, plan.rows_max, chunk);
    const size_t expanded_bytes =
        path == kPathMfma ? (size_t)n_frames * hamming_mfma_frame_bytes(plan.row_words, rows_per_frame) : 0;
    Carver sz(nullptr);
    if (!packed) sz.take<uint8_t>((size_t)n_frames * rows_per_frame * row_bytes);
    if (expanded_bytes) sz.take<uint8_t>(expanded_bytes);
    const size_t fixed = sz.total();
    void* ws;
    SFM_RETURN_IF(workspace(ctx, fixed + hamming_pairs_bytes(ctx, plan, path, chunk), &ws));
    Carver c(ws);
    const void* base = d_desc;
    size_t fstride = frame_stride;
    if (!packed) {
        uint8_t* p = c.take<uint8_t>((size_t)n_frames * rows_per_frame * row_bytes);
        fstride = (size_t)rows_per_frame * row_bytes;
        for (int f = 0; f < n_frames; ++f)
            SFM_RETURN_IF(pack_rows(ctx, static_cast<const uint8_t*>(d_desc) + f * frame_stride,
                                    rows_per_frame, (size_t)d, d, p + f * fstride, plan.row_words));
        base = p;
    }
    void* expanded = nullptr;
    if (expanded_bytes) {
        expanded = c.take<uint8_t>(expanded_bytes);
        SFM_RETURN_IF(launch_hamming_expand(ctx, base, fstride, plan.row_words, rows_per_frame,
                                            n_frames, expanded));
    }
    const size_t carve_mark = c.off;
    for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int np = std::min(chunk, n_pairs - p0);
        c.off = carve_mark;
        SFM_RETURN_IF(hamming_pairs_packed(ctx, plan, path, base, fstride, expanded, rows_per_frame,
                                           d_counts, d_pairs + 2 * (size_t)p0, np, ratio,
                                           d_matches + (size_t)p0 * cap, cap, d_match_counts + p0,
                                           c));
    }
    return SFM_OK;
}

size_t sfm_match_pairs_compact_workspace(int norm, int n_frames, int rows_per_frame, int d, int n_pairs) {
    if (norm != SFM_NORM_HAMMING || n_frames <= 0 || rows_per_frame <= 0 || d <= 0 || n_pairs <= 0) return 0;
    const int chunk = std::min(n_pairs, kPairChunk);
    Carver c(nullptr);
    c.take<sfm_dmatch>((size_t)chunk * rows_per_frame);
    return c.total() + align_up(sfm_match_pairs_workspace(norm, n_frames, rows_per_frame, d, n_pairs), 256);
}

int sfm_match_pairs_compact_dev(sfm_ctx* ctx, int norm, const void* d_desc, int n_frames,
                                size_t frame_stride, int rows_per_frame, const int32_t* d_counts, int d,
                                const int32_t* d_pairs, int n_pairs, float ratio, sfm_match16* d_out,
                                size_t out_cap, int32_t* d_pair_counts, int64_t* d_offsets) {
    SFM_RETURN_IF(check_common(ctx, norm, rows_per_frame, rows_per_frame, d));
    if (norm != SFM_NORM_HAMMING)
        return set_error(ctx, SFM_E_UNSUPPORTED, "compact match records hold Hamming distances only");
    if (rows_per_frame > 65536)
        return set_error(ctx, SFM_E_UNSUPPORTED, "compact match records index at most 65536 rows per frame");
    if (n_pairs < 0 || n_frames < 0) return set_error(ctx, SFM_E_INVALID, "negative count");
    if (!d_offsets) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (n_pairs == 0 || rows_per_frame < 2) {
        SFM_HIP_CHECK(ctx, hipMemsetAsync(d_offsets, 0, sizeof(int64_t) * ((size_t)n_pairs + 1), ctx->stream));
        if (n_pairs > 0) {
            if (!d_pair_counts) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
            SFM_HIP_CHECK(ctx, hipMemsetAsync(d_pair_counts, 0, sizeof(int32_t) * (size_t)n_pairs, ctx->stream));
        }
        return SFM_OK;
    }
    if (!d_desc || !d_pairs || !d_pair_counts || (out_cap > 0 && !d_out))
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (frame_stride < (size_t)rows_per_frame * d)
        return set_error(ctx, SFM_E_INVALID, "frame stride smaller than a frame");

    const int chunk = std::min(n_pairs, kPairChunk);
    const int cap = rows_per_frame;
    KnnPlan plan = make_knn_plan(ctx, norm, d, rows_per_frame, rows_per_frame, 2 * chunk);
    const size_t row_bytes = (size_t)plan.row_words * 4;
    const bool packed = d == plan.row_words * 4 && (frame_stride & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(d_desc) & 15) == 0;
    const HammingPath path = hamming_path(ctx, plan.row_words, plan.rows_max, chunk);
    const size_t expanded_bytes =
        path == kPathMfma ? (size_t)n_frames * hamming_mfma_frame_bytes(plan.row_words, rows_per_frame) : 0;
    Carver sz(nullptr);
    sz.take<sfm_dmatch>((size_t)chunk * cap);
    if (!packed) sz.take<uint8_t>((size_t)n_frames * rows_per_frame * row_bytes);
    if (expanded_bytes) sz.take<uint8_t>(expanded_bytes);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sz.total() + hamming_pairs_bytes(ctx, plan, path, chunk), &ws));
    Carver c(ws);
    sfm_dmatch* chunk_matches = c.take<sfm_dmatch>((size_t)chunk * cap);
    const void* base = d_desc;
    size_t fstride = frame_stride;
    if (!packed) {
        uint8_t* p = c.take<uint8_t>((size_t)n_frames * rows_per_frame * row_bytes);
        fstride = (size_t)rows_per_frame * row_bytes;
        for (int f = 0; f < n_frames; ++f)
            SFM_RETURN_IF(pack_rows(ctx, static_cast<const uint8_t*>(d_desc) + f * frame_stride,
                                    rows_per_frame, (size_t)d, d, p + f * fstride, plan.row_words));
        base = p;
    }
    void* expanded = nullptr;
    if (expanded_bytes) {
        expanded = c.take<uint8_t>(expanded_bytes);
        SFM_RETURN_IF(launch_hamming_expand(ctx, base, fstride, plan.row_words, rows_per_frame, n_frames,
                                            expanded));
    }
    const size_t carve_mark = c.off;
    for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int np = std::min(chunk, n_pairs - p0);
        c.off = carve_mark;
        SFM_RETURN_IF(hamming_pairs_packed(ctx, plan, path, base, fstride, expanded, rows_per_frame, d_counts,
                                           d_pairs + 2 * (size_t)p0, np, ratio, chunk_matches, cap,
                                           d_pair_counts + p0, c));
        SFM_RETURN_IF(launch_compact_matches(ctx, chunk_matches, d_pair_counts + p0, np, cap, p0 == 0,
                                             reinterpret_cast<long long*>(d_offsets) + p0, d_out,
                                             (long long)out_cap));
    }
    return SFM_OK;
}

int sfm_gather_segments_dev(sfm_ctx* ctx, const void* d_src, const int64_t* d_src_off,
                            const int64_t* d_dst_off, const int32_t* d_len, int n_segments, int elem_bytes,
                            void* d_dst) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_segments < 0 || elem_bytes <= 0 || (elem_bytes & 1))
        return set_error(ctx, SFM_E_INVALID, "segment count must be >= 0 and the element size even");
    if (n_segments == 0) return SFM_OK;
    if (!d_src || !d_src_off || !d_dst_off || !d_len || !d_dst)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return launch_gather_segments(ctx, d_src, reinterpret_cast<const long long*>(d_src_off),
                                  reinterpret_cast<const long long*>(d_dst_off), d_len, n_segments,
                                  elem_bytes / 2, d_dst);
}

int sfm_robust_matches(sfm_ctx* ctx, int norm, const void* prev, int n_prev, size_t prev_stride,
                       const void* curr, int n_curr, size_t curr_stride, int d, float ratio,
                       sfm_dmatch* out, int* out_count) {
    SFM_RETURN_IF(check_common(ctx, norm, n_prev, n_curr, d));
    if (!out_count) return set_error(ctx, SFM_E_INVALID, "out_count is NULL");
    *out_count = 0;
    if (n_prev < 2 || n_curr < 2) return SFM_OK;  // no ratio test can pass in both directions
    if (!prev || !curr || !out) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t elem = norm == SFM_NORM_HAMMING ? 1 : sizeof(float);
    const size_t row = (size_t)d * elem;
    if (prev_stride < row || curr_stride < row)
        return set_error(ctx, SFM_E_INVALID, "row stride smaller than the row");
    const int rows = std::max(n_prev, n_curr);
    const int cap = std::min(n_prev, n_curr);
    // two-frame stack in the staging buffer: frame 0 = prev, frame 1 = curr
    const size_t fstride = align_up((size_t)rows * row, 256);
    Carver sz(nullptr);
    sz.take<uint8_t>(2 * fstride);
    sz.take<int32_t>(8);
    sz.take<sfm_dmatch>((size_t)cap);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    uint8_t* stack = c.take<uint8_t>(2 * fstride);
    int32_t* meta = c.take<int32_t>(8);  // pairs[2], counts[2], match count
    sfm_dmatch* dm = c.take<sfm_dmatch>((size_t)cap);
    SFM_HIP_CHECK(ctx, hipMemsetAsync(stack, 0, 2 * fstride, ctx->stream));
    SFM_RETURN_IF(upload_rows(ctx, stack, prev, prev_stride, row, (size_t)n_prev));
    SFM_RETURN_IF(upload_rows(ctx, stack + fstride, curr, curr_stride, row, (size_t)n_curr));
    const int32_t h_meta[8] = {0, 1, n_prev, n_curr, 0, 0, 0, 0};
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(meta, h_meta, sizeof(h_meta), hipMemcpyHostToDevice,
                                      ctx->stream));
    SFM_RETURN_IF(sfm_match_pairs_dev(ctx, norm, stack, 2, fstride, rows, meta + 2, d, meta, 1,
                                      ratio, dm, cap, meta + 4));
    int32_t count = 0;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&count, meta + 4, sizeof(int32_t), hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pinned_used = 0;
    if (count > cap) count = cap;
    if (count > 0)
        SFM_HIP_CHECK(ctx, hipMemcpy(out, dm, sizeof(sfm_dmatch) * (size_t)count,
                                     hipMemcpyDeviceToHost));
    *out_count = count;
    return SFM_OK;
}

}  // extern "C"
