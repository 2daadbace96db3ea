// Internal declarations shared by the HIP translation units of libsfm_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "sfm_hip.h"

namespace sfm {
// Kernel families, for the optional per-kernel event timing (sfm_ctx_set_profiling).
enum KernelId {
    KID_HAMMING_KNN2 = 0,
    KID_HAMMING_PAIR,
    KID_RATIO_CROSS,
    KID_MERGE_RATIO,
    KID_KNN_FINALIZE,
    KID_PACK_ROWS,
    KID_RATIO_COMPACT,
    KID_L2_CANDIDATES,
    KID_L2_RESCORE,
    KID_L2_PACK,
    KID_LK_PAD_LEVEL0,
    KID_LK_PYR_DOWN,
    KID_LK_SCHARR,
    KID_LK_TRACK,
    KID_TRIANGULATE_DLT,
    KID_TRIANGULATE_LIST,
    KID_FM_SUBSETS,
    KID_FM_SOLVE,
    KID_FM_SCORE,
    KID_FM_SELECT,
    KID_CORNER_EIG,
    KID_CORNER_PICK,
    KID_CORNER_SORT,
    KID_CORNER_SELECT,
    KID_EM_ADVANCE,
    KID_EM_SOLVE,
    KID_EM_SCORE,
    KID_POSE,
    KID_HAMMING_MFMA,
    KID_HAMMING_EXPAND,
    KID_COMPACT_MATCHES,
    KID_L2_EXHAUSTIVE,
    KID_COUNT
};
struct TimedLaunch {
    int kid;
    hipEvent_t start, stop;
};
}  // namespace sfm

struct sfm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* ws = nullptr;       // grow-only device workspace
    size_t ws_bytes = 0;
    void* staging = nullptr;  // grow-only device buffer for the host-pointer entry points
    size_t staging_bytes = 0;
    void* pinned = nullptr;   // grow-only page-locked host buffer for gathering strided rows
    size_t pinned_bytes = 0;
    size_t pinned_used = 0;   // bytes handed out since the stream was last idle
    int32_t* stats = nullptr; // small device buffer of counters kernels add to ([0]: L2 queries sent
                              // through the exhaustive pass by the last L2 entry point); from
                              // sfm::kStatsTriBins on: the triangulation's bin counters, which its
                              // second launch leaves at zero for the next call
    bool tri_bins_dirty = false;  // a failed launch may have left those counters non-zero
    int num_cus = 256;
    std::string err;
    bool profiling = false;
    std::vector<sfm::TimedLaunch> timed;      // launches recorded since the last reset
    std::vector<hipEvent_t> event_pool;       // events free for reuse
};

namespace sfm {

constexpr int kStatsTriBins = 64;     // first of the triangulation's bin counters in sfm_ctx::stats
constexpr int kStatsTriBinCount = 1024;
constexpr size_t kStatsBytes = (kStatsTriBins + kStatsTriBinCount) * sizeof(int32_t);

int set_error(sfm_ctx* ctx, int code, const char* fmt, ...);

#define SFM_HIP_CHECK(ctx, expr)                                                        \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return sfm::set_error((ctx), SFM_E_HIP, "%s failed: %s (%s:%d)", #expr,     \
                                  hipGetErrorString(_e), __FILE__, __LINE__);           \
    } while (0)

#define SFM_RETURN_IF(expr)        \
    do {                           \
        int _rc = (expr);          \
        if (_rc != SFM_OK) return _rc; \
    } while (0)

// Brackets one kernel launch with HIP events on the context's stream when profiling is on.
struct KernelTimer {
    sfm_ctx* ctx;
    int kid;
    hipEvent_t start = nullptr, stop = nullptr;
    KernelTimer(sfm_ctx* c, int k);
    ~KernelTimer();
};

// Device workspace of at least `bytes`; contents are not preserved on growth.
int workspace(sfm_ctx* ctx, size_t bytes, void** out);
// Second buffer used by host-pointer wrappers to stage inputs/outputs.
int staging(sfm_ctx* ctx, size_t bytes, void** out);

// One host-to-device transfer of `rows` rows of row_bytes bytes (src_stride apart in host
// memory) into a dense device buffer, enqueued on the context's stream.  Strided sources are
// gathered in page-locked memory first: a pitched copy would cost one transfer per row.
int upload_rows(sfm_ctx* ctx, void* d_dst, const void* src, size_t src_stride, size_t row_bytes,
                size_t rows);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves aligned sub-buffers out of one allocation.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <typename T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return align_up(off, 256); }
};

// Looks up, once per device and with that device current, the address of the fixed scratch
// array sfm_triangulate_noalloc_dev works in (triangulate_dlt.hip).
void triangulate_scratch_setup(int device);

// ---- matching (hamming_knn2.hip, l2_knn2_mfma.hip, ratio_cross.hip) --------

// Geometry of one batched 2-NN launch.  A "problem" is one (query frame, train
// frame) direction; a pair contributes two problems.
struct KnnPlan {
    int norm;
    int d;            // bytes (Hamming) or floats (L2) per row
    int row_words;    // padded 32-bit words per packed row
    int q_per_block;  // query rows one workgroup owns
    int splits;       // train-dimension splits per problem
    int rows_max;     // query-row capacity: stride of the per-split key arrays
};

// nq_max / nt_max: largest query / train row count of any problem in the launch.
KnnPlan make_knn_plan(const sfm_ctx* ctx, int norm, int d, int nq_max, int nt_max, int n_problems);

// Zero-padded repack of `n` rows of `d_bytes` bytes (stride src_stride) into rows
// of row_words*4 bytes.
int pack_rows(sfm_ctx* ctx, const void* d_src, int n, size_t src_stride, int d_bytes,
              void* d_dst, int row_words);

struct KnnBatch {
    const void* q_base;       // packed rows, frame f at q_base + f*frame_stride
    const void* t_base;
    size_t q_frame_stride;    // bytes
    size_t t_frame_stride;
    const int32_t* d_counts_q;  // per-frame row counts or nullptr
    const int32_t* d_counts_t;
    int nq_uniform, nt_uniform;
    const int32_t* d_pairs;   // [n_pairs][2] or nullptr (single problem q->t)
    int n_problems;           // 2*n_pairs when d_pairs, else 1
};

// Writes partial top-2 keys: keys[problem][split][row][2] (uint32 for Hamming,
// uint64 for L2 candidates, see the kernels).
int launch_hamming_knn2(sfm_ctx* ctx, const KnnPlan& plan, const KnnBatch& b, uint32_t* d_keys);

// Both directions of every pair from one pass over the distance matrix (hamming_pair.hip).
struct PairPlan {
    int row_words;
    int rows_max;
    int qblocks;            // workgroups along the prev rows
    int splits;             // workgroups along the curr rows
    int batches_per_split;
};
// Rows of prev one workgroup owns, or 0 when no pair kernel exists for that row width.
int hamming_pair_block_rows(int row_words);
PairPlan make_pair_plan(const sfm_ctx* ctx, int row_words, int rows_max, int n_pairs);
// keys_row: [pair][split][rows_max][2] (prev -> curr), keys_col: [pair][qblock][rows_max][2]
// (curr -> prev; only the qblocks that hold rows of prev are written).
int launch_hamming_pairs(sfm_ctx* ctx, const PairPlan& plan, const void* packed, size_t frame_stride,
                         int rows_uniform, const int32_t* d_counts, const int32_t* d_pairs,
                         int n_pairs, uint32_t* d_keys_row, uint32_t* d_keys_col);
// Ratio test, cross-check and compaction over the partial keys of launch_hamming_pairs.
int launch_ratio_cross_pairs(sfm_ctx* ctx, const PairPlan& plan, const uint32_t* d_keys_row,
                             const uint32_t* d_keys_col, const int32_t* d_counts, int rows_uniform,
                             const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                             float* d_best_dist, sfm_dmatch* d_matches, int cap,
                             int32_t* d_match_counts);

// Both directions of every pair on the integer matrix cores (hamming_mfma.hip).
struct MfmaPlan {
    int row_words;
    int rows_max;
    int qtiles;   // workgroups along the query rows; 0: no kernel for this row width
    int splits;   // workgroups along the train rows
    int query_tiles;  // 32-query tiles per wave: 2 or 4
};
// Query rows one workgroup owns, or 0 when no matrix kernel exists for that row width.
int hamming_mfma_q_per_block(int row_words);
MfmaPlan make_mfma_plan(const sfm_ctx* ctx, int row_words, int rows_max, int n_problems);
// Bytes of one frame of `rows` rows in the signed-byte form the matrix kernel reads.
size_t hamming_mfma_frame_bytes(int row_words, int rows);
// Packed frames (row_words words per row, frame_stride bytes apart) -> signed-byte frames,
// hamming_mfma_frame_bytes apart.
int launch_hamming_expand(sfm_ctx* ctx, const void* packed, size_t frame_stride, int row_words,
                          int rows, int n_frames, void* d_rows);
// Writes keys[problem][split][rows_max][2], the layout launch_ratio_cross and
// launch_knn_finalize read with plan.splits parts per direction.  With d_pairs: both
// directions of every pair of frames of one array (d_q_rows == d_t_rows, nq_uniform ==
// nt_uniform rows per frame).  Without: the single problem d_q_rows against d_t_rows.
int launch_hamming_mfma(sfm_ctx* ctx, const MfmaPlan& plan, const void* d_q_rows,
                        const void* d_t_rows, int nq_uniform, int nt_uniform,
                        const int32_t* d_counts, const int32_t* d_pairs, int n_pairs,
                        uint32_t* d_keys);

// Merges the splits of a single problem and writes idx/dist [nq][2].
int launch_knn_finalize(sfm_ctx* ctx, const KnnPlan& plan, const uint32_t* d_keys, int nq,
                        int nt, int32_t* d_idx, float* d_dist);

// Per pair: merge, ratio test both ways, cross-check, ordered compaction.
int launch_ratio_cross(sfm_ctx* ctx, const KnnPlan& plan, const uint32_t* d_keys,
                       const int32_t* d_counts, int rows_uniform_prev, int rows_uniform_curr,
                       const int32_t* d_pairs, int n_pairs, float ratio, int32_t* d_best,
                       float* d_best_dist, sfm_dmatch* d_matches, int cap,
                       int32_t* d_match_counts);

// Appends the chunk's match lists (pair p: min(count, cap) entries of d_matches + p * cap) to
// the compact output as 6-byte records and writes d_offsets[1 .. n_pairs] (d_offsets[0] is the
// running start: set to 0 when first_chunk, else left by the chunk before).
int launch_compact_matches(sfm_ctx* ctx, const sfm_dmatch* d_matches, const int32_t* d_counts, int n_pairs,
                           int cap, bool first_chunk, long long* d_offsets, sfm_match16* d_out,
                           long long out_cap);
int launch_gather_segments(sfm_ctx* ctx, const void* d_src, const long long* d_src_off,
                           const long long* d_dst_off, const int32_t* d_len, int n_segments, int words,
                           void* d_dst);

}  // namespace sfm
