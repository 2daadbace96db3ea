// Hamming 2-nearest-neighbour search of many (query frame, train frame) problems on the
// matrix cores of gfx950.
//
// Replaces the two cv::BFMatcher(NORM_HAMMING)::knnMatch(.., 2) calls of
// DescriptorMatcher::findRobustMatches (Source Files/feature_processing.cpp:148-149) and the
// one of DescriptorMatcher::ratioMaches (:121).
//
// popcount(q ^ t) over B bits is a contraction: with every bit written as a signed unit,
// s(0) = +1 and s(1) = -1, a train row as s(t) and a query row as -s(q),
//     sum_k s(t_k) * (-s(q_k)) = #differing - #equal = 2 h - B.
// The units are 4-bit floats (E2M1: +1 = 0x2, -1 = 0xA), the densest operand the matrix pipe
// takes: v_mfma_f32_32x32x64_f8f6f4 contracts 64 bits per instruction, and the sum is a small
// integer, exact in the float accumulator.  (The block-scaled form of the instruction, which
// an earlier version used to make one differing bit worth 8192, occupies the issue port
// longer: 721 against 695 us on the default launch.)  The accumulator of train row r starts
// at  B + (r - first row of the workgroup's train range) / 8192,  so when the k loop ends it
// IS the ranking key  2 h + local train index / 8192  (a workgroup's range holds at most 8192
// rows; 2 h <= 1024 for rows of up to 64 bytes, which leaves the float its 13 fraction bits):
// no instruction builds keys, and positive
// floats order like their bit patterns, so the lists are kept with float min / median
// instructions in the main loop and with integer ones on the final keys.  Train tiles are the A operand (rows), the queries the B operand (columns):
// each lane's 16 accumulator registers are 16 train rows of ONE query, so the running
// best-two of that query stay in the lane's registers and take three instructions per two
// distances (top2_pair).  The minimum of (distance, index) keys does not depend on the
// visiting order, so the neighbours, ties included, are those of the scalar kernels
// (hamming_knn2.hip, hamming_pair.hip).
//
// The 4-bit rows are written once per frame by hamming_expand_kernel (half a byte per
// descriptor bit); a workgroup keeps its 512 queries (8 waves x 64) in registers for its
// whole life and streams 128-row train tiles global -> LDS (XOR-swizzled so that the 16-byte
// fragment reads are conflict-free), double-buffered.  Two accumulator sets alternate, one
// per 32-row half of a tile: while the matrix pipe works on one half the vector ALU folds the
// previous one into the best-two lists, a slice after every k step.
#include "sfm_internal.h"

namespace sfm {

namespace {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kTileUnit = 64;        // train ranges are whole multiples of this many rows
constexpr int kLocalBits = 13;       // a workgroup's train range holds at most 2^13 rows
constexpr int kIdxBits = 18;         // index bits of the keys the merge kernels read
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr uint32_t kEmpty = 0x7F7FFFFFu;   // tracker start (FLT_MAX), above every accumulator
constexpr float kMasked = 1e30f;           // accumulator start of a row past the range
constexpr float kRealLimit = 1e20f;
constexpr int kAhead = 2;                  // fragment reads run this many k steps ahead

struct MfmaArgs {
    const uint8_t* q_rows;   // 4-bit frames: frame f at q_rows + f * frame_bytes
    const uint8_t* t_rows;   // (the same array for frame pairs)
    size_t frame_bytes;
    const int32_t* counts;   // per-frame row counts or nullptr
    int nq_uniform, nt_uniform;
    const int32_t* pairs;    // [n_pairs][2]; problem 2p is (a -> b), 2p + 1 is (b -> a);
                             // nullptr: one problem, q_rows against t_rows
    int n_problems;
    int qtiles, splits;
    int rows_max;
    int grouped;
    uint2* keys;             // [problem][split][rows_max]
};

// Written as asm: from the min / max forms the compiler shares subexpressions between the
// median and the minimum and ends up with more instructions.
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

__device__ __forceinline__ void top2(uint32_t& m1, uint32_t& m2, uint32_t key) {
    m2 = med3_u32(m1, m2, key);
    m1 = min(m1, key);
}

// The lists of the main loop are kept on the keys as floats (non-negative, so they order like
// their bit patterns): v_med3_f32 / v_min3_f32 through the compiler, which then knows the
// readers of an MFMA result and what follows them (asm statements cost a wait state each
// here: the compiler takes an asm result for that of a transcendental instruction).
__device__ __forceinline__ float fmin3(float a, float b, float c) {
    return __builtin_fminf(__builtin_fminf(a, b), c);
}

// Two keys at once in three instructions (v_med3, v_min, v_min3) instead of four: with
// m1 <= m2, the second smallest of {m1, m2, x, y} is min(m2, median(m1, x, y)).
__device__ __forceinline__ void top2_pair(float& m1, float& m2, float x, float y) {
    m2 = __builtin_fminf(m2, __builtin_amdgcn_fmed3f(m1, x, y));
    m1 = fmin3(m1, x, y);
}

// Four keys at once in five instructions: two of the steps above with the two updates of m2
// joined into one v_min3.
__device__ __forceinline__ void top2_quad(float& m1, float& m2, float x0, float y0, float x1, float y1) {
    const float a = __builtin_amdgcn_fmed3f(m1, x0, y0);
    m1 = fmin3(m1, x0, y0);
    const float b = __builtin_amdgcn_fmed3f(m1, x1, y1);
    m2 = fmin3(m2, a, b);
    m1 = fmin3(m1, x1, y1);
}

// One thread per descriptor word: 32 bits become 32 four-bit floats (16 bytes), bit 0 -> +1
// (0x2), bit 1 -> -1 (0xA); nibble b of a row is bit b of the descriptor.
__global__ __launch_bounds__(256) void hamming_expand_kernel(const uint32_t* __restrict__ packed,
                                                             size_t frame_words, int row_words,
                                                             int rows, int n_frames,
                                                             uint4* __restrict__ out) {
    const size_t per_frame = (size_t)rows * row_words;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_frame * n_frames) return;
    const size_t f = i / per_frame, r = i % per_frame;
    const uint32_t word = packed[f * frame_words + r];
    uint32_t d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t t = (word >> (8 * q)) & 0xFFu;       // bit j of the byte -> bit 4 j
        t = (t | (t << 12)) & 0x000F000Fu;
        t = (t | (t << 6)) & 0x03030303u;
        t = (t | (t << 3)) & 0x11111111u;
        d[q] = 0x22222222u | (t << 3);                // the sign bit of nibble j
    }
    out[i] = make_uint4(d[0], d[1], d[2], d[3]);
}

// W: 32-bit words per packed row (the contraction length is 32 W; 64 per MFMA), NT: 32-query
// tiles per wave, kTileRows: train rows per LDS tile (a multiple of 64).
template <int W, int NT, int kTileRows>
__global__ __launch_bounds__(kThreads) void hamming_mfma_kernel(MfmaArgs a) {
    static_assert(W % 2 == 0, "whole 64-bit k steps");
    constexpr int KS = W / 2;                            // k steps (MFMAs) per 32 x 32 tile
    constexpr int kRowBytes = 16 * W;
    constexpr int kCpr = W;                              // 16-byte chunks per row: one per word
    constexpr int kSwz = kCpr < 16 ? kCpr - 1 : 15;
    constexpr int kTileBytes = kTileRows * kRowBytes;
    constexpr int kStagePerWave = kTileRows * kCpr / 64 / kWaves;  // 1 KiB wave-instructions per wave and tile
    static_assert((kTileRows * kCpr / 64) % kWaves == 0, "a tile must split evenly over the waves");
    static_assert(kTileRows % kTileUnit == 0, "whole 64-row units");
    constexpr int kQPerWave = 32 * NT;
    constexpr int kQPerBlock = kWaves * kQPerWave;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * kTileBytes];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    const int per_problem = a.qtiles * a.splits;
    int problem, inner;
    if (a.grouped) {  // eight problems per XCD label: a problem's train frame stays in one L2
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        problem = (slot / per_problem) * 8 + xcd;
        inner = slot % per_problem;
    } else {
        problem = blockIdx.x / per_problem;
        inner = blockIdx.x % per_problem;
    }
    if (problem >= a.n_problems) return;
    int qf = 0, tf = 0;
    if (a.pairs) {
        const int pair = problem >> 1, dir = problem & 1;
        const int fa = a.pairs[2 * pair], fb = a.pairs[2 * pair + 1];
        qf = dir ? fb : fa;
        tf = dir ? fa : fb;
    }
    const int nq = a.counts ? a.counts[qf] : a.nq_uniform;
    const int nt = a.counts ? a.counts[tf] : a.nt_uniform;
    const int qtile = inner / a.splits, split = inner % a.splits;
    const int q0 = qtile * kQPerBlock;
    if (q0 >= nq) return;  // the merge reads only rows below nq
    const uint8_t* qrows = a.q_rows + (size_t)qf * a.frame_bytes;
    const uint8_t* trows = a.t_rows + (size_t)tf * a.frame_bytes;

    // B operand: the negated units of this lane's queries, resident for the whole kernel.
    // breg[n][m] holds bits 64m + 32lh .. + 31 of query q0 + wave * kQPerWave + 32n + li.
    intx4 breg[NT][KS];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        int qi = q0 + wave * kQPerWave + 32 * n + li;
        qi = qi < nq ? qi : nq - 1;
        const intx4* src = reinterpret_cast<const intx4*>(qrows + (size_t)qi * kRowBytes);
#pragma unroll
        for (int m = 0; m < KS; ++m) {
            const intx4 v = src[2 * m + lh];
            breg[n][m] = v ^ (int)0x88888888;  // +-1 -> -+1
        }
    }
    // best two per query tile, kept twice (even and odd accumulator pairs) so that the
    // vector ALU always has independent chains to issue from; merged at the end
    float m1[NT][2], m2[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n) m1[n][0] = m1[n][1] = m2[n][0] = m2[n][1] = __uint_as_float(kEmpty);

    // this workgroup's train range: whole tiles, at most 2^kLocalBits rows
    const int units_total = (nt + kTileUnit - 1) / kTileUnit;
    const int units_per_split = (units_total + a.splits - 1) / a.splits;
    const int t_begin = split * units_per_split * kTileUnit;
    const int t_end = min(nt, t_begin + units_per_split * kTileUnit);
    const int ntiles = t_end > t_begin ? (t_end - t_begin + kTileRows - 1) / kTileRows : 0;

    // global -> LDS copy of one tile: LDS position p (16-byte chunks, linear) holds chunk
    // (p % kCpr) ^ (row & kSwz) of row p / kCpr; each wave instruction fills 64 positions.
    auto stage = [&](int k, int buf) {
        const int tbase = t_begin + k * kTileRows;
        uint8_t* dst = lds + buf * kTileBytes;
#pragma unroll
        for (int it = 0; it < kStagePerWave; ++it) {
            const int instr = wave * kStagePerWave + it;
            const int p = instr * 64 + lane;
            const int row = p / kCpr, cpos = p % kCpr;
            const int c = cpos ^ (row & kSwz);
            int trow = tbase + row;
            trow = trow < nt ? trow : nt - 1;  // rows past the end are masked below
            glds16(trows + (size_t)trow * kRowBytes + c * 16, dst + instr * 1024);
        }
    };

    if (ntiles > 0) stage(0, 0);

    // accumulator start: register r of a 32-row half tile is its train row 8(r>>2) + 4lh + (r&3);
    // it starts at the number of bits + the row's index in the workgroup's range / 8192.  The start values of
    // a half tile are formed afresh (a per-lane value + one constant per register): a vector
    // updated in place would be live across the MFMAs that start from it and cost a register
    // copy per half tile.
    constexpr float kUnit = 1.f / (float)(1 << kLocalBits);   // one index step in the key
    const float lane_base = (float)(32 * W) + (float)(4 * lh) * kUnit;
    const int scale = 0;
    __syncthreads();

    // Two accumulator sets, one per half of a tile.  While the matrix pipe accumulates one half
    // tile into one set, the vector ALU folds the other set (the previous half tile) into the
    // best-two lists: behind every MFMA the same share of it, so that the wave offers the
    // issue port one MFMA and about five vector instructions in turn.
    floatx16 acc0[NT], acc1[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[n][r] = acc1[n][r] = kMasked;

    // Pairs [p0, p1) of the eight register pairs of query tile n's accumulator; consecutive
    // pairs go to the same one of the tile's two lists (four keys in five instructions).
    auto fold_pairs = [&](const floatx16 (&acc)[NT], int n, int p0, int p1) {
        int p = p0;
#pragma unroll
        for (; p + 1 < p1; p += 2)
            top2_quad(m1[n][(p >> 1) & 1], m2[n][(p >> 1) & 1], acc[n][2 * p], acc[n][2 * p + 1],
                      acc[n][2 * p + 2], acc[n][2 * p + 3]);
        if (p < p1)
            top2_pair(m1[n][(p >> 1) & 1], m2[n][(p >> 1) & 1], acc[n][2 * p], acc[n][2 * p + 1]);
    };

    const int left = t_end - t_begin;
    // One tile: kSteps k steps (KS per 32-row half tile), the A fragments read kAhead k steps
    // ahead through a ring.  first: index of the tile's first row in the range; masked: the
    // tile may hold rows past the range (the last tile only).
    constexpr int kHalves = kTileRows / 32;
    constexpr int kSteps = kHalves * KS;
    constexpr int kRing = kAhead + 1;
    const int swz = li & kSwz;                       // 32-row pieces keep row & kSwz == li & kSwz
    const int lane_off = li * kRowBytes;
    auto tile_body = [&](const uint8_t* tile, int first, bool masked) {
        const uint8_t* rowp = tile + lane_off;
        auto frag = [&](int s) {
            const int mt = s / KS, m = s % KS;
            return *reinterpret_cast<const intx4*>(rowp + mt * (32 * kRowBytes) +
                                                   (((2 * m + lh) ^ swz) * 16));
        };
        intx4 ring[kRing];
#pragma unroll
        for (int s = 0; s < kAhead && s < kSteps; ++s) ring[s] = frag(s);
        const float first_base = lane_base + (float)first * kUnit;
#pragma clang loop unroll(full)
        for (int s = 0; s < kSteps; ++s) {
            const int mt = s / KS, m = s % KS;
            // Half way through the tile the copy of the next one has landed: retire it, so that
            // for the rest of the tile the compiler waits for single fragment reads (while a
            // global -> LDS load is in flight it waits for all outstanding LDS reads at once)
            if (s == kSteps / 2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
            if (s + kAhead < kSteps) ring[(s + kAhead) % kRing] = frag(s + kAhead);
            floatx16 (&accC)[NT] = (mt & 1) ? acc1 : acc0;
            const floatx16 (&accF)[NT] = (mt & 1) ? acc0 : acc1;
            floatx16 start;
            if (m == 0) {
                // one v_add_f32 per register, written as asm: left to itself the compiler packs
                // pairs of these adds into v_pk_add_f32 on misaligned register pairs and then
                // moves all sixteen results into place
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    asm("v_add_f32 %0, %1, %2"
                        : "=v"(start[r])
                        : "s"((float)(32 * mt + 8 * (r >> 2) + (r & 3)) * kUnit), "v"(first_base));
                if (masked) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (first + 32 * mt + 8 * (r >> 2) + 4 * lh + (r & 3) >= left) start[r] = kMasked;
                }
            }
            const intx4 av = ring[s % kRing];
            const intx8 a8 = {av[0], av[1], av[2], av[3], 0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const intx4 bv = breg[n][m];
                const intx8 b8 = {bv[0], bv[1], bv[2], bv[3], 0, 0, 0, 0};
                // cbsz = blgp = 4: both operands E2M1; scale operands 0, 0 select the unscaled
                // instruction
                accC[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    a8, b8, m == 0 ? start : accC[n], 4, 4, 0, scale, 0, scale);
                // This tile's share of the other set, whose last MFMA was issued NT MFMAs ago.
                // The barrier keeps the share behind its MFMA (the compiler otherwise gathers
                // the fold in front of the MFMAs).
                fold_pairs(accF, n, (8 * m) / KS, (8 * (m + 1)) / KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    for (int k = 0; k + 1 < ntiles; ++k) {
        const int buf = k & 1;
        stage(k + 1, buf ^ 1);
        tile_body(lds + buf * kTileBytes, k * kTileRows, false);
        __syncthreads();
    }
    if (ntiles > 0) tile_body(lds + ((ntiles - 1) & 1) * kTileBytes, (ntiles - 1) * kTileRows, true);
#pragma unroll
    for (int n = 0; n < NT; ++n) fold_pairs(acc1, n, 0, 8);

    // accumulator -> (distance << 18 | train index); the two lane halves saw different rows
    auto to_key = [&](float f) {
        const uint32_t v = (uint32_t)(f * (float)(1 << kLocalBits));   // 2 h << 13 | local, exact
        const uint32_t h = v >> (kLocalBits + 1), local = v & ((1u << kLocalBits) - 1);
        return f < kRealLimit ? (h << kIdxBits) | (uint32_t)(t_begin + (int)local) : kNoKey;
    };
    uint2* out = a.keys + ((size_t)problem * a.splits + split) * (size_t)a.rows_max;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        uint32_t k1 = to_key(m1[n][0]), k2 = to_key(m2[n][0]);
        top2(k1, k2, to_key(m1[n][1]));
        top2(k1, k2, to_key(m2[n][1]));
        const uint32_t o1 = __shfl_xor(k1, 32), o2 = __shfl_xor(k2, 32);
        top2(k1, k2, o1);
        top2(k1, k2, o2);
        const int qi = q0 + wave * kQPerWave + 32 * n + li;
        if (lh == 0 && qi < nq) out[qi] = make_uint2(k1, k2);
    }
}

}  // namespace

int hamming_mfma_q_per_block(int row_words) {
    switch (row_words) {
        case 4:
        case 8:
        case 16: return kWaves * 64;
        default: return 0;
    }
}

MfmaPlan make_mfma_plan(const sfm_ctx* ctx, int row_words, int rows_max, int n_problems) {
    MfmaPlan p{};
    p.row_words = row_words;
    p.rows_max = rows_max;
    int qpb = hamming_mfma_q_per_block(row_words);
    p.query_tiles = 2;
    // 32-byte rows: four query tiles per wave (1024 queries per workgroup) when that still
    // leaves two rounds of workgroups -- more independent accumulators per wave and half the
    // fragment reads and start values per MFMA (400 against 446 us on 32 pairs of 8192 x 8192;
    // one such pair alone is faster with the smaller workgroups)
    if (row_words == 8 && (long)n_problems * ((rows_max + 2 * qpb - 1) / (2 * qpb)) >= 2L * ctx->num_cus) {
        p.query_tiles = 4;
        qpb *= 2;
    }
    p.qtiles = qpb ? (rows_max + qpb - 1) / qpb : 0;
    const int tiles = (rows_max + kTileUnit - 1) / kTileUnit;
    const int tiles_cap = (1 << kLocalBits) / kTileUnit;   // rows per range the keys can index
    int min_splits = (tiles + tiles_cap - 1) / tiles_cap;
    // one workgroup per CU at a time: aim at four rounds of them
    const long want = 4L * ctx->num_cus;
    const long blocks = (long)p.qtiles * n_problems;
    int splits = blocks > 0 ? (int)((want + blocks - 1) / blocks) : 1;
    if (splits > tiles / 2) splits = tiles / 2;  // at least two tiles per range
    if (splits < min_splits) splits = min_splits;
    if (splits < 1) splits = 1;
    const int per = (tiles + splits - 1) / splits;
    p.splits = per > 0 ? (tiles + per - 1) / per : 1;
    if (p.splits < 1) p.splits = 1;
    return p;
}

size_t hamming_mfma_frame_bytes(int row_words, int rows) { return (size_t)rows * row_words * 16; }

int launch_hamming_expand(sfm_ctx* ctx, const void* packed, size_t frame_stride, int row_words,
                          int rows, int n_frames, void* d_rows) {
    if (rows <= 0 || n_frames <= 0) return SFM_OK;
    const size_t units = (size_t)rows * row_words * n_frames;
    const size_t grid = (units + 255) / 256;
    if (grid > 0x7fffffffUL) return set_error(ctx, SFM_E_INVALID, "expand launch too large");
    KernelTimer timer(ctx, KID_HAMMING_EXPAND);
    hipLaunchKernelGGL(hamming_expand_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream,
                       static_cast<const uint32_t*>(packed), frame_stride / 4, row_words, rows,
                       n_frames, static_cast<uint4*>(d_rows));
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int launch_hamming_mfma(sfm_ctx* ctx, const MfmaPlan& plan, const void* d_q_rows,
                        const void* d_t_rows, int nq_uniform, int nt_uniform,
                        const int32_t* d_counts, const int32_t* d_pairs, int n_pairs,
                        uint32_t* d_keys) {
    if (d_pairs && n_pairs <= 0) return SFM_OK;
    if (plan.qtiles <= 0) return set_error(ctx, SFM_E_UNSUPPORTED, "no matrix kernel for %d-word rows", plan.row_words);
    MfmaArgs a{};
    a.q_rows = static_cast<const uint8_t*>(d_q_rows);
    a.t_rows = static_cast<const uint8_t*>(d_t_rows);
    a.frame_bytes = hamming_mfma_frame_bytes(plan.row_words, nq_uniform);
    a.counts = d_counts;
    a.nq_uniform = nq_uniform;
    a.nt_uniform = nt_uniform;
    a.pairs = d_pairs;
    a.n_problems = d_pairs ? 2 * n_pairs : 1;
    a.qtiles = plan.qtiles;
    a.splits = plan.splits;
    a.rows_max = plan.rows_max;
    a.keys = reinterpret_cast<uint2*>(d_keys);
    a.grouped = a.n_problems >= 64 ? 1 : 0;
    const long padded = a.grouped ? (long)((a.n_problems + 7) / 8) * 8 : a.n_problems;
    const long grid = padded * plan.qtiles * plan.splits;
    if (grid > 0x7fffffffL) return set_error(ctx, SFM_E_INVALID, "matrix launch too large");
    KernelTimer timer(ctx, KID_HAMMING_MFMA);
    switch (plan.row_words) {
        case 4:
            hipLaunchKernelGGL((hamming_mfma_kernel<4, 2, 128>), dim3((unsigned)grid), dim3(kThreads), 0,
                               ctx->stream, a);
            break;
        case 8:
            if (plan.query_tiles == 4)
                hipLaunchKernelGGL((hamming_mfma_kernel<8, 4, 128>), dim3((unsigned)grid), dim3(kThreads),
                                   0, ctx->stream, a);
            else
                hipLaunchKernelGGL((hamming_mfma_kernel<8, 2, 128>), dim3((unsigned)grid), dim3(kThreads),
                                   0, ctx->stream, a);
            break;
        case 16:
            hipLaunchKernelGGL((hamming_mfma_kernel<16, 2, 128>), dim3((unsigned)grid), dim3(kThreads), 0,
                               ctx->stream, a);
            break;
        default:
            return set_error(ctx, SFM_E_UNSUPPORTED, "no matrix kernel for %d-word rows", plan.row_words);
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace sfm
