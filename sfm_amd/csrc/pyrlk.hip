// Pyramidal Lucas-Kanade sparse optical flow for gfx950.
//
// Replaces cv::SparsePyrLKOpticalFlow::calc as called from OptFlow::computeFlow
// (Source Files/feature_processing.cpp:241) plus the error gate at :245-248.
// The arithmetic follows OpenCV's lkpyramid.cpp / pyramids.cpp for 8-bit
// single-channel images with flags == 0:
//   - pyramid: pyrDown ([1 4 6 4 1] separable, (v+128)>>8, reflect-101), every
//     level stored with a reflect-101 border of one window;
//   - Scharr derivative of the previous image per level, stored as one 32-bit
//     word per pixel (4 Ix in the low half, 4 Iy in the high half), zero outside;
//   - tracker: one 64-lane wave per point walks the levels from the top.  A lane
//     owns up to 7 horizontally adjacent window pixels of one window row
//     ("slot"); the window of I and of its derivatives is interpolated once per
//     level with OpenCV's 14-bit fixed-point weights and then stays in VGPRs for
//     all iterations.  Each iteration reads the window of J straight from the
//     padded pyramid (two unaligned 8-byte loads per slot), forms the mismatch
//     vector and reduces it over the wave with DPP adds.
// The normal-equation sums (A11, A12, A22, b1, b2) are accumulated exactly in
// integers (OpenCV: float accumulators whose order depends on its SIMD build);
// every other operation is the float / fixed-point operation OpenCV performs, in
// the same order, with FP contraction disabled.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "sfm_internal.h"

namespace sfm {

namespace {

constexpr int kMaxLevels = 12;
constexpr int kPxPerSlot = 7;
constexpr int W_BITS = 14;

struct LkGeom {
    int levels;                 // top level index (levels+1 levels exist)
    int w[kMaxLevels], h[kMaxLevels];
    int stride[kMaxLevels];     // elements per padded row (bytes for images, words for deriv)
    size_t img_off[kMaxLevels]; // offset of pixel (0,0) of level l inside one image block
    size_t block;               // elements per image block (all levels)
    int bw, bh;                 // left / top border: window width rounded up to 16, window height
};

// t / d and t % d for 0 <= t < 2^23, 0 < d < 2^23 by a float reciprocal and one fix-up
// (an integer division costs about forty instructions).
__device__ __forceinline__ void divmod_small(int t, int d, int& q, int& r) {
    q = (int)((float)t * __frcp_rn((float)d));
    r = t - q * d;
    if (r < 0) {
        --q;
        r += d;
    } else if (r >= d) {
        ++q;
        r -= d;
    }
}

__device__ __forceinline__ int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// ---- pyramid construction ---------------------------------------------------

// Level 0 of every image: copy with a reflect-101 border.  two_sets: grid.z = 2 * n_pairs
// (even: previous image, odd: next image, each set with its own row pitch); otherwise grid.z =
// number of images of the first set.  One thread writes 16 bytes; in the interior that is one
// unaligned 16-byte load and one store.
__global__ void lk_pad_level0_kernel(const uint8_t* __restrict__ prev,
                                     const uint8_t* __restrict__ curr, size_t image_stride,
                                     size_t prev_row_stride, size_t curr_row_stride, bool two_sets, LkGeom g,
                                     uint8_t* __restrict__ pyr_prev, uint8_t* __restrict__ pyr_curr) {
    const int pair = two_sets ? blockIdx.z >> 1 : blockIdx.z, which = two_sets ? blockIdx.z & 1 : 0;
    const size_t row_stride = which ? curr_row_stride : prev_row_stride;
    const uint8_t* src = (which ? curr : prev) + (size_t)pair * image_stride;
    uint8_t* dst = (which ? pyr_curr : pyr_prev) + (size_t)pair * g.block + g.img_off[0];
    const int S = g.stride[0], w = g.w[0], h = g.h[0];
    const int per_row = S / 16, total = per_row * (h + 2 * g.bh);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int yy, xg;
        divmod_small(t, per_row, yy, xg);
        const int y = yy - g.bh, x16 = xg * 16;
        const int sy = reflect101(y, h);
        const uint8_t* srow = src + (size_t)sy * row_stride;
        uint8_t* drow = dst + (ptrdiff_t)y * S - g.bw;
        const int x = x16 - g.bw;
        uint4 v;
        // reflect-101 of a whole group that lies left (m = -x - 15) or right (m = 2(w-1) - x - 15)
        // of the image: the 16 source bytes m .. m+15 in reverse order
        const int m = x + 16 <= 0 ? -x - 15 : 2 * (w - 1) - x - 15;
        if (x >= 0 && x + 16 <= w) {
            __builtin_memcpy(&v, srow + x, 16);
        } else if ((x + 16 <= 0 || (x >= w && x + 16 <= w + g.bw)) && m >= 0 && m + 16 <= w) {
            uint4 r;
            __builtin_memcpy(&r, srow + m, 16);
            v = make_uint4(__builtin_bswap32(r.w), __builtin_bswap32(r.z), __builtin_bswap32(r.y),
                           __builtin_bswap32(r.x));
        } else {
            uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int xb = x + b;
                const uint32_t px = xb < w + g.bw ? srow[reflect101(xb, w)] : 0u;
                q[b >> 2] |= px << (8 * (b & 3));
            }
            v = make_uint4(q[0], q[1], q[2], q[3]);
        }
        *reinterpret_cast<uint4*>(drow + x16) = v;
    }
}

// pyrDown at output pixel (rx, ry) of a level whose source (with its reflect-101
// border, which holds exactly the values pyrDown's border handling asks for) is src.
__device__ __forceinline__ uint32_t pyr_down_px(const uint8_t* src, int SS, int rx, int ry) {
    const uint8_t* s = src + (ptrdiff_t)(2 * ry - 2) * SS + (2 * rx - 2);
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint8_t* r = s + (ptrdiff_t)k * SS;
        const int hsum = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
        acc += hsum * (k == 2 ? 6 : (k == 1 || k == 3) ? 4 : 1);
    }
    return (uint32_t)((acc + 128) >> 8);
}

// Horizontal [1 4 6 4 1] sums of one source row for four adjacent pyrDown outputs x .. x+3
// (all inside the level): output i takes the five source bytes 2i .. 2i+4: a byte dot product
// over four of them plus a second one that picks the fifth.  s points at source column 2x-4
// of the row.  kAligned: x is a multiple of 4 (groups inside the image); the mirrored groups
// of the border are not.
template <bool kAligned>
__device__ __forceinline__ void pyr_down_row(const uint8_t* s, uint32_t (&hs)[4]) {
    constexpr uint32_t kW = 0x04060401u;                      // weights of bytes 0..3
    constexpr uint32_t kE0 = 0x00000001u, kE2 = 0x00010000u;  // byte 0 / byte 2 alone
    // source columns 2x-2 .. 2x+8, read as four dwords from column 2x-4 (with x a multiple
    // of 4 and rows 16-byte aligned that is one aligned load; one at 2x-2 would not be)
    uint4 d;
    if (kAligned)
        d = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(s, 4));
    else
        __builtin_memcpy(&d, s, 16);
    const uint32_t q0 = __builtin_amdgcn_alignbyte(d.y, d.x, 2);  // columns 2x-2 .. 2x+1
    const uint32_t q1 = __builtin_amdgcn_alignbyte(d.z, d.y, 2);  // columns 2x+2 .. 2x+5
    const uint32_t q2 = __builtin_amdgcn_alignbyte(d.w, d.z, 2);  // columns 2x+6 .. 2x+9
    hs[0] = __builtin_amdgcn_udot4(q0, kW, __builtin_amdgcn_udot4(q1, kE0, 0u, false), false);
    hs[1] = __builtin_amdgcn_udot4(d.y, kW, __builtin_amdgcn_udot4(q1, kE2, 0u, false), false);
    hs[2] = __builtin_amdgcn_udot4(q1, kW, __builtin_amdgcn_udot4(q2, kE0, 0u, false), false);
    hs[3] = __builtin_amdgcn_udot4(d.z, kW, __builtin_amdgcn_udot4(q2, kE2, 0u, false), false);
}

constexpr int kPyrRows = 4;  // output rows per thread of lk_pyr_down_kernel

// Output rows ry0 .. ry0+3 of the four adjacent outputs x .. x+3 as packed bytes: the eleven
// source rows 2 ry0 - 2 .. 2 ry0 + 8 are summed horizontally once each and go into up to three
// output rows.  Source rows no stored output row needs (beyond 2 (h-1) + 2) are not read.
template <bool kAligned>
__device__ __forceinline__ void pyr_down_block(const uint8_t* src, int SS, int x, int ry0, int h,
                                               uint32_t (&out)[kPyrRows]) {
    uint32_t col[kPyrRows][4];
#pragma unroll
    for (int j = 0; j < kPyrRows; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) col[j][i] = 128u;  // rounding of the final >> 8
    const uint8_t* s = src + (ptrdiff_t)(2 * ry0 - 2) * SS + (2 * x - 4);
    const int last = 2 * (h - 1) + 2 - (2 * ry0 - 2);  // index of the last source row in use
#pragma unroll
    for (int r = 0; r < 2 * kPyrRows + 3; ++r) {
        if (r > 4 && r > last) break;
        uint32_t hs[4];
        pyr_down_row<kAligned>(s + (ptrdiff_t)r * SS, hs);
#pragma unroll
        for (int j = 0; j < kPyrRows; ++j) {
            const int k = r - 2 * j;
            if (k < 0 || k > 4) continue;
            const uint32_t wk = k == 2 ? 6u : (k == 1 || k == 3) ? 4u : 1u;
#pragma unroll
            for (int i = 0; i < 4; ++i) col[j][i] += wk * hs[i];
        }
    }
#pragma unroll
    for (int j = 0; j < kPyrRows; ++j)
        out[j] = (col[j][0] >> 8) | ((col[j][1] >> 8) << 8) | ((col[j][2] >> 8) << 16) | ((col[j][3] >> 8) << 24);
}

// Level l (>= 1) of both images from level l-1, including its reflect-101 border.
// One thread computes 4 x 4 output pixels of the image (rows ry0 .. ry0+3 of one 4-pixel
// group of the padded row) and stores each row to its own place and to the border rows that
// mirror it; a group in the left or right border is the reversed group of the mirrored image
// columns.  A level is taller than the window (lk_num_levels), so a border row mirrors exactly
// one image row and every padded row is written once.
__global__ void lk_pyr_down_kernel(LkGeom g, int level, bool two_sets, uint8_t* __restrict__ pyr_prev,
                                   uint8_t* __restrict__ pyr_curr) {
    const int pair = two_sets ? blockIdx.z >> 1 : blockIdx.z, which = two_sets ? blockIdx.z & 1 : 0;
    uint8_t* base = (which ? pyr_curr : pyr_prev) + (size_t)pair * g.block;
    const uint8_t* src = base + g.img_off[level - 1];
    uint8_t* dst = base + g.img_off[level];
    const int SS = g.stride[level - 1];
    const int S = g.stride[level], w = g.w[level], h = g.h[level];
    const int per_row = S / 4, total = per_row * ((h + kPyrRows - 1) / kPyrRows);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int rg, xg;
        divmod_small(t, per_row, rg, xg);
        const int ry0 = rg * kPyrRows, x4 = xg * 4;
        const int x = x4 - g.bw;
        const int m = x + 4 <= 0 ? -x - 3 : 2 * (w - 1) - x - 3;  // first column of the mirrored group
        uint32_t v[kPyrRows];
        if (x >= 0 && x + 4 <= w) {
            pyr_down_block<true>(src, SS, x, ry0, h, v);
        } else if ((x + 4 <= 0 || (x >= w && x + 4 <= w + g.bw)) && m >= 0 && m + 4 <= w) {
            pyr_down_block<false>(src, SS, m, ry0, h, v);
#pragma unroll
            for (int j = 0; j < kPyrRows; ++j) v[j] = __builtin_bswap32(v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < kPyrRows; ++j) {
                v[j] = 0;
                if (ry0 + j >= h) continue;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xb = x + b;
                    const uint32_t px = xb < w + g.bw ? pyr_down_px(src, SS, reflect101(xb, w), ry0 + j) : 0u;
                    v[j] |= px << (8 * b);
                }
            }
        }
        uint8_t* dcol = dst - g.bw + x4;
#pragma unroll
        for (int j = 0; j < kPyrRows; ++j) {
            const int y = ry0 + j;
            if (y >= h) break;
            *reinterpret_cast<uint32_t*>(dcol + (ptrdiff_t)y * S) = v[j];
            if (y >= 1 && y <= g.bh) *reinterpret_cast<uint32_t*>(dcol - (ptrdiff_t)y * S) = v[j];
            const int yb = 2 * (h - 1) - y;  // the row below the image that mirrors y
            if (y <= h - 2 && yb <= h - 1 + g.bh) *reinterpret_cast<uint32_t*>(dcol + (ptrdiff_t)yb * S) = v[j];
        }
    }
}

// Scharr derivative of the previous image, all levels in one launch (blockIdx.y = level,
// blockIdx.z = pair); zero in the border.  One thread writes four adjacent pixels as one
// 16-byte store (a wave: 1 KiB of one row) from one unaligned 8-byte load of each of the three
// source rows.
// pairs (optional): pair p reads pyramid pairs[2 p] of n_pyramids (clamped into range).
__global__ void lk_scharr_kernel(LkGeom g, const uint8_t* __restrict__ pyr_prev, const int32_t* __restrict__ pairs,
                                 int n_pyramids, uint32_t* __restrict__ deriv) {
    const int level = blockIdx.y, pair = blockIdx.z;
    const int from = pairs ? min(max(pairs[2 * pair], 0), n_pyramids - 1) : pair;
    const uint8_t* src = pyr_prev + (size_t)from * g.block + g.img_off[level];
    uint32_t* dst = deriv + (size_t)pair * g.block + g.img_off[level];
    const int S = g.stride[level], w = g.w[level], h = g.h[level];
    const int per_row = S / 4, total = per_row * (h + 2 * g.bh);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int yy, xg;
        divmod_small(t, per_row, yy, xg);
        const int y = yy - g.bh, xi = xg * 4;
        const int x0 = xi - g.bw;  // image column of the first of the four pixels
        uint32_t* drow = dst + (ptrdiff_t)y * S - g.bw;
        uint4 out = make_uint4(0, 0, 0, 0);
        if (y >= 0 && y < h && x0 >= 0 && x0 < w) {
            // padded columns x0-1 .. x0+4 of rows y-1, y, y+1 (lkpyramid.cpp calcSharrDeriv:
            // vertical pass first, then the horizontal one)
            const uint8_t* r1 = src + (ptrdiff_t)y * S + x0 - 1;
            uint2 a, b, c;
            __builtin_memcpy(&a, r1 - S, 8);
            __builtin_memcpy(&b, r1, 8);
            __builtin_memcpy(&c, r1 + S, 8);
            int sv[6], dv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int sh = 8 * (k & 3);
                const int p0 = (int)(((k < 4 ? a.x : a.y) >> sh) & 0xFF);
                const int p1 = (int)(((k < 4 ? b.x : b.y) >> sh) & 0xFF);
                const int p2 = (int)(((k < 4 ? c.x : c.y) >> sh) & 0xFF);
                sv[k] = (p0 + p2) * 3 + p1 * 10;
                dv[k] = p2 - p0;
            }
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ix = sv[i + 2] - sv[i];
                const int iy = (dv[i + 2] + dv[i]) * 3 + dv[i + 1] * 10;
                // stored times four (|4 d| <= 16320 fits 16 bits): the tracker's CV_DESCALE(.., 14)
                // of an interpolated derivative is then the high half of the weighted sum
                o[i] = x0 + i < w ? (((uint32_t)(4 * ix) & 0xFFFFu) | ((uint32_t)(4 * iy) << 16)) : 0u;
            }
            out = make_uint4(o[0], o[1], o[2], o[3]);
        }
        uint32_t* dp = drow + xi;
        __builtin_nontemporal_store(out.x, dp);
        __builtin_nontemporal_store(out.y, dp + 1);
        __builtin_nontemporal_store(out.z, dp + 2);
        __builtin_nontemporal_store(out.w, dp + 3);
    }
}

// ---- tracker ------------------------------------------------------------------

struct LkArgs {
    LkGeom g;
    const uint8_t* pyr_prev;
    const uint8_t* pyr_curr;
    const uint32_t* deriv;
    const int32_t* pairs;   // optional: pair p tracks from pyramid pairs[2p] to pyramid pairs[2p+1]
    int n_pyramids;         // of pyr_prev == pyr_curr (pair list in use)
    const float2* prev_pts;
    int n;        // points per pair
    int n_pairs;
    int win_w, win_h, segs, slots;
    int max_count;
    double epsilon;  // already squared
    float min_eig;
    float max_error;
    float lscale[kMaxLevels];
    float2* next_pts;
    uint8_t* status;
    float* err;
};

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

// N independent wave sums advanced in lock step, so that consecutive DPP adds
// belong to different chains and need no wait states between them.
template <int N>
__device__ __forceinline__ void wave_sum_n(int (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x111, 0xF>(v[i]);  // row_shr:1
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x112, 0xF>(v[i]);  // row_shr:2
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x114, 0xF>(v[i]);  // row_shr:4
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x118, 0xF>(v[i]);  // row_shr:8: lane 15 of a row = row sum
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x142, 0xA>(v[i]);  // row_bcast:15 into rows 1, 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x143, 0xC>(v[i]);  // row_bcast:31 into rows 2, 3
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_readlane(v[i], 63);
}

// Exact 64-bit sums of N per-lane int32 partials (|partial| < 2^28).  Eight
// lanes are added in 32 bits (cannot overflow), then the low 16 bits and the
// high part are reduced separately.  EARLY = false splits before any add (for
// partials up to 2^31).
template <int N, bool EARLY>
__device__ __forceinline__ void wave_sum_wide_n(const int (&p)[N], long long (&out)[N]) {
    int v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i];
    if (EARLY) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = dpp_add<0x111, 0xF>(v[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = dpp_add<0x112, 0xF>(v[i]);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = dpp_add<0x114, 0xF>(v[i]);  // lanes 7, 15 of a row: 8-lane sums
    }
    int h[2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        h[2 * i] = v[i] & 0xFFFF;
        h[2 * i + 1] = v[i] >> 16;
    }
    if (EARLY) {
        // add the two 8-lane sums of each row (lane 7 into lane 15), then rows
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x118, 0xF>(h[i]);
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x142, 0xA>(h[i]);
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x143, 0xC>(h[i]);
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = __builtin_amdgcn_readlane(h[i], 63);
    } else {
        wave_sum_n<2 * N>(h);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = (long long)h[2 * i + 1] * 65536 + h[2 * i];
}

// Sums over groups of LANES (16: a DPP row, 8: half a row) adjacent lanes, result in every
// lane of the group: two quad permutes, then the half-row and (16 only) full-row mirrors.
template <int N, int LANES>
__device__ __forceinline__ void row_sum_n(int (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0xB1, 0xF>(v[i]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x4E, 0xF>(v[i]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dpp_add<0x141, 0xF>(v[i]);  // row_half_mirror
    if (LANES == 16) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = dpp_add<0x140, 0xF>(v[i]);  // row_mirror
    }
}

// Exact per-group sums of N int32 partials as (low 16 bits, high part) halves.  With 16 lanes
// the partials stay below 2^30: one add in 32 bits (two lanes cannot overflow), then the halves
// separately.  With 8 lanes a lane holds twice as many pixels (partials up to 2^31): the
// halves are split before any add.
template <int N, int LANES>
__device__ __forceinline__ void row_sum_wide_n(const int (&p)[N], int (&lo)[N], int (&hi)[N]) {
    int h[2 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int v = LANES == 16 ? dpp_add<0xB1, 0xF>(p[i]) : p[i];
        h[2 * i] = v & 0xFFFF;
        h[2 * i + 1] = v >> 16;
    }
    if (LANES != 16) {
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0xB1, 0xF>(h[i]);
    }
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x4E, 0xF>(h[i]);
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x141, 0xF>(h[i]);
    if (LANES == 16) {
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) h[i] = dpp_add<0x140, 0xF>(h[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lo[i] = h[2 * i];
        hi[i] = h[2 * i + 1];
    }
}

// (float)(hi * 65536 + lo) with one rounding, for |hi| < 2^24 and |lo| < 2^24: both
// conversions and the scaling by 2^16 are exact, so the only rounding is the final add.
__device__ __forceinline__ float halves_to_float(int hi, int lo) {
    return (float)hi * 65536.f + (float)lo;
}

__device__ __forceinline__ uint2 load_u8x8(const uint8_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// (a[J] | b[J] << 16): the u16 pair of two vertically adjacent pixels, column J of the
// eight bytes a = {a.x, a.y} (row y) and b (row y + 1).
template <int J>
__device__ __forceinline__ uint32_t col_pair(uint2 a, uint2 b) {
    constexpr uint32_t sel = (uint32_t)(J & 3) | (0x0Cu << 8) | ((uint32_t)(4 + (J & 3)) << 16) | (0x0Cu << 24);
    return __builtin_amdgcn_perm(J < 4 ? b.x : b.y, J < 4 ? a.x : a.y, sel);
}

// OpenCV's four fixed-point bilinear weights, paired by COLUMN of the 2x2 footprint: one
// pair multiplies the pixels (x, y), (x, y+1), the other (x+1, y), (x+1, y+1).  A column
// pair of pixels then serves two neighbouring output pixels (as the right column of one and
// the left column of the next), which halves the byte shuffles against pairing by row.
struct Weights {
    uint32_t wl;  // iw00 | iw10 << 16
    uint32_t wr;  // iw01 | iw11 << 16
};

__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    const int w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    const int w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    const int w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    const int w11 = (1 << W_BITS) - w00 - w01 - w10;
    Weights w;
    w.wl = (uint32_t)w00 | ((uint32_t)w10 << 16);
    w.wr = (uint32_t)w01 | ((uint32_t)w11 << 16);  // w11 may be -1: it sits in the high half
    return w;
}

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b), c,
                                  false);
}

// Signed 16-bit dot product plus a constant that travels in an SGPR.  (The compiler's
// choice for a constant accumulator is v_mov + v_dot2c: one more vector instruction per
// interpolated value.)
__device__ __forceinline__ int sdot2_k(uint32_t a, uint32_t b, int k) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

// Weighted sum of one 2x2 footprint from its two column pairs, rounding constant included.
// Signed: iw11 = 2^14 - (the three rounded weights) can be -1.
__device__ __forceinline__ int footprint(uint32_t left, uint32_t right, const Weights& w, int round) {
    return sdot2(right, w.wr, sdot2_k(left, w.wl, round));
}

// (lo16(a) | lo16(b) << 16): two 16-bit values side by side
__device__ __forceinline__ uint32_t pack16(int a, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x05040100u);
}

// (hi16(a) | hi16(b) << 16): the high halves of two 32-bit values side by side
__device__ __forceinline__ uint32_t pack_hi16(int a, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x07060302u);
}

__device__ __forceinline__ int mad24(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    const short2_t d = __builtin_bit_cast(short2_t, a) - __builtin_bit_cast(short2_t, b);
    return __builtin_bit_cast(uint32_t, d);
}

// One slot's window, 7 pixels as three 16-bit pairs (0,1) (2,3) (4,5) and a single (6).
struct SlotWin {
    uint32_t i01, i23, i45;  // interpolated I, 5 fractional bits
    uint32_t x01, x23, x45;  // interpolated dI/dx (0 outside the window)
    uint32_t y01, y23, y45;  // interpolated dI/dy
    int i6, x6, y6;
};

// The seventh pixels of two slots side by side (slot 2 t in the low half, 2 t + 1 in the high
// half): half the registers of three singles per slot, and one 16-bit dot product per sum
// instead of two multiply-adds.
struct SlotTail {
    uint32_t i, x, y;
};

// Which of a slot's 7 pixels lie inside the window: 0xFFFF per pixel.  FULL (the window width
// is a multiple of 7, e.g. 21): a slot is inside as a whole or not at all, and its mask is
// rebuilt from one comparison where it is needed instead of living in four registers.
template <bool FULL>
struct SlotMask;
template <>
struct SlotMask<false> {
    uint32_t m01, m23, m45;
    int m6;
    __device__ __forceinline__ void set(int npx) {
        m01 = (npx > 0 ? 0xFFFFu : 0u) | (npx > 1 ? 0xFFFF0000u : 0u);
        m23 = (npx > 2 ? 0xFFFFu : 0u) | (npx > 3 ? 0xFFFF0000u : 0u);
        m45 = (npx > 4 ? 0xFFFFu : 0u) | (npx > 5 ? 0xFFFF0000u : 0u);
        m6 = npx > 6 ? -1 : 0;
    }
    __device__ __forceinline__ uint32_t p01(bool) const { return m01; }
    __device__ __forceinline__ uint32_t p23(bool) const { return m23; }
    __device__ __forceinline__ uint32_t p45(bool) const { return m45; }
    __device__ __forceinline__ int p6(bool) const { return m6; }
};
template <>
struct SlotMask<true> {
    __device__ __forceinline__ void set(int) {}
    __device__ __forceinline__ uint32_t p01(bool valid) const { return valid ? 0xFFFFFFFFu : 0u; }
    __device__ __forceinline__ uint32_t p23(bool valid) const { return valid ? 0xFFFFFFFFu : 0u; }
    __device__ __forceinline__ uint32_t p45(bool valid) const { return valid ? 0xFFFFFFFFu : 0u; }
    __device__ __forceinline__ int p6(bool valid) const { return valid ? -1 : 0; }
};

// ((s0 >> 9) | (s1 >> 9) << 16) for two footprint sums in [1, 2^23): bytes 1..2 of each
// (a shift by 8 for free in the byte shuffle), then one packed shift by 1.
__device__ __forceinline__ uint32_t pack_shr9(int s0, int s1) {
    const uint32_t p = __builtin_amdgcn_perm((uint32_t)s1, (uint32_t)s0, 0x06050201u);
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2_t, p) >> (unsigned short)1);
}

// Interpolated window of one slot (I or J) with 5 fractional bits, CV_DESCALE(..., W_BITS-5),
// as three pairs and a single.  The sums are at least 1: the only negative weight is -1.
// The two rows sit at base + o and base + o + S; base is wave-uniform and both offsets are
// 32-bit, so the loads need no 64-bit address arithmetic per lane.
// RAW6: p6 is the seventh pixel's footprint sum before the shift (the caller packs two of them).
template <bool RAW6 = false>
__device__ __forceinline__ void window_interp(uint2 a, uint2 b, const Weights& wt,
                                              uint32_t& p01, uint32_t& p23, uint32_t& p45, int& p6) {
    const uint32_t c0 = col_pair<0>(a, b), c1 = col_pair<1>(a, b), c2 = col_pair<2>(a, b),
                   c3 = col_pair<3>(a, b), c4 = col_pair<4>(a, b), c5 = col_pair<5>(a, b),
                   c6 = col_pair<6>(a, b), c7 = col_pair<7>(a, b);
    constexpr int kRound = 1 << (W_BITS - 5 - 1);
    const int s0 = footprint(c0, c1, wt, kRound), s1 = footprint(c1, c2, wt, kRound);
    const int s2 = footprint(c2, c3, wt, kRound), s3 = footprint(c3, c4, wt, kRound);
    const int s4 = footprint(c4, c5, wt, kRound), s5 = footprint(c5, c6, wt, kRound);
    const int s6 = footprint(c6, c7, wt, kRound);
    p6 = RAW6 ? s6 : s6 >> (W_BITS - 5);
    p01 = pack_shr9(s0, s1);
    p23 = pack_shr9(s2, s3);
    p45 = pack_shr9(s4, s5);
}
template <bool RAW6 = false>
__device__ __forceinline__ void window_interp(const uint8_t* base, unsigned o, int S, const Weights& wt,
                                              uint32_t& p01, uint32_t& p23, uint32_t& p45, int& p6) {
    window_interp<RAW6>(load_u8x8(base + o), load_u8x8(base + (o + (unsigned)S)), wt, p01, p23, p45, p6);
}

// ---- derivative windows formed in the tracker (21 x 21 windows) -----------------------------
// Packed 16-bit helpers: two values per register, as pairs of horizontally adjacent columns.
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2_t, a) + __builtin_bit_cast(ushort2_t, b));
}
// a * ka + b * kb per half (all results fit 16 bits; signed and unsigned agree in the low 16 bits)
template <int KA, int KB>
__device__ __forceinline__ uint32_t pk_lin_u16(uint32_t a, uint32_t b) {
    const ushort2_t ka = {(unsigned short)KA, (unsigned short)KA}, kb = {(unsigned short)KB, (unsigned short)KB};
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2_t, a) * ka + __builtin_bit_cast(ushort2_t, b) * kb);
}
// bytes 2 W, 2 W + 1 of x as two 16-bit values
template <int W>
__device__ __forceinline__ uint32_t unpack_pair(uint32_t x) {
    return __builtin_amdgcn_perm(0u, x, W ? 0x0C030C02u : 0x0C010C00u);
}
// (hi16(lo), lo16(hi)): the pair one column to the right of `lo`
__device__ __forceinline__ uint32_t next_pair(uint32_t lo, uint32_t hi) {
    return __builtin_amdgcn_alignbyte(hi, lo, 2);
}

// The four bilinear weights paired by ROW of the 2x2 footprint (the derivative path holds its
// values as pairs of adjacent columns).
struct RowWeights {
    uint32_t top;  // iw00 | iw01 << 16
    uint32_t bot;  // iw10 | iw11 << 16
};
__device__ __forceinline__ RowWeights row_weights(const Weights& w) {
    RowWeights r;
    r.top = __builtin_amdgcn_perm(w.wr, w.wl, 0x05040100u);  // (lo16(wl), lo16(wr))
    r.bot = __builtin_amdgcn_perm(w.wr, w.wl, 0x07060302u);  // (hi16(wl), hi16(wr))
    return r;
}
__device__ __forceinline__ int footprint_rows(uint32_t top, uint32_t bot, const RowWeights& w, int round) {
    return sdot2(bot, w.bot, sdot2_k(top, w.top, round));
}

// LPP = lanes per point: 64 (one point per wave), 16 (four points per wave, one per DPP row; the
// per-point scalar work is then shared by four points per instruction) or, with FLY, 8 (eight
// points per wave: a lane then takes two of the blocks described below, see the kernel).
// VERT (21 x 21 windows on 16 lanes): a lane's four slots are the same seven columns of four
// consecutive window rows (lanes 0..14: rows 4 (lane / 3) .. + 3 of segment lane % 3), so the
// lower image row of one slot is the upper row of the next and five row loads serve four
// slots instead of eight; lane 15 takes the three slots of window row 20 and loads their
// upper rows by itself.
// FLY (21 x 21 windows, the default for them, on 8 lanes or 16): the vertical layout again, and
// no derivative image is read.  Scharr and the bilinear interpolation are both exact integer
// linear maps, so a lane forms the derivatives at the 5 x 8 integer positions its four slots
// touch from 7 rows of 10 pixels of the level itself (packed 16-bit arithmetic on pairs of
// adjacent columns), zeroes the positions outside the image, as the derivative image would be
// there, and interpolates them like loaded ones.  Blocks 0..14 (one per lane on 16 lanes) own
// window rows 0-3, 4-7, 9-12, 13-16, 17-20 of a segment; block 15 owns the three slots of window
// row 8, whose two derivative rows are the last one of the group above and the first one of the
// group below: they and the two pixel rows travel to it through LDS.
// WAVES: waves per workgroup (the launch's block size is 64 WAVES).
template <int LPP, int SPL, bool FULL, bool VERT = false, bool FLY = false, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, FLY ? 3 : 1) void lk_track_kernel(LkArgs a) {
    static_assert(!VERT || (FULL && ((LPP == 16 && SPL == 4) || (FLY && LPP == 8 && SPL == 8))),
                  "vertical groups: 21 x 21 windows as sixteen blocks of four slots on 16 or 8 lanes");
    static_assert(!FLY || VERT, "the derivative path builds on the vertical layout");
    const int lane = threadIdx.x & (LPP - 1);
    // Vertical layouts: the window is sixteen blocks of four slots, block b = lane + LPP k the
    // k-th of the lane (one per lane on 16 lanes, two on 8: the work per point that does not
    // depend on the window -- weights, sums over the lanes, the 2 x 2 solve -- is then shared by
    // eight points per instruction instead of four).  Block 15 is the one whose slots sit side by side.
    constexpr int NB = VERT ? SPL / 4 : 1;
    bool own_blk[NB];
    // FLY: first window row and segment of a block (block 15: a block whose loads stay inside
    // the level; what it computes from them is replaced)
    int fly_row0_blk[NB], fly_seg_blk[NB];
    auto block_geometry = [&](int lane_) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int b = lane_ + LPP * k;
            own_blk[k] = VERT && b == 15;
            fly_row0_blk[k] = own_blk[k] ? 8 : 4 * (b / 3) + (b >= 6 ? 1 : 0);
            fly_seg_blk[k] = own_blk[k] ? 0 : b % 3;
        }
    };
    block_geometry(lane);
    const bool own_rows = own_blk[0];   // (the layouts with one block per lane)
    __shared__ uint4 xfer[FLY ? (64 * WAVES / LPP) * 3 * 7 : 1];   // per point and segment: 7 x 16 bytes for block 15
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); give each XCD one
    // contiguous eighth of the points, so that an image pair's pyramids are fetched into
    // one L2 instead of all eight.
    const int per_xcd = gridDim.x >> 3;
    const int pblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    // A wave never mixes points of two image pairs: the pyramid base addresses are then
    // wave-uniform (scalar registers, 32-bit lane offsets).
    constexpr int kPerWave = 64 / LPP;
    const int waves_per_pair = (a.n + kPerWave - 1) / kPerWave;
    const int wave_id = __builtin_amdgcn_readfirstlane(pblock * WAVES + (int)(threadIdx.x >> 6));
    const int pair = wave_id / waves_per_pair;
    if (pair >= a.n_pairs) return;  // wave-uniform
    const int local_raw = (wave_id - pair * waves_per_pair) * kPerWave + (int)(threadIdx.x & 63) / LPP;
    const bool pt_live = local_raw < a.n;
    const int pt = pair * a.n + (pt_live ? local_raw : a.n - 1);  // surplus rows repeat the last point
    const LkGeom& g = a.g;
    int from = pair, to = pair;
    if (a.pairs) {   // wave-uniform; out-of-range entries are clamped rather than followed
        from = min(max(a.pairs[2 * pair], 0), a.n_pyramids - 1);
        to = min(max(a.pairs[2 * pair + 1], 0), a.n_pyramids - 1);
    }
    const uint8_t* pyrI = a.pyr_prev + (size_t)from * g.block;
    const uint8_t* pyrJ = a.pyr_curr + (size_t)to * g.block;
    const uint32_t* pyrD = a.deriv + (size_t)pair * g.block;

    // slot geometry of this lane: (window row << 8 | first window column) per slot (FLY: formed
    // from the blocks' rows and segments where it is needed)
    int sgeom[FLY ? 1 : SPL];
    // Two blocks per lane: the windows of the first block are parked in LDS while the second
    // block's are set up, and read back slot by slot in every iteration (nine 16-byte vectors
    // per lane, lane-major: no bank conflicts, no exchange between lanes).  The registers of a
    // lane then hold one block's windows at a time, and three waves fit a SIMD.
    constexpr bool PARK = FLY && NB == 2;
    __shared__ uint4 parked[PARK ? WAVES * 9 * 64 : 1];
    uint4* const park = parked + (threadIdx.x >> 6) * (9 * 64) + (threadIdx.x & 63);
    SlotWin win[PARK ? 4 : SPL];
    SlotTail tail[FLY ? SPL / 2 : 1];   // FLY: the slots' seventh pixels, in pairs
    // slot s of the lane: from the registers, or (parked) from LDS; y45 of the four parked slots
    // travel together
    auto slot_window = [&](int s, const uint4& parked_y45) -> SlotWin {
        if constexpr (PARK) {
            if (s < 4) {
                const uint4 a0 = park[(2 * s) * 64], a1 = park[(2 * s + 1) * 64];
                SlotWin w;
                w.i01 = a0.x; w.i23 = a0.y; w.i45 = a0.z; w.x01 = a0.w;
                w.x23 = a1.x; w.x45 = a1.y; w.y01 = a1.z; w.y23 = a1.w;
                w.y45 = s == 0 ? parked_y45.x : s == 1 ? parked_y45.y : s == 2 ? parked_y45.z : parked_y45.w;
                return w;
            }
            return win[s - 4];
        } else {
            return win[s];
        }
    };
    SlotMask<FULL> msk[SPL];
#pragma unroll
    for (int s = 0; s < (FLY ? 0 : SPL); ++s) {
        if (VERT) {
            // lane 15's fourth slot does not exist: it repeats the third one's place (loads stay
            // inside the level) and counts as invalid below
            const int srow = own_rows ? 20 : 4 * (lane / 3) + s;
            const int sx0 = (own_rows ? min(s, 2) : lane % 3) * kPxPerSlot;
            sgeom[s] = (srow << 8) | sx0;
            continue;
        }
        const int slot = lane + LPP * s;
        const bool valid = slot < a.slots;
        const int sl = valid ? slot : 0;
        const int srow = sl / a.segs, sx0 = (sl % a.segs) * kPxPerSlot;
        sgeom[s] = (srow << 8) | sx0;
        msk[s].set(valid ? min(kPxPerSlot, a.win_w - sx0) : 0);
    }
    auto slot_valid = [&](int s) {
        return FLY ? !(own_blk[s / 4] && s % 4 == 1) : VERT ? !(own_rows && s == 3) : lane + LPP * s < a.slots;
    };
    // The rows of all slots of a lane from one image: up[s] / lo[s] are the upper and lower row
    // of slot s.  VERT: five loads by every lane (the first slot's upper row and every slot's
    // lower row) and two more by lane 15 alone.
    auto load_rows8 = [&](const uint8_t* base, unsigned o0, const unsigned (&soff)[SPL], int S, uint2 (&up)[SPL],
                          uint2 (&lo)[SPL]) {
        if constexpr (!VERT) {
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                up[s] = load_u8x8(base + (o0 + soff[s]));
                lo[s] = load_u8x8(base + (o0 + soff[s] + (unsigned)S));
            }
            return;
        } else if constexpr (FLY) {   // block 15: slots 0, 2, 3 side by side; its slot 1 holds rows of no meaning
            uint2 x2 = make_uint2(0, 0), x3 = x2;
            constexpr int kl = NB - 1;   // block 15 is the last block of its lane
            if (own_blk[kl]) {
                x2 = load_u8x8(base + (o0 + soff[4 * kl + 2]));
                x3 = load_u8x8(base + (o0 + soff[4 * kl + 3]));
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) up[4 * k] = load_u8x8(base + (o0 + soff[4 * k]));
#pragma unroll
            for (int s = 0; s < SPL; ++s) lo[s] = load_u8x8(base + (o0 + soff[s] + (unsigned)S));
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                up[4 * k + 1] = lo[4 * k];
                up[4 * k + 2] = k == kl && own_blk[kl] ? x2 : lo[4 * k + 1];
                up[4 * k + 3] = k == kl && own_blk[kl] ? x3 : lo[4 * k + 2];
            }
        } else {
            // lane 15's own loads first, into registers of their own: all seven loads are then in
            // flight together (a load that overwrote a copy of lo[] would have to wait for it)
            uint2 x1 = make_uint2(0, 0), x2 = x1;
            if (own_rows) {
                x1 = load_u8x8(base + (o0 + soff[1]));
                x2 = load_u8x8(base + (o0 + soff[2]));
            }
            up[0] = load_u8x8(base + (o0 + soff[0]));
#pragma unroll
            for (int s = 0; s < SPL; ++s) lo[s] = load_u8x8(base + (o0 + soff[s] + (unsigned)S));
            up[1] = own_rows ? x1 : lo[0];
            up[2] = own_rows ? x2 : lo[1];
            up[3] = lo[2];
        }
    };

    const float2 p0 = a.prev_pts[pt];
    const float half_x = (a.win_w - 1) * 0.5f, half_y = (a.win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    float out_x = 0.f, out_y = 0.f;
    int status = 1;
    float err = 0.f;

    for (int level = g.levels; level >= 0; --level) {
        const int lw = g.w[level], lh = g.h[level], S = g.stride[level];
        if constexpr (NB > 1) {
            // formed afresh at every level (a handful of instructions): kept from level to level, the
            // blocks' rows, columns and what the compiler derives from them take registers that the
            // iterations need
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            block_geometry(lane_);
        }
        // first element of the padded level: all window offsets below are >= 0
        const size_t pad0 = g.img_off[level] - (size_t)g.bh * S - g.bw;
        const uint8_t* I = pyrI + pad0;
        const uint8_t* J = pyrJ + pad0;
        const uint32_t* D = pyrD + pad0;

        float prev_x = p0.x * a.lscale[level], prev_y = p0.y * a.lscale[level];
        float next_x, next_y;
        if (level == g.levels) {
            next_x = prev_x;
            next_y = prev_y;
        } else {
            next_x = out_x * 2.f;
            next_y = out_y * 2.f;
        }
        out_x = next_x;
        out_y = next_y;

        prev_x -= half_x;
        prev_y -= half_y;
        const int ipx = (int)floorf(prev_x), ipy = (int)floorf(prev_y);
        if (ipx < -a.win_w || ipx >= lw || ipy < -a.win_h || ipy >= lh) {
            if (level == 0) {
                status = 0;
                err = 0.f;
            }
            continue;
        }
        Weights wt = bilinear_weights(prev_x - ipx, prev_y - ipy);
        // offset of each slot inside a window at this level's row pitch
        unsigned soff[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            if constexpr (FLY) {
                // a block's slots sit one below the other; block 15: slots 0, 2, 3 are the segments
                // of window row 8, slot 1 does not exist (it repeats slot 0's place and counts as
                // invalid).  Written so that only a block's first row and column are kept from one
                // level to the next.
                const int k = s / 4, j = s % 4;
                const int below = fly_row0_blk[k] * S + fly_seg_blk[k] * kPxPerSlot + j * S;
                const int beside = 8 * S + (j > 1 ? j - 1 : 0) * kPxPerSlot;
                soff[s] = (unsigned)(own_blk[k] ? beside : below);
            } else {
                soff[s] = (unsigned)((sgeom[s] >> 8) * S + (sgeom[s] & 0xFF));
            }
        }

        int pA[3] = {0, 0, 0};
        const unsigned obase = (unsigned)((ipy + g.bh) * S + ipx + g.bw);
        uint2 iu[SPL], il[SPL];
        uint4 du[SPL][2], dl[SPL][2];   // VERT: the derivative rows of the lane's slots, loaded once
        if constexpr (FLY) {
            const RowWeights rw = row_weights(wt);
            uint4* xq = xfer + ((threadIdx.x / LPP) * 3) * 7;   // this point's three segments
#pragma unroll
            for (int k = 0; k < NB; ++k) {
            const int blk = lane + LPP * k;
            const bool own = own_blk[k];
            const int fly_row0 = fly_row0_blk[k], fly_seg = fly_seg_blk[k];
            // 7 rows (window rows fly_row0 - 1 .. + 5) of the 10 pixels from one column left of the
            // segment, as aligned 16-byte loads (a padded level starts 16-byte aligned and its row
            // pitch is a multiple of 16); the byte shuffle that widens two pixels to 16 bits each
            // picks them at the lane's own byte offset.
            const uint8_t* Ia = I - S - 4;
            const unsigned ofb = obase + (unsigned)(fly_row0 * S + fly_seg * kPxPerSlot) + 3u;
            const unsigned oal = ofb & ~3u, sh = (ofb & 3u) * 0x00010001u;
            // selectors for the bytes (sh + 2k, sh + 2k + 1) of the 16 loaded, from the two dwords that hold them
            const uint32_t sel0 = 0x0C010C00u + sh, sel1 = 0x0C030C02u + sh, sel2 = 0x0C010C00u + sh,
                           sel3 = 0x0C030C02u + sh, sel4 = 0x0C010C00u + sh;
            uint32_t hp[7][5];   // pixel pairs (columns 2k, 2k+1 of the 10 in use)
            {
                uint4 raw[7];
#pragma unroll
                for (int r = 0; r < 7; ++r)
                    raw[r] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(Ia + (oal + (unsigned)(r * S)), 4));
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    hp[r][0] = __builtin_amdgcn_perm(raw[r].y, raw[r].x, sel0);
                    hp[r][1] = __builtin_amdgcn_perm(raw[r].y, raw[r].x, sel1);
                    hp[r][2] = __builtin_amdgcn_perm(raw[r].z, raw[r].y, sel2);
                    hp[r][3] = __builtin_amdgcn_perm(raw[r].z, raw[r].y, sel3);
                    hp[r][4] = __builtin_amdgcn_perm(raw[r].w, raw[r].z, sel4);
                }
            }
            // Four times the Scharr derivatives (lkpyramid.cpp calcSharrDeriv) at the block's
            // 5 rows x 8 columns, as pairs of columns (2m, 2m+1): derivative row i sits at window
            // row fly_row0 + i, derivative column j at column 7 seg + j, which is column j + 1 of hp.
            uint32_t dx[5][4], dy[5][4];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                uint32_t sv[5], dv[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    sv[c] = pk_lin_u16<12, 40>(pk_add_u16(hp[i][c], hp[i + 2][c]), hp[i + 1][c]);   // 4 (3, 10, 3)
                    dv[c] = pk_sub_i16(hp[i + 2][c], hp[i][c]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    dx[i][m] = pk_sub_i16(sv[m + 1], sv[m]);
                    dy[i][m] = pk_lin_u16<12, 40>(pk_add_u16(dv[m], dv[m + 1]), next_pair(dv[m], dv[m + 1]));
                }
            }
            // the derivative image is zero outside the image: windows that reach beyond it
            if (__any(ipx < 0 || ipx + a.win_w + 1 > lw || ipy < 0 || ipy + a.win_h + 1 > lh)) {
                const int cx0 = ipx + fly_seg * kPxPerSlot, cy0 = ipy + fly_row0;
                uint32_t cm[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    cm[m] = ((unsigned)(cx0 + 2 * m) < (unsigned)lw ? 0xFFFFu : 0u) |
                            ((unsigned)(cx0 + 2 * m + 1) < (unsigned)lw ? 0xFFFF0000u : 0u);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const uint32_t rm = (unsigned)(cy0 + i) < (unsigned)lh ? 0xFFFFFFFFu : 0u;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        dx[i][m] &= cm[m] & rm;
                        dy[i][m] &= cm[m] & rm;
                    }
                }
            }
            // Window row 8 for block 15: its upper derivative and pixel rows from the blocks of
            // rows 4-7 (their row 4 / rows 5, 6 of hp), its lower derivative row from the blocks of
            // rows 9-12 (their row 0).  A wave's LDS operations complete in order, every point has
            // its own lanes and its own part of the array, and block 15 is the last one in program
            // order: no barrier.
            if (blk >= 3 && blk < 6) {
                uint4* q = xq + fly_seg * 7;
                q[0] = make_uint4(dx[4][0], dx[4][1], dx[4][2], dx[4][3]);
                q[1] = make_uint4(dy[4][0], dy[4][1], dy[4][2], dy[4][3]);
                q[4] = make_uint4(hp[5][0], hp[5][1], hp[5][2], hp[5][3]);
                q[5] = make_uint4(hp[6][0], hp[6][1], hp[6][2], hp[6][3]);
                q[6] = make_uint4(hp[5][4], hp[6][4], 0u, 0u);
            }
            if (blk >= 6 && blk < 9) {
                uint4* q = xq + fly_seg * 7;
                q[2] = make_uint4(dx[0][0], dx[0][1], dx[0][2], dx[0][3]);
                q[3] = make_uint4(dy[0][0], dy[0][1], dy[0][2], dy[0][3]);
            }
            // Slot j reads derivative rows j, j + 1 and pixel rows j + 1, j + 2.  Block 15's slots
            // 0 and 2 find their rows in place of rows 0 .. 3 (1 .. 4 of the pixels); the upper rows
            // of its slot 3 replace row 3 (4) once slot 2 has used it, see below.
            if (k == NB - 1 && own) {
                auto put4 = [](uint32_t (&d)[4], const uint4& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
                auto put5 = [](uint32_t (&d)[5], const uint4& v, uint32_t e) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = e; };
                const uint4 t0 = xq[6], t1 = xq[7 + 6], t2 = xq[14 + 6];
                put4(dx[0], xq[0]);      put4(dy[0], xq[1]);      put4(dx[1], xq[2]);      put4(dy[1], xq[3]);
                put5(hp[1], xq[4], t0.x);                         put5(hp[2], xq[5], t0.y);
                put4(dx[2], xq[7 + 0]);  put4(dy[2], xq[7 + 1]);  put4(dx[3], xq[7 + 2]);  put4(dy[3], xq[7 + 3]);
                put5(hp[3], xq[7 + 4], t1.x);                     put5(hp[4], xq[7 + 5], t1.y);
                put4(dx[4], xq[14 + 2]); put4(dy[4], xq[14 + 3]);
                put5(hp[5], xq[14 + 5], t2.y);
            }
            int i6e = 0, x6e = 0, y6e = 0;   // an even slot's seventh sums wait for its odd partner
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k == NB - 1 && j == 3) {
                    // block 15, slot 3: its upper rows (segment 2) take the place of the rows that
                    // slot 2 has just used -- a select per register, every lane reads
                    const uint4 vx = xq[14 + 0], vy = xq[14 + 1], vh = xq[14 + 4], vt = xq[14 + 6];
                    const uint32_t nx[4] = {vx.x, vx.y, vx.z, vx.w}, ny[4] = {vy.x, vy.y, vy.z, vy.w};
                    const uint32_t nh[5] = {vh.x, vh.y, vh.z, vh.w, vt.x};
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        dx[3][m] = own ? nx[m] : dx[3][m];
                        dy[3][m] = own ? ny[m] : dy[3][m];
                    }
#pragma unroll
                    for (int c = 0; c < 5; ++c) hp[4][c] = own ? nh[c] : hp[4][c];
                }
                SlotWin& w = win[PARK ? j : 4 * k + j];
                const uint32_t(&ht)[5] = hp[j + 1];
                const uint32_t(&hb)[5] = hp[j + 2];
                const uint32_t(&xt)[4] = dx[j];
                const uint32_t(&yt)[4] = dy[j];
                const uint32_t(&xb)[4] = dx[j + 1];
                const uint32_t(&yb)[4] = dy[j + 1];
                // window pixel p has its footprint at columns p + 1, p + 2 of hp
                constexpr int kRound = 1 << (W_BITS - 5 - 1);
                int si[kPxPerSlot];
#pragma unroll
                for (int p = 0; p < kPxPerSlot; ++p) {
                    const int c = (p + 1) >> 1;
                    const uint32_t top = (p & 1) ? ht[c] : next_pair(ht[c], ht[c + 1]);
                    const uint32_t bot = (p & 1) ? hb[c] : next_pair(hb[c], hb[c + 1]);
                    si[p] = footprint_rows(top, bot, rw, kRound);
                }
                w.i01 = pack_shr9(si[0], si[1]);
                w.i23 = pack_shr9(si[2], si[3]);
                w.i45 = pack_shr9(si[4], si[5]);
                // derivative of window pixel p from derivative columns p, p + 1
                int sx[kPxPerSlot], sy[kPxPerSlot];
#pragma unroll
                for (int p = 0; p < kPxPerSlot; ++p) {
                    const int m = p >> 1;
                    const uint32_t xtp = (p & 1) ? next_pair(xt[m], xt[m + 1]) : xt[m];
                    const uint32_t xbp = (p & 1) ? next_pair(xb[m], xb[m + 1]) : xb[m];
                    const uint32_t ytp = (p & 1) ? next_pair(yt[m], yt[m + 1]) : yt[m];
                    const uint32_t ybp = (p & 1) ? next_pair(yb[m], yb[m + 1]) : yb[m];
                    sx[p] = footprint_rows(xtp, xbp, rw, 1 << (W_BITS + 1));
                    sy[p] = footprint_rows(ytp, ybp, rw, 1 << (W_BITS + 1));
                }
                w.x01 = pack_hi16(sx[0], sx[1]);
                w.x23 = pack_hi16(sx[2], sx[3]);
                w.x45 = pack_hi16(sx[4], sx[5]);
                w.y01 = pack_hi16(sy[0], sy[1]);
                w.y23 = pack_hi16(sy[2], sy[3]);
                w.y45 = pack_hi16(sy[4], sy[5]);
                if (j == 1 && k == NB - 1) {   // block 15's slot 1 does not exist: no derivatives, no contribution
                    const uint32_t keep = own ? 0u : 0xFFFFFFFFu;
                    w.x01 &= keep;
                    w.x23 &= keep;
                    w.x45 &= keep;
                    w.y01 &= keep;
                    w.y23 &= keep;
                    w.y45 &= keep;
                }
                pA[0] = sdot2(w.x45, w.x45, sdot2(w.x23, w.x23, sdot2(w.x01, w.x01, pA[0])));
                pA[1] = sdot2(w.x45, w.y45, sdot2(w.x23, w.y23, sdot2(w.x01, w.y01, pA[1])));
                pA[2] = sdot2(w.y45, w.y45, sdot2(w.y23, w.y23, sdot2(w.y01, w.y01, pA[2])));
                if (j & 1) {
                    SlotTail& tl = tail[2 * k + j / 2];
                    tl.i = pack_shr9(i6e, si[6]);
                    tl.x = pack_hi16(x6e, sx[6]);
                    tl.y = pack_hi16(y6e, sy[6]);
                    if (j == 1 && k == NB - 1) {   // (block 15's slot 1 again: the high halves)
                        const uint32_t keep = own ? 0x0000FFFFu : 0xFFFFFFFFu;
                        tl.x &= keep;
                        tl.y &= keep;
                    }
                    pA[0] = sdot2(tl.x, tl.x, pA[0]);
                    pA[1] = sdot2(tl.x, tl.y, pA[1]);
                    pA[2] = sdot2(tl.y, tl.y, pA[2]);
                } else {
                    i6e = si[6];
                    x6e = sx[6];
                    y6e = sy[6];
                }
            }
            if constexpr (PARK) {
                if (k == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        park[(2 * j) * 64] = make_uint4(win[j].i01, win[j].i23, win[j].i45, win[j].x01);
                        park[(2 * j + 1) * 64] = make_uint4(win[j].x23, win[j].x45, win[j].y01, win[j].y23);
                    }
                    park[8 * 64] = make_uint4(win[0].y45, win[1].y45, win[2].y45, win[3].y45);
                    // (the next block's loads and shuffles stay behind this block's arithmetic)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }   // blocks of the lane
        } else {
        if constexpr (VERT) {
            load_rows8(I, obase, soff, S, iu, il);
            auto load_d = [&](unsigned o, uint4 (&r)[2]) { __builtin_memcpy(r, D + o, 32); };
            // an invalid slot reads the zero border at the start of the padded level
            uint4 dx1[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, dx2[2] = {dx1[0], dx1[0]};
            if (own_rows) {
                load_d(obase + soff[1], dx1);
                load_d(obase + soff[2], dx2);
            }
            load_d(obase + soff[0], du[0]);
#pragma unroll
            for (int s = 0; s < SPL; ++s) load_d(slot_valid(s) ? obase + soff[s] + (unsigned)S : 0u, dl[s]);
            auto pick = [&](const uint4& mine, const uint4& shared) {
                return make_uint4(own_rows ? mine.x : shared.x, own_rows ? mine.y : shared.y,
                                  own_rows ? mine.z : shared.z, own_rows ? mine.w : shared.w);
            };
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                du[1][h] = pick(dx1[h], dl[0][h]);
                du[2][h] = pick(dx2[h], dl[1][h]);
                du[3][h] = pick(make_uint4(0, 0, 0, 0), dl[2][h]);   // lane 15's fourth slot: zero derivatives
            }
        }
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            SlotWin& w = win[s];
            const bool valid = slot_valid(s);
            const uint32_t m01 = msk[s].p01(valid), m23 = msk[s].p23(valid), m45 = msk[s].p45(valid);
            const int m6 = msk[s].p6(valid);
            const unsigned o = obase + soff[s];
            uint32_t i01, i23, i45;
            int i6;
            if (VERT)
                window_interp(iu[s], il[s], wt, i01, i23, i45, i6);
            else
                window_interp(I, o, S, wt, i01, i23, i45, i6);
            if (FULL) {  // a slot outside the window contributes through its zero derivatives only
                w.i01 = i01;
                w.i23 = i23;
                w.i45 = i45;
                w.i6 = i6;
            } else {
                w.i01 = i01 & m01;
                w.i23 = i23 & m23;
                w.i45 = i45 & m45;
                w.i6 = i6 & m6;
            }
            // a slot outside a FULL window reads the zero border at the start of the padded level
            // (at least 3 rows and 16 columns) instead of being masked afterwards
            const unsigned od = FULL ? (valid ? o : 0u) : o;
            uint32_t d0[8], d1[8];
            if (VERT) {
                __builtin_memcpy(d0, du[s], 32);
                __builtin_memcpy(d1, dl[s], 32);
            } else {
                __builtin_memcpy(d0, D + od, 32);
                __builtin_memcpy(d1, D + (od + (unsigned)S), 32);
            }
            // column pairs (row y, row y+1) of Ix and of Iy, then CV_DESCALE(..., W_BITS)
            uint32_t xp[8], yp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xp[j] = __builtin_amdgcn_perm(d1[j], d0[j], 0x05040100u);
                yp[j] = __builtin_amdgcn_perm(d1[j], d0[j], 0x07060302u);
            }
            // CV_DESCALE(sum, W_BITS) of the interpolated derivatives: the stored values are four
            // times the derivatives, so (4 sum + 2^15) >> 16 is the same number and sits in the
            // high half of the 32-bit sum, where the byte shuffle that pairs two pixels finds it
            int sx[kPxPerSlot], sy[kPxPerSlot];
#pragma unroll
            for (int j = 0; j < kPxPerSlot; ++j) {
                sx[j] = footprint(xp[j], xp[j + 1], wt, 1 << (W_BITS + 1));
                sy[j] = footprint(yp[j], yp[j + 1], wt, 1 << (W_BITS + 1));
            }
            w.x01 = pack_hi16(sx[0], sx[1]);
            w.x23 = pack_hi16(sx[2], sx[3]);
            w.x45 = pack_hi16(sx[4], sx[5]);
            w.x6 = sx[6] >> 16;
            w.y01 = pack_hi16(sy[0], sy[1]);
            w.y23 = pack_hi16(sy[2], sy[3]);
            w.y45 = pack_hi16(sy[4], sy[5]);
            w.y6 = sy[6] >> 16;
            if (!FULL) {
                w.x01 &= m01;
                w.x23 &= m23;
                w.x45 &= m45;
                w.x6 &= m6;
                w.y01 &= m01;
                w.y23 &= m23;
                w.y45 &= m45;
                w.y6 &= m6;
            }
            pA[0] = mad24(w.x6, w.x6, sdot2(w.x45, w.x45, sdot2(w.x23, w.x23, sdot2(w.x01, w.x01, pA[0]))));
            pA[1] = mad24(w.x6, w.y6, sdot2(w.x45, w.y45, sdot2(w.x23, w.y23, sdot2(w.x01, w.y01, pA[1]))));
            pA[2] = mad24(w.y6, w.y6, sdot2(w.y45, w.y45, sdot2(w.y23, w.y23, sdot2(w.y01, w.y01, pA[2]))));
        }
        }   // !FLY
        float A11, A12, A22;
        if (LPP == 64) {
            long long sA[3];
            wave_sum_wide_n<3, SPL == 1>(pA, sA);
            A11 = (float)sA[0] * FLT_SCALE;
            A12 = (float)sA[1] * FLT_SCALE;
            A22 = (float)sA[2] * FLT_SCALE;
        } else {
            int lo[3], hi[3];
            row_sum_wide_n<3, LPP>(pA, lo, hi);
            A11 = halves_to_float(hi[0], lo[0]) * FLT_SCALE;
            A12 = halves_to_float(hi[1], lo[1]) * FLT_SCALE;
            A22 = halves_to_float(hi[2], lo[2]) * FLT_SCALE;
        }

        float D_ = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * a.win_w * a.win_h);
        if (minEig < a.min_eig || D_ < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D_ = 1.f / D_;

        next_x -= half_x;
        next_y -= half_y;
        float prev_dx = 0.f, prev_dy = 0.f;
        // The two rows of bytes every slot reads from J, kept from one iteration to the next:
        // once the steps are below a pixel the window's integer position mostly stays where it
        // was, and the loads (what this kernel waits for most) are issued only for the points
        // whose position changed.
        uint2 ja[SPL], jb[SPL];
        int held_x = INT_MIN, held_y = INT_MIN;

        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(next_x), iny = (int)floorf(next_y);
            if (inx < -a.win_w || inx >= lw || iny < -a.win_h || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            wt = bilinear_weights(next_x - inx, next_y - iny);
            int pb[2] = {0, 0};
            const unsigned jbase = (unsigned)((iny + g.bh) * S + inx + g.bw);
            if (inx != held_x || iny != held_y) {
                load_rows8(J, jbase, soff, S, ja, jb);
                held_x = inx;
                held_y = iny;
            }
            int j6even = 0;   // FLY: an even slot's seventh sum waits for its odd partner
            uint4 parked_y45 = make_uint4(0, 0, 0, 0);
            if constexpr (PARK) parked_y45 = park[8 * 64];
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const SlotWin w = slot_window(s, parked_y45);
                uint32_t j01, j23, j45;
                int j6;
                window_interp<FLY>(ja[s], jb[s], wt, j01, j23, j45, j6);
                // mismatch J - I; the derivative pairs are 0 outside the window
                const uint32_t d01 = pk_sub_i16(j01, w.i01), d23 = pk_sub_i16(j23, w.i23),
                               d45 = pk_sub_i16(j45, w.i45);
                if constexpr (FLY) {
                    pb[0] = sdot2(d45, w.x45, sdot2(d23, w.x23, sdot2(d01, w.x01, pb[0])));
                    pb[1] = sdot2(d45, w.y45, sdot2(d23, w.y23, sdot2(d01, w.y01, pb[1])));
                    if (s & 1) {
                        const SlotTail& tl = tail[s / 2];
                        const uint32_t d6 = pk_sub_i16(pack_shr9(j6even, j6), tl.i);
                        pb[0] = sdot2(d6, tl.x, pb[0]);
                        pb[1] = sdot2(d6, tl.y, pb[1]);
                    } else {
                        j6even = j6;
                    }
                } else {
                    const int d6 = j6 - w.i6;
                    pb[0] = mad24(d6, w.x6, sdot2(d45, w.x45, sdot2(d23, w.x23, sdot2(d01, w.x01, pb[0]))));
                    pb[1] = mad24(d6, w.y6, sdot2(d45, w.y45, sdot2(d23, w.y23, sdot2(d01, w.y01, pb[1]))));
                }
            }
            float b1, b2;
            if (LPP == 64) {
                long long sb[2];
                wave_sum_wide_n<2, SPL == 1>(pb, sb);
                b1 = (float)sb[0] * FLT_SCALE;
                b2 = (float)sb[1] * FLT_SCALE;
            } else {
                int lo[2], hi[2];
                row_sum_wide_n<2, LPP>(pb, lo, hi);
                b1 = halves_to_float(hi[0], lo[0]) * FLT_SCALE;
                b2 = halves_to_float(hi[1], lo[1]) * FLT_SCALE;
            }
            const float dx = (A12 * b2 - A22 * b1) * D_;
            const float dy = (A12 * b1 - A11 * b2) * D_;
            next_x += dx;
            next_y += dy;
            out_x = next_x + half_x;
            out_y = next_y + half_y;
            if ((double)dx * dx + (double)dy * dy <= a.epsilon) break;
            if (j > 0 && fabs((double)(dx + prev_dx)) < 0.01 && fabs((double)(dy + prev_dy)) < 0.01) {
                out_x -= dx * 0.5f;
                out_y -= dy * 0.5f;
                break;
            }
            prev_dx = dx;
            prev_dy = dy;
        }

        if (status && level == 0) {
            const float ex = out_x - half_x, ey = out_y - half_y;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -a.win_w || inx >= lw || iny < -a.win_h || iny >= lh) {
                status = 0;
            } else {
                wt = bilinear_weights(ex - inx, ey - iny);
                int perr[1] = {0};
                const unsigned ebase = (unsigned)((iny + g.bh) * S + inx + g.bw);
                uint4 parked_y45 = make_uint4(0, 0, 0, 0);
                if constexpr (PARK) parked_y45 = park[8 * 64];
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    const SlotWin w = slot_window(s, parked_y45);
                    const bool valid = slot_valid(s);
                    const uint32_t m01 = msk[s].p01(valid), m23 = msk[s].p23(valid), m45 = msk[s].p45(valid);
                    const int m6 = msk[s].p6(valid);
                    const unsigned o = ebase + soff[s];
                    uint32_t j01, j23, j45;
                    int j6;
                    window_interp(J, o, S, wt, j01, j23, j45, j6);
                    const uint32_t dd[3] = {pk_sub_i16(j01, w.i01) & m01, pk_sub_i16(j23, w.i23) & m23,
                                            pk_sub_i16(j45, w.i45) & m45};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int lo = (int)(short)(dd[k] & 0xFFFF), hi = (int)dd[k] >> 16;
                        perr[0] += abs(lo) + abs(hi);
                    }
                    if constexpr (FLY) {   // the seventh pixel of I sits in its pair's half
                        const uint32_t ip = tail[s / 2].i;
                        const int i6 = (s & 1) ? (int)(ip >> 16) : (int)(ip & 0xFFFFu);
                        perr[0] += abs((j6 - i6) & m6);
                    } else {
                        perr[0] += abs((j6 - w.i6) & m6);
                    }
                }
                // Sum of |diff| stays below 2^24 for windows up to 41x41, so the float
                // accumulation of the reference is exact and equals this integer sum.
                if (LPP == 64)
                    wave_sum_n<1>(perr);
                else
                    row_sum_n<1, LPP>(perr);
                const float errval = (float)perr[0];
                err = errval * 1.f / (float)(32 * a.win_w * a.win_h);
            }
        }
    }

    if (err > a.max_error) status = 0;  // feature_processing.cpp:245-248
    if (lane == 0 && pt_live) {
        a.next_pts[pt] = make_float2(out_x, out_y);
        a.status[pt] = (uint8_t)status;
        if (a.err) a.err[pt] = err;
    }
}

int lk_num_levels(int w, int h, int win_w, int win_h, int max_level) {
    int sw = w, sh = h;
    for (int level = 0; level <= max_level; ++level) {
        sw = (sw + 1) / 2;
        sh = (sh + 1) / 2;
        if (sw <= win_w || sh <= win_h) return level;
    }
    return max_level;
}

int make_geom(int width, int height, int win_w, int win_h, int max_level, LkGeom* g) {
    if (max_level > kMaxLevels - 1) max_level = kMaxLevels - 1;
    g->levels = lk_num_levels(width, height, win_w, win_h, max_level);
    // Left border rounded up to 16 columns: every 16-byte (level 0) or 4-pixel (other levels)
    // group of a padded row then lies entirely in the border or entirely in the image, and a
    // border group is the byte-reversed copy of an image group.
    g->bw = (int)align_up((size_t)win_w, 16);
    g->bh = win_h;
    size_t off = 0;
    int w = width, h = height;
    for (int l = 0; l <= g->levels; ++l) {
        g->w[l] = w;
        g->h[l] = h;
        // 8 extra columns: a slot's 8-element loads may run past the window
        g->stride[l] = (int)align_up((size_t)w + g->bw + win_w + 8, 16);
        // every padded row starts 16-byte aligned (off is a multiple of 64).  One row of slack
        // above and below the padded level: the tracker's derivative path reads a ring of one
        // pixel around the window's footprint (what it reads there never reaches a result).
        g->img_off[l] = off + (size_t)(win_h + 1) * g->stride[l] + (size_t)g->bw;
        off = align_up(g->img_off[l] - g->bw + (size_t)(h + win_h + 1) * g->stride[l] + 64, 64);
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    g->block = align_up(off + 64, 64);
    return SFM_OK;
}

}  // namespace

}  // namespace sfm

using namespace sfm;

namespace {

struct LkBuffers {
    uint8_t* pyr_prev;
    uint8_t* pyr_curr;
    uint32_t* deriv;
};

// 21 x 21 windows: the tracker forms the derivatives itself (lk_track_kernel<16, 4, true, true,
// true>) and no derivative image exists.  SFM_LK_DERIV_IMAGE=1 (a measurement aid) selects the
// kernel that reads one.
bool lk_needs_deriv_image(int win_w, int win_h) {
    static const bool forced = [] {
        const char* e = getenv("SFM_LK_DERIV_IMAGE");
        return e && e[0] == '1';
    }();
    return forced || win_w != 21 || win_h != 21;
}

size_t lk_carve(Carver& c, const LkGeom& g, int n_pairs, bool deriv, LkBuffers* b) {
    b->pyr_prev = c.take<uint8_t>(g.block * n_pairs);
    b->pyr_curr = c.take<uint8_t>(g.block * n_pairs);
    b->deriv = deriv ? c.take<uint32_t>(g.block * n_pairs) : nullptr;
    return c.total();
}

int check_lk_args(sfm_ctx* ctx, int width, int height, size_t stride, int n, int win_w, int win_h,
                  int max_level) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (width <= 0 || height <= 0 || stride < (size_t)width)
        return set_error(ctx, SFM_E_INVALID, "bad image size or stride");
    if (n < 0) return set_error(ctx, SFM_E_INVALID, "negative point count");
    // cv::SparsePyrLKOpticalFlow::calc asserts maxLevel >= 0 and a window above 2x2
    if (max_level < 0 || win_w <= 2 || win_h <= 2)
        return set_error(ctx, SFM_E_INVALID, "maxLevel must be >= 0 and winSize above 2x2");
    const int segs = (win_w + kPxPerSlot - 1) / kPxPerSlot;
    if (win_w > 41 || win_h > 41 || segs * win_h > 256)
        return set_error(ctx, SFM_E_UNSUPPORTED, "windows above 41x41 are not supported");
    return SFM_OK;
}

// The filter the reference applies between computeFlow and findCameraPose
// (ProcesingAdds::filterPointsByStatusMask then filterPointsByBoundary, Source Files/
// feature_processing.cpp:332-358, called at Source Files/app_solver.cpp:69-70): keep the
// correspondences whose status is set and whose current point lies inside the boundary
// rectangle, both lists in step and in order.  One workgroup per image pair.
__global__ __launch_bounds__(1024) void lk_filter_kernel(const float2* __restrict__ prev, const float2* __restrict__ curr,
                                                         const uint8_t* __restrict__ status,
                                                         const int32_t* __restrict__ valid, int n, float bx, float by,
                                                         float bw, float bh, float2* __restrict__ prev_out,
                                                         float2* __restrict__ curr_out, int32_t* __restrict__ counts) {
    __shared__ int wave_count[16];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)p * n;
    const int n_in = valid ? min(max(valid[p], 0), n) : n;
    int running = 0;
    for (int i0 = 0; i0 < n_in; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        float2 a = make_float2(0.f, 0.f), b = a;
        if (i < n_in) {
            a = prev[base + i];
            b = curr[base + i];
            ok = status[base + i] != 0;
            // cv::Rect_<int>::contains(Point2f): x <= pt.x < x + width, y <= pt.y < y + height
            if (bw > 0.f && bh > 0.f) ok = ok && bx <= b.x && b.x < bx + bw && by <= b.y && b.y < by + bh;
        }
        const unsigned long long ballot = __ballot(ok);
        const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(ballot);
        __syncthreads();
        int offset = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wave_count[w];
            offset += w < wave ? c : 0;
            total += c;
        }
        if (ok) {
            prev_out[base + running + offset + prefix] = a;
            curr_out[base + running + offset + prefix] = b;
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) counts[p] = running;
}

}  // namespace

extern "C" {

int sfm_filter_tracked_dev(sfm_ctx* ctx, const float* d_prev_pts, const float* d_curr_pts, const uint8_t* d_status,
                           const int32_t* d_valid, int n_pairs, int n, int bound_x, int bound_y, int bound_w,
                           int bound_h, float* d_prev_out, float* d_curr_out, int32_t* d_counts) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_pairs < 0 || n < 0) return set_error(ctx, SFM_E_INVALID, "negative count");
    if (n_pairs == 0) return SFM_OK;
    if (!d_prev_pts || !d_curr_pts || !d_status || !d_prev_out || !d_curr_out || !d_counts)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (d_prev_out == d_prev_pts || d_curr_out == d_curr_pts)
        return set_error(ctx, SFM_E_INVALID, "the filter does not work in place");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(lk_filter_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_prev_pts), reinterpret_cast<const float2*>(d_curr_pts), d_status,
                       d_valid, n, (float)bound_x, (float)bound_y, (float)bound_w, (float)bound_h,
                       reinterpret_cast<float2*>(d_prev_out), reinterpret_cast<float2*>(d_curr_out), d_counts);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // extern "C"

namespace {

// Pads level 0 and builds the other levels for `n` images per set (one set, or a previous and a
// next set of n images each) into pyramid blocks of g.block bytes.
int lk_build_levels(sfm_ctx* ctx, const LkGeom& g, const uint8_t* d_prev, const uint8_t* d_curr, int n,
                    size_t image_stride, size_t prev_stride, size_t curr_stride, bool two_sets, uint8_t* pyr_prev,
                    uint8_t* pyr_curr) {
    const dim3 blk(256);
    const unsigned nz = (unsigned)(two_sets ? 2 * n : n);
    {
        const int S = g.stride[0];
        const dim3 grid(((S / 16) * (g.h[0] + 2 * g.bh) + 255) / 256, 1, nz);
        KernelTimer timer(ctx, KID_LK_PAD_LEVEL0);
        hipLaunchKernelGGL(lk_pad_level0_kernel, grid, blk, 0, ctx->stream, d_prev, d_curr, image_stride,
                           prev_stride, curr_stride, two_sets, g, pyr_prev, pyr_curr);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    for (int l = 1; l <= g.levels; ++l) {
        const int S = g.stride[l];
        const dim3 grid(((S / 4) * ((g.h[l] + kPyrRows - 1) / kPyrRows) + 255) / 256, 1, nz);
        KernelTimer timer(ctx, KID_LK_PYR_DOWN);
        hipLaunchKernelGGL(lk_pyr_down_kernel, grid, blk, 0, ctx->stream, g, l, two_sets, pyr_prev, pyr_curr);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    return SFM_OK;
}

// SparsePyrLKOpticalFlowImpl::calc: criteria clamping
void lk_clamp_criteria(int crit_type, int* max_count, double* epsilon) {
    if ((crit_type & SFM_TERM_COUNT) == 0)
        *max_count = 30;
    else
        *max_count = *max_count < 0 ? 0 : (*max_count > 100 ? 100 : *max_count);
    if ((crit_type & SFM_TERM_EPS) == 0)
        *epsilon = 0.01;
    else
        *epsilon = *epsilon < 0. ? 0. : (*epsilon > 10. ? 10. : *epsilon);
    *epsilon *= *epsilon;
}

// The derivative image (window sizes whose tracker reads one) and the tracker launch over
// pyramids that exist: pair p tracks from block p of pyr_prev to block p of pyr_curr, or, with a
// pair list, from block d_pairs[2p] to block d_pairs[2p+1] of the one array both pointers name.
int lk_track_pyramids(sfm_ctx* ctx, const LkGeom& g, const uint8_t* pyr_prev, const uint8_t* pyr_curr,
                      const int32_t* d_pairs, int n_pyramids, uint32_t* deriv, int n_pairs, const float* d_prev_pts,
                      int n, int win_w, int win_h, int max_count, double epsilon_sq, double min_eig_threshold,
                      float max_error, float* d_curr_pts, uint8_t* d_status, float* d_err) {
    const dim3 blk(256);
    const bool deriv_image = deriv != nullptr;
    if (deriv_image) {
        const int S = g.stride[0];
        // four groups per thread at level 0: workgroups that find nothing to do at the smaller
        // levels are then a small part of the launch
        const dim3 grid(((S / 4) * (g.h[0] + 2 * g.bh) + 1023) / 1024, g.levels + 1, n_pairs);
        KernelTimer timer(ctx, KID_LK_SCHARR);
        hipLaunchKernelGGL(lk_scharr_kernel, grid, blk, 0, ctx->stream, g, pyr_prev, d_pairs, n_pyramids, deriv);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    LkArgs a{};
    a.g = g;
    a.pyr_prev = pyr_prev;
    a.pyr_curr = pyr_curr;
    a.deriv = deriv;
    a.pairs = d_pairs;
    a.n_pyramids = n_pyramids;
    a.prev_pts = reinterpret_cast<const float2*>(d_prev_pts);
    a.n = n;
    a.n_pairs = n_pairs;
    a.win_w = win_w;
    a.win_h = win_h;
    a.segs = (win_w + kPxPerSlot - 1) / kPxPerSlot;
    a.slots = a.segs * win_h;
    a.max_count = max_count;
    a.epsilon = epsilon_sq;
    a.min_eig = (float)min_eig_threshold;
    a.max_error = max_error;
    for (int l = 0; l <= g.levels; ++l) a.lscale[l] = (float)(1. / (1 << l));
    a.next_pts = reinterpret_cast<float2*>(d_curr_pts);
    a.status = d_status;
    a.err = d_err;
    KernelTimer timer(ctx, KID_LK_TRACK);
    static const int lanes_wanted = [] {   // measurement aid: lanes per point (8 or 16)
        const char* e = getenv("SFM_LK_LANES");
        return e ? atoi(e) : 0;
    }();
    const bool eight_lanes = lanes_wanted == 8;
    if (!deriv_image && lanes_wanted != 16) {
        // 21 x 21 windows, derivatives formed in the tracker: eight lanes per point, two blocks of
        // four slots each; four waves per workgroup, eight points per wave
        const long waves = (long)n_pairs * ((n + 7) / 8);
        static const int wg_waves = [] {   // measurement aid: waves per workgroup (1, 2 or 4)
            const char* e = getenv("SFM_LK_WG_WAVES");
            return e ? atoi(e) : 1;
        }();
        // (one wave per workgroup, 866 / 858 / 845 us with four / two / one: a workgroup takes its
        // place on a SIMD as soon as one wave's registers are free, and the launch's last round of
        // workgroups leaves the chip partly empty for a shorter time)
        if (wg_waves == 4) {
            const dim3 grid((unsigned)((((waves + 3) / 4 + 7) / 8) * 8));
            hipLaunchKernelGGL((lk_track_kernel<8, 8, true, true, true, 4>), grid, dim3(256), 0, ctx->stream, a);
        } else if (wg_waves == 1) {
            const dim3 grid((unsigned)(((waves + 7) / 8) * 8));
            hipLaunchKernelGGL((lk_track_kernel<8, 8, true, true, true, 1>), grid, dim3(64), 0, ctx->stream, a);
        } else {
            const dim3 grid((unsigned)((((waves + 1) / 2 + 7) / 8) * 8));
            hipLaunchKernelGGL((lk_track_kernel<8, 8, true, true, true, 2>), grid, dim3(128), 0, ctx->stream, a);
        }
    } else if (a.slots <= 64 && a.slots > 56 && eight_lanes && win_w % kPxPerSlot == 0 && deriv_image) {
        const long waves = (long)n_pairs * ((n + 7) / 8);
        const dim3 grid((unsigned)((((waves + 3) / 4 + 7) / 8) * 8));
        hipLaunchKernelGGL((lk_track_kernel<8, 8, true>), grid, blk, 0, ctx->stream, a);
    } else if (a.slots <= 64) {
        // four waves per workgroup, four points per wave (16 lanes each); a multiple of 8
        // workgroups (see the kernel)
        const long waves = (long)n_pairs * ((n + 3) / 4);
        const dim3 grid((unsigned)((((waves + 3) / 4 + 7) / 8) * 8));
        const bool full = win_w % kPxPerSlot == 0;
#define SFM_LK_LAUNCH(LPP_, SPL_)                                                                     \
    do {                                                                                              \
        if (full)                                                                                     \
            hipLaunchKernelGGL((lk_track_kernel<LPP_, SPL_, true>), grid, blk, 0, ctx->stream, a);    \
        else                                                                                          \
            hipLaunchKernelGGL((lk_track_kernel<LPP_, SPL_, false>), grid, blk, 0, ctx->stream, a);   \
    } while (0)
        static const bool no_vert = getenv("SFM_LK_NO_VERT") != nullptr;   // measurement aid
        if (!deriv_image) {
            hipLaunchKernelGGL((lk_track_kernel<16, 4, true, true, true>), grid, blk, 0, ctx->stream, a);
        } else if (a.win_w == 21 && a.win_h == 21 && !no_vert) {
            hipLaunchKernelGGL((lk_track_kernel<16, 4, true, true>), grid, blk, 0, ctx->stream, a);
        } else
        switch ((a.slots + 15) / 16) {
            case 1: SFM_LK_LAUNCH(16, 1); break;
            case 2: SFM_LK_LAUNCH(16, 2); break;
            case 3: SFM_LK_LAUNCH(16, 3); break;
            default: SFM_LK_LAUNCH(16, 4); break;
        }
    } else {
        const long waves = (long)n_pairs * n;  // one point per wave
        const dim3 grid((unsigned)((((waves + 3) / 4 + 7) / 8) * 8));
        const bool full = win_w % kPxPerSlot == 0;
        switch ((a.slots + 63) / 64) {
            case 2: SFM_LK_LAUNCH(64, 2); break;
            case 3: SFM_LK_LAUNCH(64, 3); break;
            default: SFM_LK_LAUNCH(64, 4); break;
        }
#undef SFM_LK_LAUNCH
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // namespace

extern "C" {

size_t sfm_pyrlk_workspace(int n_pairs, int width, int height, int win_w, int win_h,
                           int max_level) {
    if (n_pairs <= 0 || width <= 0 || height <= 0 || win_w <= 2 || win_h <= 2 || max_level < 0)
        return 0;
    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    Carver c(nullptr);
    LkBuffers b;
    return lk_carve(c, g, n_pairs, lk_needs_deriv_image(win_w, win_h), &b);
}

int sfm_pyrlk_batch_dev(sfm_ctx* ctx, const uint8_t* d_prev, const uint8_t* d_curr, int n_pairs,
                        size_t image_stride, int width, int height, size_t stride,
                        const float* d_prev_pts, int n, int win_w, int win_h, int max_level,
                        int crit_type, int max_count, double epsilon, double min_eig_threshold,
                        float max_error, float* d_curr_pts, uint8_t* d_status, float* d_err) {
    SFM_RETURN_IF(check_lk_args(ctx, width, height, stride, n, win_w, win_h, max_level));
    if (n_pairs < 0) return set_error(ctx, SFM_E_INVALID, "negative pair count");
    if (n == 0 || n_pairs == 0) return SFM_OK;
    if (!d_prev || !d_curr || !d_prev_pts || !d_curr_pts || !d_status)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (n_pairs > 1 && image_stride < (size_t)height * stride)
        return set_error(ctx, SFM_E_INVALID, "image stride smaller than an image");
    if ((long long)n * n_pairs > 0x7fffffffLL / 4 || 2 * (long long)n_pairs > 65535)
        return set_error(ctx, SFM_E_INVALID, "batch too large");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    lk_clamp_criteria(crit_type, &max_count, &epsilon);

    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    Carver sz(nullptr);
    LkBuffers bufs;
    const bool deriv_image = lk_needs_deriv_image(win_w, win_h);
    const size_t need = lk_carve(sz, g, n_pairs, deriv_image, &bufs);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, need, &ws));
    Carver c(ws);
    lk_carve(c, g, n_pairs, deriv_image, &bufs);
    SFM_RETURN_IF(lk_build_levels(ctx, g, d_prev, d_curr, n_pairs, image_stride, stride, stride, true,
                                  bufs.pyr_prev, bufs.pyr_curr));
    return lk_track_pyramids(ctx, g, bufs.pyr_prev, bufs.pyr_curr, nullptr, n_pairs, bufs.deriv, n_pairs, d_prev_pts,
                             n, win_w, win_h, max_count, epsilon, min_eig_threshold, max_error, d_curr_pts, d_status,
                             d_err);
}

size_t sfm_lk_pyramid_bytes(int width, int height, int win_w, int win_h, int max_level) {
    if (width <= 0 || height <= 0 || win_w <= 2 || win_h <= 2 || max_level < 0 || win_w > 41 || win_h > 41) return 0;
    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    return g.block;
}

int sfm_lk_build_pyramids_dev(sfm_ctx* ctx, const uint8_t* d_frames, int n_frames, size_t image_stride, int width,
                              int height, size_t stride, int win_w, int win_h, int max_level, uint8_t* d_pyramids) {
    SFM_RETURN_IF(check_lk_args(ctx, width, height, stride, 0, win_w, win_h, max_level));
    if (n_frames < 0) return set_error(ctx, SFM_E_INVALID, "negative frame count");
    if (n_frames == 0) return SFM_OK;
    if (!d_frames || !d_pyramids) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (n_frames > 1 && image_stride < (size_t)height * stride)
        return set_error(ctx, SFM_E_INVALID, "image stride smaller than an image");
    if (n_frames > 65535) return set_error(ctx, SFM_E_INVALID, "batch too large");
    if (reinterpret_cast<uintptr_t>(d_pyramids) & 15)
        return set_error(ctx, SFM_E_INVALID, "the pyramid array must be 16-byte aligned");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    return lk_build_levels(ctx, g, d_frames, d_frames, n_frames, image_stride, stride, stride, false, d_pyramids,
                           d_pyramids);
}

size_t sfm_pyrlk_pyramids_workspace(int n_pairs, int width, int height, int win_w, int win_h, int max_level) {
    if (n_pairs <= 0 || width <= 0 || height <= 0 || win_w <= 2 || win_h <= 2 || max_level < 0) return 0;
    if (!lk_needs_deriv_image(win_w, win_h)) return 0;
    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    Carver c(nullptr);
    c.take<uint32_t>(g.block * n_pairs);
    return c.total();
}

int sfm_pyrlk_pyramids_dev(sfm_ctx* ctx, const uint8_t* d_pyramids, int n_pyramids, const int32_t* d_pairs,
                           int n_pairs, int width, int height, const float* d_prev_pts, int n, int win_w, int win_h,
                           int max_level, int crit_type, int max_count, double epsilon, double min_eig_threshold,
                           float max_error, float* d_curr_pts, uint8_t* d_status, float* d_err) {
    SFM_RETURN_IF(check_lk_args(ctx, width, height, (size_t)width, n, win_w, win_h, max_level));
    if (n_pairs < 0 || n_pyramids < 0) return set_error(ctx, SFM_E_INVALID, "negative count");
    if (n == 0 || n_pairs == 0) return SFM_OK;
    if (!d_pyramids || !d_pairs || !d_prev_pts || !d_curr_pts || !d_status)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (n_pyramids == 0) return set_error(ctx, SFM_E_INVALID, "pairs but no pyramids");
    if ((long long)n * n_pairs > 0x7fffffffLL / 4 || n_pairs > 65535)
        return set_error(ctx, SFM_E_INVALID, "batch too large");
    if (reinterpret_cast<uintptr_t>(d_pyramids) & 15)
        return set_error(ctx, SFM_E_INVALID, "the pyramid array must be 16-byte aligned");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    lk_clamp_criteria(crit_type, &max_count, &epsilon);
    LkGeom g;
    make_geom(width, height, win_w, win_h, max_level, &g);
    uint32_t* deriv = nullptr;
    if (lk_needs_deriv_image(win_w, win_h)) {
        Carver sz(nullptr);
        sz.take<uint32_t>(g.block * n_pairs);
        void* ws;
        SFM_RETURN_IF(workspace(ctx, sz.total(), &ws));
        Carver c(ws);
        deriv = c.take<uint32_t>(g.block * n_pairs);
    }
    return lk_track_pyramids(ctx, g, d_pyramids, d_pyramids, d_pairs, n_pyramids, deriv, n_pairs, d_prev_pts, n,
                             win_w, win_h, max_count, epsilon, min_eig_threshold, max_error, d_curr_pts, d_status,
                             d_err);
}

int sfm_pyrlk(sfm_ctx* ctx, const uint8_t* prev, const uint8_t* curr, int width, int height,
              size_t prev_stride, size_t curr_stride, const float* prev_pts, int n, int win_w, int win_h,
              int max_level, int crit_type, int max_count, double epsilon, double min_eig_threshold,
              float max_error, float* curr_pts, uint8_t* status, float* err) {
    SFM_RETURN_IF(check_lk_args(ctx, width, height, prev_stride, n, win_w, win_h, max_level));
    if (curr_stride < (size_t)width) return set_error(ctx, SFM_E_INVALID, "bad image size or stride");
    if (n == 0) return SFM_OK;
    if (!prev || !curr || !prev_pts || !curr_pts || !status)
        return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t img_bytes = (size_t)height * width;
    Carver sz(nullptr);
    sz.take<uint8_t>(img_bytes);
    sz.take<uint8_t>(img_bytes);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>(2 * (size_t)n);
    sz.take<float>((size_t)n);
    sz.take<uint8_t>((size_t)n);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver c(st);
    uint8_t* dp = c.take<uint8_t>(img_bytes);
    uint8_t* dc = c.take<uint8_t>(img_bytes);
    float* dpts = c.take<float>(2 * (size_t)n);
    float* dout = c.take<float>(2 * (size_t)n);
    float* derr = c.take<float>((size_t)n);
    uint8_t* dst = c.take<uint8_t>((size_t)n);
    // each image is gathered with its own row pitch (one may be a region of a larger image)
    SFM_RETURN_IF(upload_rows(ctx, dp, prev, prev_stride, (size_t)width, (size_t)height));
    SFM_RETURN_IF(upload_rows(ctx, dc, curr, curr_stride, (size_t)width, (size_t)height));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(dpts, prev_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice,
                                      ctx->stream));
    SFM_RETURN_IF(sfm_pyrlk_batch_dev(ctx, dp, dc, 1, img_bytes, width, height, (size_t)width, dpts,
                                      n, win_w, win_h, max_level, crit_type, max_count, epsilon,
                                      min_eig_threshold, max_error, dout, dst, derr));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(curr_pts, dout, sizeof(float) * 2 * n, hipMemcpyDeviceToHost,
                                      ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(status, dst, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (err)
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(err, derr, sizeof(float) * n, hipMemcpyDeviceToHost,
                                          ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pinned_used = 0;
    return SFM_OK;
}

}  // extern "C"
