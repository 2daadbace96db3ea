// Pyramidal Lucas-Kanade sparse optical flow for gfx950.
//
// Replaces cv::SparsePyrLKOpticalFlow::calc as called from OptFlow::computeFlow
// (Source Files/feature_processing.cpp:241) plus the error gate at :245-248.
// The arithmetic follows OpenCV's lkpyramid.cpp / pyramids.cpp for 8-bit
// single-channel images with flags == 0:
//   - pyramid: pyrDown ([1 4 6 4 1] separable, (v+128)>>8, reflect-101), every
//     level stored with a reflect-101 border of one window;
//   - Scharr derivative of the previous image per level, stored as one 32-bit
//     word per pixel (Ix in the low half, Iy in the high half), zero outside;
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
#include <cmath>

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
    int bw, bh;                 // border = window size
};

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

// Level 0 of both images of every pair: copy with a reflect-101 border.
// grid.z = 2*n_pairs (even: previous image, odd: next image).
__global__ void lk_pad_level0_kernel(const uint8_t* __restrict__ prev,
                                     const uint8_t* __restrict__ curr, size_t image_stride,
                                     size_t row_stride, LkGeom g, uint8_t* __restrict__ pyr_prev,
                                     uint8_t* __restrict__ pyr_curr) {
    const int pair = blockIdx.z >> 1, which = blockIdx.z & 1;
    const uint8_t* src = (which ? curr : prev) + (size_t)pair * image_stride;
    uint8_t* dst = (which ? pyr_curr : pyr_prev) + (size_t)pair * g.block + g.img_off[0];
    const int S = g.stride[0], w = g.w[0], h = g.h[0];
    const int y = (int)blockIdx.y - g.bh;
    if (y >= h + g.bh) return;
    const int sy = reflect101(y, h);
    const uint8_t* srow = src + (size_t)sy * row_stride;
    uint8_t* drow = dst + (ptrdiff_t)y * S - g.bw;
    for (int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4; x4 < S; x4 += gridDim.x * blockDim.x * 4) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = x4 + b - g.bw;
            const uint32_t px = x < w + g.bw ? srow[reflect101(x, w)] : 0u;
            v |= px << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(drow + x4) = v;
    }
}

// Level l (>= 1) of both images from level l-1, including its reflect-101 border.
__global__ void lk_pyr_down_kernel(LkGeom g, int level, uint8_t* __restrict__ pyr_prev,
                                   uint8_t* __restrict__ pyr_curr) {
    const int pair = blockIdx.z >> 1, which = blockIdx.z & 1;
    uint8_t* base = (which ? pyr_curr : pyr_prev) + (size_t)pair * g.block;
    const uint8_t* src = base + g.img_off[level - 1];
    uint8_t* dst = base + g.img_off[level];
    const int SS = g.stride[level - 1];
    const int S = g.stride[level], w = g.w[level], h = g.h[level];
    const int y = (int)blockIdx.y - g.bh;
    if (y >= h + g.bh) return;
    const int ry = reflect101(y, h);
    uint8_t* drow = dst + (ptrdiff_t)y * S - g.bw;
    for (int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4; x4 < S; x4 += gridDim.x * blockDim.x * 4) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = x4 + b - g.bw;
            uint32_t px = 0;
            if (x < w + g.bw) {
                const int rx = reflect101(x, w);
                // The source border already holds the reflect-101 values pyrDown asks for.
                const uint8_t* s = src + (ptrdiff_t)(2 * ry - 2) * SS + (2 * rx - 2);
                int acc = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const uint8_t* r = s + (ptrdiff_t)k * SS;
                    const int hsum = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
                    acc += hsum * (k == 2 ? 6 : (k == 1 || k == 3) ? 4 : 1);
                }
                px = (uint32_t)((acc + 128) >> 8);
            }
            v |= px << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(drow + x4) = v;
    }
}

// Scharr derivative of the previous image at one level; zero in the border.
__global__ void lk_scharr_kernel(LkGeom g, int level, const uint8_t* __restrict__ pyr_prev,
                                 uint32_t* __restrict__ deriv) {
    const int pair = blockIdx.z;
    const uint8_t* src = pyr_prev + (size_t)pair * g.block + g.img_off[level];
    uint32_t* dst = deriv + (size_t)pair * g.block + g.img_off[level];
    const int S = g.stride[level], w = g.w[level], h = g.h[level];
    const int y = (int)blockIdx.y - g.bh;
    if (y >= h + g.bh) return;
    uint32_t* drow = dst + (ptrdiff_t)y * S - g.bw;
    const bool yin = y >= 0 && y < h;
    for (int xi = blockIdx.x * blockDim.x + threadIdx.x; xi < S; xi += gridDim.x * blockDim.x) {
        const int x = xi - g.bw;
        uint32_t out = 0;
        if (yin && x >= 0 && x < w) {
            const uint8_t* r0 = src + (ptrdiff_t)(y - 1) * S + x;
            const uint8_t* r1 = r0 + S;
            const uint8_t* r2 = r1 + S;
            // vertical pass at columns x-1, x, x+1 (lkpyramid.cpp calcSharrDeriv)
            const int s_m = (r0[-1] + r2[-1]) * 3 + r1[-1] * 10, d_m = r2[-1] - r0[-1];
            const int d_0 = r2[0] - r0[0];
            const int s_p = (r0[1] + r2[1]) * 3 + r1[1] * 10, d_p = r2[1] - r0[1];
            const int ix = s_p - s_m;
            const int iy = (d_p + d_m) * 3 + d_0 * 10;
            out = ((uint32_t)ix & 0xFFFFu) | ((uint32_t)iy << 16);
        }
        drow[xi] = out;
    }
}

// ---- tracker ------------------------------------------------------------------

struct LkArgs {
    LkGeom g;
    const uint8_t* pyr_prev;
    const uint8_t* pyr_curr;
    const uint32_t* deriv;
    const float2* prev_pts;
    int n;        // points per pair
    int n_total;  // n * n_pairs
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

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}

// Sum over the 64 lanes, returned in every lane.
__device__ __forceinline__ int wave_sum(int v) {
    v = dpp_add<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF>(v);  // row_shr:8  -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __builtin_amdgcn_readlane(v, 63);
}

// Exact sum of per-lane int32 partials whose total may exceed 32 bits.
__device__ __forceinline__ long long wave_sum_wide(int v) {
    const int lo = wave_sum(v & 0xFFFF);
    const int hi = wave_sum(v >> 16);
    return (long long)hi * 65536 + lo;
}

__device__ __forceinline__ uint2 load_u8x8(const uint8_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ void unpack8(uint2 v, int (&b)[8]) {
    b[0] = v.x & 0xFF;
    b[1] = (v.x >> 8) & 0xFF;
    b[2] = (v.x >> 16) & 0xFF;
    b[3] = v.x >> 24;
    b[4] = v.y & 0xFF;
    b[5] = (v.y >> 8) & 0xFF;
    b[6] = (v.y >> 16) & 0xFF;
    b[7] = v.y >> 24;
}

struct Weights {
    int w00, w01, w10, w11;
};

__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    Weights w;
    w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    w.w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

// Interpolated window pixel, 5 fractional bits: CV_DESCALE(..., W_BITS-5).
__device__ __forceinline__ int interp_px(int p00, int p01, int p10, int p11, const Weights& w) {
    return (p00 * w.w00 + p01 * w.w01 + p10 * w.w10 + p11 * w.w11 + (1 << (W_BITS - 5 - 1))) >>
           (W_BITS - 5);
}

template <int SPL>
__global__ __launch_bounds__(256) void lk_track_kernel(LkArgs a) {
    const int lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pt >= a.n_total) return;  // wave-uniform
    const int pair = pt / a.n;
    const LkGeom& g = a.g;
    const uint8_t* pyrI = a.pyr_prev + (size_t)pair * g.block;
    const uint8_t* pyrJ = a.pyr_curr + (size_t)pair * g.block;
    const uint32_t* pyrD = a.deriv + (size_t)pair * g.block;

    // slot geometry of this lane
    int srow[SPL], sx0[SPL], snpx[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int slot = lane + 64 * s;
        const bool valid = slot < a.slots;
        const int sl = valid ? slot : 0;
        srow[s] = sl / a.segs;
        sx0[s] = (sl % a.segs) * kPxPerSlot;
        snpx[s] = valid ? min(kPxPerSlot, a.win_w - sx0[s]) : 0;
    }

    const float2 p0 = a.prev_pts[pt];
    const float half_x = (a.win_w - 1) * 0.5f, half_y = (a.win_h - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    float out_x = 0.f, out_y = 0.f;
    int status = 1;
    float err = 0.f;

    int Iw[SPL][kPxPerSlot];       // interpolated I, 5 fractional bits
    uint32_t dIw[SPL][kPxPerSlot]; // interpolated (Ix | Iy << 16)

    for (int level = g.levels; level >= 0; --level) {
        const int lw = g.w[level], lh = g.h[level], S = g.stride[level];
        const uint8_t* I = pyrI + g.img_off[level];
        const uint8_t* J = pyrJ + g.img_off[level];
        const uint32_t* D = pyrD + g.img_off[level];

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

        int pA11 = 0, pA12 = 0, pA22 = 0;
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            const ptrdiff_t o = (ptrdiff_t)(ipy + srow[s]) * S + ipx + sx0[s];
            int r0[8], r1[8];
            unpack8(load_u8x8(I + o), r0);
            unpack8(load_u8x8(I + o + S), r1);
            uint32_t d0[8], d1[8];
            {
                uint32_t tmp[8];
                __builtin_memcpy(tmp, D + o, 32);
#pragma unroll
                for (int j = 0; j < 8; ++j) d0[j] = tmp[j];
                __builtin_memcpy(tmp, D + o + S, 32);
#pragma unroll
                for (int j = 0; j < 8; ++j) d1[j] = tmp[j];
            }
#pragma unroll
            for (int j = 0; j < kPxPerSlot; ++j) {
                const bool on = j < snpx[s];
                const int ival = interp_px(r0[j], r0[j + 1], r1[j], r1[j + 1], wt);
                const int x00 = (int)(short)(d0[j] & 0xFFFF), x01 = (int)(short)(d0[j + 1] & 0xFFFF);
                const int x10 = (int)(short)(d1[j] & 0xFFFF), x11 = (int)(short)(d1[j + 1] & 0xFFFF);
                const int y00 = (int)d0[j] >> 16, y01 = (int)d0[j + 1] >> 16;
                const int y10 = (int)d1[j] >> 16, y11 = (int)d1[j + 1] >> 16;
                int ixval = (x00 * wt.w00 + x01 * wt.w01 + x10 * wt.w10 + x11 * wt.w11 +
                             (1 << (W_BITS - 1))) >> W_BITS;
                int iyval = (y00 * wt.w00 + y01 * wt.w01 + y10 * wt.w10 + y11 * wt.w11 +
                             (1 << (W_BITS - 1))) >> W_BITS;
                ixval = on ? ixval : 0;
                iyval = on ? iyval : 0;
                Iw[s][j] = on ? ival : 0;
                dIw[s][j] = ((uint32_t)ixval & 0xFFFFu) | ((uint32_t)iyval << 16);
                pA11 += ixval * ixval;
                pA12 += ixval * iyval;
                pA22 += iyval * iyval;
            }
        }
        const float A11 = (float)wave_sum_wide(pA11) * FLT_SCALE;
        const float A12 = (float)wave_sum_wide(pA12) * FLT_SCALE;
        const float A22 = (float)wave_sum_wide(pA22) * FLT_SCALE;

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

        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(next_x), iny = (int)floorf(next_y);
            if (inx < -a.win_w || inx >= lw || iny < -a.win_h || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            wt = bilinear_weights(next_x - inx, next_y - iny);
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const ptrdiff_t o = (ptrdiff_t)(iny + srow[s]) * S + inx + sx0[s];
                int r0[8], r1[8];
                unpack8(load_u8x8(J + o), r0);
                unpack8(load_u8x8(J + o + S), r1);
#pragma unroll
                for (int k = 0; k < kPxPerSlot; ++k) {
                    const int diff = interp_px(r0[k], r0[k + 1], r1[k], r1[k + 1], wt) - Iw[s][k];
                    const int dx = (int)(short)(dIw[s][k] & 0xFFFF), dy = (int)dIw[s][k] >> 16;
                    pb1 += diff * dx;  // dx == dy == 0 outside the window
                    pb2 += diff * dy;
                }
            }
            const float b1 = (float)wave_sum_wide(pb1) * FLT_SCALE;
            const float b2 = (float)wave_sum_wide(pb2) * FLT_SCALE;
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
                int perr = 0;
#pragma unroll
                for (int s = 0; s < SPL; ++s) {
                    const ptrdiff_t o = (ptrdiff_t)(iny + srow[s]) * S + inx + sx0[s];
                    int r0[8], r1[8];
                    unpack8(load_u8x8(J + o), r0);
                    unpack8(load_u8x8(J + o + S), r1);
#pragma unroll
                    for (int k = 0; k < kPxPerSlot; ++k) {
                        const int diff = interp_px(r0[k], r0[k + 1], r1[k], r1[k + 1], wt) - Iw[s][k];
                        perr += k < snpx[s] ? abs(diff) : 0;
                    }
                }
                // Sum of |diff| stays below 2^24 for windows up to 41x41, so the float
                // accumulation of the reference is exact and equals this integer sum.
                const float errval = (float)wave_sum(perr);
                err = errval * 1.f / (float)(32 * a.win_w * a.win_h);
            }
        }
    }

    if (err > a.max_error) status = 0;  // feature_processing.cpp:245-248
    if (lane == 0) {
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
    g->bw = win_w;
    g->bh = win_h;
    size_t off = 0;
    int w = width, h = height;
    for (int l = 0; l <= g->levels; ++l) {
        g->w[l] = w;
        g->h[l] = h;
        // 8 extra columns: a slot's 8-element loads may run past the window
        g->stride[l] = (int)align_up((size_t)w + 2 * win_w + 8, 4);
        g->img_off[l] = off + (size_t)win_h * g->stride[l] + (size_t)win_w;
        // keep every level's first padded element 4-aligned
        g->img_off[l] = align_up(g->img_off[l] - win_w, 4) + win_w;
        off = align_up(g->img_off[l] - win_w + (size_t)(h + win_h) * g->stride[l] + 64, 64);
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

size_t lk_carve(Carver& c, const LkGeom& g, int n_pairs, LkBuffers* b) {
    b->pyr_prev = c.take<uint8_t>(g.block * n_pairs);
    b->pyr_curr = c.take<uint8_t>(g.block * n_pairs);
    b->deriv = c.take<uint32_t>(g.block * n_pairs);
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
    return lk_carve(c, g, n_pairs, &b);
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

    // SparsePyrLKOpticalFlowImpl::calc: criteria clamping
    if ((crit_type & SFM_TERM_COUNT) == 0)
        max_count = 30;
    else
        max_count = max_count < 0 ? 0 : (max_count > 100 ? 100 : max_count);
    if ((crit_type & SFM_TERM_EPS) == 0)
        epsilon = 0.01;
    else
        epsilon = epsilon < 0. ? 0. : (epsilon > 10. ? 10. : epsilon);
    epsilon *= epsilon;

    LkArgs a{};
    make_geom(width, height, win_w, win_h, max_level, &a.g);
    const LkGeom& g = a.g;
    Carver sz(nullptr);
    LkBuffers bufs;
    const size_t need = lk_carve(sz, g, n_pairs, &bufs);
    void* ws;
    SFM_RETURN_IF(workspace(ctx, need, &ws));
    Carver c(ws);
    lk_carve(c, g, n_pairs, &bufs);

    const dim3 blk(256);
    {
        const int S = g.stride[0];
        const dim3 grid((S / 4 + 255) / 256, g.h[0] + 2 * g.bh, 2 * n_pairs);
        KernelTimer timer(ctx, KID_LK_PAD_LEVEL0);
        hipLaunchKernelGGL(lk_pad_level0_kernel, grid, blk, 0, ctx->stream, d_prev, d_curr,
                           image_stride, stride, g, bufs.pyr_prev, bufs.pyr_curr);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    for (int l = 1; l <= g.levels; ++l) {
        const int S = g.stride[l];
        const dim3 grid((S / 4 + 255) / 256, g.h[l] + 2 * g.bh, 2 * n_pairs);
        KernelTimer timer(ctx, KID_LK_PYR_DOWN);
        hipLaunchKernelGGL(lk_pyr_down_kernel, grid, blk, 0, ctx->stream, g, l, bufs.pyr_prev,
                           bufs.pyr_curr);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    for (int l = 0; l <= g.levels; ++l) {
        const int S = g.stride[l];
        const dim3 grid((S + 255) / 256, g.h[l] + 2 * g.bh, n_pairs);
        KernelTimer timer(ctx, KID_LK_SCHARR);
        hipLaunchKernelGGL(lk_scharr_kernel, grid, blk, 0, ctx->stream, g, l, bufs.pyr_prev,
                           bufs.deriv);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }

    a.pyr_prev = bufs.pyr_prev;
    a.pyr_curr = bufs.pyr_curr;
    a.deriv = bufs.deriv;
    a.prev_pts = reinterpret_cast<const float2*>(d_prev_pts);
    a.n = n;
    a.n_total = n * n_pairs;
    a.win_w = win_w;
    a.win_h = win_h;
    a.segs = (win_w + kPxPerSlot - 1) / kPxPerSlot;
    a.slots = a.segs * win_h;
    a.max_count = max_count;
    a.epsilon = epsilon;
    a.min_eig = (float)min_eig_threshold;
    a.max_error = max_error;
    for (int l = 0; l <= g.levels; ++l) a.lscale[l] = (float)(1. / (1 << l));
    a.next_pts = reinterpret_cast<float2*>(d_curr_pts);
    a.status = d_status;
    a.err = d_err;
    const int spl = (a.slots + 63) / 64;
    const dim3 grid((a.n_total + 3) / 4);
    KernelTimer timer(ctx, KID_LK_TRACK);
    switch (spl) {
        case 1: hipLaunchKernelGGL(lk_track_kernel<1>, grid, blk, 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(lk_track_kernel<2>, grid, blk, 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL(lk_track_kernel<3>, grid, blk, 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(lk_track_kernel<4>, grid, blk, 0, ctx->stream, a); break;
    }
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

int sfm_pyrlk(sfm_ctx* ctx, const uint8_t* prev, const uint8_t* curr, int width, int height,
              size_t stride, const float* prev_pts, int n, int win_w, int win_h, int max_level,
              int crit_type, int max_count, double epsilon, double min_eig_threshold,
              float max_error, float* curr_pts, uint8_t* status, float* err) {
    SFM_RETURN_IF(check_lk_args(ctx, width, height, stride, n, win_w, win_h, max_level));
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
    SFM_HIP_CHECK(ctx, hipMemcpy2DAsync(dp, width, prev, stride, width, height,
                                        hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_CHECK(ctx, hipMemcpy2DAsync(dc, width, curr, stride, width, height,
                                        hipMemcpyHostToDevice, ctx->stream));
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
    return SFM_OK;
}

}  // extern "C"
