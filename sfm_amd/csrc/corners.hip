// Shi-Tomasi corners: cv::goodFeaturesToTrack(imGray, corners, maxCorners, qualityLevel,
// minDistance) as FeatureDetector::generateFlowFeatures calls it
// (Source Files/feature_processing.cpp:87-99): no mask, 3x3 block, 3x3 Sobel, no Harris.
//
//   1. corner_eig_kernel     Sobel -> products -> 3x3 box sum -> smaller eigenvalue, one pass
//                            over the image through LDS tiles; image maximum by atomics.
//   2. corner_pick_kernel    threshold at max * qualityLevel, 3x3 local maxima of the interior,
//                            appended as sortable keys (image << 58 | value bits << 26 | y * w + x).
//   3. corner_sort_kernel    radix sort of an image's candidates, descending: by value, ties by
//                            address, both descending, as OpenCV's std::sort with greaterThanPtr
//                            orders its pointers.  One workgroup per image; only the candidates
//                            found are sorted, not the capacity of the list.
//   4. corner_rank_kernel    position -> rank image (which also carries each candidate's state).
//   5. corner_select_kernel  the greedy minimum-distance pass.  Sequentially, a candidate is
//                            kept unless an already kept one is closer than minDistance (the
//                            grid of round(minDistance) cells OpenCV searches never hides such a
//                            corner: ceil(d) - 1 <= round(d), so it lies in the 3x3 cells).
//                            Here every undecided candidate looks at the stronger candidates
//                            in its disc each round: one of them kept -> dropped; all of them
//                            dropped -> kept; otherwise it waits.  A decision is only ever
//                            taken on final information, so the fixed point is the sequential
//                            answer whatever the schedule.  Only a prefix of the sorted
//                            candidates is worked on (a candidate depends on stronger ones
//                            only): when a round leaves it settled with fewer than maxCorners
//                            kept, the next round works on twice as much; with enough kept the
//                            rounds end.  Every workgroup derives that schedule for itself
//                            from the per-round counts of the rounds before (no workgroup waits
//                            on another).  Rounds after the end exit at once; a single-wave
//                            sequential pass settles whatever a fixed number of rounds left open.
//   6. corner_emit_kernel    the kept candidates, in rank order, up to maxCorners.
//
// Every kernel takes a batch of equally sized images (one grid dimension is the image).  For a
// single 1280x720 frame most of the time are launch gaps between ~20 short kernels; a batch
// shares them.
//
// All float arithmetic is unfused and ordered as in the CPU restatement: Sobel smoothing
// (s0 + s2) * k1 + s1 * k0 with the scale 1 / (4 * 3 * 255) folded into the taps, box sums
// ((l + m) + r) per row then ((top + mid) + bottom), eig = (a + c) - sqrt((a - c)^2 + b^2).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include "sfm_internal.h"

namespace sfm {

namespace {

constexpr int kTileW = 64, kTileH = 16;  // outputs per workgroup of 256 threads
constexpr int kSelectRounds = 12;
constexpr int kMaxDistance = 128;  // largest minDistance: the disc's offsets are tabulated
// per round r: candidates still open after it, candidates kept in it
enum { C_FOUND = 0, C_OVERFLOW, C_OPEN, C_KEPT = C_OPEN + kSelectRounds, C_COUNT = C_KEPT + kSelectRounds };
constexpr int kStateShift = 30;  // rank image entries: rank | state << 30, or -1
constexpr int kPosBits = 26, kImageShift = 58, kMaxImages = 64;  // key = image | value bits | position
constexpr int kBuckets = 64;     // value ranges an image's candidates are split into before the sort
constexpr int kCtrlWords = 64 + 2 * kBuckets;  // cleared words per image: maximum, counters; bucket sizes, bucket starts
constexpr int kBucketWord = 64 - 4;  // first bucket word, counted from the counters

struct CornerArgs {
    const uint8_t* img;
    int w, h;
    size_t stride;
    int n_images, image;    // batch size; the image these pointers belong to (after image_view)
    size_t img_stride;      // bytes between images
    float2* corners;        // [image][cap_out]
    int cap_out;
    int32_t* info;          // [image][3]
    float* eig;             // [h][w]
    unsigned* max_bits;     // ordered bits of the image maximum
    unsigned* tile_max;     // [image][tile]: the maximum of every tile of corner_eig_kernel, or null
    unsigned long long* keys;     // [cap] appended candidates
    unsigned long long* sorted;   // [cap]
    int32_t* rank;          // [h][w]: -1 where no candidate, else rank | state << 30
    int32_t* state;         // [cap] 0 undecided, 1 kept, 2 dropped
    int32_t* counters;      // see the C_* indices
    int cap;
    double quality, min_distance;
    int max_corners;
};

// The arguments of image `im` of the batch: every per-image array moved to its slice.  The
// candidates of an image are sorted inside its own slice.
__device__ __forceinline__ CornerArgs image_view(CornerArgs a, int im) {
    const size_t px = (size_t)a.w * a.h;
    a.image = im;
    a.img += (size_t)im * a.img_stride;
    a.eig += (size_t)im * px;
    a.rank += (size_t)im * px;
    a.max_bits += (size_t)im * kCtrlWords;
    a.counters += (size_t)im * kCtrlWords;
    a.keys += (size_t)im * a.cap;
    a.sorted += (size_t)im * a.cap;
    a.state += (size_t)im * a.cap;
    a.corners += (size_t)im * a.cap_out;
    a.info += 3 * im;
    return a;
}

__device__ __forceinline__ unsigned key_position(unsigned long long key) { return (unsigned)(key & ((1ull << kPosBits) - 1)); }

__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

// Monotone map of float order onto unsigned order.
__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void corner_eig_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.z);
    __shared__ uint8_t s_img[kTileH + 4][kTileW + 4];
    __shared__ float2 s_d[kTileH + 2][kTileW + 2];
    __shared__ unsigned s_max;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    if (tid == 0) s_max = 0;
    for (int i = tid; i < (kTileH + 4) * (kTileW + 4); i += 256) {
        const int ty = i / (kTileW + 4), tx = i - ty * (kTileW + 4);
        const int gx = reflect101(x0 - 2 + tx, a.w), gy = reflect101(y0 - 2 + ty, a.h);
        s_img[ty][tx] = a.img[(size_t)gy * a.stride + gx];
    }
    __syncthreads();
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k0 = (float)(2.0 * scale), k1 = (float)(1.0 * scale);
    // derivatives on the tile and a ring of one; outside the image they are the derivatives
    // AT the reflected position (the box filter reflects the products, not the image)
    for (int i = tid; i < (kTileH + 2) * (kTileW + 2); i += 256) {
        const int ty = i / (kTileW + 2), tx = i - ty * (kTileW + 2);
        const int X = x0 - 1 + tx, Y = y0 - 1 + ty;
        float2 d = make_float2(0.f, 0.f);
        if (X <= a.w && Y <= a.h) {
            const int cx = reflect101(X, a.w) - (x0 - 2), cy = reflect101(Y, a.h) - (y0 - 2);
            const uint8_t* r0 = &s_img[cy - 1][cx];
            const uint8_t* r1 = &s_img[cy][cx];
            const uint8_t* r2 = &s_img[cy + 1][cx];
            const float d0 = (float)((int)r0[1] - (int)r0[-1]);
            const float d1 = (float)((int)r1[1] - (int)r1[-1]);
            const float d2 = (float)((int)r2[1] - (int)r2[-1]);
            d.x = (d0 + d2) * k1 + d1 * k0;
            const float s0 = (float)((int)r0[-1] + (int)r0[1]) * k1 + (float)r0[0] * k0;
            const float s2 = (float)((int)r2[-1] + (int)r2[1]) * k1 + (float)r2[0] * k0;
            d.y = s2 - s0;
        }
        s_d[ty][tx] = d;
    }
    __syncthreads();
    unsigned best = 0;
    const int px = tid & (kTileW - 1);
#pragma unroll
    for (int k = 0; k < kTileH / 4; ++k) {
        const int py = (tid >> 6) + 4 * k;
        const int x = x0 + px, y = y0 + py;
        if (x >= a.w || y >= a.h) continue;
        float rows[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float p[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float2 g = s_d[py + j][px + i];
                p[0][i] = g.x * g.x;
                p[1][i] = g.x * g.y;
                p[2][i] = g.y * g.y;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) rows[c][j] = (p[c][0] + p[c][1]) + p[c][2];
        }
        const float cxx = (rows[0][0] + rows[0][1]) + rows[0][2];
        const float cxy = (rows[1][0] + rows[1][1]) + rows[1][2];
        const float cyy = (rows[2][0] + rows[2][1]) + rows[2][2];
        const float aa = cxx * 0.5f, bb = cxy, cc = cyy * 0.5f;
        const float e = (aa + cc) - sqrtf((aa - cc) * (aa - cc) + bb * bb);
        a.eig[(size_t)y * a.w + x] = e;
        best = max(best, ordered_bits(e));
    }
    if (!batch.tile_max) return;   // (the response image alone was asked for)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if ((tid & 63) == 0) atomicMax(&s_max, best);
    __syncthreads();
    // One slot per tile, folded by corner_max_kernel: an atomic maximum per workgroup on the
    // image's word (900 workgroups per 1280 x 720 image, all in flight) took two thirds of
    // this kernel's time.
    if (tid == 0)
        batch.tile_max[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s_max;
}

// The image maxima from the tiles' maxima: one workgroup per image.
__global__ __launch_bounds__(256) void corner_max_kernel(CornerArgs batch, int tiles) {
    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    unsigned best = 0;
    for (int i = threadIdx.x; i < tiles; i += 256) best = max(best, batch.tile_max[(size_t)blockIdx.x * tiles + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, best);
    __syncthreads();
    if (threadIdx.x == 0) batch.max_bits[(size_t)blockIdx.x * kCtrlWords] = s_max;
}

constexpr int kPickRows = 32;  // rows per workgroup of the candidate pass (8 per wave)

__global__ __launch_bounds__(256) void corner_pick_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.z);
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = 1 + blockIdx.x * 64 + lane;
    // minMaxLoc's double maximum times the double quality level, then threshold()'s float
    const float thresh = (float)((double)from_ordered_bits(*a.max_bits) * a.quality);
    unsigned long long ballots[kPickRows / 4];
    float vals[kPickRows / 4];
    int wave_total = 0;
#pragma unroll
    for (int k = 0; k < kPickRows / 4; ++k) {
        const int y = 1 + blockIdx.y * kPickRows + wave + 4 * k;
        bool pick = false;
        float v = 0.f;
        if (x < a.w - 1 && y < a.h - 1) {
            const float* e = a.eig + (size_t)y * a.w + x;
            v = e[0] > thresh ? e[0] : 0.f;  // THRESH_TOZERO
            if (v != 0.f) {
                float m = v;  // 3x3 dilation of the thresholded image
#pragma unroll
                for (int j = -1; j <= 1; ++j)
#pragma unroll
                    for (int i = -1; i <= 1; ++i) {
                        const float u = e[j * a.w + i];
                        m = fmaxf(m, u > thresh ? u : 0.f);
                    }
                pick = v == m;
            }
        }
        vals[k] = v;
        ballots[k] = __ballot(pick);
        wave_total += (int)__popcll(ballots[k]);
    }
    if (lane == 0) s_wave[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = total ? atomicAdd(a.counters + C_FOUND, total) : 0;  // one per workgroup
    }
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
#pragma unroll
    for (int k = 0; k < kPickRows / 4; ++k) {
        const int y = 1 + blockIdx.y * kPickRows + wave + 4 * k;
        if ((ballots[k] >> lane) & 1ULL) {
            const int slot = base + (int)__popcll(ballots[k] & ((1ULL << lane) - 1ULL));
            if (slot < a.cap)
                a.keys[slot] = ((unsigned long long)a.image << kImageShift) |
                               ((unsigned long long)ordered_bits(vals[k]) << kPosBits) | (unsigned)(y * a.w + x);
            else
                a.counters[C_OVERFLOW] = 1;
        }
        base += (int)__popcll(ballots[k]);
    }
}

__device__ __forceinline__ int candidate_count(const CornerArgs& a) { return min(a.counters[C_FOUND], a.cap); }

// States are read and written past the per-CU cache: a wave may then see decisions other
// waves took earlier in the same launch.  A stale read can only show "undecided", which
// delays a decision and never changes it.  A candidate's state is kept twice: by rank (to
// walk the candidates) and inside the rank image (so that looking at a neighbouring pixel
// is one load, not a chain of two).
__device__ __forceinline__ int load_state(const CornerArgs& a, int i) {
    return __hip_atomic_load(a.state + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_state(const CornerArgs& a, int i, unsigned pos, int s) {
    __hip_atomic_store(a.state + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.rank + pos, i | (s << kStateShift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Fine value bin of a key: the value range [threshold, maximum] of the image cut into kFineBins
// equal steps of the ordered value bits, the highest values in bin 0.  Monotone in the key.
constexpr int kFineBins = 2048;
__device__ __forceinline__ int key_fine_bin(unsigned long long key, unsigned lo, float scale) {
    const unsigned vb = (unsigned)(key >> kPosBits);
    const unsigned rel = vb > lo ? vb - lo : 0u;
    // float arithmetic is monotone in rel, which is all the split needs
    const int b = (int)((float)rel * scale);
    return kFineBins - 1 - min(b, kFineBins - 1);
}

// Splits the candidates of an image (keys, in the order they were found) by value into
// `sorted`: bucket 0 (the highest values) first, any order inside a bucket.  The buckets are
// runs of fine value bins of about equal population (a histogram over kFineBins bins, cut where
// the running count passes multiples of n / kBuckets), so that the workgroups that sort them
// get equal work whatever the distribution of the values.  Sizes and starts of the buckets go
// to the image's control words.  One workgroup per image.
__global__ __launch_bounds__(1024) void corner_split_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.x);
    const int n = candidate_count(a);
    __shared__ unsigned s_fine[kFineBins];      // population, then bucket of the bin
    __shared__ unsigned s_wave_total[16];
    __shared__ unsigned s_size[kBuckets], s_next[kBuckets];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_fine[tid] = 0;
    s_fine[tid + 1024] = 0;
    if (tid < kBuckets) s_size[tid] = 0;
    __syncthreads();
    const unsigned hi = *a.max_bits;
    const unsigned lo = min(ordered_bits((float)((double)from_ordered_bits(hi) * a.quality)), hi);
    const float scale = (float)kFineBins / ((float)(hi - lo) + 1.f);
    for (int i = tid; i < n; i += 1024) atomicAdd(&s_fine[key_fine_bin(a.keys[i], lo, scale)], 1u);
    __syncthreads();
    // running count before each bin (thread t: bins 2t, 2t + 1), bucket = where it falls in n
    const unsigned c0 = s_fine[2 * tid], c1 = s_fine[2 * tid + 1];
    unsigned incl = c0 + c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave_total[wave] = incl;
    __syncthreads();
    unsigned before = incl - (c0 + c1);
    for (int w = 0; w < wave; ++w) before += s_wave_total[w];
    const unsigned per = (unsigned)(n + kBuckets - 1) / kBuckets + 1u;
    const unsigned b0 = min(before / per, (unsigned)(kBuckets - 1));
    const unsigned b1 = min((before + c0) / per, (unsigned)(kBuckets - 1));
    s_fine[2 * tid] = b0;
    s_fine[2 * tid + 1] = b1;
    if (c0) atomicAdd(&s_size[b0], c0);
    if (c1) atomicAdd(&s_size[b1], c1);
    __syncthreads();
    if (tid == 0) {
        unsigned at = 0;
        for (int b = 0; b < kBuckets; ++b) {
            a.counters[kBucketWord + b] = (int)s_size[b];
            a.counters[kBucketWord + kBuckets + b] = (int)at;
            s_next[b] = at;
            at += s_size[b];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = a.keys[i];
        a.sorted[atomicAdd(&s_next[s_fine[key_fine_bin(key, lo, scale)]], 1u)] = key;
    }
}

// One bucket of an image's candidates (in `sorted`) into descending key order (value, then
// position: OpenCV sorts by value and breaks ties by the higher address first), so that the
// whole list is in order: a stable least-significant-digit radix sort, eight passes of eight
// bits, by one workgroup of 16 waves.  A pass whose digit is the same for all keys (the bits
// that made the bucket, the image number) is a plain copy.  Every wave owns a contiguous share
// of the list; per pass it counts its digits, the workgroup turns the counts into start
// positions (digit major, wave minor), and the wave walks its share again and places every
// key: keys of a 64-key tile with the same digit find their order among themselves from
// ballots.  That is the path of a bucket above kSortLdsKeys keys (many equal values); the
// usual bucket, a few hundred keys, is sorted inside LDS by a bitonic network instead (the keys
// are all different -- they carry their position -- so an unstable sort gives the same order).
constexpr int kSortWaves = 16, kSortAhead = 4, kSortLdsKeys = 2048;
__global__ __launch_bounds__(64 * kSortWaves) void corner_sort_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.y);
    const int n = a.counters[kBucketWord + blockIdx.x];
    if (n < 2) return;  // workgroup-uniform
    const int first = a.counters[kBucketWord + kBuckets + blockIdx.x];
    __shared__ unsigned s_count[kSortWaves][256];  // then: next free position per (wave, digit)
    __shared__ unsigned s_total[256];
    __shared__ unsigned s_wave_sum[4];
    __shared__ int s_one_digit;  // all keys share this pass's digit: the pass is a plain copy
    __shared__ unsigned long long s_keys[kSortLdsKeys];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (n <= kSortLdsKeys) {
        unsigned long long* io = a.sorted + first;
        int padded = 2;
        while (padded < n) padded <<= 1;
        for (int i = tid; i < padded; i += 64 * kSortWaves) s_keys[i] = i < n ? io[i] : 0ull;  // 0 sorts last
        __syncthreads();
        for (int k = 2; k <= padded; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < padded; i += 64 * kSortWaves) {
                    const int other = i ^ j;
                    if (other > i) {
                        const unsigned long long x = s_keys[i], y = s_keys[other];
                        const bool descending = (i & k) == 0;
                        if ((x < y) == descending) {
                            s_keys[i] = y;
                            s_keys[other] = x;
                        }
                    }
                }
                __syncthreads();
            }
        for (int i = tid; i < n; i += 64 * kSortWaves) io[i] = s_keys[i];
        return;
    }
    const int share = (((n + kSortWaves - 1) / kSortWaves) + 63) & ~63;
    const int begin = min(wave * share, n), end = min(begin + share, n);
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long* src = a.sorted + first;
    unsigned long long* dst = a.keys + first;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 8 * pass;
        // lanes of the tile that hold the same digit (descending order: digit 255 first)
        auto same_digit = [&](bool valid, unsigned long long key, unsigned& digit) {
            digit = 255u - (unsigned)((key >> shift) & 255u);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long v = __ballot(bit);
                m &= bit ? v : ~v;
            }
            return m;
        };
        for (int i = lane; i < 256; i += 64) s_count[wave][i] = 0;
        if (tid == 0) s_one_digit = 0;
        // one wave touches its own row of the table only, in program order: no barrier needed
        for (int i0 = begin; i0 < end; i0 += 64 * kSortAhead) {
            unsigned long long keys[kSortAhead];  // the loads of several tiles in flight at once
#pragma unroll
            for (int u = 0; u < kSortAhead; ++u) keys[u] = i0 + 64 * u + lane < end ? src[i0 + 64 * u + lane] : 0ull;
#pragma unroll
            for (int u = 0; u < kSortAhead; ++u) {
                const bool valid = i0 + 64 * u + lane < end;
                unsigned digit;
                const unsigned long long m = same_digit(valid, keys[u], digit);
                if (valid && (m & below) == 0) s_count[wave][digit] += (unsigned)__popcll(m);
            }
        }
        __syncthreads();
        // start positions: all keys of smaller digits, then the same digit in earlier waves
        unsigned mine = 0;
        if (tid < 256) {
            for (int w = 0; w < kSortWaves; ++w) {
                const unsigned c = s_count[w][tid];
                s_count[w][tid] = mine;
                mine += c;
            }
            unsigned incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (lane == 63) s_wave_sum[wave] = incl;
            if (mine == (unsigned)n) s_one_digit = 1;
            s_total[tid] = incl - mine;  // exclusive within the wave
        }
        __syncthreads();
        if (tid < 256) {
            unsigned base = s_total[tid];
            for (int w = 0; w < wave; ++w) base += s_wave_sum[w];
            for (int w = 0; w < kSortWaves; ++w) s_count[w][tid] += base;
        }
        __syncthreads();
        if (s_one_digit) {
            for (int i = begin + lane; i < end; i += 64) dst[i] = src[i];
        } else {
            for (int i0 = begin; i0 < end; i0 += 64 * kSortAhead) {
                unsigned long long keys[kSortAhead];
#pragma unroll
                for (int u = 0; u < kSortAhead; ++u) keys[u] = i0 + 64 * u + lane < end ? src[i0 + 64 * u + lane] : 0ull;
#pragma unroll
                for (int u = 0; u < kSortAhead; ++u) {
                    const bool valid = i0 + 64 * u + lane < end;
                    unsigned digit;
                    const unsigned long long m = same_digit(valid, keys[u], digit);
                    unsigned at = 0;
                    if (valid) at = s_count[wave][digit] + (unsigned)__popcll(m & below);
                    // every lane has read the position of its digit before the digit's first lane moves it on
                    __builtin_amdgcn_wave_barrier();
                    if (valid && (m & below) == 0) s_count[wave][digit] += (unsigned)__popcll(m);
                    if (valid) dst[at] = keys[u];
                }
            }
        }
        __syncthreads();  // also orders this pass's stores before the next pass's loads
        unsigned long long* t = src;
        src = dst;
        dst = t;
    }
    // eight passes: the result is where the bucket started, in `sorted`
}

// Ranks and the first prefix to work on.
__global__ __launch_bounds__(256) void corner_rank_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.y);
    const int n = candidate_count(a);
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int i = t0; i < n; i += nt) a.rank[key_position(a.sorted[i])] = i;
}

// The disc "closer than minDistance" on the pixel grid.  OpenCV tests float dx*dx + dy*dy <
// minDistance^2 (a double); for integer offsets the sum is exact, so that is
// dx*dx + dy*dy <= limit with limit the largest integer below minDistance^2.
struct Disc {
    int reach, side, limit;
    float inv_side;
};
__device__ __forceinline__ Disc make_disc(double min_distance) {
    Disc d;
    d.reach = max((int)ceil(min_distance) - 1, 0);
    d.side = 2 * d.reach + 1;
    d.limit = (int)ceil(min_distance * min_distance) - 1;
    d.inv_side = 1.0f / (float)d.side;
    return d;
}

// Decision for candidate i from the states of the stronger candidates around it:
// 1 kept, 2 dropped, 0 cannot be told yet.  By one wave.
__device__ int decide(const CornerArgs& a, int i, unsigned pos, int lane, const Disc& d) {
    const int y = (int)(pos / (unsigned)a.w), x = (int)(pos - (unsigned)y * a.w);
    bool near_kept = false, waiting = false;
    const int cells = d.side * d.side;
    // branch-free, eight steps at a time: their loads are independent and overlap
    for (int t0 = lane; t0 < cells; t0 += 512) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + 64 * u;
            // t / side: exact, (t + 0.5) / side stays 1 / (2 side) away from every integer
            const int row = (int)(((float)t + 0.5f) * d.inv_side);
            const int dy = row - d.reach, dx = t - row * d.side - d.reach;
            const int nx = x + dx, ny = y + dy;
            const bool look = t < cells && dx * dx + dy * dy <= d.limit && nx >= 0 && ny >= 0 && nx < a.w && ny < a.h;
            const int got = __hip_atomic_load(a.rank + (size_t)min(max(ny, 0), a.h - 1) * a.w + min(max(nx, 0), a.w - 1),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[u] = look ? got : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool stronger = v[u] >= 0 && (v[u] & ((1 << kStateShift) - 1)) < i;
            const int s = v[u] >> kStateShift;
            near_kept |= stronger && s == 1;
            waiting |= stronger && s == 0;
        }
    }
    if (__any(near_kept)) return 2;
    if (__any(waiting)) return 0;
    return 1;
}

// The prefix of the sorted candidates round `round` works on, or 0 when the rounds have ended:
// replayed from what the earlier rounds counted.
__device__ int window_of_round(const CornerArgs& a, int n, int round) {
    int window = a.max_corners > 0 ? min(n, max(4 * a.max_corners, 1024)) : n;
    int kept = 0;
    for (int q = 0; q < round; ++q) {
        kept += a.counters[C_KEPT + q];
        if (a.counters[C_OPEN + q] != 0) continue;  // not settled: same prefix again
        if (window >= n || (a.max_corners > 0 && kept >= a.max_corners)) return 0;
        window = min(n, 2 * window);
    }
    return window;
}

__global__ __launch_bounds__(256) void corner_select_kernel(CornerArgs batch, int round) {
    __shared__ int s_open, s_kept;
    const CornerArgs a = image_view(batch, blockIdx.y);
    const int n = candidate_count(a);
    const int window = window_of_round(a, n, round);
    if (window == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const Disc disc = make_disc(a.min_distance);
    if (threadIdx.x == 0) s_open = s_kept = 0;
    __syncthreads();
    int open = 0, kept = 0;
    for (int i = wave; i < window; i += n_waves) {
        if (load_state(a, i) != 0) continue;
        const unsigned pos = key_position(a.sorted[i]);
        const int s = a.min_distance >= 1 ? decide(a, i, pos, lane, disc) : 1;
        open += s == 0;
        kept += s == 1;
        if (s != 0 && lane == 0) store_state(a, i, pos, s);
    }
    if (lane == 0 && open) atomicAdd(&s_open, open);
    if (lane == 0 && kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (s_open) atomicAdd(a.counters + C_OPEN + round, s_open);
    if (s_kept) atomicAdd(a.counters + C_KEPT + round, s_kept);
}

// Whatever the rounds left open, in order, by one wave.
__global__ __launch_bounds__(64) void corner_settle_kernel(CornerArgs batch) {
    const CornerArgs a = image_view(batch, blockIdx.x);
    const int n = candidate_count(a);
    if (window_of_round(a, n, kSelectRounds) == 0) return;
    const int lane = threadIdx.x;
    const Disc disc = make_disc(a.min_distance);
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        int s = load_state(a, i);
        if (s == 0) {
            const unsigned pos = key_position(a.sorted[i]);
            s = a.min_distance >= 1 ? decide(a, i, pos, lane, disc) : 1;  // every stronger one is settled
            if (lane == 0) store_state(a, i, pos, s);
            __threadfence();
        }
        kept += s == 1;
        if (a.max_corners > 0 && kept >= a.max_corners) break;
    }
}

// out[0] = corners written, out[1] = candidates, out[2] = 1 if candidates were lost to the
// capacity.
__global__ __launch_bounds__(1024) void corner_emit_kernel(CornerArgs batch) {
    __shared__ int wave_count[16];
    const CornerArgs a = image_view(batch, blockIdx.x);
    float2* corners = a.corners;
    int32_t* out = a.info;
    const int cap_out = a.cap_out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = candidate_count(a);
    int limit = cap_out;
    if (a.max_corners > 0) limit = min(limit, a.max_corners);
    int running = 0;
    for (int i0 = 0; i0 < n && running < limit; i0 += 1024) {
        const int i = i0 + tid;
        const bool ok = i < n && a.state[i] == 1;
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
        const int slot = running + offset + prefix;
        if (ok && slot < limit) {
            const unsigned pos = key_position(a.sorted[i]);
            const int y = (int)(pos / (unsigned)a.w), x = (int)(pos - (unsigned)y * a.w);
            corners[slot] = make_float2((float)x, (float)y);
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = min(running, limit);
        out[1] = a.counters[C_FOUND];
        out[2] = a.counters[C_OVERFLOW];
    }
}

int default_capacity(int w, int h) {
    // strict 3x3 local maxima fill at most a quarter of the pixels; plateaus of equal values
    // can exceed that, and are reported as overflow
    const size_t px = (size_t)w * h;
    return (int)std::min<size_t>(px, px / 4 + 1024);
}

struct CornerBuffers {
    float* eig;
    int32_t* rank;
    unsigned long long *keys, *sorted;
    int32_t* state;
    unsigned* ctrl;  // per image kCtrlWords: the maximum, then the counters
    unsigned* tile_max;  // per image: one word per tile of corner_eig_kernel
};
inline size_t eig_tiles(int w, int h) { return (size_t)((w + kTileW - 1) / kTileW) * ((h + kTileH - 1) / kTileH); }

CornerBuffers carve_corners(Carver& c, int n_images, int w, int h, int cap) {
    CornerBuffers b;
    const size_t px = (size_t)w * h * n_images, slots = (size_t)cap * n_images;
    b.eig = c.take<float>(px);
    b.rank = c.take<int32_t>(px);
    b.keys = c.take<unsigned long long>(slots);
    b.sorted = c.take<unsigned long long>(slots);
    b.state = c.take<int32_t>(slots);
    b.ctrl = c.take<unsigned>((size_t)kCtrlWords * n_images);
    b.tile_max = c.take<unsigned>(eig_tiles(w, h) * n_images);
    static_assert(4 + C_COUNT <= 64, "the counters end before the bucket words");
    return b;
}

// Enqueues the detection for a batch of images; the results land in d_corners and d_info.
int run_corners(sfm_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride, int w, int h, size_t stride,
                int max_corners, double quality, double min_distance, float* d_corners, int cap_out, int32_t* d_info,
                int cap, const CornerBuffers& b) {
    CornerArgs a{};
    a.img = d_imgs;
    a.w = w;
    a.h = h;
    a.stride = stride;
    a.n_images = n_images;
    a.img_stride = image_stride;
    a.corners = reinterpret_cast<float2*>(d_corners);
    a.cap_out = cap_out;
    a.info = d_info;
    a.eig = b.eig;
    a.max_bits = b.ctrl;
    a.tile_max = b.tile_max;
    a.keys = b.keys;
    a.sorted = b.sorted;
    a.rank = b.rank;
    a.state = b.state;
    a.counters = reinterpret_cast<int32_t*>(b.ctrl) + 4;
    a.cap = cap;
    a.quality = quality;
    a.min_distance = min_distance;
    a.max_corners = max_corners;
    hipStream_t s = ctx->stream;
    const size_t px = (size_t)w * h * n_images, slots = (size_t)cap * n_images;
    const unsigned n = (unsigned)n_images;
    SFM_HIP_CHECK(ctx, hipMemsetAsync(b.ctrl, 0, sizeof(unsigned) * kCtrlWords * n_images, s));
    SFM_HIP_CHECK(ctx, hipMemsetAsync(b.rank, 0xff, sizeof(int32_t) * px, s));
    SFM_HIP_CHECK(ctx, hipMemsetAsync(b.state, 0, sizeof(int32_t) * slots, s));
    {
        KernelTimer timer(ctx, KID_CORNER_EIG);
        hipLaunchKernelGGL(corner_eig_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH, n), dim3(256), 0,
                           s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(corner_max_kernel, dim3(n), dim3(256), 0, s, a, (int)eig_tiles(w, h));
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_CORNER_PICK);
        hipLaunchKernelGGL(corner_pick_kernel, dim3((w - 2 + 63) / 64, (h - 2 + kPickRows - 1) / kPickRows, n), dim3(256),
                           0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_CORNER_SORT);
        hipLaunchKernelGGL(corner_split_kernel, dim3(n), dim3(1024), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(corner_sort_kernel, dim3(kBuckets, n), dim3(64 * kSortWaves), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    {
        KernelTimer timer(ctx, KID_CORNER_SELECT);
        hipLaunchKernelGGL(corner_rank_kernel, dim3(256, n), dim3(256), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        // about 32 waves per CU over the whole batch, at least 32 workgroups per image
        const unsigned per_image = (unsigned)std::min(std::max(ctx->num_cus * 32 / n_images, 32), 1024);
        // the open candidates thin out round by round (a few hundred are left after the fourth);
        // the late rounds, empty more often than not, are smaller launches
        for (int r = 0; r < kSelectRounds; ++r) {
            const unsigned blocks = r < 2 ? per_image : std::max(per_image / (r < 4 ? 2u : 8u), 32u);
            hipLaunchKernelGGL(corner_select_kernel, dim3(blocks, n), dim3(256), 0, s, a, r);
            SFM_HIP_CHECK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(corner_settle_kernel, dim3(n), dim3(64), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(corner_emit_kernel, dim3(n), dim3(1024), 0, s, a);
        SFM_HIP_CHECK(ctx, hipGetLastError());
    }
    return SFM_OK;
}

int check_corner_args(sfm_ctx* ctx, const void* img, int w, int h, size_t stride, double quality,
                      double min_distance, const void* corners, int cap_out) {
    if (!img || (!corners && cap_out > 0)) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (w < 3 || h < 3 || stride < (size_t)w || (size_t)w * h > (1u << kPosBits))
        return set_error(ctx, SFM_E_INVALID, "image must be at least 3x3, stride >= width, at most 2^26 pixels");
    // OpenCV: CV_Assert(qualityLevel > 0 && minDistance >= 0 && maxCorners >= 0)
    if (!(quality > 0) || !(min_distance >= 0) || cap_out < 0)
        return set_error(ctx, SFM_E_INVALID, "qualityLevel must be > 0, minDistance and the capacity >= 0");
    if (min_distance > kMaxDistance) return set_error(ctx, SFM_E_INVALID, "minDistance above 128 is not supported");
    return SFM_OK;
}

}  // namespace

}  // namespace sfm

using namespace sfm;

extern "C" {

size_t sfm_good_features_batch_workspace(int n_images, int w, int h, int max_candidates) {
    if (w <= 0 || h <= 0 || n_images <= 0) return 0;
    const int cap = max_candidates > 0 ? max_candidates : default_capacity(w, h);
    Carver c(nullptr);
    carve_corners(c, n_images, w, h, cap);
    return c.total();
}

size_t sfm_good_features_workspace(int w, int h, int max_candidates) {
    return sfm_good_features_batch_workspace(1, w, h, max_candidates);
}

int sfm_good_features_batch_dev(sfm_ctx* ctx, const uint8_t* d_imgs, int n_images, size_t image_stride, int w, int h,
                                size_t stride, int max_corners, double quality, double min_distance,
                                int max_candidates, float* d_corners, int cap, int32_t* d_info) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (n_images < 0 || n_images > kMaxImages) return set_error(ctx, SFM_E_INVALID, "between 0 and 64 images per call");
    if (n_images == 0) return SFM_OK;
    SFM_RETURN_IF(check_corner_args(ctx, d_imgs, w, h, stride, quality, min_distance, d_corners, cap));
    if (!d_info || max_corners < 0) return set_error(ctx, SFM_E_INVALID, "d_info is NULL or maxCorners negative");
    if (n_images > 1 && image_stride < stride * (size_t)(h - 1) + (size_t)w)
        return set_error(ctx, SFM_E_INVALID, "image stride smaller than an image");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int cand_cap = max_candidates > 0 ? (int)std::min<size_t>(max_candidates, (size_t)w * h) : default_capacity(w, h);
    cand_cap = std::min(cand_cap, (1 << kStateShift) - 1);  // ranks share a word with the state
    void* ws;
    SFM_RETURN_IF(workspace(ctx, sfm_good_features_batch_workspace(n_images, w, h, cand_cap), &ws));
    Carver c(ws);
    const CornerBuffers b = carve_corners(c, n_images, w, h, cand_cap);
    return run_corners(ctx, d_imgs, n_images, image_stride, w, h, stride, max_corners, quality, min_distance, d_corners, cap,
                       d_info, cand_cap, b);
}

int sfm_good_features_dev(sfm_ctx* ctx, const uint8_t* d_img, int w, int h, size_t stride, int max_corners,
                          double quality, double min_distance, int max_candidates, float* d_corners, int cap,
                          int32_t* d_info) {
    return sfm_good_features_batch_dev(ctx, d_img, 1, 0, w, h, stride, max_corners, quality, min_distance, max_candidates,
                                       d_corners, cap, d_info);
}

int sfm_good_features(sfm_ctx* ctx, const uint8_t* img, int w, int h, size_t stride, int max_corners, double quality,
                      double min_distance, float* corners, int cap, int* count) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (count) *count = 0;
    SFM_RETURN_IF(check_corner_args(ctx, img, w, h, stride, quality, min_distance, corners, cap));
    if (!count || max_corners < 0) return set_error(ctx, SFM_E_INVALID, "count is NULL or maxCorners negative");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    Carver sz(nullptr);
    sz.take<uint8_t>((size_t)w * h);
    sz.take<float>(2 * (size_t)std::max(cap, 1));
    sz.take<int32_t>(4);
    void* st;
    SFM_RETURN_IF(staging(ctx, sz.total(), &st));
    Carver sc(st);
    uint8_t* d_img = sc.take<uint8_t>((size_t)w * h);
    float* d_corners = sc.take<float>(2 * (size_t)std::max(cap, 1));
    int32_t* d_info = sc.take<int32_t>(4);
    SFM_RETURN_IF(upload_rows(ctx, d_img, img, stride, (size_t)w, (size_t)h));
    int32_t info[3] = {0, 0, 0};
    // first with the usual candidate capacity; an image of plateaus can need every pixel
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int cand_cap = attempt == 0 ? 0 : (int)std::min<size_t>((size_t)w * h, 0x7fffffff);
        SFM_RETURN_IF(sfm_good_features_dev(ctx, d_img, w, h, (size_t)w, max_corners, quality, min_distance, cand_cap,
                                            d_corners, cap, d_info));
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
        SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!info[2]) break;
    }
    if (getenv("SFM_CORNER_TRACE")) {  // the rounds of the selection, for tuning
        int32_t c[C_COUNT];
        Carver wc(ctx->ws);
        const CornerBuffers wb = carve_corners(wc, 1, w, h, default_capacity(w, h));
        if (hipMemcpy(c, wb.ctrl + 4, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "corners: found %d open/kept per round:", c[C_FOUND]);
            for (int r = 0; r < kSelectRounds; ++r) fprintf(stderr, " %d/%d", c[C_OPEN + r], c[C_KEPT + r]);
            fprintf(stderr, "\n");
        }
    }
    ctx->pinned_used = 0;
    if (info[2]) return set_error(ctx, SFM_E_INVALID, "candidate capacity exceeded");
    if (info[0] > 0)
        SFM_HIP_CHECK(ctx, hipMemcpy(corners, d_corners, sizeof(float) * 2 * (size_t)info[0], hipMemcpyDeviceToHost));
    *count = info[0];
    return SFM_OK;
}

int sfm_min_eigen_val_dev(sfm_ctx* ctx, const uint8_t* d_img, int w, int h, size_t stride, float* d_eig) {
    if (!ctx) return set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (!d_img || !d_eig) return set_error(ctx, SFM_E_INVALID, "NULL pointer argument");
    if (w < 3 || h < 3 || stride < (size_t)w || (size_t)w * h > 0x7fffffffu)
        return set_error(ctx, SFM_E_INVALID, "image must be at least 3x3, stride >= width, below 2^31 pixels");
    SFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* ws;
    SFM_RETURN_IF(workspace(ctx, 256, &ws));
    CornerArgs a{};
    a.img = d_img;
    a.w = w;
    a.h = h;
    a.stride = stride;
    a.n_images = 1;
    a.eig = d_eig;
    a.max_bits = static_cast<unsigned*>(ws);
    a.counters = reinterpret_cast<int32_t*>(ws) + 4;
    a.tile_max = nullptr;   // only the response image is wanted
    KernelTimer timer(ctx, KID_CORNER_EIG);
    hipLaunchKernelGGL(corner_eig_kernel, dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH), dim3(256), 0,
                       ctx->stream, a);
    SFM_HIP_CHECK(ctx, hipGetLastError());
    return SFM_OK;
}

}  // extern "C"
