// Row-norm streaming variants (developer probe for l2_pack_kernel / l2_narrow_kernel):
// 40 000 rows of 128 floats -> ||row||^2 per row, (and fp16 copies), timed per launch.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));

// A: the library's shape: 64 lanes x 8 B per row, 8 rows per wave, shuffles through LDS
template <int ROWS, bool HALF, int MAXMODE>
__global__ __launch_bounds__(256) void norm_a(const float* src, int rows, float* norms, _Float16* half, float* gmax) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    floatx2 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, rows - 1);
        v[r] = *reinterpret_cast<const floatx2*>(src + (size_t)row * 128 + 2 * lane);
    }
    float largest = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        float acc = v[r].x * v[r].x;
        acc += v[r].y * v[r].y;
        if (HALF) {
            halfx2 h;
            h.x = (_Float16)(v[r].x * 4.f);
            h.y = (_Float16)(v[r].y * 4.f);
            *reinterpret_cast<halfx2*>(half + (size_t)row * 128 + 2 * lane) = h;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) norms[row] = acc;
        largest = fmaxf(largest, acc);
    }
    if (MAXMODE == 1) {
        __shared__ float s_max[4];
        if (lane == 0) s_max[threadIdx.x >> 6] = largest;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) largest = fmaxf(largest, s_max[w]);
            if (largest > *reinterpret_cast<volatile float*>(gmax)) atomicMax(reinterpret_cast<unsigned*>(gmax), __float_as_uint(largest));
        }
    } else if (MAXMODE == 2) {   // 64 slots
        __shared__ float s_max[4];
        if (lane == 0) s_max[threadIdx.x >> 6] = largest;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) largest = fmaxf(largest, s_max[w]);
            float* slot = gmax + (blockIdx.x & 63);
            if (largest > *reinterpret_cast<volatile float*>(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(largest));
        }
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_addf(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// B: 32 lanes x 16 B per row (two rows per wave instruction), ROWS rows per half wave, DPP sums
template <int ROWS, bool HALF, int MAXMODE>
__global__ __launch_bounds__(256) void norm_b(const float* src, int rows, float* norms, _Float16* half, float* gmax, float* wgmax) {
    const int lane = threadIdx.x & 63, sub = lane & 31, hw = lane >> 5;
    const int row0 = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + hw) * ROWS;
    floatx4 v[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, rows - 1);
        v[r] = *reinterpret_cast<const floatx4*>(src + (size_t)row * 128 + 4 * sub);
    }
    float largest = 0.f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r;
        float acc = v[r].x * v[r].x;
        acc += v[r].y * v[r].y;
        acc += v[r].z * v[r].z;
        acc += v[r].w * v[r].w;
        if (HALF && row < rows) {
            halfx4 h;
            h.x = (_Float16)(v[r].x * 4.f);
            h.y = (_Float16)(v[r].y * 4.f);
            h.z = (_Float16)(v[r].z * 4.f);
            h.w = (_Float16)(v[r].w * 4.f);
            *reinterpret_cast<halfx4*>(half + (size_t)row * 128 + 4 * sub) = h;
        }
        acc = dpp_addf<0xB1>(acc);    // quad_perm [1,0,3,2]
        acc = dpp_addf<0x4E>(acc);    // quad_perm [2,3,0,1]
        acc = dpp_addf<0x141>(acc);   // row_half_mirror
        acc = dpp_addf<0x140>(acc);   // row_mirror: sum of 16 lanes in each
        acc += __shfl_xor(acc, 16);   // the other row of the half wave
        if (sub == 0 && row < rows) norms[row] = acc;
        largest = fmaxf(largest, row < rows ? acc : 0.f);
    }
    if (MAXMODE) {
        __shared__ float s_max[8];
        if (sub == 0) s_max[threadIdx.x >> 5] = largest;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 8; ++w) largest = fmaxf(largest, s_max[w]);
            float* slot = gmax + (MAXMODE == 2 || MAXMODE == 4 ? (blockIdx.x & 63) : 0);
            if (MAXMODE == 5) wgmax[blockIdx.x] = largest;
            else if (MAXMODE >= 3) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(largest));
            else if (largest > *reinterpret_cast<volatile float*>(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(largest));
        }
    }
}

int main() {
    const int rows = 40000;
    float *src, *norms, *gmax;
    _Float16* half;
    hipMalloc(&src, (size_t)rows * 128 * 4);
    hipMalloc(&half, (size_t)rows * 128 * 2);
    hipMalloc(&norms, rows * 4);
    hipMalloc(&gmax, 64 * 4);
    float* wgmax;
    hipMalloc(&wgmax, 8192 * 4);
    std::vector<float> h((size_t)rows * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time_it = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, s);
            for (int k = 0; k < 100; ++k) {
                hipMemsetAsync(gmax, 0, 256, s);
                launch();
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        float n0;
        hipMemcpy(&n0, norms + 12345, 4, hipMemcpyDeviceToHost);
        printf("%-44s %.2f us per launch (with a 256-byte memset)   norms[12345]=%.4f\n", name, best * 10.f, n0);
    };
    time_it("memset only", [&] {});
#define RUN_A(R, H, M) time_it("A rows=" #R " half=" #H " max=" #M, [&] { hipLaunchKernelGGL((norm_a<R, H, M>), dim3((rows + 4 * R - 1) / (4 * R)), dim3(256), 0, s, src, rows, norms, half, gmax); })
#define RUN_B(R, H, M) time_it("B rows=" #R " half=" #H " max=" #M, [&] { hipLaunchKernelGGL((norm_b<R, H, M>), dim3((rows + 8 * R - 1) / (8 * R)), dim3(256), 0, s, src, rows, norms, half, gmax, wgmax); })
    RUN_A(8, false, 0);
    RUN_A(8, false, 1);
    RUN_A(8, false, 2);
    RUN_A(4, false, 2);
    RUN_A(16, false, 2);
    RUN_A(8, true, 2);
    RUN_B(4, false, 0);
    RUN_B(4, false, 1);
    RUN_B(4, false, 2);
    RUN_B(8, false, 2);
    RUN_B(2, false, 2);
    RUN_B(4, true, 2);
    RUN_B(8, true, 2);
    RUN_B(2, true, 2);
    RUN_B(4, false, 3);
    RUN_B(4, false, 4);
    RUN_B(4, false, 5);
    RUN_B(8, false, 3);
    RUN_B(8, false, 5);
    RUN_B(4, true, 5);
    RUN_B(8, true, 5);
    RUN_B(4, true, 0);
    RUN_B(8, true, 0);
    return 0;
}
