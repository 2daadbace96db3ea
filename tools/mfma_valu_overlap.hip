// How many vector-ALU instructions of one wave fit under its own matrix instructions on gfx950?
// One workgroup of four waves (one per SIMD) runs a loop of four independent
// v_mfma_f32_32x32x16_bf16, each followed by N VALU instructions of one kind, and reports the
// cycles per MFMA slot.  With WAVES = 8 two waves share each SIMD.
//   kind 0: v_add_f32 (two independent chains)      kind 1: v_med3_f32 chains
//   kind 2: v_accvgpr_read of registers no MFMA writes
//   kind 3: v_accvgpr_read of the accumulator the MFMA four slots back wrote
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND, int N, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, long long* cycles, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(float)(threadIdx.x + e);
        b[e] = (__bf16)(float)(threadIdx.x * 3 + e);
    }
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, y = 0.5f;
    float spare[8];
    for (int i = 0; i < 8; ++i) {
        spare[i] = i;
        asm volatile("" : "+a"(spare[i]));
    }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) {
                    if (n & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(y));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(y));
                } else if (KIND == 1) {
                    if (n & 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(x2));
                    else asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(y), "v"(x2));
                } else if (KIND == 2) {
                    float t;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(spare[n % 8]));
                    asm volatile("" ::"v"(t));
                } else {
                    float t = acc[(i + 1) & 3][n % 16];  // written three slots ago
                    asm volatile("" ::"v"(t));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(t));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int N, bool MFMA>
void run(int waves, float* out, long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<KIND, N, MFMA><<<1, 64 * waves>>>(out, cyc, iters);
    (void)hipEventRecord(e0, 0);
    k<KIND, N, MFMA><<<1, 64 * waves>>>(out, cyc, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c;
    (void)hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("kind %d N %2d mfma %d waves/simd %d: %.1f counter ticks, %.2f ns per slot\n", KIND, N, (int)MFMA,
           waves / 4, (double)c / (4.0 * iters), 1e6 * ms / (4.0 * iters));
}

template <int KIND>
void sweep(float* out, long long* cyc) {
    for (int waves : {4, 8}) {
        run<KIND, 0, true>(waves, out, cyc);
        run<KIND, 2, true>(waves, out, cyc);
        run<KIND, 4, true>(waves, out, cyc);
        run<KIND, 6, true>(waves, out, cyc);
        run<KIND, 8, true>(waves, out, cyc);
        run<KIND, 12, true>(waves, out, cyc);
        run<KIND, 4, false>(waves, out, cyc);
        run<KIND, 8, false>(waves, out, cyc);
        run<KIND, 12, false>(waves, out, cyc);
    }
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 1024 * sizeof(float));
    hipMalloc(&cyc, 64);
    sweep<0>(out, cyc);
    sweep<1>(out, cyc);
    sweep<2>(out, cyc);
    sweep<3>(out, cyc);
    return 0;
}
