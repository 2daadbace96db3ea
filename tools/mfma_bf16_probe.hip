// Checks the operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 against a host product:
// lane l supplies A[l % 32][8 (l / 32) + e] and B[8 (l / 32) + e][l % 32], e = 0..7; accumulator
// register r of lane l is D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {  // A[32][16], B[16][32], D[32][32]
    const int l = threadIdx.x, i = l % 32, kb = l / 32;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[i * 16 + 8 * kb + e];
        b[e] = (__bf16)B[(8 * kb + e) * 32 + i];
    }
    floatx16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * kb + r % 4) * 32 + i] = acc[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    srand(1);
    for (float& v : hA) v = (float)(rand() % 17 - 8);
    for (float& v : hB) v = (float)(rand() % 17 - 8);
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j];
            bad += s != hD[i * 32 + j];
        }
    printf("mismatches: %d of 1024\n", bad);
    return bad != 0;
}
