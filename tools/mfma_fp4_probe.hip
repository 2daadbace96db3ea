// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands: checks that a +-1
// contraction over K = 64 with block scales 2^6 x 2^6 is exact and that the A and B operand
// layouts pair the same k in the same (lane half, nibble) position.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_fp4_probe.hip -o tools/mfma_fp4_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void probe(const uint32_t* a_frag, const uint32_t* b_frag, float* out, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    intx8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        a[i] = (int)a_frag[lane * 4 + i];
        b[i] = (int)b_frag[lane * 4 + i];
    }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 1000.f + r;
    // cbsz = 4, blgp = 4: both operands fp4 (E2M1)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
    std::vector<int> A(32 * 64), B(32 * 64);  // bits per (row, k) and (col, k)
    srand(7);
    for (auto& v : A) v = rand() & 1;
    for (auto& v : B) v = rand() & 1;
    std::vector<uint32_t> af(64 * 4, 0), bf(64 * 4, 0);
    for (int lane = 0; lane < 64; ++lane) {
        const int rc = lane & 31, kb = (lane >> 5) * 32;
        for (int j = 0; j < 32; ++j) {
            const uint32_t na = A[rc * 64 + kb + j] ? 0xAu : 0x2u;  // bit 1 -> -1, bit 0 -> +1
            const uint32_t nb = B[rc * 64 + kb + j] ? 0x2u : 0xAu;  // negated on the query side
            af[lane * 4 + j / 8] |= na << (4 * (j % 8));
            bf[lane * 4 + j / 8] |= nb << (4 * (j % 8));
        }
    }
    uint32_t *da, *db;
    float* dout;
    hipMalloc(&da, af.size() * 4);
    hipMalloc(&db, bf.size() * 4);
    hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, af.data(), af.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bf.data(), bf.size() * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int variant = 0; variant < 2; ++variant) {
        const int sa = variant ? 133 : 127, sb = variant ? 133 : 127;  // E8M0: 2^6 or 2^0
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout, sa, sb);
        std::vector<float> out(64 * 16);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        const double scale = variant ? 4096.0 : 1.0;
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int h = 0;
                for (int k = 0; k < 64; ++k) h += A[row * 64 + k] != B[col * 64 + k];
                // sum a * (-b) = (#differing - #equal) = 2h - 64
                const double want = 1000.0 + r + scale * (2 * h - 64);
                if (out[lane * 16 + r] != (float)want) {
                    if (bad < 5) printf("variant %d lane %d r %d got %g want %g\n", variant, lane, r, out[lane * 16 + r], want);
                    ++bad;
                }
            }
        printf("variant %d (scales 2^%d): %d mismatches of 1024\n", variant, variant ? 6 : 0, bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
