// Microbenchmark: sustained issue rate of individual VALU instructions on gfx950.
// Each wave runs a long unrolled stream of independent instructions of one kind;
// the grid fills every SIMD with `WAVES` waves.  Prints wave-instructions/s and
// the implied cycles per wave64 instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 32;  // 8 independent chains x 4

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, unsigned seed) {
    unsigned v[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * (i + 1) + threadIdx.x; f[i] = (float)v[i]; }
    unsigned k = seed | 1u;
    float fk = 1.0001f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(k));
                if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 3) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 7) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 8) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 10) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[i]));
                if (OP == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 13) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"((double)fk));
                if (OP == 14) asm volatile("v_dot4_u32_u8 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 15) asm volatile("v_dot2_u32_u16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 16) asm volatile("v_dot2c_i32_i16 %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 18) asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(v[i]) : "v"(k));
                if (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 20) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 21) asm volatile("v_mad_i32_i16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 22) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(k) : "vcc");
                if (OP == 23) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 24) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 25) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(v[i]));
                if (OP == 26) asm volatile("v_ashrrev_i32 %0, 9, %0" : "+v"(v[i]));
                if (OP == 27) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 28) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 29) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"((double)fk));
                if (OP == 30) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&v[i & 6]) : "v"((double)fk));
                if (OP == 31) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"((double)fk));
                if (OP == 32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(k) : );
                if (OP == 33) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(k));
                if (OP == 34) asm volatile("s_nop 0\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
                if (OP == 35) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 36) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 37) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 38) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 65) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fk), "v"(f[(i + 1) & 7]));
                if (OP == 66) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 39) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 40) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 41) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fk), "v"(f[(i + 1) & 7]));
                if (OP == 42) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 43) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fk));
                if (OP == 44) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]));
                if (OP == 45) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 46) asm volatile("v_not_b32 %0, %0" : "+v"(v[i]));
                if (OP == 47) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 48) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 49) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 50) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i]));
                if (OP == 51) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 52) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "v"(seed));
                if (OP == 53) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                // mixed streams: one full-rate and one half-rate instruction per step
                if (OP == 54) asm volatile("v_xor_b32 %0, %0, %2\n\tv_bcnt_u32_b32 %1, %2, %1" : "+v"(v[i]), "+v"(f[i]) : "v"(k));
                if (OP == 55) asm volatile("v_xor_b32 %0, %0, %2\n\tv_xor_b32 %1, %1, %2" : "+v"(v[i]), "+v"(f[i]) : "v"(k));
                if (OP == 56) asm volatile("v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %2, %1" : "+v"(v[i]), "+v"(f[i]) : "v"(k));
                if (OP == 57) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 58) asm volatile("v_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 59) asm volatile("v_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(k));
                if (OP == 60) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(v[i]) : "v"(k) : "vcc");
                // run-length patterns of the Hamming inner loop: N xors then N popcount-accumulates
                if (OP == 62 && (i & 3) == 0) asm volatile(
                    "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4\n\t"
                    "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %4, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\tv_bcnt_u32_b32 %3, %4, %3"
                    : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(k));
                if (OP == 63 && (i & 3) == 0) asm volatile(
                    "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\t"
                    "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %4, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\tv_bcnt_u32_b32 %3, %4, %3\n\tv_xor_b32 %3, %3, %4"
                    : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(k));
                if (OP == 64 && i == 0) asm volatile(
                    "v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\t"
                    "v_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\t"
                    "v_bcnt_u32_b32 %0, %8, %0\n\tv_bcnt_u32_b32 %1, %8, %1\n\tv_bcnt_u32_b32 %2, %8, %2\n\tv_bcnt_u32_b32 %3, %8, %3\n\t"
                    "v_bcnt_u32_b32 %4, %8, %4\n\tv_bcnt_u32_b32 %5, %8, %5\n\tv_bcnt_u32_b32 %6, %8, %6\n\tv_bcnt_u32_b32 %7, %8, %7"
                    : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(k));
                if (OP == 61) asm volatile("v_xor_b32 %0, %0, %2\n\tv_min_u32 %1, %2, %1" : "+v"(v[i]), "+v"(f[i]) : "v"(k));
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i] + (unsigned)f[i];
    if (acc == 0x12345678u) out[0] = acc;
}

// N xors followed by N popcount-accumulates, each on its own register (the shape of the
// Hamming inner loop with all of a train row's xors issued before its popcounts).
template <int N>
__global__ __launch_bounds__(256) void pattern_kernel(unsigned* out, unsigned seed) {
    unsigned t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = seed * (i + 1) + threadIdx.x;
    unsigned k = seed | 1u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / N; ++u) {
#pragma unroll
            for (int i = 0; i < N; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(t[i]) : "v"(k));
#pragma unroll
            for (int i = 0; i < N; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(t[i]) : "v"(k));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc += t[i];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int N>
int run_pattern(int waves_per_simd, unsigned* d_out) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    pattern_kernel<N><<<blocks, 256>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) pattern_kernel<N><<<blocks, 256>>>(d_out, 7);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double pairs = 5.0 * blocks * 4 * (double)ITERS * 32;  // (xor, bcnt) pairs per wave
    const double per_simd_per_s = pairs / (ms * 1e-3) / 1024.0;
    printf("%dxor+%dbcnt runs  waves/SIMD=%d  %.3f ms  %.2f cycles per (xor,bcnt) pair @2.4GHz\n", N, N,
           waves_per_simd, ms / 5, 2.4e9 / per_simd_per_s);
    return 0;
}

template <int OP>
int run(const char* name, int waves_per_simd, unsigned* d_out) {
    const int blocks = 256 * waves_per_simd;  // 256 CUs, 4 waves per block -> one wave per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(d_out, 7);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) rate_kernel<OP><<<blocks, 256>>>(d_out, 7);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = 5.0 * blocks * 4 * (double)ITERS * UNROLL;  // wave-instructions
    const double per_simd_per_s = instr / (ms * 1e-3) / 1024.0;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f T lane-ops/s  %.2f cycles/wave-instr @2.4GHz\n", name,
           waves_per_simd, ms / 5, instr * 64 / (ms * 1e-3) / 1e12, 2.4e9 / per_simd_per_s);
    return 0;
}

int main(int argc, char**) {
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    const bool extra_only = argc > 1;
    for (int w : {2, 4, 8}) {
        run<65>("v_min3_f32", w, d_out);
        run<66>("v_pk_min_f16", w, d_out);
        run<39>("v_min_f32", w, d_out);
        run<40>("v_max_f32", w, d_out);
        run<41>("v_med3_f32", w, d_out);
        run<42>("v_fmac_f32", w, d_out);
        run<43>("v_sub_f32", w, d_out);
        run<44>("v_lshrrev_b32", w, d_out);
        run<45>("v_xnor_b32", w, d_out);
        run<46>("v_not_b32", w, d_out);
        run<47>("v_min_i32", w, d_out);
        run<57>("v_max_i32", w, d_out);
        run<48>("v_pk_min_u16", w, d_out);
        run<49>("v_pk_add_u16", w, d_out);
        run<58>("v_min_u16", w, d_out);
        run<59>("v_add_u16", w, d_out);
        run<50>("v_cvt_f32_u32", w, d_out);
        run<51>("v_lshl_add_u32", w, d_out);
        run<52>("v_or3_b32", w, d_out);
        run<53>("v_subrev_u32", w, d_out);
        run<60>("v_add_co_u32", w, d_out);
        run<54>("xor+bcnt (2 instr)", w, d_out);
        run<55>("xor+xor (2 instr)", w, d_out);
        run<56>("bcnt+bcnt (2 instr)", w, d_out);
        run<61>("xor+min_u32 (2 instr)", w, d_out);
        run<62>("4xor+4bcnt (per pair x4)", w, d_out);
        run<63>("3xor+4bcnt+1xor (per pair x4)", w, d_out);
        run<64>("8xor+8bcnt (per pair x8)", w, d_out);
    }
    for (int w : {2, 5, 8}) {
        run_pattern<1>(w, d_out);
        run_pattern<4>(w, d_out);
        run_pattern<8>(w, d_out);
        run_pattern<16>(w, d_out);
        run_pattern<32>(w, d_out);
    }
    if (extra_only) return 0;
    for (int w : {4, 8}) {
        run<14>("v_dot4_u32_u8", w, d_out);
        run<15>("v_dot2_u32_u16", w, d_out);
        run<16>("v_dot2c_i32_i16", w, d_out);
        run<17>("v_perm_b32", w, d_out);
        run<18>("v_alignbyte_b32", w, d_out);
        run<19>("v_mul_lo_u32", w, d_out);
        run<20>("v_mul_u32_u24", w, d_out);
        run<21>("v_mad_i32_i16", w, d_out);
        run<22>("v_cmp_lt_u32", w, d_out);
        run<23>("v_sub_u32", w, d_out);
        run<24>("v_or_b32", w, d_out);
        run<25>("v_lshlrev_b32", w, d_out);
        run<26>("v_ashrrev_i32", w, d_out);
        run<27>("v_max_u32", w, d_out);
        run<28>("v_add_f32", w, d_out);
        run<29>("v_mul_f64", w, d_out);
        run<30>("v_add_f64", w, d_out);
        run<31>("v_pk_fma_f32", w, d_out);
        run<32>("v_cndmask_b32", w, d_out);
        run<33>("v_mov_b32", w, d_out);
        run<34>("nop+v_add_dpp", w, d_out);
        run<35>("v_min3_u32", w, d_out);
        run<36>("v_xad_u32", w, d_out);
        run<37>("v_bfi_b32", w, d_out);
        run<38>("v_sad_u8", w, d_out);
    }
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", w, d_out);
        run<1>("v_bcnt_u32_b32", w, d_out);
        run<2>("v_add_u32", w, d_out);
        run<3>("v_med3_u32", w, d_out);
        run<4>("v_lshl_or_b32", w, d_out);
        run<7>("v_min_u32", w, d_out);
        run<6>("v_mad_u32_u24", w, d_out);
        run<8>("v_mad_i32_i24", w, d_out);
        run<9>("v_add3_u32", w, d_out);
        run<10>("v_bfe_u32", w, d_out);
        run<12>("v_and_b32", w, d_out);
        run<5>("v_fma_f32", w, d_out);
        run<11>("v_mul_f32", w, d_out);
        run<13>("v_fma_f64", w, d_out);
    }
    return 0;
}
