// microbench.hip -- issue-rate microbenchmarks on MI355X for the instruction classes the path tracer uses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 16384
#define UNROLL 16

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a0, float b0, unsigned u0)
{
    float x[8];
    unsigned y[8];
    double z[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = a0 + i + threadIdx.x; y[i] = u0 + i * 77 + threadIdx.x; }
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = a0 + i;
    float b = b0;
    asm volatile("s_mov_b64 vcc, 0x5555\n\ts_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
                if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
                if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 7) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : );
                if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(b) : );
                if (OP == 15) { if (i & 1) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(x[i]), "v"(b) : "s20", "s21"); else asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" : : "v"(x[i]), "v"(b) : "s22", "s23"); }
                if (OP == 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 17) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(y[i]));
                if (OP == 18) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
                if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 20) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
                if (OP == 21) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 22) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(y[i]));
                if (OP == 23) asm volatile("v_sub_f32 %0, s20, %0" : "+v"(x[i]));
                if (OP == 24) asm volatile("v_add_f32 %0, 0x3dcccccd, %0" : "+v"(x[i]));
                if (OP == 25) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(x[i]));
                if (OP == 26) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 27) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 28) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(b));
                if (OP == 29) asm volatile("v_and_b32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 30) asm volatile("v_fma_f32 %0, %0, s20, %0" : "+v"(x[i]));
                if (OP == 31) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(x[i]));
                if (OP == 32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[i]));
                if (OP == 33) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 34) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\tv_nop\n\tv_nop" : : "v"(x[i]), "v"(b) : "vcc");
                if (OP == 35) asm volatile("v_nop");
                if (OP == 36) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 37) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 38) asm volatile("v_alignbit_b32 %0, %0, %0, 15" : "+v"(y[i]));
                if (OP == 39) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 40) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 41) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(u0));
                if (OP == 42) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
                if (OP == 43) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : );
                if (OP == 44) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(z[i & 3]) : "v"(y[i]), "v"(u0) : "s20", "s21");
                if (OP == 45) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
                if (OP == 46) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 47) asm volatile("v_min_u32 %0, %0, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 48) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 49) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(y[i]), "v"(u0) : "vcc");
                if (OP == 50) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 51) asm volatile("v_cmp_lt_u64_e32 vcc, %0, %1" : : "v"(z[i & 3]), "v"(z[(i + 1) & 3]) : "vcc");
                if (OP == 52) asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(y[i]));
                if (OP == 53) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 55) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(y[i]) : "v"(u0) : "vcc");
                if (OP == 56) asm volatile("v_readfirstlane_b32 s22, %0" : : "v"(y[i]) : "s22");
                if (OP == 57) asm volatile("v_mbcnt_lo_u32_b32 %0, s20, %0" : "+v"(y[i]));
                if (OP == 58) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(u0));
                if (OP == 59) asm volatile("v_cmp_eq_u32_e64 s[22:23], %0, %1" : : "v"(y[i]), "v"(u0) : "s22", "s23");
                if (OP == 60) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 61) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 62) asm volatile("v_cndmask_b32_e64 %0, 0, %1, s[20:21]" : "+v"(x[i]) : "v"(b) : );
                if (OP == 63) asm volatile("v_add_u32 %0, 0x38d1b718, %0" : "+v"(y[i]));
                if (OP == 64) asm volatile("v_add_u32 %0, 5, %0" : "+v"(y[i]));
                if (OP == 65) asm volatile("v_sub_f32_e64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 66) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y[i]));
                if (OP == 68) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_add_f32 %2, %2, %3" : "+v"(y[i]) : "v"(u0), "v"(x[i]), "v"(b) : "vcc");
                if (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(y[i]) : "v"(u0));
                if (OP == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(b) : "vcc");
                if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(y[i]) : "v"(u0));
            }
        }
        if (OP == 11) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[2 * i]) : "v"(*(double*)&x[0])); asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[2 * i]) : "v"(*(double*)&x[0]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[2 * i]) : "v"(*(double*)&x[0])); asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&x[2 * i]) : "v"(*(double*)&x[0])); }
        }
        if (OP == 12) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(z[i]) : "v"(z[0]));
        }
        if (OP == 13) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * i]) : "v"(*(double*)&x[0]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + (float)y[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (float)z[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char* name, int waves_per_simd, float* d)
{
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; blocks/CU = waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 1.0f, 1.0001f, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 1.0f, 1.0001f, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * ITER * UNROLL;           // wave-instructions
    double per_simd_per_s = winstr / (ms * 1e-3) / (256.0 * 4);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f G wave-instr/s/SIMD  => %.2f cycles/wave-instr @2.4GHz  lane-ops %.1f T/s\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, winstr * 64 / (ms * 1e-3) / 1e12);
}

int main()
{
    float* d; hipMalloc(&d, 4);
    for (int w : {4}) {
        run<0>("v_fma_f32", w, d); run<1>("v_mul_f32", w, d); run<2>("v_add_f32", w, d); run<3>("v_mul_lo_u32", w, d);
        run<4>("v_sqrt_f32", w, d); run<5>("v_rcp_f32", w, d); run<6>("v_xor_b32", w, d); run<7>("v_cndmask_b32", w, d);
        run<8>("v_mad_u32_u24", w, d); run<9>("v_cmp_lt_f32", w, d); run<10>("v_lshl_add_u32", w, d);
        run<11>("v_pk_fma_f32", w, d); run<12>("v_fma_f64", w, d); run<13>("v_pk_mul_f32", w, d);
        run<14>("v_cndmask_e64", w, d); run<15>("v_cmp_e64 sgpr", w, d); run<16>("v_add_u32", w, d); run<17>("v_lshrrev_b32", w, d);
        run<18>("v_cvt_f32_u32", w, d); run<19>("v_max_f32", w, d); run<20>("cmp+nop+cndmask", w, d); run<21>("v_min3_f32", w, d);
        run<22>("v_xor_sdwa", w, d); run<23>("v_sub_f32 sgpr", w, d);
        run<24>("v_add_f32 literal", w, d); run<25>("v_mul_f32 inline2.0", w, d); run<26>("v_min_f32", w, d); run<27>("v_med3_f32", w, d);
        run<28>("v_mov_b32", w, d); run<29>("v_and_b32", w, d); run<30>("v_fma_f32 sgpr", w, d); run<31>("v_mul_f32 sgpr", w, d);
        run<32>("v_cvt_f32_i32", w, d); run<33>("v_sub_u32", w, d); run<34>("v_cmp+2 v_nop (x3)", w, d); run<35>("v_nop", w, d);
        run<36>("v_bfi_b32", w, d); run<37>("v_xad_u32", w, d); run<38>("v_alignbit_b32", w, d); run<39>("v_mul_u32_u24", w, d);
        run<40>("v_mul_hi_u32", w, d); run<41>("v_ldexp_f32", w, d); run<42>("v_rsq_f32", w, d);
        run<44>("v_mad_u64_u32", w, d); run<45>("v_cvt_i32_f32", w, d); run<46>("v_add3_u32", w, d); run<47>("v_min_u32", w, d);
        run<48>("v_and_or_b32", w, d); run<49>("v_cmp_lt_u32", w, d); run<50>("v_fmac_f32", w, d);
        run<51>("v_cmp_lt_u64", w, d); run<52>("v_bfe_u32", w, d); run<53>("v_lshl_or_b32", w, d); run<55>("v_add_co_u32", w, d);
        run<56>("v_readfirstlane", w, d); run<57>("v_mbcnt_lo", w, d); run<58>("v_min3_u32", w, d); run<59>("v_cmp_eq_u32_e64", w, d);
        run<60>("v_add_f32 |abs|", w, d); run<61>("v_mul_f32 -neg", w, d); run<62>("v_cndmask 0,v,s", w, d); run<63>("v_add_u32 literal", w, d);
        run<64>("v_add_u32 inline5", w, d); run<65>("v_sub_f32_e64", w, d); run<66>("v_mov_dpp row_shr", w, d);
        run<68>("cmp_u32+add_f32 (x2)", w, d);
        printf("\n");
    }
    return 0;
}
