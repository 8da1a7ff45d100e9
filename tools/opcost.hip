// per-opcode throughput on gfx950: 8 independent instances per iteration, N waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float a, float b)
{
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3f + j + 1.0f;
    float2 p[4];
    for (int j = 0; j < 4; j++) p[j] = make_float2(x[2 * j], x[2 * j + 1]);
    int sacc = 0;
    for (int i = 0; i < N_IT; i++)
    {
#define OP1(n, INS) asm volatile(INS " %0, %0, %1" : "+v"(x[n]) : "v"(a));
#define OPU(n, INS) asm volatile(INS " %0, %0" : "+v"(x[n]));
        if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b)); }
#define U8(INS) asm volatile(INS " %0, %0\n " INS " %1, %1\n " INS " %2, %2\n " INS " %3, %3\n " INS " %4, %4\n " INS " %5, %5\n " INS " %6, %6\n " INS " %7, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#define B8(INS) asm volatile(INS " %0, %0, %8\n " INS " %1, %1, %8\n " INS " %2, %2, %8\n " INS " %3, %3, %8\n " INS " %4, %4, %8\n " INS " %5, %5, %8\n " INS " %6, %6, %8\n " INS " %7, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a));
#define T8(INS) asm volatile(INS " %0, %0, %8, %9\n " INS " %1, %1, %8, %9\n " INS " %2, %2, %8, %9\n " INS " %3, %3, %8, %9\n " INS " %4, %4, %8, %9\n " INS " %5, %5, %8, %9\n " INS " %6, %6, %8, %9\n " INS " %7, %7, %8, %9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
        if (MODE == 1) { B8("v_mul_f32") }
        if (MODE == 2) { B8("v_add_f32") }
        if (MODE == 3) { U8("v_rcp_f32") }
        if (MODE == 4) { U8("v_sqrt_f32") }
        if (MODE == 5) { U8("v_exp_f32") }
        if (MODE == 6) { U8("v_floor_f32") }
        if (MODE == 7) { U8("v_cvt_i32_f32") }
        if (MODE == 8) { B8("v_max_f32") }
        if (MODE == 9) { B8("v_min_i32") }
        if (MODE == 10) { B8("v_add_u32") }
        if (MODE == 11) { T8("v_bfi_b32") }
        if (MODE == 12) { B8("v_xor_b32") }
        if (MODE == 13) { B8("v_mul_lo_u32") }
        if (MODE == 14) { T8("v_add_lshl_u32") }
        if (MODE == 15) { T8("v_div_fixup_f32") }
        if (MODE == 16) { asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a) : "vcc"); }
        if (MODE == 17) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a) : "vcc"); }
        if (MODE == 18) { // cmp + cndmask pairs (4 pairs = 8 instrs)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a) : "vcc"); }
        if (MODE == 19) { // div_scale (writes vcc)
            asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %9\n v_div_scale_f32 %1, vcc, %1, %8, %9\n v_div_scale_f32 %2, vcc, %2, %8, %9\n v_div_scale_f32 %3, vcc, %3, %8, %9\n v_div_scale_f32 %4, vcc, %4, %8, %9\n v_div_scale_f32 %5, vcc, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %9\n v_div_scale_f32 %7, vcc, %7, %8, %9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc"); }
        if (MODE == 20) { asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b) : "vcc"); }
        if (MODE == 21) { // readlane to sgpr x8
            int s0, s1, s2, s3, s4, s5, s6, s7;
            asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 3\n v_readlane_b32 %2, %10, 3\n v_readlane_b32 %3, %11, 3\n v_readlane_b32 %4, %12, 3\n v_readlane_b32 %5, %13, 3\n v_readlane_b32 %6, %14, 3\n v_readlane_b32 %7, %15, 3" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
            sacc += s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7; }
        if (MODE == 22) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"); }
        if (MODE == 23) { asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(make_float2(a, a)), "v"(make_float2(b, b))); }
        if (MODE == 24) { asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(make_float2(a, a))); }
        if (MODE == 25) { T8("v_med3_f32") }
        if (MODE == 26) { // dependent fma chain, 8 deep on one register
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b)); }
        if (MODE == 27) { // dependent rcp -> fma chain
            asm volatile("v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b)); }
        if (MODE == 28) { B8("v_mul_f32_e64") }
        if (MODE == 29) { asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a)); }
    }
    float s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    for (int j = 0; j < 4; j++) s += p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)sacc;
}
template <int MODE> void run(const char* name)
{
    float* d;
    hipMalloc(&d, 256 * 256 * 16 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-18s", name);
    for (int b : {1, 2, 4})
    {
        int grid = 256 * b;
        k<MODE><<<grid, 256>>>(d, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(d, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double instr = grid * 4.0 * N_IT * 8.0;
        double simdcycles = ms * 1e-3 * 2.4e9 * 1024;
        printf("  %dw: %5.2f", b, simdcycles / instr);
    }
    printf("   SIMD-cycles per wave64 instruction\n");
    hipFree(d);
}
int main()
{
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<2>("v_add_f32"); run<28>("v_mul_f32_e64"); run<29>("v_mov_b32");
    run<23>("v_pk_fma_f32"); run<24>("v_pk_mul_f32");
    run<3>("v_rcp_f32"); run<4>("v_sqrt_f32"); run<5>("v_exp_f32"); run<6>("v_floor_f32"); run<7>("v_cvt_i32_f32");
    run<8>("v_max_f32"); run<25>("v_med3_f32"); run<9>("v_min_i32"); run<10>("v_add_u32"); run<11>("v_bfi_b32"); run<12>("v_xor_b32");
    run<13>("v_mul_lo_u32"); run<14>("v_add_lshl_u32"); run<15>("v_div_fixup_f32"); run<19>("v_div_scale_f32"); run<20>("v_div_fmas_f32");
    run<16>("v_cmp_lt_f32"); run<17>("v_cndmask_b32"); run<18>("cmp+cndmask x4"); run<21>("v_readlane_b32"); run<22>("s_nop 0");
    run<26>("dep fma chain"); run<27>("dep rcp+fma chain");
}
