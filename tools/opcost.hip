// tools/opcost.hip — VALU issue cost per opcode on gfx950, measured so that neither loop overhead nor an assumed clock can
// pollute it (round-1's version timed 8-instruction loop bodies against an assumed 2.4 GHz):
//   * 128 independent instructions per loop iteration in ONE asm statement (16 registers x 8; a register is reused every
//     16 instructions, far beyond the dependent-issue latency), 4096 iterations (1 - 4 ms per launch: the ramp of the
//     dispatch is small against it);
//   * exactly W = 1, 2, 4, 8 waves on every SIMD: one 256-thread workgroup puts one wave on each of a CU's 4 SIMDs, the
//     dynamic LDS size admits exactly W workgroups per CU, and the grid is W x (number of CUs);
//   * cycles from the shader clock itself: every wave brackets its loop with s_memtime (tick = shader cycle,
//     MI355X_MICROARCH.md "Per-instruction cycle constants"); cycles per instruction per SIMD = mean wave delta / (W x
//     instructions). The wall time of the launch (hipEvents) then gives the EFFECTIVE clock = cycles / time.
// Output: one row per opcode, for each W: SIMD-cycles per wave64 instruction (in-kernel clock) and the effective GHz.
// The guide's figure for v_fma_f32 is 2 cycles at >= 2 waves per SIMD and 4 for one wave alone.
//
//   hipcc -O3 --offload-arch=gfx950 tools/opcost.hip -o opcost && ./opcost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define N_IT 4096

#define CHECK(x)                                                                                                                  \
    do                                                                                                                            \
    {                                                                                                                             \
        hipError_t e_ = (x);                                                                                                      \
        if (e_ != hipSuccess)                                                                                                     \
        {                                                                                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                             \
            exit(1);                                                                                                              \
        }                                                                                                                         \
    } while (0)

// sixteen registers, one instruction each
#define R16_OUT                                                                                                                   \
    "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),      \
        "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
#define L16(F)                                                                                                                    \
    F("%0") F("%1") F("%2") F("%3") F("%4") F("%5") F("%6") F("%7") F("%8") F("%9") F("%10") F("%11") F("%12") F("%13") F("%14")  \
        F("%15")

enum Mode
{
    FMA, MUL, ADD, MUL64, MOV, PKFMA, PKMUL, RCP, RSQ, SQRT, EXP, LOG, FLOOR, FRACT, CVT_I, CVT_F, MAX, MIN, MED3, MINI, ADDU,
    BFI, XOR, LSHL, MULLO, ADDLSHL, DIVFIXUP, DIVSCALE, DIVFMAS, CMP, CNDMASK, FMAC, SUB, CND_SGPR, MAX3, CMP_CND, READLANE, SNOP, LDEXP, FREXP,
    ANDOR, MAD_U24, DPP_ROW, FMA_DEP, NMODES
};

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* cycles, float a, float b)
{
    extern __shared__ char lds_pad[];
    float x[16];
    for (int j = 0; j < 16; j++)
    {
        x[j] = threadIdx.x * 1e-3f + j + 1.0f;
    }
    float2 p[8];
    for (int j = 0; j < 8; j++)
    {
        p[j] = make_float2(x[2 * j], x[2 * j + 1]);
    }
    int sacc = 0;
    unsigned long long t0 = __builtin_readcyclecounter(); // s_memtime
    for (int i = 0; i < N_IT; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
// one asm statement per 128 instructions (the compiler pads every inline-asm statement with an s_nop)
#define L128(F) L16(F) L16(F) L16(F) L16(F) L16(F) L16(F) L16(F) L16(F)
#define A1(INS)                                                                                                                   \
    if (u == 0)                                                                                                                   \
    asm volatile(L128(INS) : R16_OUT : "v"(a), "v"(b))
#define AVCC(INS)                                                                                                                 \
    if (u == 0)                                                                                                                   \
    asm volatile(L128(INS) : R16_OUT : "v"(a), "v"(b) : "vcc")
#define F_FMA(R) "v_fma_f32 " R ", " R ", %16, %17\n"
#define F_FMAC(R) "v_fmac_f32 " R ", %16, %17\n"
#define F_SUB(R) "v_sub_f32 " R ", " R ", %16\n"
#define F_MAX3(R) "v_max3_f32 " R ", " R ", %16, %17\n"
#define F_CND_SGPR(R) "v_cndmask_b32_e64 " R ", " R ", %16, %18\n"
#define F_MUL(R) "v_mul_f32 " R ", " R ", %16\n"
#define F_ADD(R) "v_add_f32 " R ", " R ", %16\n"
#define F_MUL64(R) "v_mul_f32_e64 " R ", " R ", %16\n"
#define F_MOV(R) "v_mov_b32 " R ", %16\n"
#define F_RCP(R) "v_rcp_f32 " R ", " R "\n"
#define F_RSQ(R) "v_rsq_f32 " R ", " R "\n"
#define F_SQRT(R) "v_sqrt_f32 " R ", " R "\n"
#define F_EXP(R) "v_exp_f32 " R ", " R "\n"
#define F_LOG(R) "v_log_f32 " R ", " R "\n"
#define F_FLOOR(R) "v_floor_f32 " R ", " R "\n"
#define F_FRACT(R) "v_fract_f32 " R ", " R "\n"
#define F_CVT_I(R) "v_cvt_i32_f32 " R ", " R "\n"
#define F_CVT_F(R) "v_cvt_f32_i32 " R ", " R "\n"
#define F_MAX(R) "v_max_f32 " R ", " R ", %16\n"
#define F_MIN(R) "v_min_f32 " R ", " R ", %16\n"
#define F_MED3(R) "v_med3_f32 " R ", " R ", %16, %17\n"
#define F_MINI(R) "v_min_i32 " R ", " R ", %16\n"
#define F_ADDU(R) "v_add_u32 " R ", " R ", %16\n"
#define F_BFI(R) "v_bfi_b32 " R ", " R ", %16, %17\n"
#define F_XOR(R) "v_xor_b32 " R ", " R ", %16\n"
#define F_LSHL(R) "v_lshlrev_b32 " R ", 1, " R "\n"
#define F_MULLO(R) "v_mul_lo_u32 " R ", " R ", %16\n"
#define F_ADDLSHL(R) "v_add_lshl_u32 " R ", " R ", %16, 1\n"
#define F_DIVFIXUP(R) "v_div_fixup_f32 " R ", " R ", %16, %17\n"
#define F_DIVSCALE(R) "v_div_scale_f32 " R ", vcc, " R ", %16, %17\n"
#define F_DIVFMAS(R) "v_div_fmas_f32 " R ", " R ", %16, %17\n"
#define F_CMP(R) "v_cmp_lt_f32 vcc, " R ", %16\n"
#define F_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %16, vcc\n"
#define F_CMP_CND(R) "v_cmp_lt_f32 vcc, " R ", %16\n v_cndmask_b32 " R ", " R ", %17, vcc\n"
#define F_SNOP(R) "s_nop 0\n"
#define F_LDEXP(R) "v_ldexp_f32 " R ", " R ", 1\n"
#define F_FREXP(R) "v_frexp_mant_f32 " R ", " R "\n"
#define F_ANDOR(R) "v_and_or_b32 " R ", " R ", %16, %17\n"
#define F_MAD24(R) "v_mad_u32_u24 " R ", " R ", %16, %17\n"
#define F_DPP(R) "v_add_f32_dpp " R ", " R ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            if (MODE == FMA) A1(F_FMA);
            if (MODE == FMAC) A1(F_FMAC);
            if (MODE == SUB) A1(F_SUB);
            if (MODE == MAX3) A1(F_MAX3);
            if (MODE == CND_SGPR)
            {
                if (u == 0)
                    asm volatile(L128(F_CND_SGPR) : R16_OUT : "v"(a), "v"(b), "s"(0x5555555555555555ull));
            }
            if (MODE == MUL) A1(F_MUL);
            if (MODE == ADD) A1(F_ADD);
            if (MODE == MUL64) A1(F_MUL64);
            if (MODE == MOV) A1(F_MOV);
            if (MODE == RCP) A1(F_RCP);
            if (MODE == RSQ) A1(F_RSQ);
            if (MODE == SQRT) A1(F_SQRT);
            if (MODE == EXP) A1(F_EXP);
            if (MODE == LOG) A1(F_LOG);
            if (MODE == FLOOR) A1(F_FLOOR);
            if (MODE == FRACT) A1(F_FRACT);
            if (MODE == CVT_I) A1(F_CVT_I);
            if (MODE == CVT_F) A1(F_CVT_F);
            if (MODE == MAX) A1(F_MAX);
            if (MODE == MIN) A1(F_MIN);
            if (MODE == MED3) A1(F_MED3);
            if (MODE == MINI) A1(F_MINI);
            if (MODE == ADDU) A1(F_ADDU);
            if (MODE == BFI) A1(F_BFI);
            if (MODE == XOR) A1(F_XOR);
            if (MODE == LSHL) A1(F_LSHL);
            if (MODE == MULLO) A1(F_MULLO);
            if (MODE == ADDLSHL) A1(F_ADDLSHL);
            if (MODE == DIVFIXUP) A1(F_DIVFIXUP);
            if (MODE == DIVSCALE) AVCC(F_DIVSCALE);
            if (MODE == DIVFMAS) AVCC(F_DIVFMAS);
            if (MODE == CMP) AVCC(F_CMP);
            if (MODE == CNDMASK) AVCC(F_CNDMASK);
            if (MODE == CMP_CND) AVCC(F_CMP_CND); // 32 instructions per block: counted below
            if (MODE == SNOP) A1(F_SNOP);
            if (MODE == LDEXP) A1(F_LDEXP);
            if (MODE == FREXP) A1(F_FREXP);
            if (MODE == ANDOR) A1(F_ANDOR);
            if (MODE == MAD_U24) A1(F_MAD24);
            if (MODE == DPP_ROW) A1(F_DPP);
            if (MODE == PKFMA)
            {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, "
                             "%8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 "
                             "%7, %7, %8, %9\n"
                             "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, "
                             "%8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 "
                             "%7, %7, %8, %9\n"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                             : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
            }
            if (MODE == PKMUL)
            {
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n "
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                             "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n "
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                             : "v"(make_float2(a, a)));
            }
            if (MODE == READLANE)
            {
                int s0, s1, s2, s3;
                asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 3\n v_readlane_b32 %2, %6, 3\n v_readlane_b32 %3, %7, 3\n"
                             "v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 3\n v_readlane_b32 %2, %10, 3\n v_readlane_b32 %3, %11, 3\n"
                             "v_readlane_b32 %0, %4, 5\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 5\n v_readlane_b32 %3, %7, 5\n"
                             "v_readlane_b32 %0, %8, 5\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 5\n v_readlane_b32 %3, %11, 5\n"
                             : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
                             : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
                sacc += s0 ^ s1 ^ s2 ^ s3;
            }
            if (MODE == FMA_DEP)
            {
                // one dependent chain: latency, not throughput (16 per block)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(x[0])
                             : "v"(a), "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; j++)
    {
        s += x[j];
    }
    for (int j = 0; j < 8; j++)
    {
        s += p[j].x + p[j].y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)sacc + (float)lds_pad[threadIdx.x & 15];
    if ((threadIdx.x & 63) == 0)
    {
        cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    }
}

static int g_cus = 256;

template <int MODE> void run(const char* name, int instructionsPerBlock = 16)
{
    // dynamic LDS that admits exactly W workgroups on a CU with 160 KiB (163840 B): W x size fits, (W + 1) x size does not
    const int waves[4] = {1, 2, 4, 8};
    const size_t lds[4] = {65536, 65536, 36000, 18432}; // W = 1 and 2: see below
    float* d;
    unsigned long long* c;
    CHECK(hipMalloc(&d, (size_t)g_cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&c, (size_t)g_cus * 8 * 4 * sizeof(unsigned long long)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int w = 0; w < 4; w++)
    {
        // W = 1: 64 KiB per workgroup cannot keep a second workgroup off the CU (2 x 64 <= 160), so the grid of exactly one
        // workgroup per CU relies on the dispatcher placing workgroups on empty CUs first; the spread of the per-wave cycle
        // counts below shows whether it did (a doubled-up CU runs its waves at the W = 2 rate).
        int grid = g_cus * waves[w];
        k<MODE><<<grid, 256, lds[w]>>>(d, c, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<MODE><<<grid, 256, lds[w]>>>(d, c, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)grid * 4);
        CHECK(hipMemcpy(h.data(), c, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        double sum = 0, mx = 0, mn = 1e30;
        for (auto v : h)
        {
            sum += (double)v;
            mx = v > mx ? (double)v : mx;
            mn = v < mn ? (double)v : mn;
        }
        double instr = (double)N_IT * 8 * instructionsPerBlock;
        // The SIMD issues the W waves' instructions within the span of its slowest wave (waves of one launch start together;
        // the arbiter may favour the oldest, so single waves finish early): cost per instruction = slowest wave / (W x count).
        // The mean over the waves is printed beside it; equal figures mean the waves shared the SIMD evenly.
        double perSimd = mx / instr / waves[w];
        double meanWave = sum / h.size() / instr;
        double ghz = (mx / 1e9) / (ms * 1e-3);
        printf(" | %dw %5.2f (mean wave %5.2f) %4.2f GHz", waves[w], perSimd, meanWave, ghz);
    }
    printf("\n");
    fflush(stdout);
    CHECK(hipFree(d));
    CHECK(hipFree(c));
}

int main(int argc, char** argv)
{
    (void)argc;
    (void)argv;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("# %s, %d CUs. Per opcode, for W = 1, 2, 4, 8 waves per SIMD: SIMD-cycles per wave64 instruction = slowest wave's s_memtime\n"
           "# span / (W x instructions); (mean cycles per instruction seen by one wave); effective clock = that span / launch wall time.\n",
           prop.gcnArchName, g_cus);
    run<FMA>("v_fma_f32"); run<FMAC>("v_fmac_f32 (VOP2)"); run<MUL>("v_mul_f32"); run<SUB>("v_sub_f32"); run<ADD>("v_add_f32"); run<MUL64>("v_mul_f32_e64"); run<MOV>("v_mov_b32");
    run<PKFMA>("v_pk_fma_f32"); run<PKMUL>("v_pk_mul_f32");
    run<RCP>("v_rcp_f32"); run<RSQ>("v_rsq_f32"); run<SQRT>("v_sqrt_f32"); run<EXP>("v_exp_f32"); run<LOG>("v_log_f32");
    run<FLOOR>("v_floor_f32"); run<FRACT>("v_fract_f32"); run<CVT_I>("v_cvt_i32_f32"); run<CVT_F>("v_cvt_f32_i32");
    run<LDEXP>("v_ldexp_f32"); run<FREXP>("v_frexp_mant_f32");
    run<MAX>("v_max_f32"); run<MIN>("v_min_f32"); run<MED3>("v_med3_f32"); run<MINI>("v_min_i32");
    run<ADDU>("v_add_u32"); run<XOR>("v_xor_b32"); run<LSHL>("v_lshlrev_b32"); run<BFI>("v_bfi_b32"); run<ANDOR>("v_and_or_b32");
    run<MULLO>("v_mul_lo_u32"); run<MAD_U24>("v_mad_u32_u24"); run<ADDLSHL>("v_add_lshl_u32");
    run<DIVSCALE>("v_div_scale_f32"); run<DIVFMAS>("v_div_fmas_f32"); run<DIVFIXUP>("v_div_fixup_f32");
    run<CMP>("v_cmp_lt_f32 -> vcc"); run<CNDMASK>("v_cndmask_b32 (vcc)"); run<CND_SGPR>("v_cndmask_b32 (sgpr mask)"); run<MAX3>("v_max3_f32"); run<CMP_CND>("cmp + cndmask pair", 32);
    run<DPP_ROW>("v_add_f32 dpp row_shr"); run<READLANE>("v_readlane_b32");
    run<SNOP>("s_nop 0"); run<FMA_DEP>("v_fma_f32 dependent");
    return 0;
}
