// Issue cost of the vector instructions a quantised-box slab test can be built from, as a stream of eight
// independent instructions per iteration, on as many waves per SIMD as the first argument says (default 6, what the
// measured trace kernel runs at; one workgroup of 256 threads is one wave on each SIMD of a CU, one round of workgroups).
// Reported in shader cycles per instruction and SIMD -- every wave reads the cycle counter before and after its stream,
// the longest span of any wave divided by the instructions the waves of one SIMD issue in it -- and in nanoseconds.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_ops valu_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define EIGHT(INSTR) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)

#define K(NAME, TEXT)                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iterations, unsigned long long* span) { \
        unsigned a[8];                                                                        \
        for (int i = 0; i < 8; i++) a[i] = __float_as_uint(1.0f + threadIdx.x + i);            \
        const unsigned b = __float_as_uint(1.0001f), c = __float_as_uint(0.5f);                \
        const unsigned long long began = __builtin_readcyclecounter();                        \
        for (int i = 0; i < iterations; i++) asm volatile(TEXT OPERANDS);                      \
        const unsigned long long took = __builtin_readcyclecounter() - began;                 \
        unsigned s = 0;                                                                       \
        for (int i = 0; i < 8; i++) s ^= a[i];                                                \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
        if ((threadIdx.x & 63) == 0) atomicMax(span, took);                                   \
    }
#define R8(fmt) fmt("%0") fmt("%1") fmt("%2") fmt("%3") fmt("%4") fmt("%5") fmt("%6") fmt("%7")

#define F_FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define F_ADD(r) "v_add_f32 " r ", " r ", %8\n"
#define F_MUL(r) "v_mul_f32 " r ", " r ", %8\n"
#define F_MAX(r) "v_max_f32 " r ", " r ", %8\n"
#define F_MAX3(r) "v_max3_f32 " r ", " r ", %8, %9\n"
#define F_MED3(r) "v_med3_f32 " r ", " r ", %8, %9\n"
#define F_IMAX(r) "v_max_i32 " r ", " r ", %8\n"
#define F_IMAX3(r) "v_max3_i32 " r ", " r ", %8, %9\n"
#define F_UMIN(r) "v_min_u32 " r ", " r ", %8\n"
#define F_CVT(r) "v_cvt_f32_ubyte1 " r ", " r "\n"
#define F_CVTU(r) "v_cvt_f32_u32 " r ", " r "\n"
#define F_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define F_BFE(r) "v_bfe_u32 " r ", " r ", 8, 8\n"
#define F_AND(r) "v_and_b32 " r ", " r ", %8\n"
#define F_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define F_ALIGN(r) "v_alignbit_b32 " r ", " r ", %8, 31\n"
#define F_CND(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define F_SUB(r) "v_sub_f32 " r ", " r ", %8\n"
#define F_PKFMA(r) "v_pk_fma_f16 " r ", " r ", %8, %9\n"
#define F_PKMAX(r) "v_pk_max_f16 " r ", " r ", %8\n"
#define F_MADU24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define F_MOV(r) "v_mov_b32 " r ", %8\n"
#define F_FMAMIX(r) "v_fma_mix_f32 " r ", " r ", %8, %9 op_sel_hi:[1,0,0]\n"
#define F_CVTF16(r) "v_cvt_f32_f16 " r ", " r "\n"
#define F_RCP(r) "v_rcp_f32 " r ", " r "\n"
#define F_SQRT(r) "v_sqrt_f32 " r ", " r "\n"
#define F_DIVFIX(r) "v_div_fixup_f32 " r ", " r ", %8, %9\n"
#define F_ADDU(r) "v_add_u32 " r ", " r ", %8\n"
#define F_LSHL(r) "v_lshlrev_b32 " r ", 3, " r "\n"
#define F_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8\n"

K(k_fma, R8(F_FMA)) K(k_add, R8(F_ADD)) K(k_mul, R8(F_MUL)) K(k_max, R8(F_MAX)) K(k_max3, R8(F_MAX3)) K(k_med3, R8(F_MED3))
K(k_imax, R8(F_IMAX)) K(k_imax3, R8(F_IMAX3)) K(k_umin, R8(F_UMIN)) K(k_cvt, R8(F_CVT)) K(k_cvtu, R8(F_CVTU)) K(k_perm, R8(F_PERM))
K(k_bfe, R8(F_BFE)) K(k_and, R8(F_AND)) K(k_lshlor, R8(F_LSHLOR)) K(k_align, R8(F_ALIGN)) K(k_cnd, R8(F_CND)) K(k_sub, R8(F_SUB))
K(k_rcp, R8(F_RCP)) K(k_sqrt, R8(F_SQRT)) K(k_divfix, R8(F_DIVFIX)) K(k_addu, R8(F_ADDU)) K(k_lshl, R8(F_LSHL)) K(k_bcnt, R8(F_BCNT))
K(k_pkfma, R8(F_PKFMA)) K(k_pkmax, R8(F_PKMAX)) K(k_madu24, R8(F_MADU24)) K(k_mov, R8(F_MOV)) K(k_fmamix, R8(F_FMAMIX)) K(k_cvtf16, R8(F_CVTF16))

int main(int argc, char** argv) {
    const int wavesPerSimd = argc > 1 ? atoi(argv[1]) : 6;
    int units = 256;
    CHECK(hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = units * wavesPerSimd, iterations = 4096;
    unsigned* out = nullptr;
    unsigned long long* span = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(unsigned)));
    CHECK(hipMalloc(&span, sizeof *span));
    printf("%d compute units, %d waves per SIMD, %d instructions per wave\n", units, wavesPerSimd, 8 * iterations);
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    struct Case { const char* name; void (*kernel)(unsigned*, int, unsigned long long*); };
    const Case cases[] = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max},
                          {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_max_i32", k_imax}, {"v_max3_i32", k_imax3}, {"v_min_u32", k_umin},
                          {"v_cvt_f32_ubyte1", k_cvt}, {"v_cvt_f32_u32", k_cvtu}, {"v_cvt_f32_f16", k_cvtf16}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe},
                          {"v_and_b32", k_and}, {"v_lshl_or_b32", k_lshlor}, {"v_alignbit_b32", k_align}, {"v_cndmask_b32", k_cnd},
                          {"v_mad_u32_u24", k_madu24}, {"v_mov_b32", k_mov}, {"v_fma_mix_f32", k_fmamix}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_div_fixup_f32", k_divfix}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl},
                          {"v_bcnt_u32_b32", k_bcnt}, {"v_pk_fma_f16", k_pkfma}, {"v_pk_max_f16", k_pkmax}};
    float fma = 0.0f;
    for (const Case& c : cases) {
        float best = 1e30f;
        unsigned long long cycles = ~0ull;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipMemset(span, 0, sizeof *span));
            CHECK(hipEventRecord(start));
            hipLaunchKernelGGL(c.kernel, dim3(blocks), dim3(256), 0, 0, out, iterations, span);
            CHECK(hipEventRecord(stop));
            CHECK(hipEventSynchronize(stop));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, start, stop));
            unsigned long long longest = 0;
            CHECK(hipMemcpy(&longest, span, sizeof longest, hipMemcpyDeviceToHost));
            if (ms < best) best = ms;
            if (longest < cycles) cycles = longest;
        }
        if (fma == 0.0f) fma = best;
        const double perSimd = 8.0 * iterations * wavesPerSimd;
        printf("%-20s %8.3f ms  %.2f x v_fma_f32  %.2f cycles, %.2f ns per instruction and SIMD\n", c.name, best, best / fma,
               (double)cycles / perSimd, best * 1e6 / perSimd);
    }
    return 0;
}
