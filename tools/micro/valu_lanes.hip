// How long does a vector instruction occupy its SIMD when only some lanes of the wavefront are alive?
// A stream of independent v_fma_f32 (and, second table, v_cvt_f32_ubyte0 / v_max_f32) runs under several EXEC masks;
// the time per issued instruction, relative to the full mask, tells whether empty halves or quarters of a
// wave64 are skipped.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_lanes valu_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void stream(float* out, unsigned long long mask, int iterations) {
    const int lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float b = 1.0001f, c = 0.5f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iterations; i++) {
            if (OP == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 1) {
                asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n"
                             "v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte0 %5, %5\n v_cvt_f32_ubyte0 %6, %6\n v_cvt_f32_ubyte0 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else {
                asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    const int blocks = 256 * 8, iterations = 4096;     // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float* out = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    struct Case { const char* name; unsigned long long mask; };
    const Case cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-47", 0x0000FFFFFFFFFFFFull}, {"lanes 0-31", 0x00000000FFFFFFFFull}, {"lanes 32-63", 0xFFFFFFFF00000000ull},
        {"lanes 0-15", 0xFFFFull}, {"lanes 16-31", 0xFFFF0000ull}, {"lanes 0-15 and 32-47", 0x0000FFFF0000FFFFull},
        {"every 2nd lane (32 alive)", 0x5555555555555555ull}, {"every 4th lane (16 alive)", 0x1111111111111111ull}, {"lane 0 only", 1ull},
        {"43 scattered lanes", 0xB6DB6DB6DB6DB6DBull & 0xFFFFFFFFFFFFFFFFull},
    };
    const char* ops[] = {"v_fma_f32", "v_cvt_f32_ubyte0", "v_max_f32"};
    for (int op = 0; op < 3; op++) {
        printf("%s\n", ops[op]);
        float full = 0.0f;
        for (const Case& c : cases) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(start));
                if (op == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(256), 0, 0, out, c.mask, iterations);
                else if (op == 1) hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(256), 0, 0, out, c.mask, iterations);
                else hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(256), 0, 0, out, c.mask, iterations);
                CHECK(hipEventRecord(stop));
                CHECK(hipEventSynchronize(stop));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, start, stop));
                if (ms < best) best = ms;
            }
            if (full == 0.0f) full = best;
            // instructions per SIMD: 8 waves x iterations x 8; cycles at the clock the full-mask run implies are not needed: ratios suffice
            const double perSimd = 8.0 * iterations * 8.0;
            printf("  %-28s %8.3f ms  %.3f of the full mask  (%.2f ns per instruction and SIMD)\n", c.name, best, best / full, best * 1e6 / perSimd);
        }
    }
    return 0;
}
