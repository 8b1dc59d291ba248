// f64_issue_calib.hip -- cycles of one gfx950 SIMD per wave64 fp64 instruction with 8 resident waves (the regime of
// csrc/bh_walk_f64.hpp), same method as issue_calib.hip: straight-line bodies of 64 instructions of one class, no memory,
// 256 CUs x 8 workgroups of 256 threads; cycles = wall time x 2.4 GHz / instructions issued per SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/calib/f64_issue_calib.hip -o /tmp/f64_issue_calib && /tmp/f64_issue_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)

template <int KIND>
__global__ __launch_bounds__(256) void calib(double *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 * 1.5, a2 = a0 * 0.5, a3 = a0 + 2.0;
    double b = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
    unsigned long long m0 = 0, m1 = 0;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(R16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        else if (KIND == 1) asm volatile(R16("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        else if (KIND == 2) asm volatile(R16("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));
        else if (KIND == 3) asm volatile(R16("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        else if (KIND == 4) asm volatile(R16("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        else if (KIND == 5) asm volatile(R16("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if (KIND == 6) asm volatile(R16("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        else if (KIND == 7) asm volatile(R16("v_cmp_lt_f64 %0, %2, %3\n v_cmp_lt_f64 %1, %3, %2\n v_cmp_lt_f64 %0, %4, %5\n v_cmp_lt_f64 %1, %5, %4\n") : "+s"(m0), "+s"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if (KIND == 8) asm volatile(R16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(b), "v"(c));   // one SGPR-pair operand
        else if (KIND == 9) asm volatile(R16("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + (double)(m0 ^ m1);
}

int main()
{
    const int blocks = 256 * 8, iters = 400;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const char *names[10] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_cmp_lt_f64->sgpr",
                             "v_fma_f64 (SGPR operand)", "v_rsq_f32"};
    for (int k = 0; k < 10; ++k) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (k) {
            case 0: hipLaunchKernelGGL(calib<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 1: hipLaunchKernelGGL(calib<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 2: hipLaunchKernelGGL(calib<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 3: hipLaunchKernelGGL(calib<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 4: hipLaunchKernelGGL(calib<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 5: hipLaunchKernelGGL(calib<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 6: hipLaunchKernelGGL(calib<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 7: hipLaunchKernelGGL(calib<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            case 8: hipLaunchKernelGGL(calib<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            default: hipLaunchKernelGGL(calib<9>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); break;
            }
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double insts_per_simd = 8.0 * iters * 64;              // 8 waves per SIMD
        printf("%-26s %7.3f ms   %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", names[k], ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    return 0;
}
