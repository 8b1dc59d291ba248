// Does a low-priority stream fill the tail of a high-priority kernel?  Kernel A: 16,384 one-wave workgroups that
// spin ~100 us each (two rounds of 8,192 resident waves, like the walk at N = 1M).  Kernel B: 4,096 workgroups of
// ~25 us on a second stream with the LOWEST priority, launched right after A.  Every workgroup records its start
// time; printed: when B's workgroups started relative to A's span, and the total span against A alone.
//   hipcc -O3 --offload-arch=gfx950 scripts/calib/prio_tail.hip -o /tmp/prio_tail && /tmp/prio_tail
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void spin(unsigned long long *start, unsigned long long *stop, long ticks, float *sink)
{
    const unsigned long long t0 = wall_clock64();
    float x = (float)threadIdx.x;
    while ((long)(wall_clock64() - t0) < ticks) {
        // SGPR budget like the walk: nothing special, just keep the SIMD a little busy
        for (int k = 0; k < 64; ++k) x = x * 1.0001f + 0.5f;
    }
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; stop[blockIdx.x] = wall_clock64(); }
    if (x == 12345.678f) sink[0] = x;
}

int main()
{
    const int nA = 16384, nB = 4096;
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);                  // lo = least, hi = greatest (numerically lower)
    printf("priority range: least %d greatest %d\n", lo, hi);
    hipStream_t sa, sb;
    hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi);
    hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo);
    unsigned long long *dA0, *dA1, *dB0, *dB1; float *sink;
    hipMalloc(&dA0, nA * 8); hipMalloc(&dA1, nA * 8); hipMalloc(&dB0, nB * 8); hipMalloc(&dB1, nB * 8); hipMalloc(&sink, 4);
    const long tickA = 100 * 100, tickB = 25 * 100;             // wall_clock64: 100 MHz
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: A alone; 1: A then B on the low-priority stream; 2: both on one stream, B after A
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, sa, dA0, dA1, 100, sink);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0, sa);
        hipLaunchKernelGGL(spin, dim3(nA), dim3(64), 0, sa, dA0, dA1, tickA, sink);
        if (mode == 1) hipLaunchKernelGGL(spin, dim3(nB), dim3(64), 0, sb, dB0, dB1, tickB, sink);
        if (mode == 2) hipLaunchKernelGGL(spin, dim3(nB), dim3(64), 0, sa, dB0, dB1, tickB, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> a0(nA), a1(nA), b0(nB), b1(nB);
        hipMemcpy(a0.data(), dA0, nA * 8, hipMemcpyDeviceToHost); hipMemcpy(a1.data(), dA1, nA * 8, hipMemcpyDeviceToHost);
        hipMemcpy(b0.data(), dB0, nB * 8, hipMemcpyDeviceToHost); hipMemcpy(b1.data(), dB1, nB * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(a0.begin(), a0.end());
        const unsigned long long aEnd = *std::max_element(a1.begin(), a1.end());
        printf("mode %d: A span %.1f us", mode, (aEnd - t0) / 100.0);
        if (mode) {
            std::sort(b0.begin(), b0.end());
            const unsigned long long bEnd = *std::max_element(b1.begin(), b1.end());
            printf("; B starts: first %.1f  p10 %.1f  p50 %.1f  p90 %.1f us, B end %.1f us; total span %.1f us",
                   ((long long)b0[0] - (long long)t0) / 100.0, ((long long)b0[nB / 10] - (long long)t0) / 100.0,
                   ((long long)b0[nB / 2] - (long long)t0) / 100.0, ((long long)b0[nB * 9 / 10] - (long long)t0) / 100.0,
                   ((long long)bEnd - (long long)t0) / 100.0, (std::max(aEnd, bEnd) - t0) / 100.0);
        }
        printf("\n");
    }
    return 0;
}
