// Cost of a software grid barrier on MI355X (one workgroup per CU, 256 CUs): is a persistent
// multi-phase kernel cheaper than one launch per phase (~4.7 us floor + ~1.5 us gap)?
// hipcc -O3 --offload-arch=gfx950 scripts/calib/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned *gen, unsigned nblocks, unsigned &local_gen, unsigned *err)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned target = local_gen + 1;
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { *err = 1; break; }
            }
        }
        __threadfence();
    }
    local_gen += 1;
    __syncthreads();
}

__global__ void k_barriers(unsigned *counter, unsigned *gen, unsigned *err, float *data, int phases)
{
    unsigned local_gen = 0;
    for (int p = 0; p < phases; ++p) {
        data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] += 1.0f;     // a token amount of work + memory traffic
        grid_barrier(counter, gen, gridDim.x, local_gen, err);
    }
}

__global__ void k_one(float *data) { data[(size_t)blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

int main()
{
    unsigned *ctr; float *data;
    hipMalloc(&ctr, 64); hipMalloc(&data, 1024 * 256 * 4);
    hipMemset(data, 0, 1024 * 256 * 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int nb : {64, 256, 512, 1024}) {
        for (int phases : {20, 200}) {
            hipMemset(ctr, 0, 64);
            hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, st, ctr, ctr + 1, ctr + 2, data, phases);
            hipStreamSynchronize(st);
            hipMemset(ctr, 0, 64);
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 20; ++r) {
                hipMemsetAsync(ctr, 0, 64, st);
                hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, st, ctr, ctr + 1, ctr + 2, data, phases);
            }
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
            unsigned e; hipMemcpy(&e, ctr + 2, 4, hipMemcpyDeviceToHost);
            std::printf("persistent: %4d workgroups, %3d phases: %8.1f us per launch = %6.2f us per phase (err %u)\n", nb, phases, us, us / phases, e);
        }
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 400; ++r) hipLaunchKernelGGL(k_one, dim3(nb), dim3(256), 0, st, data);
        hipStreamSynchronize(st);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 400;
        std::printf("launch per phase: %4d workgroups: %6.2f us per kernel\n", nb, us);
    }
    return 0;
}
