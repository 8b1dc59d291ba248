// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access patterns the
// walk kernel uses (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Each kernel reads a 1 GiB buffer exactly once:
//   k_sload : wave-uniform s_load_dwordx16 (64 B per load), the walk's node reads
//   k_vload : 16 B per lane coalesced global_load_dwordx4, the guide's reference case (known 1/2)
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
//                               rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v16i __attribute__((ext_vector_type(16)));
#define CONSTANT __attribute__((address_space(4)))

__global__ __launch_bounds__(256) void k_sload(const int* buf, size_t bytes_per_wave, int* out)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const char CONSTANT* p = (const char CONSTANT*)buf + wave * bytes_per_wave;
    int acc = 0;
    for (size_t o = 0; o < bytes_per_wave; o += 64) {
        const v16i r = *(const v16i CONSTANT*)(p + o);
        acc += r[0] ^ r[7] ^ r[15];
    }
    if ((threadIdx.x & 63) == 0) out[wave] = acc;
}

__global__ __launch_bounds__(256) void k_vload(const int4* buf, size_t n16, int* out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * 256) { const int4 v = buf[i]; acc += v.x ^ v.w; }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const size_t bytes = 1ull << 30;
    int *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64 << 20);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int blocks = 4096, waves = blocks * 4;
    hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(256), 0, 0, buf, bytes / waves, out);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_vload, dim3(8192), dim3(256), 0, 0, (const int4*)buf, bytes / 16, out);
    hipDeviceSynchronize();
    printf("read %zu bytes per kernel\n", bytes);
    return 0;
}
