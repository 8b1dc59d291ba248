// soffset_calib.hip -- is the SGPR byte offset of s_load_dword* (the addressing form the hand-scheduled walk uses:
// s_load_dwordx16 s[..], s[base:base+1], s_off) UNSIGNED 32-bit on gfx950?  The walk forms s_off = quad * 80 with
// s_mul_i32; if offsets >= 2^31 are taken as unsigned, forests of up to 2^32 bytes (53.6 M quads) can use the assembly
// loop, otherwise only 2^31 bytes (26.8 M quads: BASELINE config 5's capacity of 33.5 M quads does not fit).
//   hipcc -O3 --offload-arch=gfx950 scripts/calib/soffset_calib.hip -o /tmp/soffset_calib && /tmp/soffset_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void probe(const uint32_t *buf, uint32_t off, uint32_t *out)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(buf), "s"(off) : "memory");
    if (threadIdx.x == 0) out[0] = v;
}

int main()
{
    // base sits 2.5 GiB inside a 6.5 GiB allocation: both readings of an offset >= 2^31 -- base + offset (unsigned) and
    // base + offset - 2^32 (signed) -- stay inside it, so the probe cannot fault whichever the hardware implements
    const size_t region = 0x1A0000000ull, lead = 0xA0000000ull;
    char *reg = nullptr;
    uint32_t *out = nullptr;
    if (hipMalloc(&reg, region) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(reg, 0, region);
    char *base = reg + lead;
    const uint32_t offs[5] = {0x10u, 0x7FFFFFF0u, 0x80000000u, 0x90000040u, 0xF0000100u};
    for (uint32_t o : offs) {
        const uint32_t mark_u = 0xA5000000u ^ o, mark_s = 0x5A000000u ^ o;
        const long long so = (long long)(int32_t)o;
        if (so < 0) hipMemcpy(base + so, &mark_s, 4, hipMemcpyHostToDevice);
        hipMemcpy(base + (size_t)o, &mark_u, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, reinterpret_cast<const uint32_t *>(base), o, out);
        uint32_t got = 0;
        hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
        printf("offset 0x%08x: read 0x%08x -> %s\n", o, got,
               got == mark_u ? "UNSIGNED (base + offset)" : got == mark_s ? "SIGNED (base + offset - 2^32)" : "neither?");
    }
    hipFree(reg); hipFree(out);
    return 0;
}
