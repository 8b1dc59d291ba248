// bh_prims.hpp -- wave64 device helpers shared by all kernels: lane/wave ids, wave and block
// scans, wave min/max.  gfx950 only.  (Kernels live in bh_sort.hpp / bh_tree.hpp.)
//
// None of this exists in the reference (its tree is built sequentially on the host,
// project.cu:575-591); it is the machinery that lets the same tree be built on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kItems = 8;
constexpr int kTile = kBlock * kItems;     // elements per workgroup in scan / sort kernels

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---------------------------------------------------------------------------------------------
// wave-level inclusive scan on DPP row shifts and row broadcasts (v_mov_b32_dpp / v_add_u32_dpp):
// four Hillis-Steele steps inside each row of 16 lanes, then lane 15 of rows 0 and 2 is added to
// rows 1 and 3, then lane 31 to rows 2 and 3.  A `__shfl_up` step is a ds_bpermute through LDS
// (~100 cycles); the scans sit on the critical path of every latency-bound kernel of the build
// (scan_apply2 runs 32 of them in sequence), a DPP step is an ordinary VALU instruction.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    // lanes that have no source (shifted in from outside the row, or rows outside ROW_MASK) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_mov<CTRL, ROW_MASK>((uint32_t)u), hi = dpp_mov<CTRL, ROW_MASK>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));      // +0.0 where there is no source
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_mov(uint64_t u)
{
    const uint32_t lo = dpp_mov<CTRL, ROW_MASK>((uint32_t)u), hi = dpp_mov<CTRL, ROW_MASK>((uint32_t)(u >> 32));
    return ((uint64_t)hi << 32) | lo;
}

constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138;

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v)
{
    v += dpp_mov<kDppShr1, 0xF>(v);
    v += dpp_mov<kDppShr2, 0xF>(v);
    v += dpp_mov<kDppShr4, 0xF>(v);
    v += dpp_mov<kDppShr8, 0xF>(v);
    v += dpp_mov<kDppBcast15, 0xA>(v);
    v += dpp_mov<kDppBcast31, 0xC>(v);
    return v;
}

struct d3 {
    double a, b, c;
    __host__ __device__ d3 &operator+=(const d3 &o) { a += o.a; b += o.b; c += o.c; return *this; }
};
__device__ __forceinline__ d3 operator+(d3 x, const d3 &y) { x += y; return x; }

template <>
__device__ __forceinline__ d3 wave_inclusive_sum<d3>(d3 v)
{
    v.a = wave_inclusive_sum(v.a); v.b = wave_inclusive_sum(v.b); v.c = wave_inclusive_sum(v.c);
    return v;
}

template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ uint32_t zero_of<uint32_t>() { return 0u; }
template <> __device__ __forceinline__ d3 zero_of<d3>() { return d3{0.0, 0.0, 0.0}; }

// Block-wide exclusive scan of one value per thread; returns the exclusive prefix and the
// block total through `total`.  `smem` must hold kWavesPerBlock + 1 elements of T.
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T v, T *smem, T &total)
{
    T inc = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) smem[wave_id()] = inc;
    __syncthreads();
    T base = zero_of<T>();
    T tot = zero_of<T>();
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        T s = smem[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    // exclusive = the inclusive value of the lane before (lane 0 reads 0): no subtraction, so it stays
    // exact for integers and order-stable for doubles
    return base + dpp_mov<kDppWaveShr1, 0xF>(inc);
}

template <>
__device__ __forceinline__ d3 block_exclusive_sum<d3>(d3 v, d3 *smem, d3 &total)
{
    d3 inc = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) smem[wave_id()] = inc;
    __syncthreads();
    d3 base = zero_of<d3>(), tot = zero_of<d3>();
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        d3 s = smem[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    const d3 prev{dpp_mov<kDppWaveShr1, 0xF>(inc.a), dpp_mov<kDppWaveShr1, 0xF>(inc.b), dpp_mov<kDppWaveShr1, 0xF>(inc.c)};
    return base + prev;
}

// ---------------------------------------------------------------------------------------------
// min/max reduction of 2-D positions -> partial[block] = {xmin, xmax, ymin, ymax} (fp64)
// ---------------------------------------------------------------------------------------------
// (DPP as above; a lane without a source keeps its own value, which is neutral for min and max; the
// result of the whole wave ends up in lane 63 and is broadcast from there)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_self(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)u, (int)(uint32_t)u, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(u >> 32), (int)(uint32_t)(u >> 32), CTRL,
                                                              ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double wave_bcast63(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// (min / max over the wave as the reference folds its bounds -- xMin = std::min(xMin, x) from +inf, project.cu:544-551 -- : a NaN
// never wins a comparison there, so it must not win here either.  `(b < a) ? b : a` ignores a NaN that comes IN, but a lane whose OWN
// value is NaN would keep it, and whether that reaches the result depended on which lane held it: one body gone NaN (a massless
// body's 0/0 acceleration) could take the next step's root box, and with it every body, along.  Found in round 4 by a random test.)
__device__ __forceinline__ double wave_min(double v)
{
    v = (v <= (double)INFINITY) ? v : (double)INFINITY;
    auto mn = [](double a, double b) { return (b < a) ? b : a; };
    v = mn(v, dpp_mov_self<kDppShr1, 0xF>(v));
    v = mn(v, dpp_mov_self<kDppShr2, 0xF>(v));
    v = mn(v, dpp_mov_self<kDppShr4, 0xF>(v));
    v = mn(v, dpp_mov_self<kDppShr8, 0xF>(v));
    v = mn(v, dpp_mov_self<kDppBcast15, 0xA>(v));
    v = mn(v, dpp_mov_self<kDppBcast31, 0xC>(v));
    return wave_bcast63(v);
}
__device__ __forceinline__ double wave_max(double v)
{
    v = (v >= -(double)INFINITY) ? v : -(double)INFINITY;
    auto mx = [](double a, double b) { return (a < b) ? b : a; };
    v = mx(v, dpp_mov_self<kDppShr1, 0xF>(v));
    v = mx(v, dpp_mov_self<kDppShr2, 0xF>(v));
    v = mx(v, dpp_mov_self<kDppShr4, 0xF>(v));
    v = mx(v, dpp_mov_self<kDppShr8, 0xF>(v));
    v = mx(v, dpp_mov_self<kDppBcast15, 0xA>(v));
    v = mx(v, dpp_mov_self<kDppBcast31, 0xC>(v));
    return wave_bcast63(v);
}

}  // namespace bh
