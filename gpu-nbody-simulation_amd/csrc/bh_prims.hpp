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
// wave-level inclusive scan (Hillis-Steele over DPP-backed shuffles)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (l >= d) v += o;
    }
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
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        double oa = __shfl_up(v.a, d, kWave), ob = __shfl_up(v.b, d, kWave),
               oc = __shfl_up(v.c, d, kWave);
        if (l >= d) { v.a += oa; v.b += ob; v.c += oc; }
    }
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
    T exc = inc;
    // exclusive = inclusive - own (recomputed additively to stay exact for integers and
    // order-stable for doubles: shift the inclusive value down one lane)
    T prev = __shfl_up(inc, 1, kWave);
    exc = (lane_id() == 0) ? zero_of<T>() : prev;
    return base + exc;
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
    d3 prev{__shfl_up(inc.a, 1, kWave), __shfl_up(inc.b, 1, kWave), __shfl_up(inc.c, 1, kWave)};
    if (lane_id() == 0) prev = zero_of<d3>();
    return base + prev;
}

// ---------------------------------------------------------------------------------------------
// min/max reduction of 2-D positions -> partial[block] = {xmin, xmax, ymin, ymax} (fp64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { double o = __shfl_xor(v, d, kWave); v = (o < v) ? o : v; }
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) { double o = __shfl_xor(v, d, kWave); v = (v < o) ? o : v; }
    return v;
}

}  // namespace bh
