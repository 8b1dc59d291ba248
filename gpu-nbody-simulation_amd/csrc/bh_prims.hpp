// bh_prims.hpp -- wave64 device primitives for the tree build: block reductions, a three-kernel
// exclusive scan and a stable LSD radix sort of (key, value) pairs.  gfx950 only.
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
// three-kernel exclusive scan over n elements (tile = kTile per workgroup)
//   1. scan_tile_sums : bsum[b] = sum of tile b
//   2. scan_top       : one workgroup turns bsum into exclusive tile offsets, writes the total
//   3. scan_apply     : out[i] = offset[tile] + local exclusive prefix   (in == out allowed)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_tile_sums(const T *__restrict__ in, T *__restrict__ bsum,
                                                          int64_t n)
{
    __shared__ T sm[kWavesPerBlock + 1];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    T s = zero_of<T>();
#pragma unroll
    for (int k = 0; k < kItems; ++k)
        if (base + k < n) s += in[base + k];
    T tot;
    (void)block_exclusive_sum(s, sm, tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scan_top(T *__restrict__ bsum, int nb, T *__restrict__ total)
{
    __shared__ T sm[kWavesPerBlock + 1];
    T carry = zero_of<T>();
    for (int c0 = 0; c0 < nb; c0 += kBlock) {
        const int i = c0 + threadIdx.x;
        T v = (i < nb) ? bsum[i] : zero_of<T>();
        T tot;
        T ex = block_exclusive_sum(v, sm, tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scan_apply(const T *in, T *out, const T *__restrict__ bsum,
                                                      int64_t n)
{
    __shared__ T sm[kWavesPerBlock + 1];
    const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
    T v[kItems];
    T s = zero_of<T>();
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : zero_of<T>();
        s += v[k];
    }
    T tot;
    T run = bsum[blockIdx.x] + block_exclusive_sum(s, sm, tot);
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort, 8-bit digits
//   radix_hist    : counts[digit * nblocks + block] = occurrences of digit in tile `block`
//   (scan of counts, digit-major => global base of every (digit, block))
//   radix_scatter : stable scatter of the tile using per-round ballot matching
// A round handles 256 consecutive elements (thread t <-> element round*256 + t), so ranks
// follow element order and equal keys keep their input order.
// ---------------------------------------------------------------------------------------------
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void radix_hist(const KeyT *__restrict__ keys,
                                                      uint32_t *__restrict__ counts, int64_t n,
                                                      int shift, int nblocks)
{
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
        const int64_t i = base + r * kBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void radix_scatter(const KeyT *__restrict__ kin,
                                                         const uint32_t *__restrict__ vin,
                                                         KeyT *__restrict__ kout,
                                                         uint32_t *__restrict__ vout,
                                                         const uint32_t *__restrict__ offs, int64_t n,
                                                         int shift, int nblocks)
{
    __shared__ uint32_t run[kRadix];
    __shared__ uint32_t wcnt[kWavesPerBlock][kRadix];
    __shared__ uint32_t gbase[kRadix];
    const int t = threadIdx.x, w = wave_id(), l = lane_id();
    gbase[t] = offs[(int64_t)t * nblocks + blockIdx.x];
    run[t] = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; ++k) wcnt[k][t] = 0;
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * kTile;
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
#pragma unroll 1
    for (int r = 0; r < kItems; ++r) {
        const int64_t i = base + r * kBlock + t;
        const bool valid = i < n;
        KeyT key = valid ? kin[i] : (KeyT)0;
        uint32_t val = valid ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(key >> shift) & (kRadix - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rank = __popcll(peers & lt);
        if (valid && rank == 0) wcnt[w][d] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t o = run[d] + rank;
            for (int k = 0; k < w; ++k) o += wcnt[k][d];
            const int64_t dst = (int64_t)gbase[d] + o;
            kout[dst] = key;
            vout[dst] = val;
        }
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) { s += wcnt[k][t]; wcnt[k][t] = 0; }
        run[t] += s;
        __syncthreads();
    }
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
