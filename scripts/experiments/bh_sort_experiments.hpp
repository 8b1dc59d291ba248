// bh_sort_experiments.hpp -- sort variants that were measured and rejected: the
// barrier-per-round scatter of round 1 and the look-back "onesweep" sort.  NOT part of the product: included by
// csrc/bh_sort.hpp only when a scripts/ A/B build defines BHGPU_EXPERIMENTS (scripts/build_variants.sh).
#pragma once

namespace bh {

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void radix_scatter(const uint64_t *__restrict__ kin,
                                                         const uint32_t *__restrict__ vin,
                                                         uint64_t *__restrict__ kout,
                                                         uint32_t *__restrict__ vout,
                                                         const uint32_t *__restrict__ offs,
                                                         const uint32_t *__restrict__ row_total, int64_t n,
                                                         int shift, int nblocks)
{
    __shared__ uint32_t run[kRadix];
    __shared__ uint32_t wcnt[kWavesPerBlock][kRadix];
    __shared__ uint32_t gbase[kRadix];
    __shared__ uint32_t sm[kWavesPerBlock + 1];
    const int t = threadIdx.x, w = wave_id(), l = lane_id();
    {
        uint32_t all;
        const uint32_t digit_base = block_exclusive_sum(row_total[t], sm, all);
        gbase[t] = digit_base + offs[(int64_t)t * nblocks + blockIdx.x];
    }
    run[t] = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; ++k) wcnt[k][t] = 0;
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS);
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
#pragma unroll 1
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = base + r * kBlock + t;
        const bool valid = i < n;
        const uint64_t key = valid ? kin[i] : 0ull;
        const uint32_t val = valid ? vin[i] : 0u;
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


// =================================================================================================
// Single-kernel-per-pass variant ("onesweep": chained scan with decoupled look-back, Adinets &
// Merrill 2022), 2 + P launches instead of 3P.  MEASURED AND NOT THE DEFAULT (BH_SORT_ONESWEEP=1
// selects it): on MI355X the look-back chain crosses XCDs, and a pass takes 31 us against
// 5.3 + 4.9 + 19 us for histogram + row scan + scatter -- the two saved launches per pass buy
// nothing (build 0.322 vs 0.288 ms at N = 1M).  Kept as the evidence and for re-measurement.
//   radix_hist_all : global digit histograms of ALL passes from the unsorted keys (digit counts do
//                    not depend on order); also zeroes the look-back state of this step
//   radix_onesweep : per tile: local histogram -> publish -> look back over earlier tiles for the
//                    exclusive prefix of every digit -> stable scatter
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH "R2 granule"):
// a status word is ONE self-contained 32-bit value {2-bit flag, 30-bit count} written by one
// agent-scope relaxed atomic store and polled with agent-scope relaxed atomic loads -- there is no
// separate payload, so no release/acquire fence is needed and the result cannot depend on XCD
// placement.  Tile ids are drawn from an atomic counter, so every tile a workgroup waits for has
// already started (forward progress without assuming dispatch order).  Spins are bounded: on
// timeout the kernel raises `err` and finishes with garbage instead of hanging the device.
// =================================================================================================
constexpr uint32_t kStAgg = 1u << 30, kStInc = 2u << 30, kStMask = (1u << 30) - 1;
constexpr int kMaxPasses = 8;

__global__ __launch_bounds__(kBlock) void radix_hist_all(const uint64_t *__restrict__ keys, int64_t n,
                                                          int passes, uint32_t *__restrict__ ghist,
                                                          uint32_t *__restrict__ status, int64_t status_words,
                                                          uint32_t *__restrict__ tile_counter)
{
    // ghist/status/tile_counter were zeroed by radix_zero (previous launch)
    __shared__ uint32_t h[kMaxPasses][kRadix];
    for (int p = 0; p < passes; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t k = keys[i];
        for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(uint32_t)(k >> (p * kRadixBits)) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * kRadix + threadIdx.x], c);
    }
}

__global__ __launch_bounds__(kBlock) void radix_zero(uint32_t *__restrict__ ghist, uint32_t *__restrict__ status,
                                                      int64_t status_words, uint32_t *__restrict__ tile_counter)
{
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int64_t i = i0; i < status_words; i += (int64_t)gridDim.x * kBlock) status[i] = 0;
    if (i0 < kMaxPasses * kRadix) ghist[i0] = 0;
    if (i0 < kMaxPasses) tile_counter[i0] = 0;
}

__global__ __launch_bounds__(kBlock) void radix_onesweep(const uint64_t *__restrict__ kin,
                                                          const uint32_t *__restrict__ vin,
                                                          uint64_t *__restrict__ kout,
                                                          uint32_t *__restrict__ vout,
                                                          const uint32_t *__restrict__ ghist_pass,
                                                          uint32_t *__restrict__ status_pass,
                                                          uint32_t *__restrict__ tile_counter_pass,
                                                          uint32_t *__restrict__ err, int64_t n, int shift)
{
    __shared__ uint32_t run[kRadix];
    __shared__ uint32_t wcnt[kWavesPerBlock][kRadix];
    __shared__ uint32_t gbase[kRadix];
    __shared__ uint32_t hist[kRadix];
    __shared__ uint32_t sm[kWavesPerBlock + 1];
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, w = wave_id(), l = lane_id();

    if (t == 0) s_tile = atomicAdd(tile_counter_pass, 1u);
    hist[t] = 0; run[t] = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; ++k) wcnt[k][t] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t base = (int64_t)tile * kSortTile;

    // 1. local histogram
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = base + r * kBlock + t;
        if (i < n) atomicAdd(&hist[(uint32_t)(kin[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();

    // 2. publish, 3. look back (thread d owns digit d)
    const uint32_t mine = hist[t];
    uint32_t *st = status_pass + (int64_t)tile * kRadix + t;
    __hip_atomic_store(st, mine | (tile == 0 ? kStInc : kStAgg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    if (tile > 0) {
        // windowed look-back: kLook predecessor words are requested together (independent loads, one
        // round trip), then consumed nearest first until an inclusive prefix or a not-yet-published
        // word is met
        constexpr int kLook = 8;
        int64_t look = (int64_t)tile - 1;
        uint32_t spins = 0;
        bool done = false;
        while (!done) {
            uint32_t v[kLook];
#pragma unroll
            for (int j = 0; j < kLook; ++j) {
                const int64_t idx = look - j;
                v[j] = (idx >= 0) ? __hip_atomic_load(status_pass + idx * kRadix + t, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : kStInc;                       // before tile 0: inclusive prefix 0
            }
            int used = 0;
#pragma unroll
            for (int j = 0; j < kLook; ++j) {
                if (done || used != j) continue;
                const uint32_t flag = v[j] & ~kStMask;
                if (flag == 0) continue;                          // not published yet: stop consuming
                excl += v[j] & kStMask;
                used = j + 1;
                if (flag == kStInc) done = true;
            }
            look -= used;
            if (!done && used == 0) {
                if (++spins > (1u << 20)) { *err = 1; break; }    // bounded: never hang the device
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __hip_atomic_store(st, (excl + mine) | kStInc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // digit base = exclusive scan of the global histogram of this pass
    {
        uint32_t all;
        const uint32_t digit_base = block_exclusive_sum(ghist_pass[t], sm, all);
        gbase[t] = digit_base + excl;
    }
    __syncthreads();

    // 4. stable scatter of the tile (per-round ballot matching, as radix_scatter)
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
#pragma unroll 1
    for (int r = 0; r < kSortItems; ++r) {
        const int64_t i = base + r * kBlock + t;
        const bool valid = i < n;
        const uint64_t key = valid ? kin[i] : 0ull;
        const uint32_t val = valid ? vin[i] : 0u;
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
            if (dst < n) { kout[dst] = key; vout[dst] = val; }       // (dst >= n only after a spin timeout)
        }
        __syncthreads();
        uint32_t sacc = 0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) { sacc += wcnt[k][t]; wcnt[k][t] = 0; }
        run[t] += sacc;
        __syncthreads();
    }
}


}  // namespace bh
