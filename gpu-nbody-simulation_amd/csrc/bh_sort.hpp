// bh_sort.hpp -- the sort of (key, body) pairs for the tree build.  Included by the engine translation unit
// only.  Two sorts with the same result, the stable order by key:
//
//   LSD radix sort (first build after an upload, launches above 4M bodies, exact mode): 8-bit digits, three
//   launches per pass --
//     radix_hist      : counts[digit * nblocks + block] = occurrences of digit in tile `block`
//     radix_rowscan   : one workgroup per digit: exclusive prefix along its row, row total aside
//     radix_scatter_w : base(digit, block) = exclusive scan of the 256 row totals (done in-kernel) + row
//                       prefix; stable scatter of the tile, ranks within a wave's round from a per-wave LDS
//                       table of lane masks
//   bucket sort (every later build of up to 4M bodies with packed keys; see "bucket sort" below): ONE such
//   counting pass whose digit is the key's bucket among splitters from the previous build, then
//     bucket_sort_kernel : one workgroup sorts one bucket entirely in LDS.
//
// Element order inside a tile is (wave, round, lane) = tile order and ranks follow lane order, so equal keys
// keep their input order.  (Measured and rejected in rounds 1-2, code removed in round 4: a barrier-per-round scatter and
// a look-back "onesweep" sort -- DESIGN.md section 3, profiles/r02_final; also letting every scatter workgroup derive its own offsets from the count matrix -- it re-reads the
// 256 x nblocks matrix per workgroup, 37 us per pass at N = 1M against 18 + 14 us for scatter + scan then.)
#pragma once

#include "bh_prims.hpp"

namespace bh {

#ifndef BH_SORT_ITEMS
#define BH_SORT_ITEMS 8
#endif
constexpr int kSortItems = BH_SORT_ITEMS;          // keys per thread in the sort kernels
constexpr int kSortTile = kBlock * kSortItems;
constexpr int kRadixBits = 8;

// grid = 256 workgroups (one per digit): counts[d][*] -> exclusive prefix in place, total aside
__global__ __launch_bounds__(kBlock) void radix_rowscan(uint32_t *__restrict__ counts,
                                                         uint32_t *__restrict__ row_total, int nblocks)
{
    __shared__ uint32_t sm[kWavesPerBlock + 1];
    uint32_t *row = counts + (int64_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int c0 = 0; c0 < nblocks; c0 += kBlock) {
        const int b = c0 + threadIdx.x;
        const uint32_t v = (b < nblocks) ? row[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_sum(v, sm, tot);
        if (b < nblocks) row[b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

constexpr int kRadix = 1 << kRadixBits;
static_assert(kRadix == kBlock, "one thread per digit");

// Digit width of the default sort.  10 bits (the 40-bit keys of max_depth 21 in 4 passes instead of 5)
// was measured and lost at every size: build 0.241 vs 0.215 ms at N = 1M, 0.093 vs 0.090 at 65k, 3.29 vs
// 2.80 at 16.7M -- with 1,024 digits a 2,048-key tile has two keys per digit, so the digit-sorted
// write-out degenerates into 16-byte runs and the count matrix is four times larger.
constexpr int kSortBits = 8;
// ITEMS keys per thread: 8 (tiles of 2,048) when there are enough tiles to fill the GPU, 2 (tiles of
// 512) for launches of few bodies, where a workgroup's 8 sequential rounds are pure latency.
// BITS: digit width; counts[digit * nblocks + block].
// ---- bucket sort (launches of up to kBucketMaxN bodies with packed keys) --------------------------------
// The LSD sort costs five passes of dependent short kernels (~9 us per pass at any size below 1M bodies).
// Consecutive builds sort almost the same bodies, so the engine sorts in TWO steps instead:
//   (1) ONE stable counting pass whose "digit" is the bucket of the key among 256 splitters (keys_kernel
//       derives them from the previous build's sorted positions: near-equal buckets), histogram + row scan
//       + scatter as for a radix digit;
//   (2) bucket_sort_kernel: one workgroup per bucket sorts its <= kBucketCap (16,384) keys entirely in LDS (stable
//       8-bit LSD passes over the key bits that differ inside the bucket) and writes plain keys + indices.
// The result is the stable sort by key whatever the splitters are -- they only have to be sorted, which
// keys_kernel guarantees -- so (keys_sorted, perm) is bit-identical to the LSD sort's.  A bucket that does
// not fit (bodies that moved wildly, a root box that jumped and scrambled the curve) is sorted by its
// workgroup through global memory: correct, slow, and gone at the next build.
// 256 buckets up to 1M bodies, 1,024 up to 4M (BITS = 8 / 10 in the counting pass): the average bucket stays
// at most a third of what one workgroup sorts in LDS.
constexpr int kBuckets = 256, kBucketsBig = 1024;
constexpr int kBucketStartOffset = kBucketsBig + 8;        // bsum_sort: [0, nb) bucket totals, [kBucketStartOffset, +nb) bucket starts
constexpr int kDigits = 256;                              // the in-LDS sort's own 8-bit digits
constexpr int kMaxSplitSamples = 2048;                    // sample positions sorted by keys_kernel's splitter workgroup
constexpr uint64_t kKeyMask40 = (1ull << 40) - 1;
// largest j with spl[j] <= key (spl sorted, spl[0] = 0)
template <int NB>
__device__ __forceinline__ uint32_t bucket_of(uint64_t key, const uint64_t *spl)
{
    int b = 0;
#pragma unroll
    for (int s = NB / 2; s >= 1; s >>= 1) b += (spl[b + s] <= key) ? s : 0;
    return (uint32_t)b;
}

// LDS atomics of one wave instruction that hit ONE counter are served lane after lane.  On sorted data -- the
// bucket digit of a tile, the upper bytes of the keys of a bucket -- that is every instruction of the count
// phase.  A full wave whose lanes all hold the same digit therefore adds 64 through one lane, and its ranks
// are the lane numbers (no ballot matching).
// (Used in the bucket pass only -- histogram and scatter, where a tile's keys fall into one or two buckets; in
// the byte passes of bucket_sort_kernel the test cost more on moving bodies than it saved on frozen ones:
// sort 0.051 / 0.083 ms frozen / drifting with it, 0.053 / 0.074 without, N = 1M.)
__device__ __forceinline__ bool wave_same_digit(uint32_t d, bool valid)
{
    return __ballot(valid) == ~0ull && __ballot(d != (uint32_t)__builtin_amdgcn_readfirstlane((int)d)) == 0ull;
}
// (TRY = false where digits are as good as random -- the low key bytes, every pass of the LSD sort on unsorted
// data: there the two extra ballots per key cost more than they save, LSD sort 89 -> 95 us at N = 1M.)
template <bool TRY = true>
__device__ __forceinline__ void wave_count_digit(uint32_t *row, uint32_t d, bool valid)
{
    if (TRY && wave_same_digit(d, valid)) {                     // uniform branch
        if (lane_id() == 0) atomicAdd(&row[d], (uint32_t)kWave);
    } else if (valid) {
        atomicAdd(&row[d], 1u);
    }
}

// BUCKET: the digit is the key's bucket among the splitters (shift unused)
template <int ITEMS, int BITS = kRadixBits, bool BUCKET = false>
__global__ __launch_bounds__(kBlock) void radix_hist(const uint64_t *__restrict__ keys,
                                                      uint32_t *__restrict__ counts, int64_t n,
                                                      int shift, int nblocks,
                                                      const uint64_t *__restrict__ splitters = nullptr,
                                                      uint16_t *__restrict__ dig16 = nullptr,
                                                      uint32_t *__restrict__ row_total = nullptr)
{
    constexpr int R = 1 << BITS;
    __shared__ uint32_t h[R];
    __shared__ uint64_t spl[BUCKET ? R : 1];
    // A key value frequent enough to be sampled twice (bodies piled into one depth-cap cell) gets a bucket of
    // its own: the second of two equal splitters becomes value + 1, so [value, value + 1) holds equal keys only
    // -- already in their final order after the stable scatter, whatever their number.
    if (BUCKET)
        for (int d = threadIdx.x; d < R; d += kBlock) {
            const uint64_t v = splitters[d];
            spl[d] = (d > 0 && splitters[d - 1] == v) ? v + 1 : v;
        }
    for (int d = threadIdx.x; d < R; d += kBlock) h[d] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS);
    uint32_t guess = 0;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = base + r * kBlock + threadIdx.x;
        const bool valid = i < n;
        const uint64_t k = valid ? keys[i] : 0ull;
        uint32_t d;
        if (BUCKET) {
            // The bodies arrive almost in order: a key 256 places further on lies in the same bucket as the last one, or
            // the next.  Two independent LDS reads confirm that for the whole wave; the search (eight or ten
            // DEPENDENT reads) runs only where a lane's guess fails.
            const uint64_t kk = k & kKeyMask40;
            bool ok = false;
            if (r > 0) ok = spl[guess] <= kk && (guess + 1 == (uint32_t)R || kk < spl[guess + 1]);
            if (r > 0 && __ballot(valid && !ok) == 0ull) d = guess;   // uniform
            else d = bucket_of<R>(kk, spl);
            guess = d;
        } else {
            d = (uint32_t)(k >> shift) & (R - 1);
        }
        if (BUCKET && valid) dig16[i] = (uint16_t)d;            // the scatter reads it back instead of searching again
        wave_count_digit<BUCKET>(h, d, valid);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < R; d += kBlock) {
        const uint32_t v = h[d];
        counts[(int64_t)d * nblocks + blockIdx.x] = v;
        // BUCKET: the bucket totals are summed here (keys_kernel zeroed them) and the scatter adds up the few rows it
        // needs itself -- the keys arrive almost in order, a tile meets one or two buckets -- so this pass has no
        // radix_rowscan launch between its two kernels
        if (BUCKET && row_total != nullptr && v) atomicAdd(&row_total[d], v);
    }
}


// ---- scatter with wave-private ranking --------------------------------------------------------------
// The round-1 scatter (scripts/experiments) synchronised the workgroup three times per round of 256 keys (24 barriers per
// 2,048-key tile), and a workgroup's rounds are sequential: with every tile resident at once the
// kernel's time IS that chain.  Here wave w owns the contiguous keys [w * 64 * ITEMS, (w+1) * 64 * ITEMS)
// of the tile: (1) every wave counts its digits into its own LDS histogram, (2) one barrier, the
// histograms are turned into per-wave start offsets (digit base + tile prefix + earlier waves),
// (3) one barrier, then each wave scatters its rounds on its own (ranks and counts were kept in registers
// from step 1) -- the only shared state left is the wave's own offset row, and LDS operations of one wave are
// ordered.  Element order is
// (wave, round, lane) = tile order, and ranks within a round follow lane order, so the sort stays
// stable and its output is bitwise the same.
// PACK: the body index travels in the key word itself (bits 40..63; the keys of max_depth <= 21 are 40 bits,
// the launch holds <= 2^24 bodies), so a pass reads and writes ONE 8-byte array instead of a key and a
// value array (16 instead of 24 bytes per element, half the memory instructions, 16 KB less LDS per tile).
// PACK == 2: the last pass, which writes the plain key and the index to their own arrays for the tree build.
constexpr int kPackShift = 40;
constexpr uint64_t kPackKeyMask = (1ull << kPackShift) - 1;
// RAW (bucket pass only): `offs` holds the histogram's raw counts and the kernel adds up the rows it needs itself (no
// radix_rowscan launch) -- for bodies that arrive almost in order.  The exact modes never re-order their state, a tile meets
// as many buckets as it has room for, and the pass keeps its row scan (RAW = false: 29 us against 9 at N = 1M).
template <int ITEMS, int BITS = kRadixBits, int PACK = 0, bool BUCKET = false, bool RAW = BUCKET>
__global__ __launch_bounds__(kBlock) void radix_scatter_w(const uint64_t *__restrict__ kin,
                                                           const uint32_t *__restrict__ vin,
                                                           uint64_t *__restrict__ kout,
                                                           uint32_t *__restrict__ vout,
                                                           const uint32_t *__restrict__ offs,
                                                           const uint32_t *__restrict__ row_total, int64_t n,
                                                           int shift, int nblocks,
                                                           const uint16_t *__restrict__ dig16 = nullptr,
                                                           uint32_t *__restrict__ bucket_start = nullptr)
{
    static_assert(!BUCKET || PACK == 1, "the bucket pass moves packed keys");
    // (4) the tile is first sorted by digit INTO LDS, then written out in that order: the keys a digit
    // has in a tile leave as one contiguous run instead of separate partial-line writes from
    // different rounds
    constexpr int TILE = kBlock * ITEMS;
    constexpr int R = 1 << BITS, DPT = R / kBlock;          // digits per thread (thread t owns DPT*t ..)
    static_assert(R % kBlock == 0, "whole digits per thread");
    __shared__ uint32_t woff[kWavesPerBlock][R];            // counts, then running local offsets, per wave
    __shared__ int32_t gdelta[R];                           // global position - position in the sorted tile
    __shared__ uint64_t skey[TILE];
    __shared__ uint32_t sval[PACK ? 1 : TILE];
    __shared__ uint32_t sm[kWavesPerBlock + 1];
    __shared__ uint16_t sdig[BUCKET ? TILE : 1];            // BUCKET: the bucket of every key of the sorted tile
    // per wave: digit -> mask of the lanes that hold it in the current round (see bucket_sort_lds below: an LDS
    // OR + read + clear per key instead of ~8 vector instructions per digit bit of ballot matching)
    __shared__ uint64_t match[kWavesPerBlock][R];
    constexpr int kRowsTogether = 64;
    __shared__ uint32_t s_np, s_plist[kRowsTogether], s_before[kRowsTogether];
    const int t = threadIdx.x, w = wave_id(), l = lane_id();
    for (int d = t; d < R; d += kBlock)
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) { woff[k][d] = 0; match[k][d] = 0ull; }
    __syncthreads();

    const int64_t tile_base = (int64_t)blockIdx.x * TILE;
    const int64_t base = tile_base + (int64_t)w * (kWave * ITEMS);
    uint64_t key[ITEMS];
    uint32_t val[PACK ? 1 : ITEMS];
    uint32_t dig[ITEMS], rank[ITEMS], npeer[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = base + r * kWave + l;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : ~0ull;
        if (!PACK) val[r] = valid ? vin[i] : 0u;
        dig[r] = BUCKET ? (valid ? (uint32_t)dig16[i] : 0u) : (uint32_t)(key[r] >> shift) & (R - 1);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool valid = base + r * kWave + l < n;
        const uint32_t d = dig[r];
        if (BUCKET && wave_same_digit(d, valid)) {              // (uniform) a full wave inside one bucket: the usual case
            rank[r] = (uint32_t)l;
            npeer[r] = (uint32_t)kWave;
            if (l == 0) woff[w][d] += (uint32_t)kWave;
            continue;
        }
        uint64_t peers = 0;
        if (valid) (void)__hip_atomic_fetch_or(&match[w][d], 1ull << l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        if (valid) peers = match[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid) match[w][d] = 0ull;
        __builtin_amdgcn_wave_barrier();
        rank[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        npeer[r] = (uint32_t)__popcll(peers);
        if (valid && rank[r] == 0) woff[w][d] += npeer[r];      // one lane per digit; this wave's row only
    }
    __syncthreads();
    {
        // exclusive prefixes over the digits, in digit order: thread t holds digits DPT*t .. DPT*t+DPT-1
        uint32_t cnt[DPT], tot[DPT], csum = 0, tsum = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = DPT * t + j;
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < kWavesPerBlock; ++k) c += woff[k][d];
            cnt[j] = c; csum += c;
            tot[j] = row_total[d]; tsum += tot[j];
        }
        uint32_t all;
        uint32_t digit_base = block_exclusive_sum(tsum, sm, all);
        uint32_t lstart = block_exclusive_sum(csum, sm, all);
        if (BUCKET && RAW) {
            // offs holds the histogram's raw counts: the keys of bucket d in the tiles before this one are the sum of row
            // d up to this tile.  The buckets this tile meets -- one or two in steady motion, a dozen when the bodies
            // have drifted for sixteen builds since the state was last put in order -- are listed, and the waves add up
            // their rows, four rows per wave at a time; a tile that meets more than kRowsTogether buckets (disorder) lets
            // every thread add up the rows of its own digits instead.
            if (t == 0) s_np = 0u;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < DPT; ++j)
                if (cnt[j]) { const uint32_t k = atomicAdd(&s_np, 1u); if (k < (uint32_t)kRowsTogether) s_plist[k] = (uint32_t)(DPT * t + j); }
            __syncthreads();
            const uint32_t np = s_np;
            if (np <= 2u) {                                      // steady motion: the whole workgroup on one row, then the other
                for (uint32_t k = 0; k < np; ++k) {
                    const uint32_t *row = offs + (int64_t)s_plist[k] * nblocks;
                    uint32_t part = 0;
                    for (uint32_t b = (uint32_t)t; b < blockIdx.x; b += kBlock) part += row[b];
                    uint32_t total;
                    (void)block_exclusive_sum(part, sm, total);
                    if (t == 0) s_before[k] = total;
                }
                __syncthreads();
            } else if (np <= (uint32_t)kRowsTogether) {
                for (uint32_t k0 = 4u * (uint32_t)w; k0 < np; k0 += 4u * kWavesPerBlock) {   // four rows per wave at a time
                    const uint32_t *row[4];
                    uint32_t part[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int j = 0; j < 4; ++j) row[j] = offs + (int64_t)s_plist[(k0 + j < np) ? k0 + j : k0] * nblocks;
                    for (uint32_t b = (uint32_t)l; b < blockIdx.x; b += kWave) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) part[j] += row[j][b];                   // (independent loads)
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t inc = wave_inclusive_sum(part[j]);
                        if (l == kWave - 1 && k0 + j < np) s_before[k0 + j] = inc;
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = DPT * t + j;
            uint32_t before = 0;                                // keys of digit d in the tiles before this one
            if (BUCKET && RAW) {
                if (cnt[j]) {
                    const uint32_t np = s_np;
                    if (np <= (uint32_t)kRowsTogether) {
                        for (uint32_t k = 0; k < np; ++k) before = (s_plist[k] == (uint32_t)d) ? s_before[k] : before;
                    } else {
                        const uint32_t *row = offs + (int64_t)d * nblocks;
#pragma unroll 8
                        for (uint32_t b = 0; b < blockIdx.x; ++b) before += row[b];
                    }
                }
            } else {
                before = offs[(int64_t)d * nblocks + blockIdx.x];
            }
            gdelta[d] = (int32_t)(digit_base + before) - (int32_t)lstart;
            if (BUCKET && blockIdx.x == 0) bucket_start[d] = digit_base;   // where bucket d begins: bucket_sort_kernel reads it
            uint32_t run = lstart;
#pragma unroll
            for (int k = 0; k < kWavesPerBlock; ++k) { const uint32_t c = woff[k][d]; woff[k][d] = run; run += c; }
            digit_base += tot[j];
            lstart += cnt[j];
        }
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool valid = base + r * kWave + l < n;
        const uint32_t d = dig[r];
        uint32_t o = 0;
        if (valid) o = woff[w][d];                          // every peer reads the same word ...
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            skey[o + rank[r]] = key[r];
            if (!PACK) sval[o + rank[r]] = val[r];
            if (BUCKET) sdig[o + rank[r]] = (uint16_t)d;
        }
        if (valid && rank[r] == 0) woff[w][d] = o + npeer[r];   // ... before the first peer advances it
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const int64_t left = n - tile_base;
    const int count = (left < TILE) ? (int)left : TILE;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int lp = r * kBlock + t;
        if (lp < count) {
            const uint64_t k = skey[lp];
            const int64_t dst = (int64_t)lp + gdelta[BUCKET ? (uint32_t)sdig[lp] : (uint32_t)(k >> shift) & (R - 1)];
            if (PACK == 2) {
                kout[dst] = k & kPackKeyMask;
                vout[dst] = (uint32_t)(k >> kPackShift);
            } else {
                kout[dst] = k;
                if (!PACK) vout[dst] = sval[lp];
            }
        }
    }
}

// ---- step (2) of the bucket sort ------------------------------------------------------------------------
// (A root box that moves by a depth-12 cell width per step re-aligns every finer cell, so in a dense core --
// where a bucket is a few such cells -- the re-keyed splitters are only as good as random ones at that scale:
// the largest of 256 buckets was 1.5x the average on the dynamic Plummer workload at N = 1.1M.  The LDS
// buffer therefore holds three times the average bucket of the largest launch.)
constexpr int kBsThreads = 1024, kBsWaves = kBsThreads / kWave;   // 16 waves: 4 per SIMD hide the LDS round trips
constexpr int kBucketItemsMax = 12;
constexpr int kBucketCap = kBucketItemsMax * kBsThreads;      // 12,288 keys (96 KB; + 16 KB offsets + 32 KB match tables)
// (the average bucket is at most a third of what fits: the largest of the buckets of a moving workload was
// measured at 1.5x the average)
constexpr int64_t kBucketMaxN = (int64_t)kBuckets * (kBucketCap / 3);       // 1,048,576 bodies with 256 buckets
constexpr int64_t kBucketMaxNBig = (int64_t)kBucketsBig * (kBucketCap / 3);  // 4,194,304 with 1,024

// exclusive scan of the 256 values held by threads 0..255 of the 1,024-thread workgroup (every thread calls;
// threads 256.. pass 0 and ignore the result)
__device__ __forceinline__ uint32_t bs_scan256(uint32_t v, uint32_t *sm, uint32_t &total)
{
    const int w = wave_id();
    const uint32_t inc = wave_inclusive_sum(v);
    if (w < 4 && lane_id() == kWave - 1) sm[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t q = sm[k]; base += (k < w) ? q : 0u; tot += q; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// m <= 1,024 * ITEMS keys of one bucket: stable LSD byte passes, keys in registers between passes (wave w owns the
// contiguous keys [w * 64 * ITEMS, ...) as in radix_scatter_w), one LDS buffer.
// Who shares my digit?  Matching by ballots costs ~8 vector instructions per digit bit and the kernel was bound
// by exactly those (8.4 M vector instructions per launch at N = 1M, 2,050 per wave).  Instead every wave
// owns a 256-entry table of 64-bit lane masks in LDS: a lane ORs its lane bit into the entry of its digit, reads
// the entry back -- the mask of its peers -- and clears it (LDS instructions of one wave execute in order).
//
// How many passes?  A workgroup's time is its chain of passes (barriers, the 256-digit scan, the trip through LDS:
// ~4.5 us each, whatever the number of keys -- profiles/r03_final/build_ab.txt), one workgroup per CU, and the kernel
// is as slow as its slowest bucket: the keys of a core bucket of a Plummer sphere span 2^26, those of a halo bucket
// 2^38 -- four passes against five.  So the keys are taken relative to the bucket's smallest (the SPAN counts, not
// the differing bit positions: a bucket astride a quadrant boundary differs in bit 39), and a bucket whose span
// exceeds 24 bits is sorted by the TOP 24 bits of its span in three passes, after which every run of keys equal in
// those bits -- almost always a single key: <= 12,288 keys over 2^24 values -- is put in order by counting inside
// the run.  A run longer than kFixRun (keys piled up on a few values below a wide span) sends the workgroup through
// the full passes instead.  Either way the result is the stable order by key.
constexpr int kFixRun = 8;
template <int ITEMS>
__device__ __forceinline__ void bucket_sort_lds(const uint64_t *__restrict__ in, int m, uint64_t *skey,
                                                uint32_t (*woff)[kDigits], uint64_t (*match)[kDigits],
                                                uint32_t *sm, uint64_t *s_red, uint32_t *s_flag,
                                                uint64_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                uint32_t *__restrict__ reruns)
{
    const int t = threadIdx.x, w = wave_id(), l = lane_id();
    const int wbase = w * (kWave * ITEMS);
    uint64_t key[ITEMS];
    uint64_t kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = wbase + r * kWave + l;
        key[r] = (i < m) ? in[i] : ~0ull;
        if (i < m) {
            const uint64_t k = key[r] & kKeyMask40;
            kmin = (k < kmin) ? k : kmin; kmax = (k > kmax) ? k : kmax;
        }
    }
    for (int k = t; k < kBsWaves * kDigits; k += kBsThreads) { (&woff[0][0])[k] = 0; (&match[0][0])[k] = 0ull; }
    if (t == 0) *s_flag = 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint64_t a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
        kmin = (a < kmin) ? a : kmin; kmax = (b > kmax) ? b : kmax;
    }
    if (l == 0) { s_red[w] = kmin; s_red[kBsWaves + w] = kmax; }
    __syncthreads();
    kmin = ~0ull; kmax = 0ull;
#pragma unroll
    for (int k = 0; k < kBsWaves; ++k) {
        const uint64_t a = s_red[k], b = s_red[kBsWaves + k];
        kmin = (a < kmin) ? a : kmin; kmax = (b > kmax) ? b : kmax;
    }
    const uint64_t span = kmax - kmin;
    const int bits = span ? 64 - __clzll(span) : 0;
    if (bits == 0) {                                            // all keys equal (or one key): order is final
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int i = wbase + r * kWave + l;
            if (i < m) { kout[i] = key[r] & kKeyMask40; vout[i] = (uint32_t)(key[r] >> kPackShift); }
        }
        return;
    }
    // keys relative to the smallest (the body index in bits 40.. is untouched: no borrow, key >= kmin)
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) key[r] -= (wbase + r * kWave + l < m) ? kmin : 0ull;
    const int full = (bits + 7) / 8;
    int first = 0, passes = full;
    bool fix = false;
    // (two top passes and runs of up to 16 keys: 20.1 against 20.8 us at N = 1M, and twice the keys per run -- not taken)
    if (full > 3) { first = bits - 24; passes = 3; fix = true; }
    for (;;) {
        for (int p = 0; p < passes; ++p) {
            const int shift = first + 8 * p;
            // peers, ranks and counts first (kept in registers), wave totals per digit by the first peer
            uint32_t rank[ITEMS], cnt[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int i = wbase + r * kWave + l;
                const bool valid = i < m;
                const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
                uint64_t peers = 0;
                if (valid) (void)__hip_atomic_fetch_or(&match[w][d], 1ull << l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_wave_barrier();
                if (valid) peers = match[w][d];
                __builtin_amdgcn_wave_barrier();
                if (valid) match[w][d] = 0ull;
                __builtin_amdgcn_wave_barrier();
                rank[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
                cnt[r] = (uint32_t)__popcll(peers);
                if (valid && rank[r] == 0) woff[w][d] += cnt[r];    // one lane per digit; this wave's row only
            }
            __syncthreads();
            {
                uint32_t c = 0;
                if (t < kDigits) {
#pragma unroll
                    for (int k = 0; k < kBsWaves; ++k) c += woff[k][t];
                }
                uint32_t all;
                uint32_t run = bs_scan256(c, sm, all);
                if (t < kDigits) {
#pragma unroll
                    for (int k = 0; k < kBsWaves; ++k) { const uint32_t cc = woff[k][t]; woff[k][t] = run; run += cc; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int i = wbase + r * kWave + l;
                const bool valid = i < m;
                const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
                uint32_t o = 0;
                if (valid) o = woff[w][d];                          // every peer reads the same word ...
                __builtin_amdgcn_wave_barrier();
                if (valid) skey[o + rank[r]] = key[r];
                if (valid && rank[r] == 0) woff[w][d] = o + cnt[r]; // ... before the first peer advances it
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            if (p + 1 < passes) {
#pragma unroll
                for (int r = 0; r < ITEMS; ++r) {
                    const int i = wbase + r * kWave + l;
                    key[r] = (i < m) ? skey[i] : ~0ull;
                }
                for (int k = t; k < kBsWaves * kDigits; k += kBsThreads) (&woff[0][0])[k] = 0;
                __syncthreads();
            }
        }
        if (!fix) break;
        // ---- the keys are in order of their top 24 span bits, equal ones in input order: finish every run of equal
        // top bits by counting.  Position = own position + (later keys of the run that are smaller) - (earlier keys of
        // the run that are larger); keys equal in all bits keep their order.
        const uint64_t lowmask = (1ull << first) - 1ull;
        bool too_long = false;
        // (every key goes straight to its place in memory, at most kFixRun places from a coalesced store; should a run turn
        // out too long, the full sort below overwrites all of it)
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int i = t + r * kBsThreads;
            if (i < m) {
                const uint64_t word = skey[i];
                const uint64_t kk = word & kKeyMask40;
                const uint64_t top = kk >> first, low = kk & lowmask;
                int delta = 0, j;
                for (j = i - 1; j >= 0 && i - j <= kFixRun; --j) {
                    const uint64_t q = skey[j] & kKeyMask40;
                    if ((q >> first) != top) break;
                    delta -= ((q & lowmask) > low) ? 1 : 0;
                }
                too_long = too_long || (j >= 0 && i - j > kFixRun);
                for (j = i + 1; j < m && j - i <= kFixRun; ++j) {
                    const uint64_t q = skey[j] & kKeyMask40;
                    if ((q >> first) != top) break;
                    delta += ((q & lowmask) < low) ? 1 : 0;
                }
                too_long = too_long || (j < m && j - i > kFixRun);
                kout[i + delta] = kk + kmin;
                vout[i + delta] = (uint32_t)(word >> kPackShift);
            }
        }
        if (too_long) *s_flag = 1u;
        __syncthreads();
        if (*s_flag == 0u) return;
        // a long run: the full passes from the input (rare: bodies piled up on a few key values under a wide span)
        if (t == 0) atomicAdd(reruns, 1u);
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const int i = wbase + r * kWave + l;
            key[r] = (i < m) ? in[i] - kmin : ~0ull;
        }
        for (int k = t; k < kBsWaves * kDigits; k += kBsThreads) (&woff[0][0])[k] = 0;
        __syncthreads();
        first = 0; passes = full; fix = false;
    }
    for (int i = t; i < m; i += kBsThreads) {
        const uint64_t k = skey[i];
        kout[i] = (k & kKeyMask40) + kmin;
        vout[i] = (uint32_t)(k >> kPackShift);
    }
}

// One workgroup of 1,024 threads per bucket.  bucketed: the packed keys grouped by bucket (step 1's output);
// kout / vout: the sorted plain keys and body indices; kout doubles as the second buffer of the
// through-memory path.
__global__ __launch_bounds__(kBsThreads) void bucket_sort_kernel(uint64_t *__restrict__ bucketed,
                                                                  uint64_t *__restrict__ kout,
                                                                  uint32_t *__restrict__ vout,
                                                                  const uint32_t *__restrict__ bucket_total,
                                                                  const uint32_t *__restrict__ bucket_start,
                                                                  uint32_t *__restrict__ spills,
                                                                  uint32_t *__restrict__ reruns, int single_m)
{
    __shared__ uint64_t skey[kBucketCap];
    __shared__ uint32_t woff[kBsWaves][kDigits];
    __shared__ uint64_t match[kBsWaves][kDigits];
    __shared__ uint32_t sm[8];
    __shared__ uint64_t s_or[2 * kBsWaves];
    __shared__ uint32_t s_flag;
    const int t = threadIdx.x, w = wave_id(), l = lane_id();
    // where this bucket starts and how many keys it holds: the counting pass's scatter left both (wave-uniform loads)
    // (bucket_total == nullptr: a launch of at most kBucketCap keys is ONE bucket -- the keys as keys_kernel left them, no
    // counting pass in front: three launches less for the small launches whose builds are chains of 5 us launches)
    const int64_t start = bucket_total ? (int64_t)bucket_start[blockIdx.x] : 0;
    const int m = bucket_total ? (int)bucket_total[blockIdx.x] : single_m;
    if (m == 0) return;
    uint64_t *in = bucketed + start;
    uint64_t *ko = kout + start;
    uint32_t *vo = vout + start;

    if (m <= 1 * kBsThreads) { bucket_sort_lds<1>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= 2 * kBsThreads) { bucket_sort_lds<2>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= 4 * kBsThreads) { bucket_sort_lds<4>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= 5 * kBsThreads) { bucket_sort_lds<5>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= 6 * kBsThreads) { bucket_sort_lds<6>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= 8 * kBsThreads) { bucket_sort_lds<8>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }
    if (m <= kBucketCap) { bucket_sort_lds<kBucketItemsMax>(in, m, skey, woff, match, sm, s_or, &s_flag, ko, vo, reruns); return; }

    // which key bits differ inside the bucket: only those bytes need a pass
    const uint64_t k0 = in[0];
    uint64_t x = 0;
    for (int i = t; i < m; i += kBsThreads) x |= (in[i] ^ k0) & kKeyMask40;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x |= __shfl_xor(x, o);
    if (l == 0) s_or[w] = x;
    __syncthreads();
    x = 0;
#pragma unroll
    for (int k = 0; k < kBsWaves; ++k) x |= s_or[k];
    const int passes = x ? (64 - __clzll(x) + 7) / 8 : 0;

    // ---- the bucket does not fit: the same stable passes through global memory, 1,024 keys per round, by this
    // workgroup alone (all its waves share one L1: workgroup barriers order the stores and the loads)
    if (t == 0 && passes > 0) atomicAdd(spills, 1u);           // (a bucket of equal keys is only copied)
    uint64_t *src = in, *dst = ko;
    uint32_t *hist = reinterpret_cast<uint32_t *>(skey);        // [256] counts, then running bases
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (t < kDigits) hist[t] = 0;
        for (int k = t; k < kBsWaves * kDigits; k += kBsThreads) (&woff[0][0])[k] = 0;
        __syncthreads();
        for (int i = t; i < m; i += kBsThreads) atomicAdd(&hist[(uint32_t)(src[i] >> shift) & 255u], 1u);
        __syncthreads();
        {
            const uint32_t c = (t < kDigits) ? hist[t] : 0u;
            uint32_t all;
            const uint32_t e = bs_scan256(c, sm, all);
            if (t < kDigits) hist[t] = e;
        }
        __syncthreads();
        for (int i0 = 0; i0 < m; i0 += kBsThreads) {
            const int i = i0 + t;
            const bool valid = i < m;
            const uint64_t key = valid ? src[i] : ~0ull;
            const uint32_t d = (uint32_t)(key >> shift) & 255u;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t rank = __popcll(peers & lt);
            if (valid && rank == 0) woff[w][d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (valid) {
                uint32_t o = hist[d] + rank;
                for (int k = 0; k < w; ++k) o += woff[k][d];
                dst[o] = key;
            }
            __syncthreads();
            if (t < kDigits) {
                uint32_t sacc = 0;
#pragma unroll
                for (int k = 0; k < kBsWaves; ++k) { sacc += woff[k][t]; woff[k][t] = 0; }
                hist[t] += sacc;
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        uint64_t *tmp = src; src = dst; dst = tmp;
    }
    // unpack where the keys ended up (element-wise, so doing it in place in kout is safe)
#pragma unroll 8
    for (int i = t; i < m; i += kBsThreads) {
        const uint64_t k = src[i];
        ko[i] = k & kKeyMask40;
        vo[i] = (uint32_t)(k >> kPackShift);
    }
}


}  // namespace bh
