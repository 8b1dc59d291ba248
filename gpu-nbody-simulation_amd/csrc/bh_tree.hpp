// bh_tree.hpp -- device construction of the reference's quadtree.
//
// The reference inserts bodies one by one on the host (QuadInsert, project.cu:358-453).  The
// resulting tree does not depend on insertion order: a cell is subdivided iff it holds >= 2
// bodies and its depth is below the cap, all four children are allocated together
// (project.cu:410-434), and child selection uses recursively halved fp64 midpoints
// (project.cu:349-355, 417-418).  So the same tree can be derived from sorted Morton-style keys
// whose digits are produced by the SAME fp64 bisection:
//
//   keys      : per body, max_depth-1 bisection steps -> 2 bits per level (child index 0..3);
//               the same kernel histograms the first radix digit
//   sort      : stable radix sort of (key, body) -> bodies of one cell are contiguous and, inside
//               a depth-cap cell, stay in body order (the order the reference's running
//               centre-of-mass fold uses, project.cu:371-373)
//   pairs     : L[i] = levels shared by sorted neighbours i, i+1.  The subdivided cell at depth d
//               whose first body is i exists iff L[i-1] < d <= min(L[i], Dm-1); pair i "owns"
//               those cells.  An exclusive scan of the counts numbers all subdivided cells in
//               DFS pre-order ("rank").
//   nodes     : the owner of rank r writes the four children of that cell at node ids
//               1+4r .. 1+4r+3 (root = node 0): bounds by halving, leaf payloads, child links.
//   com       : exact mode: bottom-up, level by level, children summed in index order exactly as
//               ComputeMass does (project.cu:473-502).  fp32 mode: from fp64 prefix sums.
//
// Launch count matters as much as kernel quality here: on MI355X every launch costs ~6 us (4.7 us
// minimum kernel duration + boundary).  Per step at max_depth 21: bounds_final | keys |
// 5 x (hist, rowscan, scatter) | prep | scan_top2 | scan_apply2 | nodes | walk = 22 launches
// (it was 40).
#pragma once

#include <type_traits>

#include "bh_prims.hpp"
#include "bh_sort.hpp"
#include "bh_nodes.hpp"
#include "bh_bounds.hpp"

namespace bh {

// every kCoarse-th sorted key: a 32 KB index at N = 1M that stays hot in L2, so the range search
// of a LARGE cell costs ~12 hot loads + 8 more instead of ~40 cold ones (those searches are the
// critical path of the node kernel: the few cells near the root)
constexpr int kCoarse = 256;

__device__ __forceinline__ int shared_levels(uint64_t a, uint64_t b, int Dm)
{
    const uint64_t x = a ^ b;
    if (x == 0) return Dm;
    return (__clzll((long long)x) - (64 - 2 * Dm)) >> 1;
}

// ---- root box: ComputeRootBounds, project.cu:536-573 ------------------------------------------
template <typename Real2>
__global__ __launch_bounds__(kBlock) void bounds_partial(const Real2 *__restrict__ pos, int64_t n,
                                                          double *__restrict__ partial)
{
    __shared__ double sm[4][kWavesPerBlock];
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double x = (double)pos[i].x, y = (double)pos[i].y;
        xlo = (x < xlo) ? x : xlo;  xhi = (xhi < x) ? x : xhi;
        ylo = (y < ylo) ? y : ylo;  yhi = (yhi < y) ? y : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (lane_id() == 0) { sm[0][wave_id()] = xlo; sm[1][wave_id()] = xhi; sm[2][wave_id()] = ylo; sm[3][wave_id()] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            xlo = (sm[0][w] < xlo) ? sm[0][w] : xlo;  xhi = (xhi < sm[1][w]) ? sm[1][w] : xhi;
            ylo = (sm[2][w] < ylo) ? sm[2][w] : ylo;  yhi = (yhi < sm[3][w]) ? sm[3][w] : yhi;
        }
        partial[4 * blockIdx.x + 0] = xlo; partial[4 * blockIdx.x + 1] = xhi;
        partial[4 * blockIdx.x + 2] = ylo; partial[4 * blockIdx.x + 3] = yhi;
    }
}

__global__ void bounds_slots_reset(double *slots)                // one wave: records back to +-inf
{
    double *w = slots + 4 * threadIdx.x;
    w[0] = INFINITY; w[1] = -INFINITY; w[2] = INFINITY; w[3] = -INFINITY;
}

// final reduction + the padding of project.cu:553-570; also clears the per-step counters.
// from_walk: the partials are what the previous walk's epilogue left.  A walk whose tree had outgrown node_capacity
// returned at once and wrote none: the box and the overflow flag stay as they are (the bodies have not moved), so every
// later step of the same bh_step call is a no-op too and bh_sync / bh_download report BH_ERR_CAPACITY.
__global__ __launch_bounds__(kBlock) void bounds_final(const double *__restrict__ partial, int nb,
                                                        double *__restrict__ box, TreeCounters *ctr, int Dm, int from_walk)
{
    __shared__ double sm[4][kWavesPerBlock];
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
#pragma unroll 8
    for (int i = threadIdx.x; i < nb; i += kBlock) {
        const double a = partial[4 * i], b = partial[4 * i + 1], c = partial[4 * i + 2], d = partial[4 * i + 3];
        xlo = (a < xlo) ? a : xlo;  xhi = (xhi < b) ? b : xhi;
        ylo = (c < ylo) ? c : ylo;  yhi = (yhi < d) ? d : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (lane_id() == 0) { sm[0][wave_id()] = xlo; sm[1][wave_id()] = xhi; sm[2][wave_id()] = ylo; sm[3][wave_id()] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            xlo = (sm[0][w] < xlo) ? sm[0][w] : xlo;  xhi = (xhi < sm[1][w]) ? sm[1][w] : xhi;
            ylo = (sm[2][w] < ylo) ? sm[2][w] : ylo;  yhi = (yhi < sm[3][w]) ? sm[3][w] : yhi;
        }
        if (!(from_walk && ctr->overflow)) {
            const double ex = xhi - xlo, ey = yhi - ylo;
            const double span = (ex < ey) ? ey : ex;
            double pad = 0.1 * span;
            if (span == 0.0) pad = 1e-6;
            box[0] = xlo - pad; box[1] = xhi + pad; box[2] = ylo - pad; box[3] = yhi + pad;
            write_key_consts(box, Dm);
            ctr->overflow = 0;
        }
        ctr->n_internal = 0;
        ctr->visits = 0; ctr->interactions = 0; ctr->wave_nodes = 0; ctr->wave_quads = 0; ctr->wave_accepts = 0;
    }
}

// ---- keys: DetermineChild (project.cu:348-356) applied max_depth-1 times -----------------------
__device__ __forceinline__ int pick_child(double x, double y, double mx, double my)
{
    if (x <  mx && y <  my) return 0;
    if (x >= mx && y <  my) return 1;
    if (x <  mx && y >= my) return 2;
    return 3;
}

__device__ __forceinline__ void descend(int c, double mx, double my, double &x0, double &x1,
                                        double &y0, double &y1)
{
    if (c & 1) x0 = mx; else x1 = mx;
    if (c & 2) y0 = my; else y1 = my;
}

// Hilbert curve as a 4-state machine over the geometric quadrant c = (y >= mid)*2 + (x >= mid)
// (derived from the classic xy2d definition and verified against it): digit h = H[state][c],
// next state = S[state][c], two bits per entry.
//   H = {0,3,1,2} {0,1,3,2} {2,3,1,0} {2,1,3,0}     S = {1,2,0,0} {0,1,3,1} {2,0,2,3} {3,3,1,2}
constexpr uint32_t kHilbertH = 0x361EB49Cu, kHilbertS = 0x9FE27409u;
__host__ __device__ __forceinline__ int hilbert_digit(int state, int c) { return (int)((kHilbertH >> (2 * (4 * state + c))) & 3u); }
__host__ __device__ __forceinline__ int hilbert_next(int state, int c) { return (int)((kHilbertS >> (2 * (4 * state + c))) & 3u); }

// HILBERT (fp32 mode): the key digits follow the Hilbert curve instead of the child index.  The
// cells -- and therefore the tree -- are the same (a quadtree cell is one contiguous key range on
// either curve); only the order of the four siblings inside a quad and the order of the bodies
// change.  64 consecutive bodies then form a more compact patch, and a wavefront touches ~7 %
// fewer distinct nodes (measured with the oracle: U64 13.99 -> 13.08 Plummer, 10.79 -> 9.97
// uniform).  Exact mode keeps child-index order (the reference's depth-cap fold follows it).
//
// DetermineChild with two compares instead of four -- !(x < mx) is (x >= mx) -- as long as every
// compare is ordered: finite body, finite box.  Anything else (a body at infinity blows the box up,
// NaN coordinates) takes the reference's four tests, which send unordered compares to child 3 and so
// collapse such bodies into one cell instead of spreading them over a tree of their own.
template <bool HILBERT>
__device__ __forceinline__ uint64_t key_of(double x, double y, double x0, double x1, double y0, double y1, int Dm)
{
    uint64_t k = 0;
    int state = 0;
    const bool ordered = HILBERT && isfinite(x) && isfinite(y) && isfinite(x0) && isfinite(x1) && isfinite(y0) &&
                         isfinite(y1);
    if (ordered) {
        for (int l = 0; l < Dm; ++l) {
            const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
            const bool bx = !(x < mx), by = !(y < my);
            const int c = (bx ? 1 : 0) | (by ? 2 : 0);
            k = (k << 2) | (uint64_t)hilbert_digit(state, c);
            state = hilbert_next(state, c);
            if (bx) x0 = mx; else x1 = mx;
            if (by) y0 = my; else y1 = my;
        }
    } else {
        for (int l = 0; l < Dm; ++l) {
            const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
            const int c = pick_child(x, y, mx, my);
            if (HILBERT) {
                k = (k << 2) | (uint64_t)hilbert_digit(state, c);
                state = hilbert_next(state, c);
            } else {
                k = (k << 2) | (uint64_t)c;
            }
            descend(c, mx, my, x0, x1, y0, y1);
        }
    }
    return k;
}

// The same key without the bisection, for almost every body.  The reference's cell boundaries are midpoints of
// midpoints, each rounded once: boundary k of the finest grid is within Dm * ulp(M)/2 of x0 + k * w / 2^Dm
// (M = max |box corner|, w = the box width; the halving is exact and averaging does not amplify the operands'
// errors).  t = (x - x0) * (2^Dm / w), computed in fp64, is within 2^(Dm-50) cells of the exact quotient.  So
// when t is farther than `margin` = (Dm + 8) * 2^-53 * 2^Dm * M / w from an integer, every comparison of the
// bisection is decided the same way and floor(t) IS the cell; the rest (a body within ~2^-28 cell widths of a
// grid line, a body on the box edge, non-finite input: ~1 body in 10^8) takes key_of.  The curve digits
// then come from the integer cell coordinates.  Same keys, bit for bit (tests: every tree comparison against
// the oracle, and test_fast_keys_equal_the_bisection_keys on grid-line and box-edge positions).
// Three levels of the curve per table look-up: entry [state << 6 | iy3 << 3 | ix3] = next state << 6 | the three digits
// (the state machine above run over the three bit pairs, high bits first).  256 bytes, built by every key workgroup in
// LDS (one entry per thread).  A level by itself is ~14 vector instructions (two bit extractions, two variable shifts of
// the packed tables, the 64-bit shift-or of the key): 280 of keys_kernel's ~300 per body at 20 levels; three levels per
// look-up are ~9 and one dependent LDS read.
__device__ __forceinline__ uint32_t hilbert3_entry(uint32_t t)
{
    int state = (int)(t >> 6);
    const uint32_t iy3 = (t >> 3) & 7u, ix3 = t & 7u;
    uint32_t digits = 0;
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        const int c = (int)(((ix3 >> l) & 1u) | (((iy3 >> l) & 1u) << 1));
        digits = (digits << 2) | (uint32_t)hilbert_digit(state, c);
        state = hilbert_next(state, c);
    }
    return ((uint32_t)state << 6) | digits;
}

template <bool HILBERT>
__device__ __forceinline__ uint64_t key_of_fast(double x, double y, double x0, double x1, double y0, double y1, int Dm,
                                                double scale_x, double scale_y, double margin_x, double margin_y,
                                                const uint8_t *hil3 = nullptr)
{
    const double side = (double)(1u << Dm);
    const double tx = (x - x0) * scale_x, ty = (y - y0) * scale_y;
    const double fx = floor(tx), fy = floor(ty);
    const double rx = tx - fx, ry = ty - fy;
    // (written so that a NaN or an infinity anywhere makes the test fail)
    const bool sure = rx > margin_x && rx < 1.0 - margin_x && ry > margin_y && ry < 1.0 - margin_y &&
                      fx >= 0.0 && fx < side && fy >= 0.0 && fy < side;
    if (!sure) return key_of<HILBERT>(x, y, x0, x1, y0, y1, Dm);
    const uint32_t ix = (uint32_t)fx, iy = (uint32_t)fy;
    uint64_t k = 0;
    int state = 0;
    int l = Dm - 1;
    if (HILBERT && hil3 != nullptr) {
        for (; (l + 1) % 3 != 0; --l) {                          // the one or two levels above a multiple of three
            const int c = (int)(((ix >> l) & 1u) | (((iy >> l) & 1u) << 1));
            k = (k << 2) | (uint64_t)hilbert_digit(state, c);
            state = hilbert_next(state, c);
        }
        uint32_t st = (uint32_t)state;
        for (; l >= 2; l -= 3) {
            const uint32_t e = hil3[(st << 6) | (((iy >> (l - 2)) & 7u) << 3) | ((ix >> (l - 2)) & 7u)];
            k = (k << 6) | (uint64_t)(e & 63u);
            st = e >> 6;
        }
        return k;
    }
    if (!HILBERT) {
        // child-index order (the exact modes): digit = (x >= mid) | (y >= mid) << 1 per level, i.e. the bits of ix on the even
        // and those of iy on the odd places of the key -- five shift-and-mask steps per axis instead of a loop over the levels
        auto spread = [](uint32_t v) -> uint64_t {
            uint64_t u = v;
            u = (u | (u << 16)) & 0x0000FFFF0000FFFFull;
            u = (u | (u << 8)) & 0x00FF00FF00FF00FFull;
            u = (u | (u << 4)) & 0x0F0F0F0F0F0F0F0Full;
            u = (u | (u << 2)) & 0x3333333333333333ull;
            u = (u | (u << 1)) & 0x5555555555555555ull;
            return u;
        };
        return spread(ix) | (spread(iy) << 1);
    }
    for (; l >= 0; --l) {
        const int c = (int)(((ix >> l) & 1u) | (((iy >> l) & 1u) << 1));
        k = (k << 2) | (uint64_t)hilbert_digit(state, c);
        state = hilbert_next(state, c);
    }
    return k;
}

// PACK: the body index is written into bits 40..63 of the key word instead of the index array (bh_sort.hpp).
// samples != nullptr (bucket sort, bh_sort.hpp): ns / 64 extra workgroups produce the nb (256 or 1,024) splitters
// of this build -- the keys, in THIS build's box, of the positions that stood at the ranks j * n / nb of the previous
// build's sorted order (ns >= nb of them, powers of two), sorted; splitter 0 is 0.  Bodies move little between two builds, so
// the splitters cut the new keys into near-equal buckets, whatever happened to the root box in between.
template <typename Real2, bool HILBERT, bool PACK = false, bool FROM_SLOTS = false>
__global__ __launch_bounds__(kBlock) void keys_kernel(const Real2 *__restrict__ pos,
                                                       double *box_global,
                                                       uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ idx, int64_t n, int Dm,
                                                       const float2 *__restrict__ samples = nullptr,
                                                       uint64_t *__restrict__ splitters = nullptr, int nb = 0,
                                                       int ns = 0, double *slots = nullptr, TreeCounters *ctr = nullptr,
                                                       uint32_t *__restrict__ zero_words = nullptr, int n_zero = 0,
                                                       const uint32_t *__restrict__ sample_perm = nullptr)
{
    if (blockIdx.x == 0)                                         // (the bucket totals the histogram launch adds into: bh_sort.hpp)
        for (int k = threadIdx.x; k < n_zero; k += kBlock) zero_words[k] = 0u;
    // FROM_SLOTS: the root box is not in memory yet -- the previous walk left its bounds in kBoundSlots
    // records (bh_bounds.hpp).  Every workgroup reduces them and pads the box as bounds_final does
    // (project.cu:553-570), workgroup 0 also writes it out and clears the step's counters for the kernels that
    // follow; prep_kernel, two launches on, puts the slots back to +-inf for the next walk.  (A counter of readers
    // that let the last workgroup do that here made this kernel 51 us instead of 10: 4,100 atomics on one word.)
    __shared__ double s_box[FROM_SLOTS ? 8 : 1];
    __shared__ int s_empty;
    if (FROM_SLOTS) {
        if (threadIdx.x < kWave) {
            const double *sl = slots + 4 * threadIdx.x;
            const double xlo = wave_min(sl[0]), xhi = wave_max(sl[1]), ylo = wave_min(sl[2]), yhi = wave_max(sl[3]);
            // No workgroup folded anything (every record still +-inf): the previous walk returned at once because its tree
            // had outgrown node_capacity (`if (ctr->overflow) return`), so the bodies have not moved.  Keep the box in
            // memory and leave the overflow flag standing -- this build's walk then does nothing either, and the state
            // stays the last good one until bh_sync / bh_download report BH_ERR_CAPACITY (a box of {+inf, -inf}
            // gave every body the same key, a short tree that did NOT overflow, and a step integrated with garbage forces).
            const bool empty = xlo > xhi;
            if (threadIdx.x == 0) s_empty = empty ? 1 : 0;
            if (threadIdx.x < 2) {                               // lane a: axis a (four fp64 divisions in a row otherwise)
                const int a = threadIdx.x;
                if (empty) {
                    s_box[2 * a] = box_global[2 * a]; s_box[2 * a + 1] = box_global[2 * a + 1];
                    s_box[4 + a] = box_global[4 + a]; s_box[6 + a] = box_global[6 + a];
                } else {
                    const double ex = xhi - xlo, ey = yhi - ylo;
                    const double span = (ex < ey) ? ey : ex;
                    double pad = 0.1 * span;
                    if (span == 0.0) pad = 1e-6;
                    s_box[2 * a] = (a ? ylo : xlo) - pad; s_box[2 * a + 1] = (a ? yhi : xhi) + pad;
                    write_key_consts_axis(s_box, a, Dm);
                }
            }
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (!s_empty) {
#pragma unroll
                for (int k = 0; k < 8; ++k) box_global[k] = s_box[k];
                ctr->overflow = 0;
            }
            ctr->n_internal = 0;
            ctr->visits = 0; ctr->interactions = 0; ctr->wave_nodes = 0; ctr->wave_quads = 0; ctr->wave_accepts = 0;
        }
    }
    // (a template parameter, not a run-time test: read through a pointer that is either in LDS or in memory the eight values
    // become flat loads per lane; kept apart they are LDS broadcasts here and scalar loads there)
    double x0, x1, y0, y1, bk4, bk5, bk6, bk7;
    if (FROM_SLOTS) {
        x0 = s_box[0]; x1 = s_box[1]; y0 = s_box[2]; y1 = s_box[3]; bk4 = s_box[4]; bk5 = s_box[5]; bk6 = s_box[6]; bk7 = s_box[7];
    } else {
        x0 = box_global[0]; x1 = box_global[1]; y0 = box_global[2]; y1 = box_global[3];
        bk4 = box_global[4]; bk5 = box_global[5]; bk6 = box_global[6]; bk7 = box_global[7];
    }
    // (sample_perm: the exact modes keep no sorted copy of the positions -- their samples are pos[previous build's perm[rank]])
    const int nsb = (samples != nullptr || sample_perm != nullptr) ? ns / kWave : 0;      // extra workgroups in front of the key workgroups
    if ((int)blockIdx.x < nsb) {
        // ns sample positions -> keys -> ranks by counting -> every (ns / nb)-th in rank order is a splitter.
        // Every one of the ns / 64 sample workgroups forms all ns keys (cheap: one multiply per axis with the
        // box's precomputed scale -- not the exact bisection, a splitter does not have to be the key of
        // anything, the splitters only have to be sorted) and ranks 64 of them, four threads per sample, so a
        // sample workgroup does not outlast a key workgroup.  (Oversampling evens the buckets out when the
        // re-keyed samples are only as good as random ones; the largest bucket decides bucket_sort_kernel's time.)
        __shared__ uint64_t sk[kMaxSplitSamples];
        const int t = threadIdx.x;
        // (a sample workgroup shares its CU with key workgroups, so its time is its instruction count: ns, nb are
        // powers of two -- shifts, not divisions.  All Dm levels: when close encounters have blown the root box up,
        // the bodies sit in a corner of it that 12 levels do not resolve, every splitter would be the same key
        // and one bucket would get everything.)
        const int Ds = Dm, lg_ns = 31 - __clz(ns), lg_os = lg_ns - (31 - __clz(nb));
        const double side = (double)(1u << Ds), top = side - 1.0, down = 1.0 / (double)(1u << (Dm - Ds));
        const double sx = (bk4 > 0.0) ? bk4 * down : side / (x1 - x0), sy = (bk5 > 0.0) ? bk5 * down : side / (y1 - y0);
        float2 qs[kMaxSplitSamples / kBlock];                    // all of this thread's sample loads in flight at once
#pragma unroll
        for (int k = 0; k < kMaxSplitSamples / kBlock; ++k) {
            const int j = t + k * kBlock;
            qs[k] = float2{0.f, 0.f};
            if (j < ns) {
                const int64_t rk = ((int64_t)j * n) >> lg_ns;
                if (sample_perm != nullptr) { const Real2 pp = pos[sample_perm[rk]]; qs[k] = float2{(float)pp.x, (float)pp.y}; }
                else qs[k] = samples[rk];
            }
        }
#pragma unroll
        for (int k = 0; k < kMaxSplitSamples / kBlock; ++k) {
            const int j = t + k * kBlock;
            if (j >= ns) break;
            uint64_t mine = 0;
            if (j > 0) {
                const float2 q = qs[k];
                const double fx = fmin(fmax(((double)q.x - x0) * sx, 0.0), top);
                const double fy = fmin(fmax(((double)q.y - y0) * sy, 0.0), top);
                const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
                int state = 0;
                for (int l = Ds - 1; l >= 0; --l) {
                    const int cc = (int)(((ix >> l) & 1u) | (((iy >> l) & 1u) << 1));
                    mine = (mine << 2) | (uint64_t)(HILBERT ? hilbert_digit(state, cc) : cc);
                    state = hilbert_next(state, cc);
                }
            }
            sk[j] = (mine << 11) | (uint64_t)j;                 // the sample index makes the values distinct
        }
        __syncthreads();
        const int sidx = (int)blockIdx.x * kWave + (t >> 2), part = t & 3, span = ns >> 2;
        const uint64_t tagged = sk[sidx];
        int rank = 0;
#pragma unroll 16
        for (int q = part * span; q < (part + 1) * span; ++q) rank += (sk[q] < tagged) ? 1 : 0;
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        if (part == 0 && (rank & ((1 << lg_os) - 1)) == 0) splitters[rank >> lg_os] = tagged >> 11;
        return;
    }
    __shared__ uint8_t s_hil3[HILBERT ? 256 : 1];
    if (HILBERT) {
        static_assert(kBlock == 256, "one table entry per thread");
        s_hil3[threadIdx.x] = (uint8_t)hilbert3_entry(threadIdx.x);
        __syncthreads();
    }
    const int64_t i = ((int64_t)blockIdx.x - nsb) * kBlock + threadIdx.x;
    if (i >= n) return;
    // (round 4: the exact modes take the one-multiply look-up too -- the proof is about the bisection's comparisons, not about the
    // order of the digits; bodies it cannot vouch for fall back to key_of, the reference's four-test DetermineChild)
    const uint64_t k = key_of_fast<HILBERT>((double)pos[i].x, (double)pos[i].y, x0, x1, y0, y1, Dm, bk4, bk5, bk6, bk7,
                                            HILBERT ? s_hil3 : nullptr);
    if (PACK) {
        keys[i] = k | ((uint64_t)i << kPackShift);
    } else {
        keys[i] = k;
        idx[i] = (uint32_t)i;
    }
}

// ---- after the sort: cell counts per sorted body (+ fp32: sorted copies and prefix-sum terms) ----
// One workgroup per scan tile (kTile consecutive sorted bodies); element k*256 + t belongs to
// thread t, so every access is coalesced.  Also produces the tile sums of both scans, so no
// separate reduction launches are needed.
// (Real2/Real: the state's type; SReal2/SReal: the type of the sorted copies the walk reads)
template <bool EXACT, int ITEMS, typename Real2, typename Real, typename SReal2 = Real2, typename SReal = Real>
__global__ __launch_bounds__(kBlock) void prep_kernel(const uint64_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ perm,
                                                       const Real2 *__restrict__ pos,
                                                       const Real *__restrict__ mass,
                                                       uint32_t *__restrict__ cnt,
                                                       uint32_t *__restrict__ bsum_u32,
                                                       SReal2 *__restrict__ spos, SReal *__restrict__ smass,
                                                       d3 *__restrict__ terms, d3 *__restrict__ bsum_d3,
                                                       uint64_t *__restrict__ coarse, int64_t n, int Dm,
                                                       double *slots_reset = nullptr)
{
    if (slots_reset != nullptr && blockIdx.x == 0 && threadIdx.x < kBoundSlots) {   // (keys_kernel has consumed them: bh_bounds.hpp)
        double *w = slots_reset + 4 * threadIdx.x;
        w[0] = INFINITY; w[1] = -INFINITY; w[2] = INFINITY; w[3] = -INFINITY;
    }
    __shared__ uint32_t smu[kWavesPerBlock + 1];
    __shared__ d3 smd[kWavesPerBlock + 1];
    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS);
    uint32_t csum = 0;
    d3 tsum{0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
            uint32_t c = 0;
            if (i + 1 < n) {
                const uint64_t kc = keys[i];
                const int L = shared_levels(kc, keys[i + 1], Dm);
                const int Lp = (i == 0) ? -1 : shared_levels(keys[i - 1], kc, Dm);
                const int hi = (L < Dm - 1) ? L : Dm - 1;
                c = (hi > Lp) ? (uint32_t)(hi - Lp) : 0u;
            }
            cnt[i] = c;
            csum += c;
            if (!EXACT && (i & (kCoarse - 1)) == 0) coarse[i / kCoarse] = keys[i];   // sampled index
            if (!EXACT) {
                const uint32_t b = perm[i];
                const Real2 p = pos[b];
                const Real m = mass[b];
                spos[i] = SReal2{static_cast<SReal>(p.x), static_cast<SReal>(p.y)};
                smass[i] = static_cast<SReal>(m);
                const d3 t{(double)m, (double)m * (double)p.x, (double)m * (double)p.y};
                // fp32 state: the sorted copies ARE the state's values, scan_apply2 forms the same terms from
                // them (12 bytes read instead of 24 written here and 24 read there)
                if (!std::is_same<Real2, SReal2>::value) terms[i] = t;
                tsum += t;
            }
        } else if (!EXACT && i == n) {
            if (!std::is_same<Real2, SReal2>::value) terms[n] = d3{0.0, 0.0, 0.0};
        }
    }
    uint32_t utot;
    (void)block_exclusive_sum(csum, smu, utot);
    if (threadIdx.x == 0) bsum_u32[blockIdx.x] = utot;
    if (!EXACT) {
        d3 dtot;
        (void)block_exclusive_sum(tsum, smd, dtot);
        if (threadIdx.x == 0) bsum_d3[blockIdx.x] = dtot;
    }
}

// block 0: exclusive scan of the u32 tile sums (total -> n_internal); block 1: the d3 tile sums
__global__ __launch_bounds__(kBlock) void scan_top2(uint32_t *__restrict__ bsum_u32,
                                                     d3 *__restrict__ bsum_d3, int nb,
                                                     TreeCounters *ctr)
{
    if (blockIdx.x == 0) {
        __shared__ uint32_t sm[kWavesPerBlock + 1];
        uint32_t carry = 0;
        for (int c0 = 0; c0 < nb; c0 += kBlock) {
            const int i = c0 + threadIdx.x;
            const uint32_t v = (i < nb) ? bsum_u32[i] : 0u;
            uint32_t tot;
            const uint32_t ex = block_exclusive_sum(v, sm, tot);
            if (i < nb) bsum_u32[i] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) ctr->n_internal = carry;
    } else {
        __shared__ d3 sm[kWavesPerBlock + 1];
        d3 carry{0.0, 0.0, 0.0};
        for (int c0 = 0; c0 < nb; c0 += kBlock) {
            const int i = c0 + threadIdx.x;
            const d3 v = (i < nb) ? bsum_d3[i] : d3{0.0, 0.0, 0.0};
            d3 tot;
            const d3 ex = block_exclusive_sum(v, sm, tot);
            if (i < nb) bsum_d3[i] = carry + ex;
            carry += tot;
        }
    }
}

// per tile: cnt -> exclusive offsets (ranks), fp32: terms -> exclusive prefix sums (n+1 entries);
// row by row (256 consecutive elements per block scan), so accesses are coalesced.
// FOLD: bsum_* hold the raw tile TOTALS written by prep_kernel (at most 8 * kBlock of them) and every
// workgroup sums the tiles before it itself -- one launch less for launches of few bodies, where
// scan_top2 is nothing but its ~5 us of launch; workgroup 0 publishes the cell count.
// TSRC: the prefix-sum terms (m, m x, m y) are formed from the sorted fp32 copies instead of read from `terms`
template <bool EXACT, int ITEMS, bool FOLD, bool TSRC = false>
__global__ __launch_bounds__(kBlock) void scan_apply2(uint32_t *__restrict__ cnt,
                                                       const uint32_t *__restrict__ bsum_u32,
                                                       d3 *__restrict__ terms,
                                                       const d3 *__restrict__ bsum_d3, int nbs, int64_t n,
                                                       uint32_t *__restrict__ cell_first,
                                                       int64_t internal_cap, TreeCounters *ctr,
                                                       const float2 *__restrict__ spos = nullptr,
                                                       const float *__restrict__ smass = nullptr)
{
    __shared__ uint32_t smu[kWavesPerBlock + 1];
    __shared__ d3 smd[kWavesPerBlock + 1];
    const int64_t base = (int64_t)blockIdx.x * (kBlock * ITEMS);
    uint32_t ucarry;
    d3 dcarry{0.0, 0.0, 0.0};
    if (FOLD) {
        // thread t holds the tiles t, t + 256, ...: a fixed summation tree, the same for every workgroup
        uint32_t vb = 0, va = 0;
        d3 db{0.0, 0.0, 0.0};
        for (int t = threadIdx.x; t < nbs; t += kBlock) {
            const uint32_t v = bsum_u32[t];
            va += v;
            if (t < (int)blockIdx.x) {
                vb += v;
                if (!EXACT) db += bsum_d3[t];
            }
        }
        uint32_t uall;
        (void)block_exclusive_sum(vb, smu, ucarry);
        (void)block_exclusive_sum(va, smu, uall);
        if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_internal = uall;
        if (!EXACT) (void)block_exclusive_sum(db, smd, dcarry);
    } else {
        ucarry = bsum_u32[blockIdx.x];
        if (!EXACT) dcarry = bsum_d3[blockIdx.x];
    }
#pragma unroll 1
    for (int k = 0; k < ITEMS; ++k) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        const uint32_t v = (i < n) ? cnt[i] : 0u;
        uint32_t utot;
        const uint32_t uex = block_exclusive_sum(v, smu, utot);
        if (i < n) {
            const uint32_t r0 = ucarry + uex;
            cnt[i] = r0;
            // rank -> first body of the cell: lets the node kernel run one thread per CELL (the
            // body that starts a whole chain of nested cells would otherwise process them serially)
            for (uint32_t j = 0; j < v; ++j)
                if ((int64_t)(r0 + j) < internal_cap) cell_first[r0 + j] = (uint32_t)i;
        }
        ucarry += utot;
        if (!EXACT) {
            d3 t{0.0, 0.0, 0.0};
            if (TSRC) {
                if (i < n) {
                    const float2 p = spos[i];
                    const float m = smass[i];
                    t = d3{(double)m, (double)m * (double)p.x, (double)m * (double)p.y};
                }
            } else if (i <= n) {
                t = terms[i];
            }
            d3 dtot;
            const d3 dex = block_exclusive_sum(t, smd, dtot);
            if (i <= n) terms[i] = dcarry + dex;
            dcarry += dtot;
        }
    }
}

// first index j in [lo, hi) with (keys[j] >> sh) >= target   (keys sorted)
__device__ __forceinline__ int64_t lower_bound_prefix(const uint64_t *__restrict__ keys, int64_t lo,
                                                      int64_t hi, int sh, uint64_t target)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> sh) < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// End of the cell with depth-d prefix `pfx` that starts at body i: first j in (i, hi] outside it.
// keys[i+1] is known to be inside.  Galloping then bisection: O(log(cell size)) loads, not
// O(log N) -- most cells hold a handful of bodies.
__device__ __forceinline__ int64_t cell_end(const uint64_t *__restrict__ keys, int64_t i, int64_t hi,
                                            int sh, uint64_t pfx)
{
    int64_t a = i + 1, step = 1, b;
    for (;;) {
        b = a + step;
        if (b >= hi) { b = hi; break; }
        if ((keys[b] >> sh) != pfx) break;
        a = b;
        step <<= 1;
    }
    return lower_bound_prefix(keys, a + 1, b, sh, pfx + 1);
}

// BH_PRECISION_F64 (bh_walk_f64.hpp) -- what the node kernel stores in NodeD::size for that walk: the acceptance criterion size / (sqrt(d2) + 1e-15) < theta
// (project.cu:634, 643) solved for d2.  A cell so small that size / theta < 1e-15 is accepted at every distance (-1 < d2);
// a non-finite size accepts nobody, as every comparison with a NaN fails (+inf < d2 never holds).
__device__ __forceinline__ double f64_walk_threshold(double size, double theta)
{
    const double s = size / theta - 1e-15;
    if (s >= 0.0) return s * s;
    if (s < 0.0) return -1.0;
    return (double)INFINITY;
}

// BH_PRECISION_F64_EXACT (bh_walk_exact.hpp) -- the SAME criterion solved for d2 EXACTLY: the smallest double T with
//     size / (sqrt(T) + 1e-15) < theta        (every operation correctly rounded, as the reference computes it)
// sqrt, + and / are monotone under rounding, so the lanes that accept a node are exactly those with d2 >= T: the walk
// decides with one comparison what the reference decides with a square root and a division, bit for bit the same
// decisions.  NaN when no distance accepts (theta or size not finite; d2 >= NaN never holds -- and a NaN d2 fails
// against every T, as NaN < theta fails in the reference).  Found by galloping from the real-valued solution over
// the doubles' bit patterns (ordered as integers for non-negative values) and bisecting: a handful of evaluations.
__device__ __forceinline__ double exact_walk_threshold(double size, double theta)
{
    auto acc = [&](uint64_t bits) { return size / (sqrt(__longlong_as_double((long long)bits)) + 1e-15) < theta; };
    constexpr uint64_t kInf = 0x7ff0000000000000ull;
    if (!acc(kInf)) return __longlong_as_double(0x7ff8000000000000ll);
    if (acc(0ull)) return 0.0;
    uint64_t lo = 0ull, hi = kInf;                               // acc(lo) false, acc(hi) true
    const double s = size / theta - 1e-15;
    double t0 = (s > 0.0) ? s * s : 0.0;
    uint64_t g = (t0 < (double)INFINITY) ? (uint64_t)__double_as_longlong(t0) : kInf - 1;
    if (g < 1ull) g = 1ull;
    if (acc(g)) {
        hi = g;
        for (uint64_t step = 1;; step <<= 1) {
            if (hi - lo <= step) break;
            const uint64_t c = hi - step;
            if (acc(c)) hi = c; else { lo = c; break; }
        }
    } else {
        lo = g;
        for (uint64_t step = 1;; step <<= 1) {
            if (hi - lo <= step) break;
            const uint64_t c = lo + step;
            if (acc(c)) { hi = c; break; } else lo = c;
        }
    }
    while (hi - lo > 1) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (acc(mid)) hi = mid; else lo = mid;
    }
    return __longlong_as_double((long long)hi);
}

// ---- exact-mode nodes kernel: one thread per subdivided cell writes its four children ---------------
// NodeD/LinkD + self_node / pending (subdivided children per cell) for the bottom-up pass.  Round 1 ran one
// thread per sorted neighbour pair, and the pair that starts a chain of nested cells handled them in turn:
// the thread of sorted body 0 walked its whole chain (20 cells whose range searches span the array) while
// the launch waited -- 2.0 ms at N = 1M.  Now, as in the fp32 kernel, rank r is one thread: its first body
// comes from cell_first[r], its depth from its position in that body's chain.  Thread 0 also writes the
// root when nothing is subdivided (n <= 1 or max_depth == 1).  Same arithmetic per cell as before: the box
// by halving from the root along the key's digits (bitwise the reference's xmin..ymax), depth-cap cells
// folded in body order (project.cu:360-382).
__global__ __launch_bounds__(kBlock) void nodes_exact_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ cell_first,
    const double2 *__restrict__ pos, const double *__restrict__ mass,
    const double *__restrict__ box, int64_t n, int Dm, int64_t internal_cap, NodeD *__restrict__ gd,
    LinkD *__restrict__ ld, int32_t *__restrict__ self_node, int32_t *__restrict__ cell_depth,
    uint32_t *__restrict__ pending, TreeCounters *ctr, double walk_theta, int exact_thr)
{
    // walk_theta > 0: the `size` slot of every node carries the walk's d2 threshold instead -- BH_PRECISION_F64's
    // (f64_walk_threshold) or, with exact_thr, the exact one of the bit-exact walk (the four children of a cell are
    // almost always the same size: one search per cell)
    double memo_size = -1.0, memo_thr = 0.0;
    auto size_slot = [&](double size) {
        if (!(walk_theta > 0.0)) return size;
        if (!exact_thr) return f64_walk_threshold(size, walk_theta);
        if (!(size == memo_size)) { memo_size = size; memo_thr = exact_walk_threshold(size, walk_theta); }
        return memo_thr;
    };
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = ctr->n_internal;

    if (t == 0 && total == 0) {
        // ---- root only: empty root takes the body (project.cu:398-406) or the root itself is a
        //      depth-cap cell (max_depth == 1, project.cu:360-382)
        double m = 0.0, cx = 0.0, cy = 0.0;
        int occ = -1;
        if (n >= 1 && Dm == 0) {
            for (int64_t j = 0; j < n; ++j) {
                const uint32_t b = perm[j];
                const double bm = mass[b], bx = pos[b].x, by = pos[b].y;
                cx = (m * cx + bm * bx) / (m + bm);
                cy = (m * cy + bm * by) / (m + bm);
                m += bm;
            }
            occ = (n == 1) ? -(int)perm[0] - 2 : -1;
        } else if (n == 1) {
            const uint32_t b = perm[0];
            m = mass[b]; cx = pos[b].x; cy = pos[b].y;
            occ = (int)b;
        }
        const double ex = box[1] - box[0], ey = box[3] - box[2];
        gd[0] = NodeD{cx, cy, m, size_slot((ex > ey) ? ex : ey)};
        ld[0] = LinkD{-1, occ};
        return;
    }
    if ((int64_t)total > internal_cap) {
        if (t == 0) ctr->overflow = 1;
        return;
    }
    // (the workgroup's 1,024 nodes are contiguous in memory: they are assembled in LDS and written out with coalesced
    // stores -- field by field from the owning lanes they were 24 stores per cell, each spread over 64 cache lines)
    __shared__ double s_nd[kBlock * 16];                         // per thread: four nodes x {cx, cy, m, size}
    __shared__ int32_t s_ln[kBlock * 8];                         // per thread: four links x {child, occ}
    const int64_t r_block = (int64_t)blockIdx.x * kBlock;
    if (r_block >= (int64_t)total) return;                       // uniform per workgroup
    const bool active = t < (int64_t)total;
    const uint32_t r = (uint32_t)t;
    if (active) {
    const int64_t i = (int64_t)cell_first[r];
    const uint64_t key = keys[i];
    const int Lp = (i == 0) ? -1 : shared_levels(keys[i - 1], key, Dm);
    const int d = Lp + 1 + (int)(r - off[i]);               // this cell's depth

    double x0 = box[0], x1 = box[1], y0 = box[2], y1 = box[3];
    for (int l = 0; l < d; ++l) {
        const int c = (int)((key >> (2 * (Dm - 1 - l))) & 3);
        descend(c, (x0 + x1) / 2, (y0 + y1) / 2, x0, x1, y0, y1);
    }

    const int sh = 2 * (Dm - d);                     // bits below the depth-d prefix
    const uint64_t pfx = (d == 0) ? 0ull : (key >> sh);
    const int shc = sh - 2;
    int64_t e = n;
    int64_t b[5];
    bool settled = false;
    if (d != 0) {
        // Most cells hold a handful of bodies: the next eight keys, read in ONE round of independent loads, settle the end
        // and the three child boundaries of every cell of at most eight bodies (as in nodes_fast_kernel); the searches
        // below are chains of dependent loads, four of them one after the other.
        const uint64_t cell_hi = (pfx + 1) << sh;                // keys are sorted: outside the cell <=> key >= cell_hi
        uint64_t nk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nk[j] = (i + 1 + j < n) ? keys[i + 1 + j] : ~0ull;
        int inside = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) inside += (nk[j] < cell_hi) ? 1 : 0;
        if (inside < 8) {
            e = i + 1 + inside;
            int below1 = 0, below2 = 0, below3 = 0;              // bodies of the cell in children < 1, < 2, < 3
            {
                const uint32_t dg = (uint32_t)(key >> shc) & 3u;
                below1 += dg < 1u; below2 += dg < 2u; below3 += dg < 3u;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t dg = (j < inside) ? (uint32_t)(nk[j] >> shc) & 3u : 3u;
                below1 += dg < 1u; below2 += dg < 2u; below3 += dg < 3u;
            }
            b[1] = i + below1; b[2] = i + below2; b[3] = i + below3;
            settled = true;
        } else {
            e = cell_end(keys, i, n, sh, pfx);
        }
    }
    b[0] = i; b[4] = e;
    if (!settled) {
        // the three boundaries in lock-step: three independent loads per round instead of three searches one after the
        // other (this kernel lasts as long as its longest chain of dependent loads -- the few cells next to the root,
        // whose searches span the whole array: ~60 loads in a row before, ~20 now)
        int64_t lo3[3] = {i, i, i}, hi3[3] = {e, e, e};
        while ((lo3[0] < hi3[0]) | (lo3[1] < hi3[1]) | (lo3[2] < hi3[2])) {
            int64_t mid[3];
            uint64_t km[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mid[c] = (lo3[c] + hi3[c]) >> 1;
                km[c] = (lo3[c] < hi3[c]) ? keys[mid[c]] : 0ull;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (lo3[c] < hi3[c]) {
                    if ((km[c] >> shc) < ((pfx << 2) | (uint64_t)(c + 1))) lo3[c] = mid[c] + 1; else hi3[c] = mid[c];
                }
            }
        }
        b[1] = lo3[0]; b[2] = lo3[1]; b[3] = lo3[2];
    }

    const double mx = (x0 + x1) / 2.0, my = (y0 + y1) / 2.0;
    const int32_t quad = 1 + 4 * (int32_t)r;     // id of child 0

    if (d == 0) {                                  // root record (mass/COM come bottom-up)
        const double ex = x1 - x0, ey = y1 - y0;
        gd[0].size = size_slot((ex > ey) ? ex : ey);
        ld[0] = LinkD{quad, -1};
        self_node[0] = 0;
    }
    cell_depth[r] = d;
    uint32_t n_sub = 0;                            // subdivided children: what the bottom-up pass waits for
    // what the four children need from memory, all of it requested before any of it is used (perm first, then the
    // bodies / the keys and ranks behind it): two rounds of independent loads instead of a chain per child
    int64_t bc4[4], nc4[4];
    uint32_t bi1[4], offc[4];
    uint64_t kprev[4], kcur[4];
    bool sub4[4], one4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bc4[c] = b[c]; nc4[c] = b[c + 1] - b[c];
        sub4[c] = nc4[c] > 1 && d + 1 < Dm;
        one4[c] = nc4[c] == 1 && d + 1 < Dm;
        bi1[c] = one4[c] ? perm[bc4[c]] : 0u;
        offc[c] = sub4[c] ? off[bc4[c]] : 0u;
        kcur[c] = sub4[c] ? keys[bc4[c]] : 0ull;
        kprev[c] = (sub4[c] && bc4[c] > 0) ? keys[bc4[c] - 1] : 0ull;
    }
    double m1[4];
    double2 p1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        m1[c] = one4[c] ? mass[bi1[c]] : 0.0;
        p1[c] = one4[c] ? pos[bi1[c]] : double2{0.0, 0.0};
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double cx0 = (c & 1) ? mx : x0, cx1 = (c & 1) ? x1 : mx;
        const double cy0 = (c & 2) ? my : y0, cy1 = (c & 2) ? y1 : my;
        const double ex = cx1 - cx0, ey = cy1 - cy0;
        const double size = (ex > ey) ? ex : ey;
        const int64_t bc = bc4[c], nc = nc4[c];
        const int32_t node = quad + c;
        double m = 0.0, cx = 0.0, cy = 0.0;
        int32_t child = -1, occ = -1;
        if (nc == 0) {
            // empty leaf: blank child of project.cu:422-428
        } else if (d + 1 == Dm) {
            // depth-cap cell, project.cu:360-382: running mean in body order
            for (int64_t j = bc; j < bc + nc; ++j) {
                const uint32_t bi = perm[j];
                const double bm = mass[bi], bx = pos[bi].x, by = pos[bi].y;
                cx = (m * cx + bm * bx) / (m + bm);
                cy = (m * cy + bm * by) / (m + bm);
                m += bm;
            }
            occ = (nc == 1) ? (-(int32_t)perm[bc] - 2) : -1;
        } else if (nc == 1) {
            // single body in an undivided cell, project.cu:398-406
            m = m1[c]; cx = p1[c].x; cy = p1[c].y;
            occ = (int32_t)bi1[c];
        } else {
            // subdivided cell: its rank follows from its first body and depth
            const int Lpc = (bc == 0) ? -1 : shared_levels(kprev[c], kcur[c], Dm);
            const uint32_t rc = offc[c] + (uint32_t)((d + 1) - (Lpc + 1));
            child = 1 + 4 * (int32_t)rc;
            self_node[rc] = node;
            ++n_sub;
        }
        double *sn = s_nd + threadIdx.x * 16 + 4 * c;
        sn[0] = cx; sn[1] = cy; sn[2] = m; sn[3] = size_slot(size);
        s_ln[threadIdx.x * 8 + 2 * c] = child; s_ln[threadIdx.x * 8 + 2 * c + 1] = occ;
    }
    pending[r] = n_sub;
    }   // active
    __syncthreads();
    {
        const int64_t left = (int64_t)total - r_block;
        const int cells = (left < kBlock) ? (int)left : kBlock;
        double2 *dn = reinterpret_cast<double2 *>(gd + 1 + 4 * r_block);           // 32-byte aligned: node 1 + 4 r
        const double2 *sn = reinterpret_cast<const double2 *>(s_nd);
        for (int k = threadIdx.x; k < cells * 8; k += kBlock) dn[k] = sn[k];
        int2 *dl = reinterpret_cast<int2 *>(ld + 1 + 4 * r_block);
        const int2 *sl = reinterpret_cast<const int2 *>(s_ln);
        for (int k = threadIdx.x; k < cells * 4; k += kBlock) dl[k] = sl[k];
    }
}

// ---- fp32 nodes kernel ------------------------------------------------------------------------------
// Same ownership scheme as nodes_kernel, specialised for the throughput path:
//   * a window of the sorted keys (the workgroup's 256 bodies, one predecessor and a 768-key halo)
//     is cached in LDS: the range searches of almost every cell stay inside it (a cell with more
//     than ~800 bodies falls back to global loads -- there are only a few thousand such cells);
//   * no fp64 box tracking: fp32 only needs the MAC threshold (size/theta)^2, and the cell size at
//     depth d is root_size * 2^-d to far better than fp32 resolution;
//   * single bodies come from the sorted copies (neighbouring addresses), not through perm;
//   * one thread per CELL and coalesced LDS-staged stores (see the kernel); LDS (36 KB per
//     workgroup: key window + staging) bounds residency at 4 workgroups per CU.
// A depth-cap cell holding more bodies than this is aggregated as the reference does even with
// reference_compat off: a direct sum over it would cost O(bodies) per visiting body, and a
// degenerate input (one body at infinity collapses every key) would turn one step into O(N^2).
constexpr int kMaxBucket = 1024;
constexpr int kKeyHalo = 768;
constexpr int kKeyWin = kBlock + 1 + kKeyHalo;

// FULL_AUX: the {first body, count} record of EVERY node is written (bh_export_tree asks for it and re-runs
// this kernel); the step itself only needs those of bucket leaves -- 32 of the 112 bytes a cell writes, and
// 8 of the 28 KB of LDS staging.  (Asking for 7 resident workgroups instead of 5 made it SLOWER, 55 -> 58 us at N = 1M:
// the 72-register budget costs more than the extra residency returns.)
template <bool COMPAT, bool FULL_AUX = false>
__global__ __launch_bounds__(kBlock, 5) void nodes_fast_kernel(
    const uint64_t *__restrict__ keys, const uint64_t *__restrict__ coarse,
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ cell_first,
    const float2 *__restrict__ spos,
    const float *__restrict__ smass, const double *__restrict__ box, const d3 *__restrict__ psum,
    int64_t n64, int Dm, double theta, int64_t internal_cap, QuadF *__restrict__ qf,
    NodeAux *__restrict__ aux, TreeCounters *ctr)
{
    // ONE THREAD PER SUBDIVIDED CELL (rank r): its first body comes from cell_first[r], its depth
    // from its position in that body's chain.  32-bit indices throughout (bh_create caps n < 2^31).
    // LDS is used twice: first as the key window of the searches, then -- after a barrier -- as the
    // staging area of the quads (28 KB instead of 36 KB per workgroup: 5 resident workgroups per CU)
    constexpr int kStageBytes = kBlock * (FULL_AUX ? 28 : 20) * 4, kWinBytes = kKeyWin * 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kStageBytes > kWinBytes ? kStageBytes : kWinBytes];
    uint64_t *wkeys = reinterpret_cast<uint64_t *>(lds_raw);
    __shared__ int32_t s_wlo;
    __shared__ double s_thr0;                                   // (size/theta)^2 at depth 0: one fp64 division per workgroup
    // The 256 quads of a workgroup are contiguous in memory (quad = rank + 1): they are assembled
    // in LDS and written out with coalesced 16-byte stores.  Storing field by field from the
    // owning lanes costs 24 scattered 4-byte stores per cell -- 18 M separate L2 write requests at
    // N = 1M, which was most of this kernel's time.
    int32_t *stage_q = reinterpret_cast<int32_t *>(lds_raw);             // kBlock * 20 dwords
    int32_t *stage_a = stage_q + kBlock * 20;                            // kBlock * 8 dwords
    const int32_t n = (int32_t)n64;
    const uint32_t total = ctr->n_internal;
    float *qw = reinterpret_cast<float *>(qf);                  // 20 dwords per quad
    int32_t *qi = reinterpret_cast<int32_t *>(qf);
    auto put = [&](int32_t quad, int slot, float cx, float cy, float m, float thr, int32_t child) {   // direct
        const int64_t o = (int64_t)quad * 20;
        qw[o + 2 * slot] = cx; qw[o + 2 * slot + 1] = cy; qw[o + 8 + slot] = m; qw[o + 12 + slot] = thr;
        qi[o + 16 + slot] = child;
    };
    auto put_lds = [&](int slot, float cx, float cy, float m, float thr, int32_t child) {             // staged
        int32_t *o = stage_q + threadIdx.x * 20;
        o[2 * slot] = __float_as_int(cx); o[2 * slot + 1] = __float_as_int(cy);
        o[8 + slot] = __float_as_int(m); o[12 + slot] = __float_as_int(thr); o[16 + slot] = child;
    };

    if (total == 0) {
        // root only (n <= 1, or max_depth == 1)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int k = 0; k < 4; ++k) { put(0, k, 0.f, 0.f, 0.f, -1.f, -1); aux[k] = NodeAux{0, 0}; }
            if (n >= 1) {
                const d3 t = psum[n];
                const bool one = (n == 1);
                const float cx = one ? spos[0].x : (float)(t.b / t.a), cy = one ? spos[0].y : (float)(t.c / t.a);
                const float m = (t.a > 1e-15) ? (one ? smass[0] : (float)t.a) : 0.f;
                const bool bucket = (n > 1) && (n <= kMaxBucket) && !COMPAT;
                put(0, 0, cx, cy, m, bucket ? INFINITY : 0.f, bucket ? -2 : -1);
                aux[0] = NodeAux{0, (int32_t)n};
            }
        }
        return;
    }
    if ((int64_t)total > internal_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctr->overflow = 1;
        return;
    }
    const uint32_t r_block = blockIdx.x * kBlock;
    if (r_block >= total) return;                               // uniform per workgroup

    // this thread's cell: issue its two loads now, they do not depend on the key window below
    const uint32_t r = r_block + threadIdx.x;
    const int32_t i_pre = (r < total) ? (int32_t)cell_first[r] : 0;
    const uint32_t off_pre = (r < total) ? off[i_pre] : 0u;

    // key window starting one body before the first cell of this workgroup
    if (threadIdx.x == 0) {
        s_wlo = i_pre - 1;
        const double ex0 = box[1] - box[0], ey0 = box[3] - box[2];
        const double q0 = ((ex0 > ey0) ? ex0 : ey0) / theta;   // size/theta at depth 0
        s_thr0 = q0 * q0;
    }
    __syncthreads();
    const int32_t wlo = s_wlo;
    for (int k = threadIdx.x; k < kKeyWin; k += kBlock) {
        const int32_t j = wlo + k;
        wkeys[k] = (j >= 0 && j < n) ? keys[j] : 0ull;
    }
    __syncthreads();
    // (a select between the two pointers compiles to ONE flat load from a generic address -- through the vector memory
    // path even when the address is in LDS, with 64-bit address arithmetic; almost every read is inside the window, so
    // the window is read with a plain LDS instruction and the rare lane outside it is patched up under a uniform test)
    auto K = [&](int32_t j) -> uint64_t {
        const int32_t k = j - wlo;
        const bool in = (uint32_t)k < (uint32_t)kKeyWin;
        uint64_t v = wkeys[in ? k : 0];
        if (__ballot(!in) != 0ull) {
            asm volatile("" : "+v"(v));                         // (keeps the compiler from merging the two loads into a flat one again)
            if (!in) v = keys[j];
        }
        return v;
    };

    // first j in [lo, hi) with (key_j >> sh) >= target, for NT ascending targets at once.  The kernel's time
    // is the longest chain of DEPENDENT loads of any thread -- the few large cells near the root search
    // ranges of 10^5 keys through L2 -- so every round probes the three quartile points of every search
    // together (3 * NT independent loads in flight, log4 instead of log2 rounds; round 1's one-probe
    // bisection of the end and of the three child boundaries one after the other was a chain of ~80 loads).
    // Large ranges are narrowed on the sampled index first: after that the range is < 2*kCoarse wide.
    auto search = [&](auto nt_tag, auto keyfn, int32_t lo0, int32_t hi0, int sh, const uint64_t *target, int32_t *res) {
        constexpr int NT = decltype(nt_tag)::value;
        int32_t lo[NT], hi[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) { lo[c] = lo0; hi[c] = hi0; }
        if (hi0 - lo0 > 4 * kCoarse) {
            // first sample index s in [cl, ch) with (coarse[s] >> sh) >= target
            int32_t cl[NT], ch[NT];
            bool more = true;
#pragma unroll
            for (int c = 0; c < NT; ++c) { cl[c] = (lo0 + kCoarse - 1) / kCoarse; ch[c] = (hi0 - 1) / kCoarse + 1; }
            while (more) {
                uint64_t k1[NT], k2[NT], k3[NT];
                int32_t p1[NT], p2[NT], p3[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const int32_t len = ch[c] - cl[c];
                    p1[c] = cl[c] + len / 4; p2[c] = cl[c] + len / 2; p3[c] = cl[c] + (int32_t)((3 * (int64_t)len) / 4);
                    const bool on = len > 0;
                    k1[c] = on ? coarse[p1[c]] : 0ull; k2[c] = on ? coarse[p2[c]] : 0ull; k3[c] = on ? coarse[p3[c]] : 0ull;
                }
                more = false;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    if (ch[c] - cl[c] > 0) {
                        if ((k1[c] >> sh) >= target[c]) ch[c] = p1[c];
                        else if ((k2[c] >> sh) >= target[c]) { cl[c] = p1[c] + 1; ch[c] = p2[c]; }
                        else if ((k3[c] >> sh) >= target[c]) { cl[c] = p2[c] + 1; ch[c] = p3[c]; }
                        else cl[c] = p3[c] + 1;
                    }
                    more = more || (ch[c] - cl[c] > 0);
                }
            }
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int32_t up = (cl[c] * kCoarse < hi0) ? cl[c] * kCoarse : hi0;             // key[up] >= target (or up == hi)
                const int32_t dn = ((cl[c] - 1) * kCoarse > lo0) ? (cl[c] - 1) * kCoarse : lo0; // key[dn] < target (or dn == lo)
                lo[c] = dn; hi[c] = up;
            }
        }
        // (the fine part, mostly in the LDS window: plain bisection -- the kernel is vector-issue bound there, and
        // three probes per round are four times the instructions of one for half the rounds)
        bool more = true;
        while (more) {
            uint64_t k2[NT];
            int32_t p2[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                p2[c] = (int32_t)(((uint32_t)lo[c] + (uint32_t)hi[c]) >> 1);
                k2[c] = (hi[c] > lo[c]) ? keyfn(p2[c]) : 0ull;
            }
            more = false;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                if (hi[c] > lo[c]) {
                    if ((k2[c] >> sh) >= target[c]) hi[c] = p2[c]; else lo[c] = p2[c] + 1;
                }
                more = more || (hi[c] > lo[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) res[c] = lo[c];
    };
    auto lower_bound = [&](int32_t lo, int32_t hi, int sh, uint64_t target) -> int32_t {
        int32_t r;
        search(std::integral_constant<int, 1>{}, K, lo, hi, sh, &target, &r);
        return r;
    };

    float res_cx[4], res_cy[4], res_m[4], res_thr[4];
    int32_t res_child[4], res_bc[4], res_nc[4];
    if (r < total) {
    const int32_t i = i_pre;
    const uint64_t key = K(i);
    const int Lp = (i == 0) ? -1 : shared_levels(K(i - 1), key, Dm);
    const int d = Lp + 1 + (int)(r - off_pre);                  // this cell's depth

    const double thr0 = s_thr0;

    const int sh = 2 * (Dm - d);
    const uint64_t pfx = (d == 0) ? 0ull : (key >> sh);
    int32_t e;
    // Most cells hold a handful of bodies: the next eight keys, read in one round, settle the end AND the three
    // child boundaries of every cell of at most eight bodies without a loop (the searches below are divergent
    // per-lane loops of dependent LDS reads; a wave is as slow as its largest cell, but it no longer also pays
    // three or four rounds for each of its small ones).
    int32_t bnd[5];
    bool have_bnd = false;
    const int shc = sh - 2;
    if (d == 0) e = n;
    else {
        const uint64_t cell_hi = (pfx + 1) << sh;                // keys are sorted: outside the cell <=> key >= cell_hi
        uint64_t nk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nk[j] = 0ull;
        // (wave-uniform test: when every lane's eight keys lie inside the window -- nearly always -- they are eight plain LDS
        // reads at constant offsets from one address, not eight guarded reads with a branch and a fallback each)
        const int32_t k1 = i + 1 - wlo;
        if (__ballot(!(k1 >= 0 && k1 + 8 <= kKeyWin && i + 8 < n)) == 0ull) {
#pragma unroll
            for (int j = 0; j < 8; ++j) nk[j] = wkeys[k1 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) nk[j] = (i + 1 + j < n) ? K(i + 1 + j) : ~0ull;
        }
        int inside = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) inside += (nk[j] < cell_hi) ? 1 : 0;
        if (inside < 8) {
            e = i + 1 + inside;
            int below1 = 0, below2 = 0, below3 = 0;              // bodies of the cell in children < 1, < 2, < 3
            {
                const uint32_t dg = (uint32_t)(key >> shc) & 3u;
                below1 += dg < 1u; below2 += dg < 2u; below3 += dg < 3u;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t dg = (j < inside) ? (uint32_t)(nk[j] >> shc) & 3u : 3u;
                below1 += dg < 1u; below2 += dg < 2u; below3 += dg < 3u;
            }
            bnd[1] = i + below1; bnd[2] = i + below2; bnd[3] = i + below3;
            have_bnd = true;
        }
    }
    if (d != 0 && !have_bnd) {                     // nine bodies or more: galloping, then bisection
        int32_t a = i + 8, step = 8, b;
        for (;;) {
            b = (step > n - a) ? n : a + step;                   // no 32-bit overflow
            if (b >= n) { b = n; break; }
            if (step > 2 * kCoarse) break;                       // large cell: let the index finish
            if ((K(b) >> sh) != pfx) break;
            a = b; step <<= 1;
        }
        // keys in (a, b) may be inside or outside; key[a] is inside.  (If the gallop stopped
        // because the cell is large, b is not yet known to be outside: search up to n.)
        if (step > 2 * kCoarse && b < n && (K(b) >> sh) == pfx) { a = b; b = n; }
        e = lower_bound(a + 1, b, sh, pfx + 1);
    }
    const int32_t quad = (int32_t)r + 1;

    if (d == 0) {                                                // root record in quad 0
        for (int k = 1; k < 4; ++k) { put(0, k, 0.f, 0.f, 0.f, -1.f, -1); aux[k] = NodeAux{0, 0}; }
        const d3 t = psum[n];
        if (t.a > 1e-15) put(0, 0, (float)(t.b / t.a), (float)(t.c / t.a), (float)t.a, (float)thr0, quad);
        else put(0, 0, 0.f, 0.f, 0.f, -1.f, -1);
        aux[0] = NodeAux{0, n};
    }
    // (size/theta)^2 of the children (depth d+1): exact power-of-four scaling
    const float thr_child = (float)ldexp(thr0, -2 * (d + 1));

    // child boundaries first, then every global load of the four children at once: the loads are
    // independent of each other, and this kernel's time is the length of its per-thread chain of
    // dependent memory accesses (3 rounds of resident workgroups x one chain at N = 1M)
    bnd[0] = i; bnd[4] = e;
    // every key this cell still needs lies in [i - 1, e): when that range is inside the window for all the wave's cells
    // -- nearly always -- the reads below are plain LDS reads (KF), without the guard and the fallback of K
    auto KF = [&](int32_t j) -> uint64_t { return wkeys[j - wlo]; };
    const bool cells_in_window = __ballot(!(i - 1 >= wlo && e - wlo <= kKeyWin)) == 0ull;   // uniform
    if (!have_bnd) {
        const uint64_t tg[3] = {(pfx << 2) | 1ull, (pfx << 2) | 2ull, (pfx << 2) | 3ull};
        if (cells_in_window) {
            // branch-free bisection of the three child boundaries inside the window: every round reads one key per
            // boundary whether or not that search is still open (a closed one re-reads its end point: at worst entry
            // kKeyWin, inside the LDS block) and moves its bounds by selects -- the guarded form above compiles to
            // an EXEC-mask branch around every read and every update
            int32_t lo3[3] = {i, i, i}, hi3[3] = {e, e, e};
            const uint64_t t3[3] = {tg[0] << shc, tg[1] << shc, tg[2] << shc};
            while ((hi3[0] > lo3[0]) | (hi3[1] > lo3[1]) | (hi3[2] > lo3[2])) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int32_t pm = (int32_t)(((uint32_t)lo3[c] + (uint32_t)hi3[c]) >> 1);
                    const uint64_t km = wkeys[pm - wlo];
                    const bool open = hi3[c] > lo3[c], ge = km >= t3[c];
                    hi3[c] = (open && ge) ? pm : hi3[c];
                    lo3[c] = (open && !ge) ? pm + 1 : lo3[c];
                }
            }
            bnd[1] = lo3[0]; bnd[2] = lo3[1]; bnd[3] = lo3[2];
        }
        else search(std::integral_constant<int, 3>{}, K, i, e, shc, tg, &bnd[1]);
    }
    d3 ps[5];
    uint32_t offc[4];
    uint64_t kprev[4], kcur[4];
    float2 p1[4];
    float m1[4];
    // (only what the child's kind needs: most cells sit at the bottom of the tree, their children are single
    // bodies or empty, and a 24-byte prefix sum per boundary is the kernel's largest gather)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        bool left = c > 0 && bnd[c] - bnd[c - 1] > 1, right = c < 4 && bnd[c + 1] - bnd[c] > 1;
        ps[c] = (left || right) ? psum[bnd[c]] : d3{1.0, 1.0, 1.0};
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int32_t bc = bnd[c];
        const int32_t ncc = bnd[c + 1] - bc;
        const bool sub = ncc > 1 && d + 1 < Dm;                  // a subdivided child: its rank comes from off and its keys
        offc[c] = sub ? off[bc] : 0u;
        if (cells_in_window) {
            kcur[c] = sub ? KF(bc) : 0ull;
            kprev[c] = (sub && bc > 0) ? KF(bc - 1) : 0ull;
        } else {
            kcur[c] = sub ? K(bc) : 0ull;
            kprev[c] = (sub && bc > 0) ? K(bc - 1) : 0ull;
        }
        p1[c] = (ncc == 1) ? spos[bc] : float2{0.f, 0.f};
        m1[c] = (ncc == 1) ? smass[bc] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int32_t bc = bnd[c], bn = bnd[c + 1];
        const int32_t nc = bn - bc;
        float cx = 0.f, cy = 0.f, m = 0.f, thr = 0.f;            // leaves: thr = 0 (see the walk)
        int32_t child = -1;
        if (nc == 1) {
            cx = p1[c].x; cy = p1[c].y; m = m1[c];
        } else if (nc > 1) {
            const d3 lo_s = ps[c], hi_s = ps[c + 1];
            const double mm = hi_s.a - lo_s.a;
            // one reciprocal (v_rcp_f64 + two Newton steps, ~1 ulp of fp64) instead of two fp64 divisions of ~35
            // instructions each -- eight per cell were half of this kernel's vector instructions; the values
            // are rounded to fp32 right here
            double inv = __builtin_amdgcn_rcp(mm);
            inv = fma(fma(-mm, inv, 1.0), inv, inv);
            inv = fma(fma(-mm, inv, 1.0), inv, inv);
            m = (float)mm; cx = (float)((hi_s.b - lo_s.b) * inv); cy = (float)((hi_s.c - lo_s.c) * inv);
            if (d + 1 == Dm) {                                   // depth-cap cell
                if (!COMPAT && nc <= kMaxBucket) { thr = INFINITY; child = -(4 * quad + c) - 2; }
            } else {                                             // subdivided cell
                const int Lpc = (bc == 0) ? -1 : shared_levels(kprev[c], kcur[c], Dm);
                child = (int32_t)(offc[c] + (uint32_t)((d + 1) - (Lpc + 1))) + 1;
                thr = thr_child;
            }
        }
        if (!(m > 1e-15f)) { cx = 0.f; cy = 0.f; m = 0.f; thr = 0.f; child = -1; }   // project.cu:617
        res_cx[c] = cx; res_cy[c] = cy; res_m[c] = m; res_thr[c] = thr; res_child[c] = child;
        res_bc[c] = bc; res_nc[c] = nc;
    }
    }   // r < total

    __syncthreads();                                            // the key window is dead from here on
    if (r < total) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            put_lds(c, res_cx[c], res_cy[c], res_m[c], res_thr[c], res_child[c]);
            if (FULL_AUX) {
                stage_a[threadIdx.x * 8 + 2 * c] = res_bc[c];
                stage_a[threadIdx.x * 8 + 2 * c + 1] = res_nc[c];
            } else if (res_child[c] <= -2) {                     // a bucket leaf: the walk reads its body range
                aux[4 * ((int64_t)r + 1) + c] = NodeAux{res_bc[c], res_nc[c]};
            }
        }
    }
    // coalesced write-out of this workgroup's quads [r_block + 1, r_block + 1 + cells)
    __syncthreads();
    const uint32_t cells = (total - r_block < (uint32_t)kBlock) ? total - r_block : (uint32_t)kBlock;
    {
        typedef int32_t v4 __attribute__((ext_vector_type(4)));
        v4 *dq = reinterpret_cast<v4 *>(qi + ((int64_t)r_block + 1) * 20);
        const v4 *sq = reinterpret_cast<const v4 *>(stage_q);
        for (uint32_t k = threadIdx.x; k < cells * 5; k += kBlock) dq[k] = sq[k];
        if (FULL_AUX) {
            v4 *da = reinterpret_cast<v4 *>(aux + 4 * ((int64_t)r_block + 1));
            const v4 *sa = reinterpret_cast<const v4 *>(stage_a);
            for (uint32_t k = threadIdx.x; k < cells * 2; k += kBlock) da[k] = sa[k];
        }
    }
}

// ---- exact bottom-up pass: ComputeMass, project.cu:473-502, ONE launch (small trees) -------------------
// The reference recurses (children before parents).  Round 1 ran one launch per depth (9 at the
// reference's cap, 31 at cap 32), each ~5 us of launch floor -- most of the build at N = 1,024.  Here every thread starts at one subdivided
// cell; a cell whose four children are all leaves (pending == 0 after the node kernel) is summed at once,
// and the thread then climbs: it decrements its parent's count of unfinished subdivided children and
// whoever brings it to zero sums the parent.  Each cell is summed by exactly ONE thread, from the stored
// values of its children in child order 0..3 starting from 0.0 -- the reference's order (project.cu:483-495)
// -- so the result does not depend on which thread arrives last: bit-identical to the per-level pass.
// Inter-workgroup visibility (the per-XCD L2s are not coherent): the decrement is an agent-scope
// acquire-release RMW, the children's values are read with agent-scope loads after it.
__global__ __launch_bounds__(kBlock) void com_up_kernel(NodeD *__restrict__ gd, const LinkD *__restrict__ ld,
                                                         const int32_t *__restrict__ self_node,
                                                         uint32_t *__restrict__ pending,
                                                         const TreeCounters *__restrict__ ctr, int64_t internal_cap)
{
    const int64_t r0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = ctr->n_internal;
    if (r0 >= (int64_t)total || (int64_t)total > internal_cap) return;
    if (pending[r0] != 0) return;                     // a descendant's thread will come by
    uint32_t r = (uint32_t)r0;
    for (;;) {
        const int32_t node = self_node[r];
        const int32_t quad = 1 + 4 * (int32_t)r;
        double tot = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double cm = __hip_atomic_load(&gd[quad + c].m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double cx = __hip_atomic_load(&gd[quad + c].cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double cy = __hip_atomic_load(&gd[quad + c].cy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot += cm;
            sx += cm * cx;
            sy += cm * cy;
        }
        if (tot > 0.0) { sx /= tot; sy /= tot; }
        __hip_atomic_store(&gd[node].cx, sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&gd[node].cy, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&gd[node].m, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == 0) break;                            // the root
        const uint32_t parent = (uint32_t)(node - 1) >> 2;      // node = 1 + 4 * parent rank + child index
        const uint32_t left = __hip_atomic_fetch_sub(&pending[parent], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (left != 1u) break;                        // other subdivided children are still on their way
        r = parent;
    }
}

// The same pass, one launch per depth (deepest first): what large trees use.  com_up_kernel's agent-scope
// acquire-release RMW costs a write-back / invalidate of the non-coherent per-XCD L2s per cell: measured
// 15 us at N = 1,024, 50 us at 65,536 (= nine launches of this kernel), 1.75 ms at N = 1M against ~0.15 ms for
// twenty launches of this one.  Same sums in the same order either way.  (Round 4 built the pass in 2-3 launches -- a block of
// consecutive pre-order ranks holds whole subtrees, the cells it cannot finish are one chain of ancestors, lists of those chains
// are finished over 64 x larger ranges per launch -- once as a separate kernel and once started inside nodes_exact_kernel:
// 107 and 136 us against these 112, each bitwise equal; and two depths per launch, the shallower one recomputing its
// subdivided children from the grandchildren: build 0.277 ms against 0.257.  profiles/r04_f64/com_ab.txt says where the time went.
// And for the smallest trees ONE workgroup, a barrier per depth: 0.120 ms per step at N = 1,024 against the climb's 0.121, 0.199 against
// 0.155 at N = 2,048 -- a level is a round trip either way.)
__global__ __launch_bounds__(kBlock) void com_level_kernel(NodeD *__restrict__ gd, const int32_t *__restrict__ self_node,
                                                            const int32_t *__restrict__ cell_depth,
                                                            const TreeCounters *__restrict__ ctr, int64_t internal_cap,
                                                            int depth)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = ctr->n_internal;
    if (r >= (int64_t)total || (int64_t)total > internal_cap) return;
    if (cell_depth[r] != depth) return;
    const int32_t node = self_node[r];
    const int32_t quad = 1 + 4 * (int32_t)r;
    double tot = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const NodeD ch = gd[quad + c];
        tot += ch.m;
        sx += ch.m * ch.cx;
        sy += ch.m * ch.cy;
    }
    if (tot > 0.0) { sx /= tot; sy /= tot; }
    gd[node].cx = sx; gd[node].cy = sy; gd[node].m = tot;
}

}  // namespace bh
