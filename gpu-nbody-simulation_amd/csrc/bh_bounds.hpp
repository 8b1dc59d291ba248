// bh_bounds.hpp -- workgroup min/max of positions, shared by the tree build and both walk kernels.
#pragma once

#include "bh_prims.hpp"

namespace bh {

// Block-level min/max of one position per thread -> partial[block]; called from the walk kernels'
// epilogue so that the NEXT step's root box needs no pass over the bodies.  `partial` points at
// this workgroup's four doubles.  All threads of the block must call it (it synchronises).
// `slots` (may be null): kBoundSlots running {xlo, xhi, ylo, yhi} records that the workgroups of a launch fold their
// bounds into with four atomics each (slot = workgroup index mod kBoundSlots).  The next build's keys_kernel reduces
// those 64 records in every one of its workgroups -- 2 KB from L2 -- instead of waiting for a one-workgroup launch
// that reduces thousands of partials (bounds_final: a 4 us dependent launch feeding 64 bytes); prep_kernel, two launches
// on, puts the slots back to +-inf (no reader counter: it was 4,100 atomics on one word).  Records that are still all
// +-inf when keys_kernel reads them mean that the walk returned at once on an overflowed tree: the box in memory stays.
// min / max are exact and order-free: the box is the same, bit for bit.
constexpr int kBoundSlots = 64;
__device__ __forceinline__ void bounds_to_slot(double xlo, double xhi, double ylo, double yhi, double *slots, uint32_t group)
{
    double *s = slots + 4 * (group & (uint32_t)(kBoundSlots - 1));
    atomicMin(s + 0, xlo); atomicMax(s + 1, xhi); atomicMin(s + 2, ylo); atomicMax(s + 3, yhi);
}

__device__ __forceinline__ void block_bounds_to_partial(bool valid, double x, double y,
                                                        double *__restrict__ partial, double *slots = nullptr)
{
    __shared__ double sm[4][kWavesPerBlock];
    double xlo = valid ? x : INFINITY, xhi = valid ? x : -INFINITY;
    double ylo = valid ? y : INFINITY, yhi = valid ? y : -INFINITY;
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (lane_id() == 0) { sm[0][wave_id()] = xlo; sm[1][wave_id()] = xhi; sm[2][wave_id()] = ylo; sm[3][wave_id()] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            xlo = (sm[0][w] < xlo) ? sm[0][w] : xlo;  xhi = (xhi < sm[1][w]) ? sm[1][w] : xhi;
            ylo = (sm[2][w] < ylo) ? sm[2][w] : ylo;  yhi = (yhi < sm[3][w]) ? sm[3][w] : yhi;
        }
        partial[0] = xlo; partial[1] = xhi; partial[2] = ylo; partial[3] = yhi;
        if (slots) bounds_to_slot(xlo, xhi, ylo, yhi, slots, blockIdx.x);
    }
}

// The constants of keys_kernel's one-multiply cell lookup (bh_tree.hpp, key_of_fast), written behind the root
// box by whoever sets it: box[4], box[5] = 2^Dm / width per axis; box[6], box[7] = the distance from a grid
// line, in cells, beyond which the lookup is provably the bisection's result (2.0 = never: degenerate or
// non-finite box, or a box so far from the origin that its grid lines are not resolved to a quarter cell).
__device__ __forceinline__ void write_key_consts_axis(double *box, int a, int Dm)
{
    const double side = (double)(1u << Dm);
    const double eps = (double)(Dm + 8) * 1.1102230246251565e-16 * side;
    const double lo = box[2 * a], hi = box[2 * a + 1];
    const double w = hi - lo, big = fmax(fabs(lo), fabs(hi));
    const double scale = side / w, margin = eps * (big / w);
    const bool ok = isfinite(scale) && scale > 0.0 && margin < 0.25;      // (a NaN margin fails the compare)
    box[4 + a] = ok ? scale : 0.0;
    box[6 + a] = ok ? margin : 2.0;
}
__device__ __forceinline__ void write_key_consts(double *__restrict__ box, int Dm)
{
    write_key_consts_axis(box, 0, Dm);
    write_key_consts_axis(box, 1, Dm);
}

}  // namespace bh
