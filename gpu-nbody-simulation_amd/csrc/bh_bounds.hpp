// bh_bounds.hpp -- workgroup min/max of positions, shared by the tree build and both walk kernels.
#pragma once

#include "bh_prims.hpp"

namespace bh {

// Block-level min/max of one position per thread -> partial[block]; called from the walk kernels'
// epilogue so that the NEXT step's root box needs no pass over the bodies.  `partial` points at
// this workgroup's four doubles.  All threads of the block must call it (it synchronises).
__device__ __forceinline__ void block_bounds_to_partial(bool valid, double x, double y,
                                                        double *__restrict__ partial)
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
    }
}

}  // namespace bh
