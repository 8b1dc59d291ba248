// bh_nodes.hpp -- device node records shared by the tree build and the two walk kernels.
#pragma once

#include <stdint.h>

namespace bh {

// ---- node records ------------------------------------------------------------------------------
// exact mode: 32 B geometry + 8 B links, fp64 (what the walk reads per visit: 40 B)
struct NodeD { double cx, cy, m, size; };
struct LinkD { int32_t child; int32_t occ; };   // child: first of 4 contiguous children or -1
                                                // occ  : reference PARTICLE_INDEX (caller order)
// fp32 mode: one 32-byte record; a sibling quad is one 128-byte line.
//   kind of node        count   child            thr                 walk behaviour
//   empty cell          0       -1               -1                  skipped (integer test)
//   single body         1       -1               -1                  always accepted (d2 > -1)
//   subdivided cell     >= 2    first child >=0  (size/theta)^2      MAC per body, opened by the rest
//   bucket (depth-cap   >= 2    -(node id) - 2   +inf                never accepted, "opened" by every
//   cell, compat off)                                                 live body -> summed body by body
//   aggregate (depth-   >= 2    -1               -1                  always accepted as one point mass
//   cap cell, compat on)                                             (project.cu:360-382)
// A non-empty cell whose mass is <= 1e-15 is stored as empty: the reference skips it and its whole
// subtree (project.cu:617).
struct alignas(32) NodeF {
    float cx, cy, m, thr;
    int32_t child;
    int32_t first, count;      // sorted body range of the cell
    int32_t pad;
};

struct TreeCounters {
    uint32_t n_internal;       // I  (written by scan_top)
    uint32_t overflow;         // 1 if I > internal capacity
    uint32_t pad[2];
    unsigned long long visits, interactions;
    unsigned long long wave_nodes;   // nodes evaluated by wavefronts (one count per wave per node)
};

}  // namespace bh
