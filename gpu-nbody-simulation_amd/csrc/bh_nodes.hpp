// bh_nodes.hpp -- device node records shared by the tree build and the two walk kernels.
#pragma once

#include <stdint.h>

namespace bh {

// ---- node records ------------------------------------------------------------------------------
// exact mode: 32 B geometry + 8 B links, fp64 (what the walk reads per visit: 40 B)
struct NodeD { double cx, cy, m, size; };
struct LinkD { int32_t child; int32_t occ; };   // child: first of 4 contiguous children or -1
                                                // occ  : reference PARTICLE_INDEX (caller order)
// fp32 mode: one 32-byte record; a sibling quad is one 128-byte line
struct alignas(32) NodeF {
    float cx, cy, m, thr;      // thr = (size/theta)^2; -1 for leaves (always accepted)
    int32_t child;             // first of 4 contiguous children, or -1
    int32_t occ;               // SORTED index of the single occupant, or -1
    int32_t first, count;      // sorted body range of the cell
};

struct TreeCounters {
    uint32_t n_internal;       // I  (written by scan_top)
    uint32_t overflow;         // 1 if I > internal capacity
    uint32_t pad[2];
    unsigned long long visits, interactions;
};

}  // namespace bh
