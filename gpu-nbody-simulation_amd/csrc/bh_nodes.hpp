// bh_nodes.hpp -- device node records shared by the tree build and the two walk kernels.
#pragma once

#include <stdint.h>

namespace bh {

// ---- exact mode: 32 B geometry + 8 B links, fp64 (what the walk reads per visit: 40 B) ----------
// node ids: root = 0; the four children of the subdivided cell with pre-order rank r are 1+4r..1+4r+3
struct NodeD { double cx, cy, m, size; };
struct LinkD { int32_t child; int32_t occ; };   // child: first of 4 contiguous children or -1
                                                // occ  : reference PARTICLE_INDEX (caller order)

// ---- fp32 mode: one 80-byte structure-of-arrays record per SIBLING QUAD = 20 B per node ----------
// quad 0 holds the root in slot 0 (slots 1..3 empty); quad r+1 holds the four children of the
// subdivided cell with pre-order rank r.  node id = 4*quad + slot.  The walk reads a quad with one
// s_load_dwordx16 + one s_load_dwordx4.
//   kind of node        m      child             thr                 walk behaviour (accept iff d2 > thr)
//   empty cell          0      -1                any                 skipped (integer test on m)
//   single body         > 0    -1                0                   accepted unless d2 == 0 (= itself)
//   subdivided cell     > 0    child quad >= 1   (size/theta)^2      MAC per body, opened by the rest
//   bucket (depth-cap   > 0    -(node id) - 2    +inf                never accepted, "opened" by every
//   cell, compat off)                                                 live body -> summed body by body
//   aggregate (depth-   > 0    -1                0                   accepted as one point mass
//   cap cell, compat on)                                             (project.cu:360-382)
// A non-empty cell whose mass is <= 1e-15 is stored as empty: the reference skips it and its whole
// subtree (project.cu:617).
struct alignas(16) QuadF {
    float xy[8];               // x0,y0, x1,y1, x2,y2, x3,y3: (x,y) adjacent -> one v_pk_add per child
    float m[4], thr[4];
    int32_t child[4];
};
static_assert(sizeof(QuadF) == 80, "QuadF is 20 dwords");

// per node, outside the walk's stream: sorted body range of the cell (export, bucket leaves)
struct NodeAux { int32_t first, count; };

struct TreeCounters {
    uint32_t n_internal;       // I  (written by the scan)
    uint32_t overflow;         // 1 if I > internal capacity
    uint32_t sort_spills;      // buckets of the bucket sort that did not fit LDS (cumulative; bh_sort.hpp)
    uint32_t sort_reruns;      // buckets whose three top-byte passes met a long run of equal top bits and were sorted again in full
    unsigned long long visits, interactions;
    unsigned long long wave_nodes;   // nodes evaluated by wavefronts (one count per wave per node)
    unsigned long long wave_quads;   // sibling quads loaded by wavefronts (one count per wave per quad)
    unsigned long long wave_accepts; // fp64 walks: nodes some lane took a force term from (one count per wave per node)
};

}  // namespace bh
