// bh_walk_fast.h -- launcher interface of the fp32 walk (bh_walk_fast.hip is its own translation
// unit so that it can be compiled with FMA contraction while the tree build and the exact walk
// are compiled with -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bh {

struct QuadF;
struct NodeAux;
struct TreeCounters;

struct WalkFastArgs {
    const QuadF *quads;        // quad 0 = root
    const NodeAux *aux;        // per node: sorted body range (bucket leaves)
    const float2 *spos;        // positions in sorted (Hilbert) order
    const float *smass;        // masses in sorted order (bucket leaves only)
    const uint32_t *perm;      // sorted index -> caller index
    float2 *pos, *vel;         // caller-order state (updated when integrate && !to_sorted);
                               // double2 arrays when state64 (mixed precision)
    int32_t state64;
    float4 *sstate;            // sorted-order output {x, y, vx, vy} per body (integrate && to_sorted): ONE exchange buffer
    float2 *acc_out;           // caller-order accelerations, may be null
    TreeCounters *ctr;
    double *partial;           // per-workgroup min/max of the new positions, may be null
    int64_t lo, hi;            // sorted range walked by this launch
    float G, dt;
    int integrate, to_sorted;
    uint32_t nblocks, pad0;        // nblocks: filled by the launcher
    const void *bucket_consts;     // device block {aux, spos, smass, 0}: the assembly loop's bucket path reads its pointers here
    uint32_t *group_cost;          // per 64-body group: loop iterations of its walk (load-balancing weight), may be null
    int32_t pair_limit;            // one-wave walk: two stack entries per iteration while sp <= pair_limit
                                   // (116 - 3 * (max_depth - 1), never negative: the 128-entry stack bound, bh_walk_fast.hip)
    uint64_t *timeline;            // -DBHGPU_EXPERIMENTS builds: per-wave {start, end, hw id, cost, clock start, clock end} (scripts/walk_timeline.py)
    // forest walk (distributed step): besides the local tree (root quad 0) the bodies walk the
    // locally-essential trees received from the peers, whose root quads sit at
    // forest_base + t * let_cap for every t != self_rank, t < n_trees.  n_trees == 0: local tree only.
    int32_t n_trees, self_rank;
    // part: 0 = the whole forest in one launch; 1 = the local tree only, raw sums parked in acc_part
    // (sorted order), nothing else written; 2 = the received LETs only, continuing from acc_part, then
    // the usual epilogue.  Lets the LET all_to_all overlap the local walk.
    int32_t part;
    float2 *acc_part;
    int64_t forest_base, let_cap;
    double *slots;                 // bh_bounds.hpp: running bounds records of this launch's workgroups (with `partial`), may be null
    uint32_t *body_counts;         // counting variant (BH_FLAG_WALK_STATS): accepted force evaluations per body, added
                                   // atomically at the body's device slot (the engine zeroes it); may be null
};

// split: 1 = one wave per 64 bodies; 2/4/8 = that many waves share each 64-body group (few bodies; more than 8 -> 8).
// The launch writes one `partial` entry per workgroup: per 256 bodies, or per 64 when split > 1
// (walk_fast_split_effective tells which applies).
// use_asm: take the hand-scheduled loop where it applies (32-bit byte offsets into the quad array: the
// caller checks that the forest is smaller than 2 GiB and the bodies fewer than 2^28).
hipError_t launch_walk_fast(const WalkFastArgs &a, bool lds_stack, bool stats, int split, bool use_asm, hipStream_t st);
inline bool walk_fast_split_effective(const WalkFastArgs &a, bool lds_stack, int split)
{
    return split > 1 && !lds_stack && a.n_trees <= 56;
}

}  // namespace bh
