// bh_migrate.hpp -- device side of body migration and ORB re-balancing for the LET decomposition
// (include/bhgpu.h: bh_orb_histogram, bh_migrate_pack, bh_migrate_unpack).  The reference keeps every body
// on one GPU (project.cu:918-1024); this is new design.
//
//   orb_hist_kernel        : weighted histogram of the local bodies for every region of one level of the cut tree
//   migrate_classify_kernel: destination rank of every body -> (key = rank, value = slot) for ONE stable
//                            radix pass of bh_sort.hpp, which groups the slots by destination
//   migrate_pack_kernel    : the bodies in that order -> 6-double records in the send buffer
//   migrate_unpack_kernel  : received records -> state arrays
#pragma once

#include "../../include/bhgpu.h"
#include "bh_prims.hpp"

namespace bh {

constexpr int kMigrateRecord = 6;   // doubles per body: x, y, vx, vy, mass, id

// Rank range reached after at most max_level cuts: returns its first rank, *cut = index of the cut that
// splits it further (valid while *nr > 1), *nr = number of ranks in it.
__host__ __device__ inline int orb_descend(const bh_orb_cuts &c, double x, double y, int max_level, int *cut, int *nr)
{
    int r0 = 0, n = c.world, k = 0;
    for (int level = 0; n > 1 && level < max_level; ++level) {
        const int nl = n / 2;
        const double v = c.axis[k] ? y : x;
        if (v < c.value[k]) { k += 1; n = nl; }                 // left subtree: its nl - 1 cuts follow directly
        else { k += nl; r0 += nl; n -= nl; }                    // right subtree: after those
    }
    *cut = k; *nr = n;
    return r0;
}

template <typename Real2>
__global__ __launch_bounds__(kBlock) void orb_hist_kernel(const Real2 *__restrict__ pos,
                                                           const uint32_t *__restrict__ perm,
                                                           const uint32_t *__restrict__ group_cost, int64_t n,
                                                           bh_orb_cuts cuts, int level,
                                                           unsigned long long *__restrict__ hist)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;     // sorted index when perm != nullptr
    if (s >= n) return;
    const int64_t b = perm ? (int64_t)perm[s] : s;
    const double x = (double)pos[b].x, y = (double)pos[b].y;
    int k, nr;
    (void)orb_descend(cuts, x, y, level, &k, &nr);
    if (nr <= 1) return;                                               // this region is one rank already
    const int ax = cuts.axis[k];
    const double lo = cuts.box[2 * ax], hi = cuts.box[2 * ax + 1];
    const double t = ((ax ? y : x) - lo) / (hi - lo) * (double)BH_ORB_BINS;
    int bin = (t >= 0.0) ? ((t < (double)BH_ORB_BINS) ? (int)t : BH_ORB_BINS - 1) : 0;   // NaN -> 0
    unsigned long long w = 1ull;
    if (group_cost) { const uint32_t g = group_cost[s >> 6]; w = g ? g : 1u; }
    atomicAdd(&hist[(size_t)k * BH_ORB_BINS + bin], w);
}

template <typename Real2>
__global__ __launch_bounds__(kBlock) void migrate_classify_kernel(const Real2 *__restrict__ pos, int64_t n,
                                                                   bh_orb_cuts cuts, uint64_t *__restrict__ keys,
                                                                   uint32_t *__restrict__ vals)
{
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= n) return;
    int k, nr;
    const int r = orb_descend(cuts, (double)pos[b].x, (double)pos[b].y, 64, &k, &nr);
    keys[b] = (uint64_t)r;
    vals[b] = (uint32_t)b;
}

template <typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void migrate_pack_kernel(const uint32_t *__restrict__ order,
                                                               const Real2 *__restrict__ pos,
                                                               const Real2 *__restrict__ vel,
                                                               const Real *__restrict__ mass,
                                                               const uint32_t *__restrict__ orig,
                                                               const int64_t *__restrict__ gid, int64_t n,
                                                               double *__restrict__ send)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const uint32_t b = order[j];
    const Real2 p = pos[b], v = vel[b];
    double *o = send + (size_t)kMigrateRecord * j;
    o[0] = (double)p.x; o[1] = (double)p.y; o[2] = (double)v.x; o[3] = (double)v.y; o[4] = (double)mass[b];
    o[5] = (double)gid[orig ? orig[b] : b];                            // ids are kept in caller order
}

template <typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void migrate_unpack_kernel(const double *__restrict__ recv, int64_t n,
                                                                 Real2 *__restrict__ pos, Real2 *__restrict__ vel,
                                                                 Real *__restrict__ mass, float2 *__restrict__ acc,
                                                                 int64_t *__restrict__ gid)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const double *r = recv + (size_t)kMigrateRecord * j;
    pos[j] = Real2{(Real)r[0], (Real)r[1]};
    vel[j] = Real2{(Real)r[2], (Real)r[3]};
    mass[j] = (Real)r[4];
    acc[j] = float2{0.f, 0.f};
    gid[j] = (int64_t)r[5];
}

__global__ __launch_bounds__(kBlock) void iota_i64_kernel(int64_t *__restrict__ a, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] = i;
}

}  // namespace bh
