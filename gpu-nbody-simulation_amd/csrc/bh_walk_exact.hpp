// bh_walk_exact.hpp -- fp64 theta-walk that reproduces the reference's forces BIT FOR BIT, fused
// with the integrator.  Replaces computeForces / computeForcesGpu (project.cu:593-675, 679-793)
// and updateAccVelPos (project.cu:819-836).  Compiled with -ffp-contract=off.
//
// One wavefront walks the tree for 64 Morton-adjacent bodies (one body per lane).  The traversal
// state is wave-uniform and lives in LDS: one entry per tree level = {first child of the quad,
// 64-bit mask of the lanes that opened the parent, next child to visit}.  Children are visited
// 3,2,1,0 -- the reference's pop order (it pushes 0..3 on a LIFO, project.cu:662-668) -- and a
// child's whole subtree is finished before its sibling starts.  The wave therefore visits the
// UNION of its lanes' private walks in the reference's DFS order, and each lane, masked to the
// nodes its own walk would pop, adds its terms in exactly the reference's order: same fp64
// operations, same order, same bits.  Node loads are wave-uniform (scalar loads), so a visit
// costs one 40-byte read per wave instead of 64 divergent gathers.
#pragma once

#include "bh_tree.hpp"

namespace bh {

constexpr int kExactLevels = 34;   // max_depth <= 32 -> at most 32 stacked levels
extern "C" __device__ int exact_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

template <bool COMPAT, bool STATS>
__global__ __launch_bounds__(kBlock) void walk_exact_kernel(
    const NodeD *__restrict__ gd, const LinkD *__restrict__ ld, const uint32_t *__restrict__ perm,
    double2 *__restrict__ pos, double2 *__restrict__ vel, const double *__restrict__ mass,
    double2 *__restrict__ force_out, int64_t lo, int64_t hi, double theta, double G, double dt,
    int integrate, TreeCounters *ctr, double *__restrict__ partial, double *slots, int bpw)
{
    // bpw: bodies per wavefront, a power of two <= 64 (lanes bpw .. 63 idle).  A wave's walk is ONE dependent chain over the
    // union of its lanes' walks; a launch of few bodies leaves the GPU empty however it is cut, so the engine gives every
    // wave fewer bodies -- 1 at N <= 4,096 -- and the chain shrinks to one body's walk (config 1, N = 1,024: 0.30 -> see
    // DESIGN.md section 4).  Every lane still adds its own terms in the reference's order: same bits.
    // The traversal stack -- one entry per tree level: {quad of the level, next child to visit, lanes that walk it} -- lives
    // in four VGPRs, entry k in lane k (round 3; it was three LDS arrays written by lane 0 and read back through
    // v_readfirstlane: a round trip through LDS on every visited node of a walk that is one long dependent chain).
    int32_t v_quad = 0, v_next = 0, v_mlo = 0, v_mhi = 0;

    if (ctr->overflow) return;
    const int lane = lane_id();
    const int64_t s = lo + ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * bpw + lane;
    const bool valid = lane < bpw && s < hi;
    const int64_t body = valid ? (int64_t)perm[s] : -1;
    const double2 p = valid ? pos[body] : double2{0.0, 0.0};
    const double mi = valid ? mass[body] : 0.0;
    const double Gm = G * mi;                      // (G * masses[i]) * nodeMass, project.cu:651
    double fx = 0.0, fy = 0.0;
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0;

    // evaluate one node for the lanes in `live`; returns the mask of lanes that must open it
    auto visit = [&](int32_t node, uint64_t live, int32_t &child_out) -> uint64_t {
        // (both requests of a node issued together and waited for once; the compiler would fetch the mass and the links first
        // and the centre only after the empty test -- two scalar round trips per visited node)
        typedef int32_t v8i __attribute__((ext_vector_type(8)));
        typedef int32_t v2i __attribute__((ext_vector_type(2)));
        v8i qa; v2i ka;
        {
            const NodeD *pq = gd + node;
            const LinkD *pk = ld + node;
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n\t"
                         "s_load_dwordx2 %1, %3, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(qa), "=&s"(ka) : "s"(pq), "s"(pk) : "memory");
        }
        const NodeD q{__hiloint2double(qa[1], qa[0]), __hiloint2double(qa[3], qa[2]), __hiloint2double(qa[5], qa[4]),
                      __hiloint2double(qa[7], qa[6])};
        const LinkD k{ka[0], ka[1]};
        child_out = k.child;
        if (q.m <= 1e-15) return 0;                // project.cu:617
        const bool mine = (live >> lane) & 1ull;
        const bool leaf = k.child < 0;             // all four children -1, project.cu:623-626
        const double dx = q.cx - p.x;
        const double dy = q.cy - p.y;
        const double d2 = dx * dx + dy * dy;
        const double d = sqrt(d2) + 1e-15;         // project.cu:634
        const bool accept = leaf || (q.size / d < theta);   // project.cu:643
        bool self = false;
        if (leaf) {
            self = ((int64_t)k.occ == body);
            if (COMPAT) self = self || ((int64_t)k.occ + 2 == -body);   // project.cu:646
        }
        if (mine && accept && !self) {
            const double f = (Gm * q.m) / d2;      // project.cu:651
            const double ux = dx / d, uy = dy / d; // project.cu:654-655
            fx += f * ux;
            fy += f * uy;
        }
        if (STATS) {
            n_vis += __popcll(live);
            ++n_wave;
            n_int += __popcll(__ballot(mine && accept && !self));
        }
        if (leaf) return 0;
        return __ballot(mine && !accept);
    };

    auto push = [&](int at, int32_t quad, uint64_t mask) {
        v_quad = exact_writelane_i32(quad, at, v_quad);
        v_next = exact_writelane_i32(3, at, v_next);
        v_mlo = exact_writelane_i32((int32_t)(uint32_t)mask, at, v_mlo);
        v_mhi = exact_writelane_i32((int32_t)(uint32_t)(mask >> 32), at, v_mhi);
    };
    int sp = -1;
    {
        int32_t child;
        const uint64_t open = visit(0, __ballot(valid), child);
        if (open != 0 && child >= 0) { sp = 0; push(0, child, open); }
    }
    while (sp >= 0) {
        const int32_t c = __builtin_amdgcn_readlane(v_next, sp);
        if (c < 0) { --sp; continue; }
        const int32_t quad = __builtin_amdgcn_readlane(v_quad, sp);
        const uint64_t live = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_mhi, sp) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane(v_mlo, sp);
        v_next = exact_writelane_i32(c - 1, sp, v_next);
        int32_t child;
        const uint64_t open = visit(quad + c, live, child);
        if (open != 0 && child >= 0 && sp + 1 < kExactLevels) { ++sp; push(sp, child, open); }
    }

    double2 np = p;
    if (valid) {
        force_out[body] = double2{fx, fy};
        if (integrate) {
            // updateAccVelPos, project.cu:827-834
            const double ax = fx / mi, ay = fy / mi;
            double2 v = vel[body];
            v.x += ax * dt;  v.y += ay * dt;
            vel[body] = v;
            np.x += v.x * dt;  np.y += v.y * dt;
            pos[body] = np;
        }
    }
    // min/max of the new positions per workgroup: the next step's root box needs no body pass
    if (partial) block_bounds_to_partial(valid, np.x, np.y, partial + 4 * (size_t)blockIdx.x, slots);
    if (STATS) {
        // one atomic per wave
        if (lane == 0) {
            atomicAdd(&ctr->visits, n_vis);
            atomicAdd(&ctr->interactions, n_int);
            atomicAdd(&ctr->wave_nodes, n_wave);
        }
    }
}

}  // namespace bh
