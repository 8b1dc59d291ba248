// bh_walk_f64.hpp -- fp64 theta-walk for THROUGHPUT (BH_PRECISION_F64): the reference's arithmetic type
// (project.cu:38-65) with the walk design of the fp32 kernel instead of the reference's visiting order.
// Replaces computeForcesGpu (project.cu:679-793) + updateAccVelPos (project.cu:819-836) like bh_walk_exact.hpp,
// and runs on exactly the tree the exact mode builds (keys by fp64 bisection, stable sort, cells, bottom-up
// centre-of-mass pass in the reference's child order): every node -- box size, mass, centre of mass, occupant --
// is BITWISE the reference's.  What differs from BH_PRECISION_F64_EXACT is the walk only:
//   * free visiting order.  The four children of an opened cell are four consecutive NodeD / LinkD records
//     (node ids 1+4r .. 4+4r): one 128-byte + one 32-byte scalar load per opened cell instead of one 40-byte load
//     per visited node, and all four are evaluated before the next load is issued.  The first child some lane
//     opens stays in scalar registers and is the next cell taken (no push / pop); the others go to a
//     register-lane stack (entry k in lane k of three VGPRs, 128 entries: depth-first needs <= 3 * max_depth + 1).
//     A body's terms are therefore added in another order than the reference's: results agree with the oracle
//     to summation rounding (tests: <= 1e-12 relative), not bit for bit.
//   * 1/d by v_rsq_f64 and one third-order correction step instead of IEEE sqrt and three divisions per interaction
//     (project.cu:634, 651-655): d = d2 * rsqrt(d2) + 1e-15 carries the reference's offset (it shifts the
//     acceptance criterion of a near cell by up to 1e-9 relative, so it is kept), the criterion is the
//     reference's `size / d < theta` in the form size < theta * d, and the force is G m_i M d_vec / (d2 * d) with
//     1 / d = y (1 - 1e-15 y), y = rsqrt(d2) (second order: 1e-20).  Acceptance decisions can differ from the
//     oracle's only where size / d is within ~3e-16 of theta: the tests find identical per-body interaction counts.
//   * self skip, empty-node cut-off, depth-cap aggregation: the reference's rules unchanged (occupant index,
//     `mass <= 1e-15`, project.cu:617, 646), so reference_compat means what it means in the exact mode.
// Explicit fma() throughout: this header is compiled with -ffp-contract=off like the rest of the engine unit.
#pragma once

#include "bh_tree.hpp"

namespace bh {

#define BH64_CONSTANT __attribute__((address_space(4)))
extern "C" __device__ int bh64_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

typedef int32_t w64_v16i __attribute__((ext_vector_type(16)));
typedef int32_t w64_v8i __attribute__((ext_vector_type(8)));

struct Quad64 {            // four sibling nodes as the scalar loads deliver them
    w64_v16i a, b;         // NodeD x 4: {cx, cy, m, size} each, 8 dwords per node
    w64_v8i l;             // LinkD x 4: {child, occ}
};

__device__ __forceinline__ double w64_f64(int32_t lo, int32_t hi)
{
    return __hiloint2double(hi, lo);
}

template <bool COMPAT, bool STATS>
__global__ __launch_bounds__(kBlock) void walk_f64_kernel(
    const NodeD *__restrict__ gd, const LinkD *__restrict__ ld, const uint32_t *__restrict__ perm,
    double2 *__restrict__ pos, double2 *__restrict__ vel, const double *__restrict__ mass,
    double2 *__restrict__ force_out, int64_t lo, int64_t hi, double theta, double G, double dt,
    int integrate, TreeCounters *ctr, double *__restrict__ partial, uint32_t *__restrict__ body_counts)
{
    if (ctr->overflow) return;
    const int lane = lane_id();
    const int64_t s = lo + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = s < hi;
    const int64_t body = valid ? (int64_t)perm[s] : -1;
    const double2 p = valid ? pos[body] : double2{0.0, 0.0};
    const double mi = valid ? mass[body] : 0.0;
    double sx = 0.0, sy = 0.0;                       // sum of M * d_vec / (d2 * d)
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0, n_quad = 0;
    uint32_t my_int = 0;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const char BH64_CONSTANT *cg = (const char BH64_CONSTANT *)gd;
    const char BH64_CONSTANT *cl = (const char BH64_CONSTANT *)ld;
    auto load_quad = [&](int32_t first) {
        Quad64 q;
        q.a = *(const w64_v16i BH64_CONSTANT *)(cg + (int64_t)first * 32);
        q.b = *(const w64_v16i BH64_CONSTANT *)(cg + (int64_t)first * 32 + 64);
        q.l = *(const w64_v8i BH64_CONSTANT *)(cl + (int64_t)first * 8);
        return q;
    };
#pragma clang diagnostic pop

    int32_t v_base = 0, v_lo = 0, v_hi = 0, v_base2 = 0, v_lo2 = 0, v_hi2 = 0;   // register-lane stack, 128 entries
    int sp = 0;
    bool h_free = false;                              // hand-off slot of the quad being evaluated
    int32_t h_idx = -1;
    uint64_t h_mask = 0;

    auto push = [&](int32_t child, uint64_t open) {
        if (sp < kWave) {
            v_base = bh64_writelane_i32(child, sp, v_base);
            v_lo = bh64_writelane_i32((int32_t)(uint32_t)open, sp, v_lo);
            v_hi = bh64_writelane_i32((int32_t)(uint32_t)(open >> 32), sp, v_hi);
        } else if (sp < 2 * kWave) {
            v_base2 = bh64_writelane_i32(child, sp - kWave, v_base2);
            v_lo2 = bh64_writelane_i32((int32_t)(uint32_t)open, sp - kWave, v_lo2);
            v_hi2 = bh64_writelane_i32((int32_t)(uint32_t)(open >> 32), sp - kWave, v_hi2);
        }
        ++sp;                                         // (beyond 128 cannot happen: 3 * 31 + 1 entries at max_depth 32)
    };
    auto pop = [&](int32_t &base, uint64_t &mask) {
        --sp;
        if (sp < kWave) {
            base = __builtin_amdgcn_readlane(v_base, sp);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi, sp) << 32) | (uint32_t)__builtin_amdgcn_readlane(v_lo, sp);
        } else {
            base = __builtin_amdgcn_readlane(v_base2, sp - kWave);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi2, sp - kWave) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(v_lo2, sp - kWave);
        }
    };

    // one node for the lanes in `mask` (all arguments wave-uniform)
    auto eval = [&](double cx, double cy, double m, double size, int32_t child, int32_t occ, uint64_t mask) {
        if (m <= 1e-15) return;                                   // project.cu:617
        const bool mine = (mask >> lane) & 1ull;
        const bool leaf = child < 0;                              // project.cu:623-626
        const double dx = cx - p.x, dy = cy - p.y;
        const double d2 = fma(dx, dx, dy * dy);
        // y = 1 / sqrt(d2): v_rsq_f64 is good to ~2^-26; with e = 1 - d2 y0^2 (|e| <~ 3e-8), 1 / sqrt(1 - e) =
        // 1 + e/2 + 3 e^2 / 8 + O(e^3) -- ONE third-order step reaches fp64 rounding (the neglected term is
        // 5 e^3 / 16 ~ 1e-23), five instructions instead of the eight of two Newton steps
#if defined(BH_F64_RSQ32) && BH_F64_RSQ32
        const double y0 = (double)__builtin_amdgcn_rsqf((float)d2);     // A/B: fp32 seed (2^-23) -- the same step still reaches 1e-20
#else
        const double y0 = __builtin_amdgcn_rsq(d2);
#endif
        const double e = fma(-(d2 * y0), y0, 1.0);
        const double y = fma(y0, e * fma(0.375, e, 0.5), y0);
        const double d = fma(d2, y, 1e-15);                       // sqrt(d2) + 1e-15, project.cu:634
        const bool accept = leaf || (size < theta * d);           // size / d < theta, project.cu:643
        bool self = false;
        if (leaf) {
            self = ((int64_t)occ == body);
            if (COMPAT) self = self || ((int64_t)occ + 2 == -body);   // project.cu:646
        }
        const bool take = mine && accept && !self;
        const uint64_t takem = __ballot(take);
        if (takem != 0) {                                         // (a cell every lane opens: nine fp64 instructions saved)
            // M / (d2 * d):  1 / d2 = y * y,  1 / d = 1 / (sqrt(d2) + 1e-15) = y * (1 - 1e-15 * y) to second order
            const double inv_d = fma(-1e-15 * y, y, y);
            const double wgt = take ? m * (y * y) * inv_d : 0.0;
            sx = fma(wgt, dx, sx);
            sy = fma(wgt, dy, sy);
        }
        const uint64_t open = leaf ? 0ull : __ballot(mine && !accept);
        if (STATS) { n_vis += __popcll(mask); ++n_wave; n_int += __popcll(takem); my_int += take ? 1u : 0u; }
        if (open != 0) {
            if (h_free) { h_idx = child; h_mask = open; h_free = false; }
            else push(child, open);
        }
    };
    auto node_of = [&](const Quad64 &q, int k, double &cx, double &cy, double &m, double &size) {
        const w64_v16i &t = (k < 2) ? q.a : q.b;
        const int o = (k & 1) * 8;
        cx = w64_f64(t[o + 0], t[o + 1]); cy = w64_f64(t[o + 2], t[o + 3]);
        m = w64_f64(t[o + 4], t[o + 5]); size = w64_f64(t[o + 6], t[o + 7]);
    };

    // the root (node 0) alone, then quads of four siblings
    {
        const NodeD r = gd[0];
        const LinkD k = ld[0];
        h_free = true; h_idx = -1;
        eval(r.cx, r.cy, r.m, r.size, k.child, k.occ, __ballot(valid));
    }
    int32_t na = h_idx;
    uint64_t nam = h_mask;
    for (;;) {
        int32_t base;
        uint64_t mask;
        if (na >= 0) { base = na; mask = nam; }
        else if (sp > 0) pop(base, mask);
        else break;
        const Quad64 q = load_quad(base);
        if (STATS) ++n_quad;
        h_free = true; h_idx = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double cx, cy, m, size;
            node_of(q, k, cx, cy, m, size);
            eval(cx, cy, m, size, q.l[2 * k], q.l[2 * k + 1], mask);
        }
        na = h_idx; nam = h_mask;
    }
    h_free = false;

    double2 np = p;
    if (valid) {
        const double gm = G * mi;                                 // (G * masses[i]) * nodeMass / d2 * d_vec / d, project.cu:651-658
        const double fx = gm * sx, fy = gm * sy;
        force_out[body] = double2{fx, fy};
        if (integrate) {
            const double ax = G * sx, ay = G * sy;                // F / m_i (updateAccVelPos, project.cu:827-834)
            double2 v = vel[body];
            v.x = fma(ax, dt, v.x);  v.y = fma(ay, dt, v.y);
            vel[body] = v;
            np.x = fma(v.x, dt, np.x);  np.y = fma(v.y, dt, np.y);
            pos[body] = np;
        }
        if (STATS && body_counts) body_counts[body] = my_int;
    }
    if (partial) block_bounds_to_partial(valid, np.x, np.y, partial + 4 * (size_t)blockIdx.x);
    if (STATS && lane == 0) {
        atomicAdd(&ctr->visits, n_vis);
        atomicAdd(&ctr->interactions, n_int);
        atomicAdd(&ctr->wave_nodes, n_wave);
        atomicAdd(&ctr->wave_quads, n_quad);
    }
}

}  // namespace bh
