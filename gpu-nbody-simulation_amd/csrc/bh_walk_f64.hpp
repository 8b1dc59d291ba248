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

// One wavefront per workgroup: a body group's walk takes 0.3x .. 3x the mean, and a workgroup's slot is only
// re-used when its LAST wave has finished -- with four waves per workgroup the mean occupancy of this kernel was
// 5.5 of 8 waves per SIMD (SQ_WAVE_CYCLES), with one it is the dispatcher's to fill wave by wave.
#ifndef BH_F64_BLOCK
#define BH_F64_BLOCK 64
#endif
constexpr int kF64Block = BH_F64_BLOCK;

// DEEP: trees deeper than 21 levels need more than 64 stack entries (3 * (max_depth - 1) + 1): a second register-lane
// tier, and a test on every push and pop that the usual depth does without.
template <bool COMPAT, bool STATS, bool DEEP = false>
__global__ __launch_bounds__(kF64Block) void walk_f64_kernel(
    const NodeD *__restrict__ gd, const LinkD *__restrict__ ld, const uint32_t *__restrict__ perm,
    double2 *__restrict__ pos, double2 *__restrict__ vel, const double *__restrict__ mass,
    double2 *__restrict__ force_out, int64_t lo, int64_t hi, double theta, double G, double dt,
    int integrate, TreeCounters *ctr, double *__restrict__ partial, uint32_t *__restrict__ body_counts, double *slots)
{
    if (ctr->overflow) return;
    const int lane = lane_id();
    const int64_t s = lo + (int64_t)blockIdx.x * kF64Block + threadIdx.x;
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
    // (all three requests of a quad are issued together and waited for once: left to itself the compiler issues the second
    // half and the links only after the first half has arrived and its first node has passed the empty test -- two round
    // trips per quad)
    auto load_quad = [&](int32_t first) {
        Quad64 q;
        const char BH64_CONSTANT *pn = cg + (int64_t)first * 32;
        const char BH64_CONSTANT *pl = cl + (int64_t)first * 8;
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\t"
                     "s_load_dwordx16 %1, %3, 0x40\n\t"
                     "s_load_dwordx8 %2, %4, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(q.a), "=&s"(q.b), "=&s"(q.l)
                     : "s"(pn), "s"(pl)
                     : "memory");
        return q;
    };
    // (Touching the next quad's three cache lines with scalar loads as soon as the hand-off child is known -- gfx9 has no
    // scalar prefetch; the touches went to a register above the compiler's allocation -- made the walk SLOWER, 0.940 ->
    // 0.985 ms: as in the fp32 loop, a scalar-memory request is the most expensive instruction there is.)

    int32_t v_base = 0, v_lo = 0, v_hi = 0, v_base2 = 0, v_lo2 = 0, v_hi2 = 0;   // register-lane stack, 128 entries
    int sp = 0;
    int32_t h_idx = 0;                                // hand-off slot of the quad being evaluated: < 0 = free (a child index is > 0)
    uint64_t h_mask = 0;

    auto push = [&](int32_t child, uint64_t open) {
        if (!DEEP || sp < kWave) {
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
        if (!DEEP || sp < kWave) {
            base = __builtin_amdgcn_readlane(v_base, sp);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi, sp) << 32) | (uint32_t)__builtin_amdgcn_readlane(v_lo, sp);
        } else {
            base = __builtin_amdgcn_readlane(v_base2, sp - kWave);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi2, sp - kWave) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(v_lo2, sp - kWave);
        }
    };

    // one node for the lanes in `mask` (all arguments wave-uniform).  The kernel is vector-issue bound (every fp64
    // instruction costs 5.3 cycles per wave, v_rsq_f64 17: scripts/calib/f64_issue_calib.hip), so the per-node work is
    // written out instruction by instruction and everything wave-uniform stays on the scalar side: the empty-node
    // and leaf tests are integer tests on SGPRs, the lane sets are 64-bit masks straight from v_cmp (no per-lane
    // booleans), and the force is accumulated under EXEC = the accepting lanes instead of through selects.
    const int32_t body32 = (int32_t)body;                         // (perm is 32-bit; -1 on padding lanes)
    const int32_t compat32 = -body32 - 2;                         // occ + 2 == -body, project.cu:646
    const double c0375 = 0.375, tiny = 1e-15, ntiny = -1e-15;
    auto eval = [&](double cx, double cy, double m, double size, int32_t child, int32_t occ, uint64_t mask) {
        // m <= 1e-15 (project.cu:617) on the bit pattern, with scalar integer compares: for doubles that are not
        // NaN, a <= b  <=>  the same on their sign-magnitude integers (bits of 1e-15: 0x3CD203AF'9EE75616); a node's
        // mass is a sum of the bodies' masses or +0.0
        {
            const int32_t mh = __double2hiint(m);
            if (__builtin_expect(mh <= 0x3CD203AF, 0)) {            // (nested: one s_cmp + branch on the usual path, no booleans in SGPR pairs)
                if (mh < 0x3CD203AF) return;
                if ((uint32_t)__double2loint(m) <= 0x9EE75616u) return;
            }
        }
        const bool leaf = child < 0;                              // project.cu:623-626
        double dx, dy, d2, y, t, e;
        // y = 1 / sqrt(d2): v_rsq_f64 is good to ~2^-26; with e = 1 - d2 y0^2 (|e| <~ 3e-8), 1 / sqrt(1 - e) =
        // 1 + e/2 + 3 e^2 / 8 + O(e^3) -- ONE third-order step reaches fp64 rounding (the neglected term is
        // 5 e^3 / 16 ~ 1e-23), five instructions instead of the eight of two Newton steps.  (An fp32 seed,
        // v_cvt + v_rsq_f32 + v_cvt, costs the same 17 cycles as v_rsq_f64: measured, no gain.)
        asm volatile(
            "v_add_f64 %[dx], %[cx], -%[px]\n\t"
            "v_add_f64 %[dy], %[cy], -%[py]\n\t"
            "v_mul_f64 %[d2], %[dy], %[dy]\n\t"
            "v_fma_f64 %[d2], %[dx], %[dx], %[d2]\n\t"
            "v_rsq_f64 %[y], %[d2]\n\t"
            "s_nop 0\n\t"                                          // gfx94x/95x: one wait state between a TRANS result and its use
            "v_mul_f64 %[t], %[d2], -%[y]\n\t"
            "v_fma_f64 %[e], %[t], %[y], 1.0\n\t"
            "v_fma_f64 %[t], %[e], %[k], 0.5\n\t"
            "v_mul_f64 %[e], %[e], %[t]\n\t"
            "v_fma_f64 %[y], %[y], %[e], %[y]"
            : [dx] "=&v"(dx), [dy] "=&v"(dy), [d2] "=&v"(d2), [y] "=&v"(y), [t] "=&v"(t), [e] "=&v"(e)
            : [cx] "s"(cx), [cy] "s"(cy), [px] "v"(p.x), [py] "v"(p.y), [k] "s"(c0375));
        uint64_t takem, open;
        if (leaf) {
            uint64_t self = __builtin_amdgcn_ballot_w64(occ == body32);
            if (COMPAT) self |= __builtin_amdgcn_ballot_w64(occ == compat32);
            takem = mask & ~self;
            open = 0;
        } else {
            // d = sqrt(d2) + 1e-15 (project.cu:634) as d2 * y + 1e-15; size / d < theta (project.cu:643) as size < theta * d
            uint64_t acc;
            asm volatile(
                "v_fma_f64 %[t], %[d2], %[y], %[tiny]\n\t"
                "v_mul_f64 %[t], %[theta], %[t]\n\t"
                "v_cmp_lt_f64_e64 %[acc], %[size], %[t]"
                : [t] "=&v"(t), [acc] "=s"(acc)
                : [d2] "v"(d2), [y] "v"(y), [tiny] "v"(tiny), [theta] "s"(theta), [size] "s"(size));
            takem = mask & acc;
            open = mask & ~acc;
        }
        if (takem != 0) {                                         // (a cell every lane opens: seven fp64 instructions saved)
            // M / (d2 * d):  1 / d2 = y * y,  1 / d = 1 / (sqrt(d2) + 1e-15) = y * (1 - 1e-15 * y) to second order;
            // added for the accepting lanes only (the others may hold inf / NaN here: a body's own leaf has d2 = 0)
            uint64_t saved;
            asm volatile(
                "v_mul_f64 %[t], %[y], %[ntiny]\n\t"
                "v_mul_f64 %[e], %[y], %[y]\n\t"
                "v_fma_f64 %[t], %[t], %[y], %[y]\n\t"
                "v_mul_f64 %[e], %[e], %[m]\n\t"
                "v_mul_f64 %[t], %[e], %[t]\n\t"
                "s_and_saveexec_b64 %[saved], %[takem]\n\t"
                "v_fma_f64 %[sx], %[t], %[dx], %[sx]\n\t"
                "v_fma_f64 %[sy], %[t], %[dy], %[sy]\n\t"
                "s_mov_b64 exec, %[saved]"
                : [t] "=&v"(t), [e] "=&v"(e), [sx] "+v"(sx), [sy] "+v"(sy), [saved] "=&s"(saved)
                : [y] "v"(y), [ntiny] "s"(ntiny), [m] "s"(m), [takem] "s"(takem), [dx] "v"(dx), [dy] "v"(dy)
                : "scc");
        }
        if (STATS) { n_vis += __popcll(mask); ++n_wave; n_int += __popcll(takem); my_int += (uint32_t)((takem >> lane) & 1ull); }
        if (open != 0) {
            if (h_idx < 0) { h_idx = child; h_mask = open; }
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
        h_idx = -1;
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
        h_idx = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double cx, cy, m, size;
            node_of(q, k, cx, cy, m, size);
            eval(cx, cy, m, size, q.l[2 * k], q.l[2 * k + 1], mask);
        }
        na = h_idx; nam = h_mask;
    }

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
    if (partial) {                                                // min/max of the new positions per workgroup (next root box)
        if (kF64Block == kWave) {
            const double xlo = wave_min(valid ? np.x : (double)INFINITY), xhi = wave_max(valid ? np.x : -(double)INFINITY);
            const double ylo = wave_min(valid ? np.y : (double)INFINITY), yhi = wave_max(valid ? np.y : -(double)INFINITY);
            if (lane == 0) {
                double *o = partial + 4 * (size_t)blockIdx.x;
                o[0] = xlo; o[1] = xhi; o[2] = ylo; o[3] = yhi;
                if (slots) bounds_to_slot(xlo, xhi, ylo, yhi, slots, blockIdx.x);
            }
        } else {
            block_bounds_to_partial(valid, np.x, np.y, partial + 4 * (size_t)blockIdx.x, slots);
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&ctr->visits, n_vis);
        atomicAdd(&ctr->interactions, n_int);
        atomicAdd(&ctr->wave_nodes, n_wave);
        atomicAdd(&ctr->wave_quads, n_quad);
    }
}

}  // namespace bh
