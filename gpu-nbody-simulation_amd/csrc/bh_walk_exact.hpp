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

// THR (the product): the node's `size` slot holds the EXACT d2 threshold of its acceptance test (exact_walk_threshold,
// bh_tree.hpp), so the test is one comparison instead of sqrt + add + divide -- the same decisions bit for bit -- and the
// square root and the three divisions of an accepted term run only on nodes some lane accepts, through the SAME
// instruction sequences the compiler expands sqrt() and `/` to, minus their range handling (v_div_scale / v_div_fmas /
// v_div_fixup, the 2^256 pre-scaling of sqrt), which is the identity on the operand ranges checked first; dx / d and
// dy / d share the refined reciprocal of d.  Any lane outside those ranges sends the wave through the plain expressions
// for that node.  THR = false (BH_FLAG_WALK_PORTABLE; the node kernel then stores the size) is the walk written as the
// reference writes it: tests/test_gpu_exact.py holds the two bit-identical on every input, extreme ranges included.
// ASM (the product, with THR): the evaluation of one node as ONE block of gfx950 assembly -- the lanes that walk the node,
// then those that accept it, then those that take its term are selected by narrowing EXEC (s_and_b64 / the compare's
// mask) instead of by per-lane booleans, the three Newton chains (sqrt, 1 / d2, 1 / d) are interleaved, and nothing is
// copied between registers: 50 vector instructions for an accepted node against the compiler's 66 for the C++ statement
// of the same operations (THR without ASM: the counting variant runs it, and the three are compared bit for bit).
template <bool COMPAT, bool STATS, bool THR = true, bool ASM = false>
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
    // operand ranges on which the range handling of IEEE division and sqrt is the identity (v_div_scale acts when the
    // exponents of numerator and denominator differ by >= 768 or either is within ~2^53 of the ends of the format)
    constexpr double kLaneLo = 0x1p-150, kLaneHi = 0x1p150, kTiny = 0x1p-200, kHuge = 0x1p400;
    const bool lane_safe = THR && fabs(Gm) >= kLaneLo && fabs(Gm) <= kLaneHi;
    const uint64_t lanes_safe = __ballot(lane_safe);
    const int32_t body_lo = (int32_t)body, body_alt = (int32_t)(-body - 2);     // (bodies < 2^31: bh_create)
    double fx = 0.0, fy = 0.0;
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0;

    // evaluate one node -- its record `q` and links `k` in scalar registers -- for the lanes in `live`; returns the mask of
    // lanes that must open it
    auto eval = [&](const NodeD q, const LinkD k, uint64_t live, int32_t &child_out) -> uint64_t {
        child_out = k.child;
        if (ASM) {
            // status 1: some lane that takes the node has an operand outside the ranges -- nothing was added, the C++
            // statement below evaluates the node with the plain expressions
            int32_t status;
            uint64_t open, sav, tm;
            double dx, dy, d2, ta, tb, tc, td, te, tf, tg;
            asm volatile(
                // the node's mass from its high word: below 1e-15 = 0x3CD203AF'9EE75616 the node is empty (project.cu:617)
                // -- every second child of the tree's last level --, above it and below 2^150 the short sequences apply;
                // the one high word in between, negative numbers and NaNs go to the C++ statement
                "s_mov_b64 %[sav], exec\n\t"
                "s_mov_b32 %[st], 0\n\t"
                "s_mov_b64 %[open], 0\n\t"
                "s_cmp_lt_u32 %[mhi], 0x3cd203af\n\t"
                "s_cbranch_scc1 9f\n\t"
                "s_sub_u32 %[st], %[mhi], 0x3cd203b0\n\t"
                "s_cmp_lt_u32 %[st], 0xc7dfc50\n\t"             // 0x49500000 - 0x3cd203b0
                "s_mov_b32 %[st], 0\n\t"
                "s_cbranch_scc0 8f\n\t"
                "s_and_b64 exec, exec, %[live]\n\t"
                "v_add_f64 %[dx], %[cx], -%[px]\n\t"
                "v_add_f64 %[dy], %[cy], -%[py]\n\t"
                "v_mul_f64 %[ta], %[dx], %[dx]\n\t"
                "v_mul_f64 %[tb], %[dy], %[dy]\n\t"
                "v_add_f64 %[d2], %[ta], %[tb]\n\t"
                "s_cmp_lt_i32 %[child], 0\n\t"
                "s_cbranch_scc1 1f\n\t"
                // subdivided cell: lanes at or beyond the threshold accept it (project.cu:634, 643), the others open it
                "v_cmp_ge_f64 vcc, %[d2], %[thr]\n\t"
                "s_andn2_b64 %[open], exec, vcc\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 9f\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                // leaf: every lane takes it but its occupant (project.cu:623-626, 646)
                "s_mov_b64 %[open], 0\n\t"
                "v_cmp_ne_u32 vcc, %[occ], %[body]\n\t"
                "v_cmp_ne_u32 %[tm], %[occ], %[alt]\n\t"
                "s_and_b64 vcc, vcc, %[tm]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 9f\n"
                "2:\n\t"
                // the operand ranges of the short sequences, over the lanes that take the node
                "v_cmp_ge_f64 vcc, %[ta], %[tiny2]\n\t"
                "v_cmp_ge_f64 %[tm], %[tb], %[tiny2]\n\t"
                "s_and_b64 vcc, vcc, %[tm]\n\t"
                "v_cmp_le_f64 %[tm], %[d2], %[huge]\n\t"
                "s_and_b64 vcc, vcc, %[tm]\n\t"
                "s_and_b64 vcc, vcc, %[lsafe]\n\t"
                "s_andn2_b64 %[tm], exec, vcc\n\t"
                "s_cbranch_scc1 8f\n\t"
                // sqrt(d2) in td (y, g, h: one coupled step, two residual corrections), 1 / d2 in tb, G m_i m in tc
                "v_rsq_f64 %[ta], %[d2]\n\t"
                "v_rcp_f64 %[tb], %[d2]\n\t"
                "v_mul_f64 %[tc], %[gm], %[m]\n\t"
                "v_mul_f64 %[td], %[d2], %[ta]\n\t"
                "v_mul_f64 %[te], %[ta], 0.5\n\t"
                "v_fma_f64 %[ta], -%[d2], %[tb], 1.0\n\t"
                "v_fma_f64 %[tf], -%[te], %[td], 0.5\n\t"
                "v_fma_f64 %[tb], %[tb], %[ta], %[tb]\n\t"
                "v_fma_f64 %[td], %[td], %[tf], %[td]\n\t"
                "v_fma_f64 %[te], %[te], %[tf], %[te]\n\t"
                "v_fma_f64 %[ta], -%[d2], %[tb], 1.0\n\t"
                "v_fma_f64 %[tf], -%[td], %[td], %[d2]\n\t"
                "v_fma_f64 %[tb], %[tb], %[ta], %[tb]\n\t"
                "v_fma_f64 %[td], %[tf], %[te], %[td]\n\t"
                "v_mul_f64 %[ta], %[tc], %[tb]\n\t"
                "v_fma_f64 %[tf], -%[td], %[td], %[d2]\n\t"
                "v_fma_f64 %[tg], -%[d2], %[ta], %[tc]\n\t"
                "v_fma_f64 %[td], %[tf], %[te], %[td]\n\t"
                "v_fma_f64 %[ta], %[tg], %[tb], %[ta]\n\t"      // ta = (G m_i m) / d2, project.cu:651
                "v_add_f64 %[td], %[td], %[eps]\n\t"            // td = sqrt(d2) + 1e-15, project.cu:634
                "v_rcp_f64 %[tb], %[td]\n\t"
                "s_nop 0\n\t"
                "v_fma_f64 %[tc], -%[td], %[tb], 1.0\n\t"
                "v_fma_f64 %[tb], %[tb], %[tc], %[tb]\n\t"
                "v_fma_f64 %[tc], -%[td], %[tb], 1.0\n\t"
                "v_fma_f64 %[tb], %[tb], %[tc], %[tb]\n\t"
                "v_mul_f64 %[tc], %[dx], %[tb]\n\t"
                "v_mul_f64 %[te], %[dy], %[tb]\n\t"
                "v_fma_f64 %[tf], -%[td], %[tc], %[dx]\n\t"
                "v_fma_f64 %[tg], -%[td], %[te], %[dy]\n\t"
                "v_fma_f64 %[tc], %[tf], %[tb], %[tc]\n\t"      // dx / d, project.cu:654
                "v_fma_f64 %[te], %[tg], %[tb], %[te]\n\t"      // dy / d, project.cu:655
                "v_mul_f64 %[tc], %[ta], %[tc]\n\t"
                "v_mul_f64 %[te], %[ta], %[te]\n\t"
                "v_add_f64 %[fx], %[fx], %[tc]\n\t"
                "v_add_f64 %[fy], %[fy], %[te]\n\t"
                "s_branch 9f\n"
                "8:\n\t"
                "s_mov_b32 %[st], 1\n"
                "9:\n\t"
                "s_mov_b64 exec, %[sav]"
                : [st] "=&s"(status), [open] "=&s"(open), [sav] "=&s"(sav), [tm] "=&s"(tm), [dx] "=&v"(dx), [dy] "=&v"(dy),
                  [d2] "=&v"(d2), [ta] "=&v"(ta), [tb] "=&v"(tb), [tc] "=&v"(tc), [td] "=&v"(td), [te] "=&v"(te),
                  [tf] "=&v"(tf), [tg] "=&v"(tg), [fx] "+v"(fx), [fy] "+v"(fy)
                : [live] "s"(live), [cx] "s"(q.cx), [cy] "s"(q.cy), [m] "s"(q.m), [mhi] "s"(__double2hiint(q.m)), [thr] "s"(q.size), [child] "s"(k.child),
                  [occ] "s"(k.occ), [px] "v"(p.x), [py] "v"(p.y), [gm] "v"(Gm), [body] "v"(body_lo),
                  [alt] "v"(COMPAT ? body_alt : body_lo), [tiny2] "s"(kTiny * kTiny), [huge] "s"(kHuge), [eps] "s"(1e-15),
                  [lsafe] "s"(lanes_safe)
                : "vcc", "scc");
            if (status == 0) return open;
        }
        if (q.m <= 1e-15) return 0;                // project.cu:617
        const bool mine = (live >> lane) & 1ull;
        const bool leaf = k.child < 0;             // all four children -1, project.cu:623-626
        const double dx = q.cx - p.x;
        const double dy = q.cy - p.y;
        const double d2 = dx * dx + dy * dy;
        bool accept;
        double d = 0.0;
        if (THR) {
            accept = leaf || (d2 >= q.size);       // q.size = the exact threshold: project.cu:634, 643 in one comparison
        } else {
            d = sqrt(d2) + 1e-15;                  // project.cu:634
            accept = leaf || (q.size / d < theta); // project.cu:643
        }
        bool self = false;
        if (leaf) {
            self = ((int64_t)k.occ == body);
            if (COMPAT) self = self || ((int64_t)k.occ + 2 == -body);   // project.cu:646
        }
        if (mine && accept && !self) {
            const double num = Gm * q.m;
            bool fast = false;
            if (THR && !ASM) {
                const bool safe = lane_safe && q.m <= kLaneHi && fabs(dx) >= kTiny && fabs(dy) >= kTiny && d2 <= kHuge;
                fast = __ballot(!safe) == 0ull;    // (of the lanes that take this node)
            }
            if (fast) {
                // sqrt(d2): y ~ 1/sqrt, g ~ sqrt, h ~ 1/(2 sqrt); one coupled step, two residual corrections
                const double y = __builtin_amdgcn_rsq(d2);
                double g = d2 * y, h = y * 0.5;
                const double r = __builtin_fma(-h, g, 0.5);
                g = __builtin_fma(g, r, g);
                double e = __builtin_fma(-g, g, d2);
                h = __builtin_fma(h, r, h);
                g = __builtin_fma(e, h, g);
                e = __builtin_fma(-g, g, d2);
                g = __builtin_fma(e, h, g);
                const double dd = g + 1e-15;       // project.cu:634
                // a / b = fma(a - b q, r, q) with q = a r and r = 1/b after two Newton steps
                auto recip = [](double b) {
                    double r0 = __builtin_amdgcn_rcp(b);
                    double t = __builtin_fma(-b, r0, 1.0);
                    r0 = __builtin_fma(r0, t, r0);
                    t = __builtin_fma(-b, r0, 1.0);
                    return __builtin_fma(r0, t, r0);
                };
                auto quot = [](double a, double b, double rb) {
                    const double q0 = a * rb;
                    const double rem = __builtin_fma(-b, q0, a);
                    return __builtin_fma(rem, rb, q0);
                };
                const double r2 = recip(d2), rd = recip(dd);
                const double f = quot(num, d2, r2);                           // project.cu:651
                const double ux = quot(dx, dd, rd), uy = quot(dy, dd, rd);    // project.cu:654-655
                fx += f * ux;
                fy += f * uy;
            } else {
                if (THR) d = sqrt(d2) + 1e-15;
                const double f = num / d2;             // project.cu:651
                const double ux = dx / d, uy = dy / d; // project.cu:654-655
                if (ASM) {
                    // (fx and fy are written by assembly only in this kernel -- here under the taking lanes' EXEC, as the
                    // compiler has it in this branch: with a C++ `+=` beside the block above it shuttles both sums
                    // between two register pairs on every visit, eight v_mov_b64)
                    asm volatile("v_add_f64 %[fx], %[fx], %[tx]\n\tv_add_f64 %[fy], %[fy], %[ty]"
                                 : [fx] "+v"(fx), [fy] "+v"(fy) : [tx] "v"(f * ux), [ty] "v"(f * uy));
                } else {
                    fx += f * ux;
                    fy += f * uy;
                }
            }
        }
        if (STATS) {
            n_vis += __popcll(live);
            ++n_wave;
            n_int += __popcll(__ballot(mine && accept && !self));
        }
        if (leaf) return 0;
        return __ballot(mine && !accept);
    };

    // The level being walked -- its quad, the next child to visit, the lanes that walk it -- is wave-uniform state in
    // scalar registers; the stack (one lane of four VGPRs per entry) holds the levels above it, written when a child is
    // opened and read when a level is exhausted.  A level whose last child is the one being opened is not stacked at all.
    auto push = [&](int at, int32_t quad, int32_t next, uint64_t mask) {
        v_quad = exact_writelane_i32(quad, at, v_quad);
        v_next = exact_writelane_i32(next, at, v_next);
        v_mlo = exact_writelane_i32((int32_t)(uint32_t)mask, at, v_mlo);
        v_mhi = exact_writelane_i32((int32_t)(uint32_t)(mask >> 32), at, v_mhi);
    };
    typedef int32_t v16i __attribute__((ext_vector_type(16)));
    typedef int32_t v8i __attribute__((ext_vector_type(8)));
    typedef int32_t v4i __attribute__((ext_vector_type(4)));
    typedef int32_t v2i __attribute__((ext_vector_type(2)));
    int sp = -1;
    int32_t cur_quad = 0, cur_c = -1;
    uint64_t cur_live = 0;
    {
        v8i qa; v2i ka;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\t"
                     "s_load_dwordx2 %1, %3, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(qa), "=&s"(ka) : "s"(gd), "s"(ld) : "memory");
        int32_t child;
        const uint64_t open = eval(NodeD{__hiloint2double(qa[1], qa[0]), __hiloint2double(qa[3], qa[2]),
                                         __hiloint2double(qa[5], qa[4]), __hiloint2double(qa[7], qa[6])},
                                   LinkD{ka[0], ka[1]}, __ballot(valid), child);
        if (open != 0 && child >= 0) { cur_quad = child; cur_c = 3; cur_live = open; }
    }
    // Children 3,2 and 1,0 of a quad are the two halves of one 128-byte line (bh_engine.hip allocates the node arrays so):
    // a pair comes in ONE scalar request, and its second child -- visited right after the first unless that one was
    // opened -- waits for nothing.  A level resumed after a descent reads its pair again.
    v16i qa = {};
    v4i ka = {};
    bool have_pair = false;
    bool more = cur_c >= 0;
    while (more) {
        if (!have_pair) {
            const int32_t base = cur_quad + (cur_c & ~1);
            const NodeD *pq = gd + base;
            const LinkD *pk = ld + base;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\t"
                         "s_load_dwordx4 %1, %3, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(qa), "=&s"(ka) : "s"(pq), "s"(pk) : "memory");
        }
        const int32_t parity = cur_c;                            // odd: the upper half of the pair
        have_pair = (cur_c & 1) != 0;
        --cur_c;
        // (scalar selects, spelled out: the compiler turns a select between halves of the register tuple into indexed moves
        // through vector registers)
        NodeD q;
        LinkD k;
        asm("s_bitcmp1_b32 %[c], 0\n\t"
            "s_cselect_b64 %[cx], %[h0], %[l0]\n\t"
            "s_cselect_b64 %[cy], %[h1], %[l1]\n\t"
            "s_cselect_b64 %[m], %[h2], %[l2]\n\t"
            "s_cselect_b64 %[sz], %[h3], %[l3]\n\t"
            "s_cselect_b32 %[ch], %[hc], %[lc]\n\t"
            "s_cselect_b32 %[oc], %[ho], %[lo]"
            : [cx] "=&s"(q.cx), [cy] "=&s"(q.cy), [m] "=&s"(q.m), [sz] "=&s"(q.size), [ch] "=&s"(k.child), [oc] "=&s"(k.occ)
            : [c] "s"(parity), [h0] "s"(__hiloint2double(qa[9], qa[8])), [h1] "s"(__hiloint2double(qa[11], qa[10])),
              [h2] "s"(__hiloint2double(qa[13], qa[12])), [h3] "s"(__hiloint2double(qa[15], qa[14])),
              [l0] "s"(__hiloint2double(qa[1], qa[0])), [l1] "s"(__hiloint2double(qa[3], qa[2])),
              [l2] "s"(__hiloint2double(qa[5], qa[4])), [l3] "s"(__hiloint2double(qa[7], qa[6])),
              [hc] "s"(ka[2]), [ho] "s"(ka[3]), [lc] "s"(ka[0]), [lo] "s"(ka[1])
            : "scc");
        int32_t child;
        const uint64_t open = eval(q, k, cur_live, child);
        if (open != 0 && child >= 0 && (cur_c < 0 || sp + 1 < kExactLevels)) {      // (the bound cannot bind: max_depth <= 32)
            if (cur_c >= 0) {
                ++sp;
                push(sp, cur_quad, cur_c, cur_live);
            }
            cur_quad = child; cur_c = 3; cur_live = open;
            have_pair = false;
        } else if (cur_c < 0) {
            // level exhausted: back to the one above it
            if (sp < 0) more = false;
            else {
                cur_quad = __builtin_amdgcn_readlane(v_quad, sp);
                cur_c = __builtin_amdgcn_readlane(v_next, sp);
                cur_live = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_mhi, sp) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane(v_mlo, sp);
                --sp;
                have_pair = false;
            }
        }
    }

    if (ASM) {
        // (the epilogue stores {fx, fy} as one 16-byte tuple; left to itself the register allocator keeps that tuple through
        // the loop and moves the sums in and out of it around every assembly block -- eight v_mov_b64 per visited node.  One
        // explicit copy here ends the loop's registers' life)
        double ox, oy;
        asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(ox), "=&v"(oy) : "v"(fx), "v"(fy));
        fx = ox; fy = oy;
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
