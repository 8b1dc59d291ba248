// bh_walk_fast.hip -- fp32 theta-walk + integrator, the throughput path (BASELINE configs
// "fp32").  Replaces computeForcesGpu (project.cu:679-793) and updateAccVelPos
// (project.cu:819-836); designed for CDNA4 wave64, not translated from them.
//
//   * one wavefront = 64 Morton-adjacent bodies, one per lane; the traversal state is
//     wave-uniform, so node reads are scalar loads (s_load_dwordx8 through the scalar data cache)
//     broadcast to all lanes for free -- the reference's per-thread walk re-reads each 96-byte
//     node once per body (project.cu:726), this reads a 32-byte record once per wave;
//   * the four children of a subdivided cell are contiguous (one 128-byte line) and are evaluated
//     together: one dependent memory round trip per opened cell instead of one per node;
//   * a stack entry is {first child, 64-bit lane mask of the bodies that opened the parent}.  The
//     default stack lives in three VGPRs addressed by lane (v_writelane/v_readlane): entry k sits
//     in lane k, so push/pop are single VALU instructions with no LDS round trip.  The LDS
//     variant (BH_FLAG_LDS_STACK, and automatically for max_depth > 21 where 64 entries do not
//     suffice) keeps the same entries in LDS; DESIGN.md quotes the measured difference.
//   * MAC per body exactly as the reference's (size/dist < theta, evaluated per lane), in the
//     algebraically equal form d2 > (size/theta)^2 with the right side precomputed per node.
//   * force per accepted node: G*M*d/(|d|^3) through v_rsq_f32; the reference's 1e-15 offset on
//     dist (project.cu:634) is below fp32 resolution and omitted; a node at distance exactly 0
//     (the body itself, or an exactly coincident body) contributes nothing.
//   * epilogue: a = G*sum, v += a*dt, p += v*dt written back in caller order (scatter through
//     perm), or into the sorted arrays for the multi-GPU exchange.
#include "bh_prims.hpp"
#include "bh_nodes.hpp"
#include "bh_walk_fast.h"

namespace bh {

#define BH_CONSTANT __attribute__((address_space(4)))

// clang 22 exposes v_readlane_b32 as a builtin but not v_writelane_b32; bind the LLVM intrinsic.
extern "C" __device__ int bh_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

template <typename T>
__device__ __forceinline__ const T BH_CONSTANT *as_constant(const T *p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const T BH_CONSTANT *)p;
#pragma clang diagnostic pop
}

// one 32-byte node through the scalar data cache (s_load_dwordx8): the address is wave-uniform
typedef int32_t v8i __attribute__((ext_vector_type(8)));
typedef int32_t v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ NodeF load_node(const NodeF BH_CONSTANT *p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const v8i r = *(const v8i BH_CONSTANT *)p;
#pragma clang diagnostic pop
    NodeF q;
    q.cx = __int_as_float(r[0]); q.cy = __int_as_float(r[1]);
    q.m = __int_as_float(r[2]);  q.thr = __int_as_float(r[3]);
    q.child = r[4]; q.first = r[5]; q.count = r[6]; q.pad = r[7];
    return q;
}

typedef int32_t v32i __attribute__((ext_vector_type(32)));
__device__ __forceinline__ void load_quad(const NodeF BH_CONSTANT *p, NodeF (&q)[4])
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const v32i r = *(const v32i BH_CONSTANT *)p;
#pragma clang diagnostic pop
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        q[k].cx = __int_as_float(r[8 * k + 0]); q[k].cy = __int_as_float(r[8 * k + 1]);
        q[k].m = __int_as_float(r[8 * k + 2]);  q[k].thr = __int_as_float(r[8 * k + 3]);
        q[k].child = r[8 * k + 4]; q[k].first = r[8 * k + 5]; q[k].count = r[8 * k + 6];
        q[k].pad = r[8 * k + 7];
    }
}

constexpr int kLdsStackDepth = 128;   // 3*31+4 entries worst case

// The loop body is written so that hipcc emits straight-line code per child: the push is an
// unconditional write of {child, open mask} into the slot above the top followed by
// `sp += (open != 0)`, so there is no control-flow merge (and none of the register copies it
// costs) around the stack registers; lane masks stay in SGPR pairs (inverse_ballot / ballot);
// the only branch per child is the integer "cell is empty" test.
template <bool LDS_STACK, bool STATS>
__global__ __launch_bounds__(kBlock) void walk_fast_kernel(WalkFastArgs a)
{
    __shared__ int32_t s_base[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];
    __shared__ uint64_t s_mask[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];

    if (a.ctr->overflow) return;
    const int lane = lane_id(), w = wave_id();
    const int64_t s = a.lo + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = s < a.hi;
    const float2 p = valid ? a.spos[s] : float2{0.f, 0.f};
    float ax = 0.f, ay = 0.f;
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0;

    const NodeF BH_CONSTANT *nodes = as_constant(a.nodes);
    const float2 BH_CONSTANT *cpos = as_constant(a.spos);
    const float BH_CONSTANT *cmass = as_constant(a.smass);

    int32_t v_base = 0, v_lo = 0, v_hi = 0;      // register-lane stack: entry k lives in lane k
    int sp = 0;                                   // wave-uniform

    auto eval = [&](const NodeF q, const uint64_t mask) {
        if (q.count == 0) return;                           // empty cell (project.cu:617)
        const float dx = q.cx - p.x, dy = q.cy - p.y;
        const float d2 = fmaf(dx, dx, dy * dy);
        // lane masks as 64-bit scalars: a v_cmp result IS its ballot, so the algebra below is SALU
        const uint64_t farm = __ballot(d2 > q.thr);         // leaves: thr = -1; buckets: +inf
        // d2 > 0 is the self skip (project.cu:646): a single-body leaf carries the body's own
        // position.  It also drops an exactly coincident second body, where the reference divides
        // by zero (inf*0 -> NaN, project.cu:651-658): fp32 positions are quantised, so that case
        // is reachable here and one NaN would poison the root box of every later step.
        const uint64_t accm = mask & farm & __ballot(d2 > 0.f);
        const uint64_t open = mask & ~farm;                 // never set for leaves
        const float ri = __builtin_amdgcn_rsqf(d2);
        const float wgt = __builtin_amdgcn_inverse_ballot_w64(accm) ? q.m * ri * ri * ri : 0.f;
        ax = fmaf(wgt, dx, ax);
        ay = fmaf(wgt, dy, ay);
        if (STATS) { n_vis += __popcll(mask); ++n_wave; n_int += __popcll(accm); }
        if (LDS_STACK) {
            if (lane == 0) { s_base[w][sp] = q.child; s_mask[w][sp] = open; }
        } else {
            v_base = bh_writelane_i32(q.child, sp, v_base);
            v_lo = bh_writelane_i32((int32_t)(uint32_t)open, sp, v_lo);
            v_hi = bh_writelane_i32((int32_t)(uint32_t)(open >> 32), sp, v_hi);
        }
        sp += (open != 0) ? 1 : 0;
    };

    // depth-cap cell holding several bodies (compat off): summed body by body for the lanes that
    // reached it; self and exactly coincident bodies contribute nothing (d2 == 0)
    auto bucket = [&](const NodeF q, const uint64_t mask) {
        for (int32_t j = q.first; j < q.first + q.count; ++j) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            const v2i ob = *(const v2i BH_CONSTANT *)(cpos + j);      // scalar loads: j is uniform
            const float om = *(const float BH_CONSTANT *)(cmass + j);
#pragma clang diagnostic pop
            const float dx = __int_as_float(ob[0]) - p.x, dy = __int_as_float(ob[1]) - p.y;
            const float d2 = fmaf(dx, dx, dy * dy);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const uint64_t okm = mask & __ballot(d2 > 0.f);
            const float wgt = __builtin_amdgcn_inverse_ballot_w64(okm) ? om * ri * ri * ri : 0.f;
            ax = fmaf(wgt, dx, ax);
            ay = fmaf(wgt, dy, ay);
            if (STATS) n_int += __popcll(okm);
        }
    };

    eval(load_node(nodes), __ballot(valid));

    while (sp > 0) {
        --sp;
        int32_t base;
        uint64_t mask;
        if (LDS_STACK) {
            base = __builtin_amdgcn_readfirstlane(s_base[w][sp]);
            const uint64_t m = s_mask[w][sp];
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(m >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)m);
        } else {
            base = __builtin_amdgcn_readlane(v_base, sp);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi, sp) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(v_lo, sp);
        }
        if (base < 0) {                                     // bucket reference: -(node id) - 2
            bucket(load_node(nodes + (-base - 2)), mask);
            continue;
        }
        // the sibling quad is one 128-byte line: both s_load_dwordx16 are issued before any use
        NodeF q[4];
        load_quad(nodes + base, q);
        eval(q[0], mask);
        eval(q[1], mask);
        eval(q[2], mask);
        eval(q[3], mask);
    }

    if (valid) {
        const float gx = a.G * ax, gy = a.G * ay;
        const uint32_t body = a.perm[s];
        if (a.acc_out) a.acc_out[body] = float2{gx, gy};
        if (a.integrate) {
            float2 v = a.vel[body];
            v.x = fmaf(gx, a.dt, v.x);
            v.y = fmaf(gy, a.dt, v.y);
            float2 np{fmaf(v.x, a.dt, p.x), fmaf(v.y, a.dt, p.y)};
            if (a.to_sorted) {
                a.svel[s] = v;
                a.spos_out[s] = np;
            } else {
                a.vel[body] = v;
                a.pos[body] = np;
            }
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&a.ctr->visits, n_vis);
        atomicAdd(&a.ctr->interactions, n_int);
        atomicAdd(&a.ctr->wave_nodes, n_wave);
    }
}

template <bool L, bool S>
static hipError_t launch(const WalkFastArgs &a, hipStream_t st)
{
    const int64_t cnt = a.hi - a.lo;
    if (cnt <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((cnt + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((walk_fast_kernel<L, S>), dim3(grid), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_walk_fast(const WalkFastArgs &a, bool lds_stack, bool stats, hipStream_t st)
{
    if (lds_stack) return stats ? launch<true, true>(a, st) : launch<true, false>(a, st);
    return stats ? launch<false, true>(a, st) : launch<false, false>(a, st);
}

}  // namespace bh
