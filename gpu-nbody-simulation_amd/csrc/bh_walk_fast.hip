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
//     dist (project.cu:634) is below fp32 resolution and omitted.
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
__device__ __forceinline__ NodeF load_node(const NodeF BH_CONSTANT *p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const v8i r = *(const v8i BH_CONSTANT *)p;
#pragma clang diagnostic pop
    NodeF q;
    q.cx = __int_as_float(r[0]); q.cy = __int_as_float(r[1]);
    q.m = __int_as_float(r[2]);  q.thr = __int_as_float(r[3]);
    q.child = r[4]; q.occ = r[5]; q.first = r[6]; q.count = r[7];
    return q;
}

constexpr int kLdsStackDepth = 128;   // 3*31+4 entries worst case

template <bool LDS_STACK, bool STATS, bool BUCKETS>
__global__ __launch_bounds__(kBlock) void walk_fast_kernel(WalkFastArgs a)
{
    __shared__ int32_t s_base[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];
    __shared__ uint64_t s_mask[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];

    if (a.ctr->overflow) return;
    const int lane = lane_id(), w = wave_id();
    const int64_t s = a.lo + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = s < a.hi;
    const float2 p = valid ? a.spos[s] : float2{0.f, 0.f};
    const int32_t self = valid ? (int32_t)s : -2;
    float ax = 0.f, ay = 0.f;
    unsigned long long n_vis = 0, n_int = 0;

    const NodeF BH_CONSTANT *nodes = as_constant(a.nodes);

    // register-lane stack
    int32_t v_base = 0, v_lo = 0, v_hi = 0;
    int sp = 0;

    auto push = [&](int32_t base, uint64_t mask) {
        if (LDS_STACK) {
            if (sp < kLdsStackDepth) {
                if (lane == 0) { s_base[w][sp] = base; s_mask[w][sp] = mask; }
                ++sp;
            }
        } else {
            if (sp < kWave) {
                v_base = bh_writelane_i32(base, sp, v_base);
                v_lo = bh_writelane_i32((int32_t)(uint32_t)mask, sp, v_lo);
                v_hi = bh_writelane_i32((int32_t)(uint32_t)(mask >> 32), sp, v_hi);
                ++sp;
            }
        }
    };

    auto eval = [&](const NodeF q, uint64_t mask) {
        if (!(q.m > 1e-15f)) return;                        // empty cell, project.cu:617
        const bool live = (mask >> lane) & 1ull;
        if (BUCKETS && q.count > 1 && q.child < 0) {
            // depth-cap cell holding several bodies: direct sum over its members, self excluded
            for (int32_t j = q.first; j < q.first + q.count; ++j) {
                const float2 o = a.spos[j];
                const float om = a.smass[j];
                const float dx = o.x - p.x, dy = o.y - p.y;
                const float d2 = dx * dx + dy * dy;
                const float ri = __builtin_amdgcn_rsqf(d2);
                const bool ok = live && (j != self);
                const float wgt = ok ? om * ri * ri * ri : 0.f;
                ax = fmaf(wgt, dx, ax);
                ay = fmaf(wgt, dy, ay);
            }
            if (STATS) { n_vis += __popcll(mask); n_int += (unsigned long long)__popcll(mask) * (q.count); }
            return;
        }
        const float dx = q.cx - p.x, dy = q.cy - p.y;
        const float d2 = fmaf(dx, dx, dy * dy);
        const bool far = d2 > q.thr;                        // leaves carry thr = -1
        const bool acc = live && far && (q.occ != self);    // self skip, project.cu:646
        const float ri = __builtin_amdgcn_rsqf(d2);
        const float wgt = acc ? q.m * ri * ri * ri : 0.f;
        ax = fmaf(wgt, dx, ax);
        ay = fmaf(wgt, dy, ay);
        if (STATS) { n_vis += __popcll(mask); n_int += __popcll(__ballot(acc)); }
        if (q.child >= 0) {
            const uint64_t open = __ballot(live && !far);
            if (open) push(q.child, open);
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
        const NodeF q0 = load_node(nodes + base + 0);
        const NodeF q1 = load_node(nodes + base + 1);
        const NodeF q2 = load_node(nodes + base + 2);
        const NodeF q3 = load_node(nodes + base + 3);
        eval(q0, mask);
        eval(q1, mask);
        eval(q2, mask);
        eval(q3, mask);
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
    }
}

template <bool L, bool S, bool B>
static hipError_t launch(const WalkFastArgs &a, hipStream_t st)
{
    const int64_t cnt = a.hi - a.lo;
    if (cnt <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((cnt + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((walk_fast_kernel<L, S, B>), dim3(grid), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_walk_fast(const WalkFastArgs &a, bool lds_stack, bool stats, bool buckets,
                            hipStream_t st)
{
    const int key = (lds_stack ? 4 : 0) | (stats ? 2 : 0) | (buckets ? 1 : 0);
    switch (key) {
    case 0: return launch<false, false, false>(a, st);
    case 1: return launch<false, false, true>(a, st);
    case 2: return launch<false, true, false>(a, st);
    case 3: return launch<false, true, true>(a, st);
    case 4: return launch<true, false, false>(a, st);
    case 5: return launch<true, false, true>(a, st);
    case 6: return launch<true, true, false>(a, st);
    default: return launch<true, true, true>(a, st);
    }
}

}  // namespace bh
