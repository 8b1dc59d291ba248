// bh_let.hpp -- locally-essential-tree (LET) extraction for the distributed step.
// The reference is single-GPU; this is new design.
//
// Each rank owns a set of bodies -- a contiguous range of the Hilbert order of the initial positions
// (distributed.partition_hilbert), so its 64-body wave groups are the groups a single GPU would form
// -- and builds its LOCAL tree with the usual pipeline, but under the GLOBAL root box, so its cells
// are cells of the global grid.  The force on a body is the sum of its walks over the W local trees
// (a forest); per-body MAC as everywhere else.  For the remote trees a rank does not need whole
// trees, only the part its bodies can open.  A rank describes where its bodies are by kLetBoxes
// bounding boxes (of consecutive slices of its bodies: a Hilbert range is a compact but not a
// rectangular region, one box would overlap the neighbours'); then
//
//   a node of rank r's tree can be opened by SOME body of peer q only if
//        dist^2(node COM, some box of q) <= (size/theta)^2 = thr
//   (every body of q is at least that far away; a body opens a node iff d^2 <= thr).
//
// That test is local to the node, so the extraction is two passes over the quads with no traversal (round 3; rounds
// 1-2 ran five: mark, count, row scan, apply, pack -- ~35 us of launch chain per step at 131k bodies per rank):
//   let_mark_alloc: the thread of quad k decides, for each of its four nodes, which peers can open it -- that is the
//                need mask of the node's CHILD quad -- and hands every (peer, child quad) pair its slot in that
//                peer's LET: one counter per peer, one atomic per wave and peer (the lanes of a wave are ranked by
//                ballots).  Slot 0 is the root quad's.  Which slot a quad gets depends on the order the waves
//                arrive in; nothing else does -- a receiver follows child links, so its walk visits the same nodes
//                in the same order and the forces are bit-identical run to run (tests/test_gpu_let.py).
//   let_pack   : copy every needed quad into the peer's send block, child links rewritten to the
//                RECEIVER's index space (its forest array places the LET of sender r at
//                local_quads + r * let_cap), links to quads the peer cannot open cut (-1: the node
//                is then always accepted by that peer's bodies), buckets turned into aggregates; its first
//                workgroup also folds the counters into the running maxima / overflow flag and clears them.
// (A quad whose ancestor chain is cut is packed but unreachable: a few percent of waste, no
// traversal needed.)  Correctness does not depend on the domains being compact or disjoint --
// only the LET sizes do.
#pragma once

#include "bh_nodes.hpp"
#include "bh_prims.hpp"
#include "bh_bounds.hpp"

namespace bh {

constexpr int kMaxWorld = 64;
constexpr int kLetBoxes = 8;       // bounding boxes per rank
constexpr int kLetBoxParts = 16;   // first step: partial blocks per box
static_assert(kMaxWorld <= kWave, "let_box_kernel clears one counter per lane");

struct LetCounters {
    uint32_t count[kMaxWorld];   // quads packed for each peer: the LARGEST value of any build since the counters
                                 // were last read (bh_let_counts) or configured -- a check every N steps must
                                 // see an overflow or a near-overflow of ANY step in between, not of the last one
    uint32_t overflow;           // STICKY: some LET exceeded let_cap, or the local tree outgrew node_capacity
                                 // (its send blocks are then stale), in some build since the last read
    uint32_t pad[3];
};

// all_bounds: (world * kLetBoxes) x {xmin, xmax, ymin, ymax} raw (unpadded) bounds of slices of every
// rank's bodies.  One wave: global box with the reference's padding (project.cu:553-570).
__global__ __launch_bounds__(kWave) void let_box_kernel(const double *__restrict__ all_bounds, int nboxes,
                                                         double *__restrict__ box, TreeCounters *ctr, LetCounters *lc,
                                                         int Dm)
{
    if (blockIdx.x != 0) return;
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int r = threadIdx.x; r < nboxes; r += kWave) {
        const double a = all_bounds[4 * r], b = all_bounds[4 * r + 1], c = all_bounds[4 * r + 2], d = all_bounds[4 * r + 3];
        xlo = (a < xlo) ? a : xlo;  xhi = (xhi < b) ? b : xhi;
        ylo = (c < ylo) ? c : ylo;  yhi = (yhi < d) ? d : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    (void)lc;                                              // (sticky: cleared by bh_let_configure / bh_let_counts only)
    if (threadIdx.x != 0) return;
    const double ex = xhi - xlo, ey = yhi - ylo;
    const double span = (ex < ey) ? ey : ex;
    double pad = 0.1 * span;
    if (span == 0.0) pad = 1e-6;
    box[0] = xlo - pad; box[1] = xhi + pad; box[2] = ylo - pad; box[3] = yhi + pad;
    write_key_consts(box, Dm);
    ctr->n_internal = 0; ctr->overflow = 0;
    ctr->visits = 0; ctr->interactions = 0; ctr->wave_nodes = 0; ctr->wave_quads = 0;
}

// First step (no partials from a walk yet): min/max of kLetBoxes * kLetBoxParts CONTIGUOUS slices of
// the bodies in caller order (distributed.partition_hilbert hands them over in Hilbert order, so a
// slice is a compact region; any order is correct).
template <typename Real2>
__global__ __launch_bounds__(kBlock) void let_slice_bounds_kernel(const Real2 *__restrict__ pos, int64_t n,
                                                                   double *__restrict__ partial)
{
    __shared__ double sm[4][kWavesPerBlock];
    const int64_t nb = gridDim.x;
    const int64_t lo = n * blockIdx.x / nb, hi = n * (blockIdx.x + 1) / nb;
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const double x = (double)pos[i].x, y = (double)pos[i].y;
        xlo = (x < xlo) ? x : xlo;  xhi = (xhi < x) ? x : xhi;
        ylo = (y < ylo) ? y : ylo;  yhi = (yhi < y) ? y : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (lane_id() == 0) { sm[0][wave_id()] = xlo; sm[1][wave_id()] = xhi; sm[2][wave_id()] = ylo; sm[3][wave_id()] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            xlo = (sm[0][w] < xlo) ? sm[0][w] : xlo;  xhi = (xhi < sm[1][w]) ? sm[1][w] : xhi;
            ylo = (sm[2][w] < ylo) ? sm[2][w] : ylo;  yhi = (yhi < sm[3][w]) ? sm[3][w] : yhi;
        }
        double *o = partial + 4 * (size_t)blockIdx.x;
        o[0] = xlo; o[1] = xhi; o[2] = ylo; o[3] = yhi;
    }
}

// kLetBoxes boxes of this rank: box k = min/max over the k-th run of consecutive partials (partials
// come from the walk's epilogue -- one per group of sorted bodies, so a run is a stretch of the Hilbert
// order -- or from let_slice_bounds_kernel).  One wave per box.
__global__ __launch_bounds__(kWave) void let_local_bounds_kernel(const double *__restrict__ partial, int nb,
                                                                  double *__restrict__ lbounds)
{
    const int k = blockIdx.x;
    const int lo = (int)((int64_t)nb * k / kLetBoxes), hi = (int)((int64_t)nb * (k + 1) / kLetBoxes);
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += kWave) {
        const double a = partial[4 * i], b = partial[4 * i + 1], c = partial[4 * i + 2], d = partial[4 * i + 3];
        xlo = (a < xlo) ? a : xlo;  xhi = (xhi < b) ? b : xhi;
        ylo = (c < ylo) ? c : ylo;  yhi = (yhi < d) ? d : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (threadIdx.x == 0) {
        double *o = lbounds + 4 * k;                       // an empty run stays (+inf, -inf): matches nothing
        o[0] = xlo; o[1] = xhi; o[2] = ylo; o[3] = yhi;
    }
}

// ONE LANE PER NODE (four lanes per local quad k in [0, n_quads)): the need mask and the LET slots of the node's CHILD
// quad.  (One thread per quad walked its four nodes x W peers x 9 boxes in sequence: 18.6 us at 94k quads, the
// second-longest kernel of a rank's step at 135k bodies; the launch is far too small to fill the GPU, so its time is
// the length of one thread's chain.)  slots[p] counts the quads peer p gets BEYOND the root quad, which is slot 0;
// let_pack_kernel clears it.
__global__ __launch_bounds__(kBlock) void let_mark_alloc_kernel(const QuadF *__restrict__ qf,
                                                                 const double *__restrict__ all_bounds, int world,
                                                                 int rank, const TreeCounters *__restrict__ ctr,
                                                                 int64_t internal_cap, uint64_t *__restrict__ needmask,
                                                                 uint32_t *__restrict__ slots,
                                                                 uint32_t *__restrict__ outidx, int64_t outidx_stride)
{
    __shared__ float sbox[kMaxWorld * kLetBoxes][4];
    __shared__ float sall[kMaxWorld][4];                     // a peer's boxes taken together: a cheap first test
    for (int i = threadIdx.x; i < world * kLetBoxes; i += kBlock) {
        // outward-rounded float copies of the peers' boxes (the test must stay conservative)
        const double *b = all_bounds + 4 * i;
        sbox[i][0] = __double2float_rd(b[0]); sbox[i][1] = __double2float_ru(b[1]);
        sbox[i][2] = __double2float_rd(b[2]); sbox[i][3] = __double2float_ru(b[3]);
    }
    __syncthreads();
    if (threadIdx.x < world) {
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int b = 0; b < kLetBoxes; ++b) {
            const float *bx = sbox[threadIdx.x * kLetBoxes + b];
            x0 = fminf(x0, bx[0]); x1 = fmaxf(x1, bx[1]); y0 = fminf(y0, bx[2]); y1 = fmaxf(y1, bx[3]);
        }
        sall[threadIdx.x][0] = x0; sall[threadIdx.x][1] = x1; sall[threadIdx.x][2] = y0; sall[threadIdx.x][3] = y1;
    }
    __syncthreads();
    const uint32_t total = ctr->n_internal;
    if ((int64_t)total > internal_cap) return;             // (uniform; let_pack_kernel raises the flag)
    const int64_t node = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // node id = 4 * quad + slot
    const int64_t k = node >> 2;
    const int sl = (int)(node & 3);
    const bool live = k <= (int64_t)total;                 // quads 0..total
    const uint64_t everyone = (world >= 64 ? ~0ull : ((1ull << world) - 1)) & ~(1ull << rank);
    if (node == 0) {                                       // every peer gets the root, in slot 0
        needmask[0] = everyone;
        for (int p = 0; p < world; ++p) outidx[(int64_t)p * outidx_stride] = 0u;
    }
    int32_t child = -1;
    uint64_t mask = 0;
    if (live) {
        const QuadF *q = qf + k;
        child = q->child[sl];
        if (child >= 1) {                                  // (leaf, empty or bucket: no child quad)
            const float cx = q->xy[2 * sl], cy = q->xy[2 * sl + 1];
            const float thr = q->thr[sl] * 1.0001f;        // guard band for fp32 rounding of d^2
            for (int p = 0; p < world; ++p) {
                if (p == rank) continue;
                {
                    const float dx = fmaxf(fmaxf(sall[p][0] - cx, cx - sall[p][1]), 0.f);
                    const float dy = fmaxf(fmaxf(sall[p][2] - cy, cy - sall[p][3]), 0.f);
                    if (!(dx * dx + dy * dy <= thr)) continue; // out of reach of everything peer p holds
                }
                bool near = false;
#pragma unroll
                for (int b = 0; b < kLetBoxes; ++b) {
                    const float *bx = sbox[p * kLetBoxes + b];
                    const float dx = fmaxf(fmaxf(bx[0] - cx, cx - bx[1]), 0.f);
                    const float dy = fmaxf(fmaxf(bx[2] - cy, cy - bx[3]), 0.f);
                    near = near || (dx * dx + dy * dy <= thr); // an empty box (inf) gives inf: no
                }
                if (near) mask |= 1ull << p;
            }
            needmask[child] = mask;
        } else child = -1;
    }
    // slots: per peer, the wave's nodes are ranked by lane and ONE lane draws the wave's share
    const uint64_t lt = (1ull << lane_id()) - 1ull;
    for (int p = 0; p < world; ++p) {
        const uint64_t b = __ballot((mask >> p) & 1ull);
        if (b == 0) continue;                              // (uniform)
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(&slots[p], (uint32_t)__popcll(b));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)base) + 1u;       // slot 0 is the root's
        if ((mask >> p) & 1ull) outidx[(int64_t)p * outidx_stride + child] = base + (uint32_t)__popcll(b & lt);
    }
}

// thread per quad: copy it into the send block of every peer that needs it
__global__ __launch_bounds__(kBlock) void let_pack_kernel(const QuadF *__restrict__ qf,
                                                           const uint64_t *__restrict__ needmask,
                                                           const uint32_t *__restrict__ outidx,
                                                           int64_t outidx_stride, int world, int rank,
                                                           const TreeCounters *__restrict__ ctr,
                                                           int64_t internal_cap, QuadF *__restrict__ send,
                                                           uint32_t let_cap, int64_t recv_base,
                                                           uint32_t *__restrict__ slots, LetCounters *lc)
{
    const uint32_t total = ctr->n_internal;
    if (blockIdx.x == 0 && (int)threadIdx.x < world) {
        // this build's LET sizes -> running maxima and the sticky overflow flag; counters cleared for the next build
        const uint32_t size = ((int)threadIdx.x == rank) ? 0u : 1u + slots[threadIdx.x];
        slots[threadIdx.x] = 0u;
        if (size > lc->count[threadIdx.x]) lc->count[threadIdx.x] = size;      // (one writer per peer)
        if (size > let_cap) lc->overflow = 1;
        // a local tree that outgrew node_capacity leaves the send blocks untouched (both kernels return early):
        // the peers would walk the previous step's LET.  Report it where they look.
        if (ctr->overflow || (int64_t)total > internal_cap) lc->overflow = 1;
    }
    if ((int64_t)total > internal_cap) return;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k > (int64_t)total) return;
    const uint64_t mask = needmask[k];
    if (mask == 0) return;
    const QuadF q = qf[k];
    uint64_t cmask[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cmask[s] = (q.child[s] >= 1) ? needmask[q.child[s]] : 0ull;
    for (int p = 0; p < world; ++p) {
        if (!((mask >> p) & 1ull)) continue;
        const uint32_t o = outidx[(int64_t)p * outidx_stride + k];
        if (o >= let_cap) continue;                       // (overflow: flagged above)
        QuadF out = q;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int32_t c = q.child[s];
            if (c >= 1) {
                // child quad needed by p -> its slot in p's forest; otherwise cut: p's bodies all
                // accept this node (d^2 > thr for every point of p's box)
                // (a slot past let_cap -- overflow, flagged above -- is cut too, so that a
                // receiver that walks before the host has seen the flag never leaves its block)
                int32_t link = -1;
                if ((cmask[s] >> p) & 1ull) {
                    const uint32_t co = outidx[(int64_t)p * outidx_stride + c];
                    if (co < let_cap) link = (int32_t)(recv_base + (int64_t)co);
                }
                out.child[s] = link;
            } else if (c <= -2) {
                out.child[s] = -1;                        // bucket -> aggregate for remote bodies
                out.thr[s] = 0.f;
            }
        }
        send[(int64_t)p * let_cap + o] = out;
    }
}

}  // namespace bh
