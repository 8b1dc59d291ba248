// bh_tree.hpp -- device construction of the reference's quadtree.
//
// The reference inserts bodies one by one on the host (QuadInsert, project.cu:358-453).  The
// resulting tree does not depend on insertion order: a cell is subdivided iff it holds >= 2
// bodies and its depth is below the cap, all four children are allocated together
// (project.cu:410-434), and child selection uses recursively halved fp64 midpoints
// (project.cu:349-355, 417-418).  So the same tree can be derived from sorted Morton-style keys
// whose digits are produced by the SAME fp64 bisection:
//
//   keys      : per body, max_depth-1 bisection steps -> 2 bits per level (child index 0..3)
//   sort      : stable radix sort of (key, body) -> bodies of one cell are contiguous and, inside
//               a depth-cap cell, stay in body order (the order the reference's running
//               centre-of-mass fold uses, project.cu:371-373)
//   pairs     : L[i] = levels shared by sorted neighbours i, i+1.  The subdivided cell at depth d
//               whose first body is i exists iff L[i-1] < d <= min(L[i], Dm-1); pair i "owns"
//               those cells.  An exclusive scan of the counts numbers all subdivided cells in
//               DFS pre-order ("rank").
//   nodes     : the owner of rank r writes the four children of that cell at node ids
//               1+4r .. 1+4r+3 (root = node 0): bounds by halving, leaf payloads, child links.
//   com       : exact mode: bottom-up, level by level, children summed in index order exactly as
//               ComputeMass does (project.cu:473-502).  fp32 mode: from fp64 prefix sums.
//
// Depth convention: d = 0 is the root ("file depth", TraverseTreeToFile's first column);
// Dm = max_depth-1 is the depth of the cap cells.  Everything here is compiled with
// -ffp-contract=off.
#pragma once

#include "bh_prims.hpp"
#include "bh_nodes.hpp"

namespace bh {

__device__ __forceinline__ int shared_levels(uint64_t a, uint64_t b, int Dm)
{
    const uint64_t x = a ^ b;
    if (x == 0) return Dm;
    return (__clzll((long long)x) - (64 - 2 * Dm)) >> 1;
}

// ---- root box: ComputeRootBounds, project.cu:536-573 ------------------------------------------
template <typename Real2>
__global__ __launch_bounds__(kBlock) void bounds_partial(const Real2 *__restrict__ pos, int64_t n,
                                                          double *__restrict__ partial)
{
    __shared__ double sm[4][kWavesPerBlock];
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
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
        partial[4 * blockIdx.x + 0] = xlo; partial[4 * blockIdx.x + 1] = xhi;
        partial[4 * blockIdx.x + 2] = ylo; partial[4 * blockIdx.x + 3] = yhi;
    }
}

__global__ __launch_bounds__(kBlock) void bounds_final(const double *__restrict__ partial, int nb,
                                                        double *__restrict__ box)
{
    __shared__ double sm[4][kWavesPerBlock];
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += kBlock) {
        const double a = partial[4 * i], b = partial[4 * i + 1], c = partial[4 * i + 2], d = partial[4 * i + 3];
        xlo = (a < xlo) ? a : xlo;  xhi = (xhi < b) ? b : xhi;
        ylo = (c < ylo) ? c : ylo;  yhi = (yhi < d) ? d : yhi;
    }
    xlo = wave_min(xlo); xhi = wave_max(xhi); ylo = wave_min(ylo); yhi = wave_max(yhi);
    if (lane_id() == 0) { sm[0][wave_id()] = xlo; sm[1][wave_id()] = xhi; sm[2][wave_id()] = ylo; sm[3][wave_id()] = yhi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            xlo = (sm[0][w] < xlo) ? sm[0][w] : xlo;  xhi = (xhi < sm[1][w]) ? sm[1][w] : xhi;
            ylo = (sm[2][w] < ylo) ? sm[2][w] : ylo;  yhi = (yhi < sm[3][w]) ? sm[3][w] : yhi;
        }
        const double ex = xhi - xlo, ey = yhi - ylo;
        const double span = (ex < ey) ? ey : ex;
        double pad = 0.1 * span;
        if (span == 0.0) pad = 1e-6;
        box[0] = xlo - pad; box[1] = xhi + pad; box[2] = ylo - pad; box[3] = yhi + pad;
    }
}

// ---- keys: DetermineChild (project.cu:348-356) applied max_depth-1 times -----------------------
__device__ __forceinline__ int pick_child(double x, double y, double mx, double my)
{
    if (x <  mx && y <  my) return 0;
    if (x >= mx && y <  my) return 1;
    if (x <  mx && y >= my) return 2;
    return 3;
}

__device__ __forceinline__ void descend(int c, double mx, double my, double &x0, double &x1,
                                        double &y0, double &y1)
{
    if (c & 1) x0 = mx; else x1 = mx;
    if (c & 2) y0 = my; else y1 = my;
}

template <typename Real2>
__global__ __launch_bounds__(kBlock) void keys_kernel(const Real2 *__restrict__ pos,
                                                       const double *__restrict__ box,
                                                       uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ idx, int64_t n, int Dm)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = (double)pos[i].x, y = (double)pos[i].y;
    double x0 = box[0], x1 = box[1], y0 = box[2], y1 = box[3];
    uint64_t k = 0;
    for (int l = 0; l < Dm; ++l) {
        const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
        const int c = pick_child(x, y, mx, my);
        k = (k << 2) | (uint64_t)c;
        descend(c, mx, my, x0, x1, y0, y1);
    }
    keys[i] = k;
    idx[i] = (uint32_t)i;
}

// ---- pairs: how many subdivided cells does sorted body i start? --------------------------------
__global__ __launch_bounds__(kBlock) void pairs_kernel(const uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ cnt, int64_t n, int Dm)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
    if (i + 1 < n) {
        const uint64_t k = keys[i];
        const int L = shared_levels(k, keys[i + 1], Dm);
        const int Lp = (i == 0) ? -1 : shared_levels(keys[i - 1], k, Dm);
        const int hi = (L < Dm - 1) ? L : Dm - 1;
        c = (hi > Lp) ? (uint32_t)(hi - Lp) : 0u;
    }
    cnt[i] = c;
}

// first index j in [lo, hi) with (keys[j] >> sh) >= target   (keys sorted)
__device__ __forceinline__ int64_t lower_bound_prefix(const uint64_t *__restrict__ keys, int64_t lo,
                                                      int64_t hi, int sh, uint64_t target)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> sh) < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- root record when nothing is subdivided (n <= 1, or max_depth == 1) ------------------------
template <bool EXACT, bool COMPAT, typename Real2, typename Real>
__global__ void root_only_kernel(const Real2 *__restrict__ pos, const Real *__restrict__ mass,
                                 const uint32_t *__restrict__ perm, const double *__restrict__ box,
                                 int64_t n, int Dm, double theta, NodeD *gd, LinkD *ld, NodeF *nf,
                                 const TreeCounters *ctr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ctr->n_internal != 0) return;
    double m = 0.0, cx = 0.0, cy = 0.0;
    int occ = -1;
    if (n >= 1 && Dm == 0) {              // root itself is a depth-cap cell, project.cu:360-382
        for (int64_t j = 0; j < n; ++j) {
            const uint32_t b = perm[j];
            const double bm = (double)mass[b], bx = (double)pos[b].x, by = (double)pos[b].y;
            cx = (m * cx + bm * bx) / (m + bm);
            cy = (m * cy + bm * by) / (m + bm);
            m += bm;
        }
        occ = (n == 1) ? (EXACT ? -(int)perm[0] - 2 : 0) : -1;
    } else if (n == 1) {                  // empty root takes the body, project.cu:398-406
        const uint32_t b = perm[0];
        m = (double)mass[b]; cx = (double)pos[b].x; cy = (double)pos[b].y;
        occ = EXACT ? (int)b : 0;
    }
    const double ex = box[1] - box[0], ey = box[3] - box[2];
    const double size = (ex > ey) ? ex : ey;
    if (EXACT) {
        gd[0] = NodeD{cx, cy, m, size};
        ld[0] = LinkD{-1, occ};
    } else {
        NodeF r;
        r.cx = (float)cx; r.cy = (float)cy; r.m = (float)m; r.thr = -1.0f;
        r.child = -1; r.first = 0; r.count = (m > 1e-15) ? (int32_t)n : 0; r.pad = 0;
        if (n > 1 && !COMPAT) { r.child = -2; r.thr = INFINITY; }     // root itself is a bucket
        nf[0] = r;
    }
}

// ---- nodes: the owner of each subdivided cell writes its four children --------------------------
// EXACT: NodeD/LinkD + self_node/cell_depth for the bottom-up pass.
// !EXACT: NodeF complete (COM from the fp64 prefix sums psum[0..n], psum[j] = sum over sorted < j).
template <bool EXACT, bool COMPAT, typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void nodes_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm,
    const uint32_t *__restrict__ off, const Real2 *__restrict__ pos, const Real *__restrict__ mass,
    const double *__restrict__ box, const d3 *__restrict__ psum, int64_t n, int Dm, double theta,
    int64_t internal_cap, NodeD *__restrict__ gd, LinkD *__restrict__ ld, NodeF *__restrict__ nf,
    int32_t *__restrict__ self_node, int32_t *__restrict__ cell_depth, TreeCounters *ctr)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i + 1 >= n) return;
    const uint64_t key = keys[i];
    const int L = shared_levels(key, keys[i + 1], Dm);
    const int Lp = (i == 0) ? -1 : shared_levels(keys[i - 1], key, Dm);
    const int dlo = Lp + 1;
    const int dhi = (L < Dm - 1) ? L : Dm - 1;
    if (dhi < dlo) return;
    const uint32_t total = ctr->n_internal;
    if ((int64_t)total > internal_cap) {
        if (i == 0) ctr->overflow = 1;
        return;
    }
    const uint32_t r0 = off[i];

    double x0 = box[0], x1 = box[1], y0 = box[2], y1 = box[3];
    for (int l = 0; l < dlo; ++l) {
        const int c = (int)((key >> (2 * (Dm - 1 - l))) & 3);
        descend(c, (x0 + x1) / 2, (y0 + y1) / 2, x0, x1, y0, y1);
    }
    const double inv_theta = 1.0 / theta;

    int64_t hi = n;
    for (int d = dlo; d <= dhi; ++d) {
        const uint32_t r = r0 + (uint32_t)(d - dlo);
        const int sh = 2 * (Dm - d);                 // bits below the depth-d prefix
        const uint64_t pfx = (d == 0) ? 0ull : (key >> sh);
        const int64_t e = (d == 0) ? n : lower_bound_prefix(keys, i + 1, hi, sh, pfx + 1);
        hi = e;
        const int shc = sh - 2;
        int64_t b[5];
        b[0] = i; b[4] = e;
        for (int c = 1; c < 4; ++c) b[c] = lower_bound_prefix(keys, b[c - 1], e, shc, (pfx << 2) | (uint64_t)c);

        const double mx = (x0 + x1) / 2.0, my = (y0 + y1) / 2.0;
        const int32_t quad = 1 + 4 * (int32_t)r;

        if (d == 0) {                                  // root record (static part)
            const double ex = x1 - x0, ey = y1 - y0;
            const double size = (ex > ey) ? ex : ey;
            if (EXACT) {
                gd[0].size = size;
                ld[0] = LinkD{quad, -1};
                self_node[0] = 0;
                cell_depth[0] = 0;
            } else {
                NodeF rt;
                const d3 t = psum[n];
                rt.m = (float)t.a; rt.cx = (float)(t.b / t.a); rt.cy = (float)(t.c / t.a);
                const double q = size * inv_theta;
                rt.thr = (float)(q * q);
                rt.child = quad; rt.first = 0; rt.count = (int32_t)n; rt.pad = 0;
                if (!(t.a > 1e-15)) { rt.child = -1; rt.count = 0; }
                nf[0] = rt;
            }
        }

        for (int c = 0; c < 4; ++c) {
            const double cx0 = (c & 1) ? mx : x0, cx1 = (c & 1) ? x1 : mx;
            const double cy0 = (c & 2) ? my : y0, cy1 = (c & 2) ? y1 : my;
            const double ex = cx1 - cx0, ey = cy1 - cy0;
            const double size = (ex > ey) ? ex : ey;
            const int64_t bc = b[c], nc = b[c + 1] - b[c];
            const int32_t node = quad + c;
            double m = 0.0, cx = 0.0, cy = 0.0;
            int32_t child = -1, occ = -1;
            bool internal = false, bucket = false;
            if (nc == 0) {
                // empty leaf: blank child of project.cu:422-428
            } else if (d + 1 == Dm) {
                // depth-cap cell, project.cu:360-382: running mean in body order
                if (EXACT) {
                    for (int64_t j = bc; j < bc + nc; ++j) {
                        const uint32_t bi = perm[j];
                        const double bm = (double)mass[bi], bx = (double)pos[bi].x, by = (double)pos[bi].y;
                        cx = (m * cx + bm * bx) / (m + bm);
                        cy = (m * cy + bm * by) / (m + bm);
                        m += bm;
                    }
                    occ = (nc == 1) ? (-(int32_t)perm[bc] - 2) : -1;
                } else {
                    const d3 lo_s = psum[bc], hi_s = psum[bc + nc];
                    m = hi_s.a - lo_s.a;
                    if (nc == 1) { const uint32_t bi = perm[bc]; cx = (double)pos[bi].x; cy = (double)pos[bi].y; m = (double)mass[bi]; }
                    else { cx = (hi_s.b - lo_s.b) / m; cy = (hi_s.c - lo_s.c) / m; }
                    occ = (nc == 1) ? (int32_t)bc : -1;
                    bucket = (nc > 1) && !COMPAT;
                }
            } else if (nc == 1) {
                // single body in an undivided cell, project.cu:398-406
                const uint32_t bi = perm[bc];
                m = (double)mass[bi]; cx = (double)pos[bi].x; cy = (double)pos[bi].y;
                occ = EXACT ? (int32_t)bi : (int32_t)bc;
            } else {
                // subdivided cell: its rank follows from its first body and depth
                internal = true;
                const int Lpc = (bc == 0) ? -1 : shared_levels(keys[bc - 1], keys[bc], Dm);
                const uint32_t rc = off[bc] + (uint32_t)((d + 1) - (Lpc + 1));
                child = 1 + 4 * (int32_t)rc;
                if (EXACT) {
                    self_node[rc] = node;
                    cell_depth[rc] = d + 1;
                } else {
                    const d3 lo_s = psum[bc], hi_s = psum[bc + nc];
                    m = hi_s.a - lo_s.a;
                    cx = (hi_s.b - lo_s.b) / m; cy = (hi_s.c - lo_s.c) / m;
                }
            }
            if (EXACT) {
                gd[node] = NodeD{cx, cy, m, size};
                ld[node] = LinkD{child, occ};
            } else {
                NodeF q;
                q.cx = (float)cx; q.cy = (float)cy; q.m = (float)m;
                if (internal) { const double s = size * inv_theta; q.thr = (float)(s * s); }
                else if (bucket) { q.thr = INFINITY; child = -node - 2; }
                else q.thr = -1.0f;
                q.child = child; q.first = (int32_t)bc; q.count = (int32_t)nc; q.pad = 0;
                if (nc > 0 && !(m > 1e-15)) { q.child = -1; q.count = 0; q.thr = -1.0f; }
                nf[node] = q;
            }
        }
        // descend into the child that holds body i (the next cell of this owner's chain)
        if (d < dhi) {
            const int c = (int)((key >> (2 * (Dm - 1 - d))) & 3);
            descend(c, mx, my, x0, x1, y0, y1);
        }
    }
}

// ---- exact bottom-up pass: ComputeMass, project.cu:473-502, one launch per depth ----------------
__global__ __launch_bounds__(kBlock) void com_level_kernel(NodeD *__restrict__ gd,
                                                            const LinkD *__restrict__ ld,
                                                            const int32_t *__restrict__ self_node,
                                                            const int32_t *__restrict__ cell_depth,
                                                            const TreeCounters *__restrict__ ctr,
                                                            int64_t internal_cap, int depth)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = ctr->n_internal;
    if (r >= (int64_t)total || (int64_t)total > internal_cap) return;
    if (cell_depth[r] != depth) return;
    const int32_t node = self_node[r];
    const int32_t quad = 1 + 4 * (int32_t)r;
    double tot = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const NodeD ch = gd[quad + c];
        tot += ch.m;
        sx += ch.m * ch.cx;
        sy += ch.m * ch.cy;
    }
    if (tot > 0.0) { sx /= tot; sy /= tot; }
    gd[node].cx = sx; gd[node].cy = sy; gd[node].m = tot;
}

// ---- fp32 mode helpers ----------------------------------------------------------------------------
// sorted copies for the walk + the (m, m*x, m*y) terms of the prefix sums
template <typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void gather_sorted_kernel(const uint32_t *__restrict__ perm,
                                                                const Real2 *__restrict__ pos,
                                                                const Real *__restrict__ mass,
                                                                Real2 *__restrict__ spos,
                                                                Real *__restrict__ smass,
                                                                d3 *__restrict__ terms, int64_t n)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s > n) return;
    if (s == n) { terms[n] = d3{0.0, 0.0, 0.0}; return; }
    const uint32_t i = perm[s];
    const Real2 p = pos[i];
    const Real m = mass[i];
    spos[s] = p;
    smass[s] = m;
    terms[s] = d3{(double)m, (double)m * (double)p.x, (double)m * (double)p.y};
}

}  // namespace bh
