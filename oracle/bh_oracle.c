/*
 * bh_oracle.c -- CPU oracle for the Barnes-Hut step.  TEST INFRASTRUCTURE ONLY.
 * See bh_oracle.h for the rules on who may load this and for the parity status (pinned).
 *
 * Every function restates, in plain C, the arithmetic of the reference function it
 * cites, operation for operation and in the same order, so that fp64 results are
 * bit-identical to the reference compiled without FMA contraction (x86-64 g++ -O2).
 * Build with -ffp-contract=off.
 */
#include "bh_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BHO_UNCAPPED_LIMIT 200

/* ---- root box: project.cu:536-573 ------------------------------------------------- */
void bho_root_bounds(const double *pos, int64_t n, double out[4])
{
    double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        double x = pos[2 * i], y = pos[2 * i + 1];
        /* std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a */
        xlo = (x < xlo) ? x : xlo;
        xhi = (xhi < x) ? x : xhi;
        ylo = (y < ylo) ? y : ylo;
        yhi = (yhi < y) ? y : yhi;
    }
    double ex = xhi - xlo, ey = yhi - ylo;
    double span = (ex < ey) ? ey : ex;
    double pad = 0.1 * span;
    if (span == 0.0) pad = 1e-6;
    out[0] = xlo - pad;
    out[1] = xhi + pad;
    out[2] = ylo - pad;
    out[3] = yhi + pad;
}

/* ---- insertion: project.cu:343-453 / main_approach_2.cpp:73-144 -------------------- */
typedef struct {
    bho_node *nodes;
    int64_t count, cap;
    const double *pos, *mass;
    int max_depth; /* <=0: uncapped */
    int err;
} build_ctx;

static void blank_node(bho_node *q, double x0, double x1, double y0, double y1)
{
    q->child[0] = q->child[1] = q->child[2] = q->child[3] = -1;
    q->comx = q->comy = q->mass = 0.0;
    q->xmin = x0; q->xmax = x1; q->ymin = y0; q->ymax = y1;
    q->particle = -1;
}

/* project.cu:348-356 */
static int pick_child(double x, double y, const bho_node *q)
{
    double mx = (q->xmin + q->xmax) / 2;
    double my = (q->ymin + q->ymax) / 2;
    if (x <  mx && y <  my) return 0;
    if (x >= mx && y <  my) return 1;
    if (x <  mx && y >= my) return 2;
    return 3;
}

static void insert_body(build_ctx *c, int64_t body, int64_t ni, int depth)
{
    if (c->err) return;
    const double bx = c->pos[2 * body], by = c->pos[2 * body + 1], bm = c->mass[body];

    if (c->max_depth > 0 && depth >= c->max_depth) {
        /* depth-cap aggregation, project.cu:360-382: running mass-weighted mean */
        bho_node *q = &c->nodes[ni];
        double m0 = q->mass, x0 = q->comx, y0 = q->comy;
        q->comx = (m0 * x0 + bm * bx) / (m0 + bm);
        q->comy = (m0 * y0 + bm * by) / (m0 + bm);
        q->mass += bm;
        q->particle = (m0 == 0) ? (double)(-1 * body - 2) : -1.0;
        return;
    }
    if (c->max_depth <= 0 && depth > BHO_UNCAPPED_LIMIT) { c->err = -2; return; }

    bho_node q = c->nodes[ni]; /* working copy, written back where the reference does */
    int empty_leaf = (q.child[0] == -1 && q.child[1] == -1 && q.child[2] == -1 &&
                      q.child[3] == -1 && q.mass == 0.0);
    if (empty_leaf) {
        q.comx = bx; q.comy = by; q.mass = bm; q.particle = (double)body;
        c->nodes[ni] = q;
        return;
    }

    /* occupied leaf -> split.  project.cu:408 tests `> -1`, ma2.cpp:108 tests `!= -1`. */
    int occupied = (c->max_depth > 0) ? (q.particle > -1) : (q.particle != -1);
    if (q.mass > 0.0 && occupied) {
        if (c->count + 4 > c->cap) { c->err = -1; return; }
        for (int k = 0; k < 4; ++k) {
            double mx = (q.xmin + q.xmax) / 2.0;
            double my = (q.ymin + q.ymax) / 2.0;
            bho_node *ch = &c->nodes[c->count];
            if (k == 0)      blank_node(ch, q.xmin, mx, q.ymin, my);
            else if (k == 1) blank_node(ch, mx, q.xmax, q.ymin, my);
            else if (k == 2) blank_node(ch, q.xmin, mx, my, q.ymax);
            else             blank_node(ch, mx, q.xmax, my, q.ymax);
            q.child[k] = (double)c->count;
            c->count++;
        }
        double ox = q.comx, oy = q.comy;
        int64_t old_body = (int64_t)q.particle;
        q.comx = 0.0; q.comy = 0.0; q.mass = 0.0; q.particle = -1;
        c->nodes[ni] = q;
        int oc = pick_child(ox, oy, &q);
        insert_body(c, old_body, (int64_t)q.child[oc], depth + 1);
    }
    int nc = pick_child(bx, by, &q);
    insert_body(c, body, (int64_t)q.child[nc], depth + 1);
}

/* ---- mass / COM pass: project.cu:473-502 ------------------------------------------- */
static void mass_pass(bho_node *nodes, int64_t ni, double *m_out, double *x_out, double *y_out)
{
    bho_node *q = &nodes[ni];
    if (q->child[0] == -1) {
        *m_out = q->mass; *x_out = q->comx; *y_out = q->comy;
        return;
    }
    double tot = 0.0, sx = 0.0, sy = 0.0;
    for (int k = 0; k < 4; ++k) {
        if (q->child[k] != -1) {
            double cm, cx, cy;
            mass_pass(nodes, (int64_t)q->child[k], &cm, &cx, &cy);
            tot += cm;
            sx += cm * cx;
            sy += cm * cy;
        }
    }
    if (tot > 0.0) { sx /= tot; sy /= tot; }
    q->mass = tot; q->comx = sx; q->comy = sy;
    *m_out = tot; *x_out = sx; *y_out = sy;
}

int64_t bho_build_tree(const double *pos, const double *mass, int64_t n, int max_depth,
                       bho_node *nodes, int64_t cap)
{
    if (cap < 1) return -1;
    double box[4];
    bho_root_bounds(pos, n, box);
    build_ctx c = { nodes, 0, cap, pos, mass, max_depth, 0 };
    blank_node(&nodes[0], box[0], box[1], box[2], box[3]);
    c.count = 1;
    for (int64_t i = 0; i < n; ++i) {
        insert_body(&c, i, 0, 1); /* root is depth 1, project.cu:587 */
        if (c.err) return c.err;
    }
    double m, x, y;
    mass_pass(nodes, 0, &m, &x, &y);
    return c.count;
}

/* ---- theta walk: project.cu:593-675 / main_approach_2.cpp:261-343 ------------------ */
void bho_compute_forces_range(const bho_node *nodes, const double *pos, const double *mass,
                              int64_t lo, int64_t hi, double theta, double G,
                              int compat_self_skip, double *forces, bho_walk_stats *stats)
{
    int64_t scap = 1024, *stack = (int64_t *)malloc(sizeof(int64_t) * scap);
    uint64_t visits = 0, inter = 0;
    int32_t deepest = 0;
    for (int64_t i = lo; i < hi; ++i) {
        double fx = 0.0, fy = 0.0;
        const double px = pos[2 * i], py = pos[2 * i + 1];
        int64_t top = 0;
        stack[top++] = 0;
        while (top > 0) {
            if (top > deepest) deepest = (int32_t)top;
            const bho_node *q = &nodes[stack[--top]];
            double qm = q->mass;
            if (qm <= 1e-15) continue;
            visits++;
            int64_t occ = (int64_t)q->particle;
            int leaf = (q->child[0] == -1 && q->child[1] == -1 && q->child[2] == -1 &&
                        q->child[3] == -1);
            double dx = q->comx - px;
            double dy = q->comy - py;
            double d2 = dx * dx + dy * dy;
            double d = sqrt(d2) + 1e-15;
            double ex = q->xmax - q->xmin, ey = q->ymax - q->ymin;
            double size = (ex > ey) ? ex : ey;
            if (leaf || (size / d < theta)) {
                if (leaf) {
                    if (occ == i) continue;
                    if (compat_self_skip && (occ + 2) == -i) continue;
                }
                double f = (G * mass[i] * qm) / d2;
                double ux = dx / d, uy = dy / d;
                fx += f * ux;
                fy += f * uy;
                inter++;
            } else {
                if (top + 4 > scap) {
                    scap *= 2;
                    stack = (int64_t *)realloc(stack, sizeof(int64_t) * scap);
                }
                for (int k = 0; k < 4; ++k) {
                    int64_t ci = (int64_t)q->child[k];
                    if (ci != -1) stack[top++] = ci;
                }
            }
        }
        forces[2 * i] = fx;
        forces[2 * i + 1] = fy;
    }
    free(stack);
    if (stats) { stats->visits = visits; stats->interactions = inter; stats->max_stack = deepest; }
}

void bho_compute_forces(const bho_node *nodes, const double *pos, const double *mass,
                        int64_t n, double theta, double G, int compat_self_skip,
                        double *forces, bho_walk_stats *stats)
{
    bho_compute_forces_range(nodes, pos, mass, 0, n, theta, G, compat_self_skip, forces, stats);
}

/* ---- the same walk with per-body diagnostics (test infrastructure of the fp32 parity tests) ----------
 * Forces are computed by exactly the statements of bho_compute_forces_range (project.cu:593-675) -- the tests
 * check that they come out bit-identical -- and next to them, per body i:
 *   counts[i]   accepted force evaluations (the `inter++` of the walk above, per body)
 *   abs_sum[i]  sum over accepted nodes of |F_j| = G m_i M_j / d_j^2
 *   coord[i]    sum over accepted CELLS (and, if pr, leaves) of |F_j| * (|comx| + |comy| + pr * (|px| + |py|)) / d_j : how much of the
 *               force is carried by differences of nearly equal coordinates (a node's centre and a body's
 *               position are stored as fp32 values on the device: d_j is known to 2^-24 * that sum)
 *   flip[i]     sum over BORDERLINE subdivided cells of |F(cell accepted) - F(cell opened)|.  A cell is borderline
 *               for body i when its criterion size/d < theta (project.cu:643) is decided by less than the
 *               relative uncertainty fp32 arithmetic has about d:  |d - size/theta| <= d * tol,
 *               tol = 2^-23 * ((|comx| + |comy| + pr * (|px| + |py|)) / d + 4).   An fp32 walk may decide those
 *               cells, and only those, the other way; each such flip changes the body's force by exactly that
 *               cell's multipole error, which is what is summed here (F(opened) = the ordinary walk of the
 *               cell's subtree).  flip[i] == 0: the fp32 walk must accept exactly the oracle's node set.
 * pr = pos_rounded: 1 when the device rounds the body positions to fp32 itself (mixed precision), 0 when
 * the inputs already are fp32 values. */
static void diag_subtree_force(const bho_node *nodes, int64_t start, int64_t i, double px, double py,
                               double Gmi, double theta, int compat_self_skip, int64_t **stack, int64_t *scap,
                               double *ofx, double *ofy)
{
    /* the walk of bho_compute_forces_range below node `start`, which is treated as opened */
    double fx = 0.0, fy = 0.0;
    int64_t top = 0;
    const bho_node *r = &nodes[start];
    for (int k = 0; k < 4; ++k) {
        int64_t ci = (int64_t)r->child[k];
        if (ci != -1) (*stack)[top++] = ci;
    }
    while (top > 0) {
        const bho_node *q = &nodes[(*stack)[--top]];
        double qm = q->mass;
        if (qm <= 1e-15) continue;
        int64_t occ = (int64_t)q->particle;
        int leaf = (q->child[0] == -1 && q->child[1] == -1 && q->child[2] == -1 && q->child[3] == -1);
        double dx = q->comx - px, dy = q->comy - py;
        double d2 = dx * dx + dy * dy;
        double d = sqrt(d2) + 1e-15;
        double ex = q->xmax - q->xmin, ey = q->ymax - q->ymin;
        double size = (ex > ey) ? ex : ey;
        if (leaf || (size / d < theta)) {
            if (leaf) {
                if (occ == i) continue;
                if (compat_self_skip && (occ + 2) == -i) continue;
            }
            double f = (Gmi * qm) / d2;
            fx += f * (dx / d);
            fy += f * (dy / d);
        } else {
            if (top + 4 > *scap) {
                *scap *= 2;
                *stack = (int64_t *)realloc(*stack, sizeof(int64_t) * (size_t)*scap);
            }
            for (int k = 0; k < 4; ++k) {
                int64_t ci = (int64_t)q->child[k];
                if (ci != -1) (*stack)[top++] = ci;
            }
        }
    }
    *ofx = fx; *ofy = fy;
}

/* every body below node `start` (itself excluded by `i`): the direct sum the device forms for a depth-cap cell */
static uint32_t diag_subtree_direct(const bho_node *nodes, int64_t start, int64_t i, double px, double py,
                                    double Gmi, int64_t **stack, int64_t *scap, double *ofx, double *ofy,
                                    double *oabs)
{
    double fx = 0.0, fy = 0.0, fa = 0.0;
    uint32_t cnt = 0;
    int64_t top = 0;
    (*stack)[top++] = start;
    while (top > 0) {
        const bho_node *q = &nodes[(*stack)[--top]];
        if (q->mass <= 1e-15) continue;
        int leaf = (q->child[0] == -1 && q->child[1] == -1 && q->child[2] == -1 && q->child[3] == -1);
        if (leaf) {
            if ((int64_t)q->particle == i) continue;
            double dx = q->comx - px, dy = q->comy - py;
            double d2 = dx * dx + dy * dy;
            double d = sqrt(d2) + 1e-15;
            double f = (Gmi * q->mass) / d2;
            fx += f * (dx / d);
            fy += f * (dy / d);
            fa += f;
            cnt++;
            continue;
        }
        if (top + 4 > *scap) {
            *scap *= 2;
            *stack = (int64_t *)realloc(*stack, sizeof(int64_t) * (size_t)*scap);
        }
        for (int k = 0; k < 4; ++k) {
            int64_t ci = (int64_t)q->child[k];
            if (ci != -1) (*stack)[top++] = ci;
        }
    }
    *ofx = fx; *ofy = fy; *oabs = fa;
    return cnt;
}

/* cap_depth > 0 (uncapped trees only): a subdivided cell at that depth (root = 1) is treated as the device treats
 * it when reference_compat is off -- a depth-cap BUCKET, summed body by body for every body that reaches it,
 * whatever the acceptance criterion says (deviation iv) -- in forces[] and counts[];
 * cap[i] (may be NULL) receives the summed magnitude of what that changes for body i against the plain walk. */
void bho_compute_forces_diag(const bho_node *nodes, const double *pos, const double *mass,
                             int64_t lo, int64_t hi, double theta, double G, int compat_self_skip,
                             int pos_rounded, int cap_depth, double *forces, uint32_t *counts, double *abs_sum,
                             double *coord, double *flip, double *cap)
{
    int64_t scap = 1024, *stack = (int64_t *)malloc(sizeof(int64_t) * scap);
    int32_t *dstack = (int32_t *)malloc(sizeof(int32_t) * scap);
    int64_t scap2 = 1024, *stack2 = (int64_t *)malloc(sizeof(int64_t) * scap2);
    const double ulp23 = ldexp(1.0, -23);
    for (int64_t i = lo; i < hi; ++i) {
        double fx = 0.0, fy = 0.0, asum = 0.0, csum = 0.0, fsum = 0.0, capsum = 0.0;
        uint32_t cnt = 0;
        const double px = pos[2 * i], py = pos[2 * i + 1];
        const double pabs = pos_rounded ? fabs(px) + fabs(py) : 0.0;
        int64_t top = 0;
        dstack[top] = 1;
        stack[top++] = 0;
        while (top > 0) {
            const int64_t qi = stack[--top];
            const int32_t depth = dstack[top];
            const bho_node *q = &nodes[qi];
            double qm = q->mass;
            if (qm <= 1e-15) continue;
            int64_t occ = (int64_t)q->particle;
            int leaf = (q->child[0] == -1 && q->child[1] == -1 && q->child[2] == -1 &&
                        q->child[3] == -1);
            double dx = q->comx - px;
            double dy = q->comy - py;
            double d2 = dx * dx + dy * dy;
            double d = sqrt(d2) + 1e-15;
            double ex = q->xmax - q->xmin, ey = q->ymax - q->ymin;
            double size = (ex > ey) ? ex : ey;
            const double cabs = fabs(q->comx) + fabs(q->comy) + pabs;
            if (!leaf && cap_depth > 0 && depth == cap_depth) {
                /* the device's bucket: every body of the cell, one by one */
                double bx, by, ba, ox, oy;
                const uint32_t bc = diag_subtree_direct(nodes, qi, i, px, py, G * mass[i], &stack2, &scap2, &bx, &by, &ba);
                if (size / d < theta) { double f = (G * mass[i] * qm) / d2; ox = f * (dx / d); oy = f * (dy / d); }
                else diag_subtree_force(nodes, qi, i, px, py, G * mass[i], theta, compat_self_skip, &stack2, &scap2, &ox, &oy);
                capsum += sqrt((bx - ox) * (bx - ox) + (by - oy) * (by - oy));
                fx += bx; fy += by; cnt += bc; asum += ba;
                csum += pos_rounded ? ba * (cabs / d) : 0.0;
                continue;
            }
            if (!leaf) {
                const double tol = ulp23 * (cabs / d + 4.0);
                if (fabs(d - size / theta) <= d * tol) {
                    double f = (G * mass[i] * qm) / d2;
                    double ax = f * (dx / d), ay = f * (dy / d), ox, oy;
                    diag_subtree_force(nodes, qi, i, px, py, G * mass[i], theta, compat_self_skip, &stack2,
                                       &scap2, &ox, &oy);
                    fsum += sqrt((ax - ox) * (ax - ox) + (ay - oy) * (ay - oy));
                }
            }
            if (leaf || (size / d < theta)) {
                if (leaf) {
                    if (occ == i) continue;
                    if (compat_self_skip && (occ + 2) == -i) continue;
                }
                double f = (G * mass[i] * qm) / d2;
                double ux = dx / d, uy = dy / d;
                fx += f * ux;
                fy += f * uy;
                cnt++;
                asum += f;
                /* (a single body's leaf carries that body's position: an exact fp32 value unless the device rounds) */
                csum += (leaf && !pos_rounded) ? 0.0 : f * (cabs / d);
            } else {
                if (top + 4 > scap) {
                    scap *= 2;
                    stack = (int64_t *)realloc(stack, sizeof(int64_t) * scap);
                    dstack = (int32_t *)realloc(dstack, sizeof(int32_t) * scap);
                }
                for (int k = 0; k < 4; ++k) {
                    int64_t ci = (int64_t)q->child[k];
                    if (ci != -1) { dstack[top] = depth + 1; stack[top++] = ci; }
                }
            }
        }
        forces[2 * i] = fx;
        forces[2 * i + 1] = fy;
        if (counts) counts[i] = cnt;
        if (abs_sum) abs_sum[i] = asum;
        if (coord) coord[i] = csum;
        if (flip) flip[i] = fsum;
        if (cap) cap[i] = capsum;
    }
    free(stack);
    free(dstack);
    free(stack2);
}

/* ---- direct sum: main_approach_1.cpp:53-75 ----------------------------------------- */
void bho_direct_forces(const double *pos, const double *mass, int64_t n, double G,
                       double *forces)
{
    for (int64_t i = 0; i < n; ++i) {
        double sx = 0.0, sy = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            if (i == j) continue;
            double d2 = 0.0;
            double dx = pos[2 * j] - pos[2 * i];
            d2 += dx * dx;
            double dy = pos[2 * j + 1] - pos[2 * i + 1];
            d2 += dy * dy;
            double d = sqrt(d2);
            double k = G * mass[i] * mass[j] / (d2 * d);
            sx += k * dx;
            sy += k * dy;
        }
        forces[2 * i] = sx;
        forces[2 * i + 1] = sy;
    }
}

/* ---- integrator: project.cu:795-817 ------------------------------------------------ */
void bho_integrate(const double *forces, const double *mass, int64_t n, double dt,
                   double *acc, double *vel, double *pos)
{
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 2; ++k) acc[2 * i + k] = forces[2 * i + k] / mass[i];
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 2; ++k) vel[2 * i + k] += acc[2 * i + k] * dt;
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 2; ++k) pos[2 * i + k] += vel[2 * i + k] * dt;
}

/* ---- step loop: project.cu:883-910 / main_approach_1.cpp:139-148 ------------------- */
int bho_run(double *pos, double *vel, const double *mass, int64_t n, int nsteps,
            int max_depth, double theta, double G, double dt, int use_direct)
{
    double *f = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    double *a = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    int64_t cap = 0;
    bho_node *nodes = NULL;
    if (!use_direct) {
        int levels = (max_depth > 0) ? max_depth : BHO_UNCAPPED_LIMIT;
        cap = 1 + 4 * n * (int64_t)(levels < 64 ? levels : 64);
        if (max_depth > 0 && max_depth < 15) {
            int64_t full = 0, w = 1;
            for (int l = 0; l < max_depth; ++l) { full += w; w *= 4; }
            if (full < cap) cap = full;
        }
        if (cap > 16 * n + 1024) cap = 16 * n + 1024;
        nodes = (bho_node *)malloc(sizeof(bho_node) * (size_t)cap);
    }
    int rc = 0;
    for (int s = 0; s < nsteps && rc == 0; ++s) {
        if (use_direct) {
            bho_direct_forces(pos, mass, n, G, f);
        } else {
            int64_t nn = bho_build_tree(pos, mass, n, max_depth, nodes, cap);
            if (nn < 0) { rc = (int)nn; break; }
            bho_compute_forces(nodes, pos, mass, n, theta, G, max_depth > 0, f, NULL);
        }
        bho_integrate(f, mass, n, dt, a, vel, pos);
    }
    free(f); free(a); free(nodes);
    return rc;
}

/* ---- DFS pre-order export: traversal order of project.cu:504-534 -------------------- */
int64_t bho_export_preorder(const bho_node *nodes, int64_t n_nodes, bho_node *out,
                            int32_t *depth)
{
    if (n_nodes < 1) return 0;
    int64_t *st = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_nodes + 4));
    int32_t *sd = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_nodes + 4));
    int64_t top = 0, k = 0;
    st[top] = 0; sd[top] = 0; top++;
    while (top > 0) {
        --top;
        int64_t ni = st[top];
        int32_t d = sd[top];
        out[k] = nodes[ni];
        if (depth) depth[k] = d;
        k++;
        for (int c = 3; c >= 0; --c) { /* push reversed so child 0 is visited first */
            int64_t ci = (int64_t)nodes[ni].child[c];
            if (ci != -1) { st[top] = ci; sd[top] = d + 1; top++; }
        }
    }
    free(st); free(sd);
    return k;
}

/* ---- text dump: project.cu:504-534 -------------------------------------------------- */
int64_t bho_write_tree_text(const bho_node *nodes, int64_t n_nodes, const double *pos,
                            const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    bho_node *ord = (bho_node *)malloc(sizeof(bho_node) * (size_t)n_nodes);
    int32_t *dep = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_nodes);
    int64_t k = bho_export_preorder(nodes, n_nodes, ord, dep);
    for (int64_t i = 0; i < k; ++i) {
        const bho_node *q = &ord[i];
        fprintf(fp, "%d %g %g %g %g %g", dep[i], q->xmin, q->xmax, q->ymin, q->ymax, q->mass);
        int64_t occ = (int64_t)q->particle;
        if (occ != -1) {
            /* the reference indexes positions[occ] even for occ <= -2 (out of bounds);
             * the body meant is -(occ+2), whose true position is printed here */
            int64_t b = (occ >= 0) ? occ : -(occ + 2);
            fprintf(fp, " occupantIndex=%lld occupantPos=(%g,%g)", (long long)occ,
                    pos[2 * b], pos[2 * b + 1]);
        } else if (q->mass > 0) {
            fprintf(fp, " occupantIndex=%lld occupantPos=(%g,%g)", (long long)occ, q->comx,
                    q->comy);
        }
        fputc('\n', fp);
    }
    fclose(fp);
    free(ord); free(dep);
    return k;
}

/* ---- work-sharing calibration ---------------------------------------- */
uint64_t bho_group_union_visits(const bho_node *nodes, int64_t n_nodes, const double *pos,
                                const int64_t *order, int64_t n, int group, double theta)
{
    int64_t *mark = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_nodes);
    for (int64_t i = 0; i < n_nodes; ++i) mark[i] = -1;
    int64_t scap = 1024, *stack = (int64_t *)malloc(sizeof(int64_t) * scap);
    uint64_t total = 0;
    for (int64_t g0 = 0; g0 < n; g0 += group) {
        int64_t g1 = g0 + group < n ? g0 + group : n;
        for (int64_t s = g0; s < g1; ++s) {
            int64_t i = order[s];
            const double px = pos[2 * i], py = pos[2 * i + 1];
            int64_t top = 0;
            stack[top++] = 0;
            while (top > 0) {
                int64_t ni = stack[--top];
                const bho_node *q = &nodes[ni];
                if (q->mass <= 1e-15) continue;
                if (mark[ni] != g0) { mark[ni] = g0; total++; }
                int leaf = (q->child[0] == -1);
                double dx = q->comx - px, dy = q->comy - py;
                double d = sqrt(dx * dx + dy * dy) + 1e-15;
                double ex = q->xmax - q->xmin, ey = q->ymax - q->ymin;
                double size = (ex > ey) ? ex : ey;
                if (leaf || (size / d < theta)) continue;
                if (top + 4 > scap) {
                    scap *= 2;
                    stack = (int64_t *)realloc(stack, sizeof(int64_t) * scap);
                }
                for (int k = 0; k < 4; ++k) {
                    int64_t ci = (int64_t)q->child[k];
                    if (ci != -1) stack[top++] = ci;
                }
            }
        }
    }
    free(mark); free(stack);
    return total;
}
