/*
 * bh_oracle.h -- CPU oracle for the Barnes-Hut step.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's CPU algorithm (not a copy of its
 * source).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product (libbhgpu.so) never links, loads or calls anything here.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
 * against the .npz fixtures under tests/golden/, which were produced by the reference's own code compiled
 * in the development container (recipe: oracle/Makefile target `ref`, driver
 * oracle/ref_driver.cpp, generator scripts/make_golden.py).
 *
 * Reference paths are relative to /root/reference/implementation/.
 */
#ifndef BH_ORACLE_H
#define BH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One quadtree node, field-for-field the reference's 12-double `Quadrant`
 * (project.cu:46-65): children[4] (-1 = none), COM x/y, mass, xmin,xmax,ymin,ymax,
 * particle index (>=0 single body, -1 internal/empty/multi, <=-2 = -idx-2 single body
 * in a depth-cap cell). */
typedef struct bho_node {
    double child[4];
    double comx, comy, mass;
    double xmin, xmax, ymin, ymax;
    double particle;
} bho_node;

typedef struct bho_walk_stats {
    uint64_t visits;        /* nodes popped with mass > 1e-15                      */
    uint64_t interactions;  /* accepted body-node force evaluations                 */
    int32_t  max_stack;     /* deepest explicit stack seen                          */
} bho_walk_stats;

/* project.cu:536-573.  out = {xmin, xmax, ymin, ymax}.  pos is AoS [n][2]. */
void bho_root_bounds(const double *pos, int64_t n, double out[4]);

/* project.cu:343-453 + 473-502 + 575-591 (max_depth >= 1: depth-capped insert, root at
 * depth 1) or main_approach_2.cpp:73-175 (max_depth <= 0: uncapped).  Sequential
 * insertion in body order, then the recursive mass/COM pass.  Node numbering is the
 * reference's (insertion order).  Returns the node count, or -1 if `cap` nodes do not
 * suffice, or -2 if the uncapped tree exceeds 200 levels (coincident bodies). */
int64_t bho_build_tree(const double *pos, const double *mass, int64_t n, int max_depth,
                       bho_node *nodes, int64_t cap);

/* project.cu:593-675 (compat_self_skip=1: `occ==i || occ+2==-i`) and
 * main_approach_2.cpp:261-343 (compat_self_skip=0: `occ==i`).  forces is AoS [n][2]
 * and holds FORCE (m_i included), as in the reference.  stats may be NULL. */
void bho_compute_forces(const bho_node *nodes, const double *pos, const double *mass,
                        int64_t n, double theta, double G, int compat_self_skip,
                        double *forces, bho_walk_stats *stats);

/* Same walk restricted to bodies [lo, hi): used to time a bounded sample. */
void bho_compute_forces_range(const bho_node *nodes, const double *pos, const double *mass,
                              int64_t lo, int64_t hi, double theta, double G,
                              int compat_self_skip, double *forces, bho_walk_stats *stats);

/* The walk of bho_compute_forces_range (same statements, bit-identical forces) with per-body diagnostics for
 * the fp32 parity tests: accepted-node count, sum of |F_j|, the part of it carried by differences of nearly equal
 * fp32 coordinates, and the total multipole error of the cells whose acceptance criterion an fp32 walk may
 * decide the other way ("borderline"; 0 = the fp32 walk must accept exactly this node set).  See bh_oracle.c.
 * cap_depth > 0 (on an uncapped tree): subdivided cells at that depth (root = 1) are summed body by body, as the
 * device's depth-cap buckets are (reference_compat off); cap[] tells how much that changed per body; with
 * cap_depth == 0 the forces are the pinned walk's, bit for bit.
 * Any of counts / abs_sum / coord / flip / cap may be NULL.  All arrays are indexed by body like `forces`. */
void bho_compute_forces_diag(const bho_node *nodes, const double *pos, const double *mass,
                             int64_t lo, int64_t hi, double theta, double G, int compat_self_skip,
                             int pos_rounded, int cap_depth, double *forces, uint32_t *counts, double *abs_sum,
                             double *coord, double *flip, double *cap);

/* main_approach_1.cpp:53-75: O(N^2) direct sum, no softening. */
void bho_direct_forces(const double *pos, const double *mass, int64_t n, double G,
                       double *forces);

/* project.cu:795-817: a = F/m; v += a*dt; p += v*dt (three separate passes). */
void bho_integrate(const double *forces, const double *mass, int64_t n, double dt,
                   double *acc, double *vel, double *pos);

/* project.cu:865-916 without the file I/O: nsteps x (build, walk, integrate).
 * max_depth as in bho_build_tree; use_direct != 0 runs main_approach_1.cpp's loop
 * (ma1.cpp:139-148) instead.  scratch nodes are allocated internally.
 * Returns 0, or a negative bho_build_tree error. */
int bho_run(double *pos, double *vel, const double *mass, int64_t n, int nsteps,
            int max_depth, double theta, double G, double dt, int use_direct);

/* DFS pre-order export, children in index order (the traversal of project.cu:504-534):
 * out[k] receives the k-th visited node, depth[k] its depth (root 0).  Node numbering
 * independent, so two trees with the same topology export identically. */
int64_t bho_export_preorder(const bho_node *nodes, int64_t n_nodes, bho_node *out,
                            int32_t *depth);

/* project.cu:504-534: writes the text dump (ostream default formatting == "%g").
 * compat_oob != 0 reproduces nothing for particle <= -2 (the reference reads out of
 * bounds there): the body's true position is printed instead, as SURVEY 8(c) states.
 * Returns the number of lines, or -1 if the file cannot be opened. */
int64_t bho_write_tree_text(const bho_node *nodes, int64_t n_nodes, const double *pos,
                            const char *path);

/* Work-sharing calibration: for groups of `group` consecutive bodies in
 * the order given by `order` (a permutation, e.g. Morton order), the number of distinct
 * nodes popped by any member, summed over groups. */
uint64_t bho_group_union_visits(const bho_node *nodes, int64_t n_nodes, const double *pos,
                                const int64_t *order, int64_t n, int group, double theta);

#ifdef __cplusplus
}
#endif
#endif
