/*
 * bhgpu.h -- C-ABI of libbhgpu.so, the MI355X-native Barnes-Hut step.
 *
 * The reference (DavidSevic/gpu-nbody-simulation) has NO library interface: its three
 * programs are configured by -D macros and by editing source (README.md:9-18), and the only
 * stable contract is file level.  This header is therefore the boundary a
 * maintainer would bind instead of calling runSimulationGpu() (project.cu:918-1024): each
 * entry point cites the reference function(s) it replaces.  Paths are relative to
 * /root/reference/implementation/.  INTEGRATION.md shows the ctypes stub and the three-line
 * change to a C++ main().
 *
 * Conventions
 *   - plain C types only; every call returns 0 on success or a negative bh_status;
 *     bh_last_error(ctx) gives the text.  No exceptions cross the boundary.
 *   - the caller owns every host buffer; the library owns all device memory.
 *   - host arrays use the reference's layout: positions/velocities AoS double[n][2],
 *     masses double[n] (project.cu:38-43).  Body order is the caller's order on every
 *     download, whatever order the device keeps internally.
 *   - one context per device; a context is not thread-safe; there are no globals.
 */
#ifndef BHGPU_H
#define BHGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHGPU_ABI_VERSION 4   /* 2: bh_stats_t carries per-kernel-group times and algorithmic bytes;
                                 3: bh_get_interaction_counts, bh_build_info, bh_step_times,
                                    bh_stats_t.let_*_ms;
                                 4: bh_stats_t.wave_accepts, .walk_launches (the struct grew) */

typedef enum bh_status {
    BH_OK = 0,
    BH_ERR_ARG = -1,         /* bad argument (null pointer, n > capacity, ...)            */
    BH_ERR_DEVICE = -2,      /* HIP runtime error; text in bh_last_error                  */
    BH_ERR_NO_DEVICE = -3,   /* no usable GPU: the product has no CPU fallback            */
    BH_ERR_CAPACITY = -4,    /* tree needs more nodes than node_capacity                  */
    BH_ERR_STATE = -5,       /* call order (e.g. step before upload)                      */
    BH_ERR_IO = -6           /* cannot open/write a file                                  */
} bh_status;

typedef enum bh_precision {
    /* fp64 state, fp64 arithmetic in the reference's operation order, no FMA contraction:
     * results are bit-identical to the reference CPU path (project.cu:865-916). */
    BH_PRECISION_F64_EXACT = 0,
    /* fp32 state and arithmetic (BASELINE config "fp32"): the throughput mode. */
    BH_PRECISION_F32 = 1,
    /* fp64 state, fp32 forces (BASELINE config "fp64 positions / fp32 forces"): positions,
     * velocities and masses are kept and integrated in fp64, the tree keys and the centres of mass
     * are computed from the fp64 positions, the theta-walk runs in fp32 on rounded copies exactly
     * as in BH_PRECISION_F32.  For runs where a step's displacement is below the fp32 resolution
     * of the coordinates.  Single-GPU step and the LET distributed step; not the replicated one. */
    BH_PRECISION_MIXED = 2,
    /* fp64 end to end like the reference (project.cu:38-65) at THROUGHPUT: the tree is the exact mode's, node
     * for node bitwise the reference's; the walk takes the fp32 kernel's design (a hand-written gfx950 loop, four sibling
     * nodes per scalar load, free visiting order, the acceptance criterion of project.cu:643 as one compare on d^2, 1/d by
     * v_rsq_f64 + ONE Newton step instead of sqrt and three divisions).  Forces agree
     * with the reference CPU path to summation rounding (<= 1e-12 relative, same per-body interaction counts), not
     * bit for bit; trajectories therefore diverge from it as fast as the dynamics amplify 1e-15.  reference_compat,
     * max_depth and the empty-node cut-off mean what they mean in BH_PRECISION_F64_EXACT.  Single GPU. */
    BH_PRECISION_F64 = 3
} bh_precision;

/* bh_config.flags */
#define BH_FLAG_WALK_STATS   (1u << 0)  /* count visits/interactions in the walk (slower)   */
#define BH_FLAG_LDS_STACK    (1u << 1)  /* fp32 walk: LDS traversal stack instead of the
                                           register-lane stack (A/B switch)  */
#define BH_FLAG_WALK_NO_SPLIT (1u << 2) /* fp32 walk: always one wavefront per 64 bodies.  By
                                           default a launch of few bodies (<= ~100k: small N, or
                                           one rank's share) lets 4 or 8 wavefronts share each
                                           64-body group, level by level; same nodes, same
                                           per-body criterion, but another order of the fp32
                                           sums, so a body's last bits then depend on how many
                                           bodies its launch walks (still reproducible run to
                                           run).  Set this to rule that out.  BH_PRECISION_F64 likewise:
                                           launches of few bodies give a wavefront fewer than 64 of them
                                           (shorter walks), which changes the order of a body's fp64 sum;
                                           the flag pins 64.  (The bit-exact mode does the same and needs no
                                           flag: its order is the reference's whoever shares the wave.)   */

#define BH_FLAG_WALK_PORTABLE (1u << 3) /* fp32 and BH_PRECISION_F64 walks: the C++ traversal loop instead of the
                                           hand-scheduled gfx950 assembly loop.  Same operations
                                           in the same order -- results are bit-identical; kept
                                           as the readable statement of the loop and for tests.
                                           BH_PRECISION_F64_EXACT: the walk written as the reference
                                           writes it (sqrt, size / d < theta, three divisions per
                                           term; the nodes then carry sizes instead of exact d2
                                           thresholds) -- bit-identical again.                  */

/* Replaces the compile-time configuration of project.cu:1-11, 27-35, 60-62. */
typedef struct bh_config {
    int64_t  capacity;          /* max bodies (N_BODIES, project.cu:1-3)                    */
    double   theta;             /* THETA, project.cu:60 (0.5)                               */
    double   G;                 /* project.cu:27 (6.67e-11)                                 */
    double   dt;                /* DELTA_T, project.cu:29 (1.0)                             */
    int32_t  max_depth;         /* QUADTREE_MAX_DEPTH, project.cu:61 (10); root is depth 1;
                                   1..32                                                    */
    int32_t  precision;         /* bh_precision                                             */
    int32_t  reference_compat;  /* 1: depth-cap aggregation and the `occ+2 == -i` self skip
                                   exactly as project.cu:360-382, 646 (self-interaction
                                   artefact included).  0: F64_EXACT keeps the aggregation but
                                   uses main_approach_2.cpp's `occ == i` skip only (with
                                   max_depth 32 this is the uncapped tree of ma2.cpp); F32
                                   sums a depth-cap cell holding several (<= 1024) bodies body
                                   by body (bucket leaf), so no body interacts with its own
                                   cell; larger cells (degenerate inputs) are aggregated.     */
    int32_t  device;            /* HIP device ordinal                                       */
    int32_t  n_threads;         /* N_THREADS, project.cu:5-7, 703: at most this many bodies are walked
                                   at a time (rounded up to whole 256-thread workgroups; the passes run
                                   one after the other, as the reference's threads stride over the bodies).
                                   0 = all at once.  The launch SHAPE within a pass is CDNA4's, not the
                                   reference's (a12 is replaced, not kept).                              */
    uint32_t flags;             /* BH_FLAG_*                                                */
    int64_t  node_capacity;     /* 0 = automatic (8*capacity + 1024 nodes)                  */
} bh_config;

/* The reference's 12-double `Quadrant` (project.cu:46-65), as exported by bh_export_tree.
 * child[k] is an index INTO THE EXPORTED ARRAY (DFS pre-order), or -1. */
typedef struct bh_tree_node {
    double child[4];
    double comx, comy, mass;
    double xmin, xmax, ymin, ymax;
    double particle;
} bh_tree_node;

typedef struct bh_stats_t {
    int64_t  n_bodies;
    int64_t  n_nodes;            /* nodes of the last tree built                            */
    int64_t  n_internal;         /* subdivided cells of the last tree                       */
    int64_t  steps_done;
    uint64_t visits;             /* BH_FLAG_WALK_STATS: body-node visits of the last walk   */
    uint64_t interactions;       /* BH_FLAG_WALK_STATS: accepted force evaluations          */
    uint64_t wave_nodes;         /* BH_FLAG_WALK_STATS: nodes evaluated, counted once per
                                    wavefront (= distinct nodes per 64-body group)           */
    double   last_step_ms;       /* HIP-event time of the last bh_step call / nsteps        */
    double   build_ms;           /* bounds+keys+sort+nodes+COM of the last timed step       */
    double   walk_ms;            /* walk+integrate kernel of the last timed step            */
    uint64_t device_bytes;       /* device memory held by the context                       */
    /* per kernel group of the last timed step (HIP events on the context's stream):
     * the reference times "GPU parallel computation" as a whole (project.cu:957, 1008)          */
    double   keys_ms;            /* root box + keys (bounds_final, keys_kernel)              */
    double   sort_ms;            /* radix passes (+ the state re-ordering when it ran)       */
    double   scan_ms;            /* cell counts, sorted copies, ranks, prefix sums           */
    double   nodes_ms;           /* node records (+ the bottom-up mass pass in exact mode)   */
    /* algorithmic bytes of that step: what each kernel must read and write once            */
    uint64_t build_bytes;        /* keys + sort passes + scan + nodes                        */
    uint64_t walk_bytes;         /* walk + integrate: 44 B per body + 20 B per node a
                                    wavefront evaluates (needs BH_FLAG_WALK_STATS, else 0)    */
    uint64_t wave_quads;         /* BH_FLAG_WALK_STATS: sibling quads loaded, counted once per
                                    wavefront (the walk's memory round trips)                 */
    uint64_t sort_spill_buckets; /* buckets of the bucket sort that did not fit on chip and were sorted
                                    through memory, since bh_create (0 in steady motion)      */
    /* the last bh_let_build (distributed step), HIP events on the context's stream (ABI 3)  */
    double   let_tree_ms;        /* global box + the local tree under it                     */
    double   let_pack_ms;        /* marking, numbering and packing the peers' LETs           */
    uint64_t sort_rerun_buckets; /* buckets of the bucket sort whose short sort (the top 24 bits of the keys' span, then
                                    runs of equal top bits by counting) met a run of more than 8 keys and was repeated
                                    with all byte passes, since bh_create (0 unless bodies pile up)          */
    /* ABI 4 */
    uint64_t wave_accepts;       /* BH_FLAG_WALK_STATS, the fp64 precisions: nodes some lane of the wavefront took a term from, counted once
                                    per wavefront -- the nodes that pay the reciprocal square root and the force (the
                                    others stop at the compare); 0 in the other precisions                          */
    uint64_t walk_launches;      /* walk kernel launches of the last bh_step / bh_compute_forces step: 1, or the passes
                                    of n_threads (project.cu:703)                                              */
} bh_stats_t;

typedef struct bh_ctx bh_ctx;

/* --- lifetime ---------------------------------------------------------------------------
 * Replaces the cudaMalloc block of runSimulationGpu (project.cu:932-940) and its cudaFree
 * block (:1014-1019).  Fails with BH_ERR_NO_DEVICE when no GPU is present. */
int bh_create(const bh_config *cfg, bh_ctx **out);
void bh_destroy(bh_ctx *ctx);
/* ctx may be NULL: then the text of the last failed bh_create on this thread. */
const char *bh_last_error(const bh_ctx *ctx);
int bh_abi_version(void);
/* What this binary was built from: "digest=<16 hex digits of the device sources> flags=<extra compiler flags>",
 * stamped by the build (gpu_nbody_simulation_amd/build.py, scripts/build_variants.sh).  bench.py prints it, so a
 * line measured on an A/B variant or a stale library says so (the reference has no counterpart: project.cu is
 * rebuilt by nvcc for every run, first_scaling_script.sh:30). */
const char *bh_build_info(void);

/* --- state ------------------------------------------------------------------------------
 * bh_upload replaces the three cudaMemcpy H2D of project.cu:943-945 (and, on the caller's
 * side, the arrays filled by loadSimulationDataFromText, project.cu:103-161).
 * bh_download replaces the per-step D2H of positions (project.cu:1010) and additionally
 * returns velocities, which the reference never exposes.  vel may be NULL.
 * Masses: BH_PRECISION_F64_EXACT reproduces the reference bit for bit for POSITIVE masses (however small).  A mass of
 * exactly 0.0 is accepted, but QuadInsert takes a leaf whose mass is 0.0 for empty (project.cu:395-397): a massless
 * body is overwritten by a later arrival and subdivides an earlier one, an insertion-order-dependent tree that
 * this build does not reproduce -- here a massless body occupies its leaf like any other (and pulls nobody). */
int bh_upload(bh_ctx *ctx, const double *pos, const double *vel, const double *mass, int64_t n);
int bh_download(bh_ctx *ctx, double *pos, double *vel);

/* bh_initialize replaces initializeGpu (project.cu:304-341): bodies are generated on the device
 * by a counter-based generator, reproducible for a given (seed, n).  kind 0 = the reference's box
 * distribution: masses in [lower_m, higher_m], positions in [lower_p, higher_p]^2, velocities in
 * [lower_v, higher_v]^2, each range log-uniform when both bounds are positive and linear otherwise
 * (generateRandomGpu, project.cu:84-97).  kind 1 = projected Plummer sphere (BASELINE config 3):
 * scale lower_p, truncation radius higher_p, equal masses higher_m, zero velocities.
 * bh_download_masses returns the masses (the caller supplied them in bh_upload otherwise). */
int bh_initialize(bh_ctx *ctx, int64_t n, uint64_t seed, int32_t kind, double lower_m, double higher_m,
                  double lower_p, double higher_p, double lower_v, double higher_v);
int bh_download_masses(bh_ctx *ctx, double *mass);

/* --- the hot path -----------------------------------------------------------------------
 * bh_step: nsteps x { buildTree (project.cu:575-591), computeForcesGpu (:679-793),
 * updateAccVelPos (:819-836) }, i.e. the body of the step loop project.cu:955-1011, with
 * the tree built on the device instead of the host.  Asynchronous on the context's stream;
 * bh_sync / bh_download / bh_stats wait for it. */
int bh_step(bh_ctx *ctx, int32_t nsteps);
int bh_sync(bh_ctx *ctx);

/* The same three stages one at a time, for per-stage parity checks:
 * bh_build_tree   = buildTree            (project.cu:575-591)
 * bh_compute_forces = buildTree + computeForces/computeForcesGpu (project.cu:593-675, 679-793);
 *                   state is not advanced
 * bh_get_forces   -> forces[n][2], FORCE with m_i included as in the reference's `forces`
 * bh_get_accel    -> forces / m_i (updateAccelerations, project.cu:795-801)               */
int bh_build_tree(bh_ctx *ctx);
int bh_compute_forces(bh_ctx *ctx);
int bh_get_forces(bh_ctx *ctx, double *forces);
int bh_get_accel(bh_ctx *ctx, double *accel);

/* Per-body count of accepted force evaluations of the last walk -- the reference's walk has no such output; it is
 * the `inter++`-per-body of computeForces (project.cu:651-658 executed once per accepted node), which the parity
 * tests compare with the oracle's body by body: equal counts = the same acceptance decisions (project.cu:643).
 * Needs BH_FLAG_WALK_STATS; fp32, mixed precision and BH_PRECISION_F64; caller order. */
int bh_get_interaction_counts(bh_ctx *ctx, uint32_t *counts);

/* --- tree output ------------------------------------------------------------------------
 * bh_export_tree: the tree of the last bh_build_tree/bh_compute_forces/bh_step in DFS
 * pre-order with children in index order -- the visiting order of TraverseTreeToFile
 * (project.cu:504-534).  depth may be NULL.  *n_nodes receives the node count even when cap
 * is too small (then BH_ERR_CAPACITY).
 * bh_write_quadtree_file: TraverseTreeToFile itself (same text format, project.cu:509-526);
 * for occupant indices <= -2, where the reference reads out of bounds, the body's true
 * position is printed. */
int bh_export_tree(bh_ctx *ctx, bh_tree_node *nodes, int32_t *depth, int64_t cap,
                   int64_t *n_nodes);
int bh_write_quadtree_file(bh_ctx *ctx, const char *path);

/* --- measurement (replaces the std::chrono timers of project.cu:985-1007) ----------------*/
int bh_stats(bh_ctx *ctx, bh_stats_t *out);
/* Per step of the last bh_step call (at most 4,096 of them): step_ms[s] = end of step s-1's walk (or the start of
 * the call) to the end of step s's walk, walk_ms[s] = that step's walk + integrate kernel -- HIP events on the
 * context's stream.  The reference accumulates one total per run (project.cu:985-1007); a spread needs the steps.
 * *n_out receives the number of steps available even when cap is too small; step_ms / walk_ms may be NULL. */
int bh_step_times(bh_ctx *ctx, double *step_ms, double *walk_ms, int32_t cap, int32_t *n_out);

/* --- multi-GPU plumbing -----------------------------------------------------------------
 * The reference is single-GPU.  One process per GPU owns a contiguous range [lo, hi) of
 * the sorted (space-filling-curve order) bodies: bh_step then walks and integrates only that range, and the
 * host exchanges the updated ranges (torch.distributed all_gather over RCCL) through the
 * device pointers below.  The exchange buffers stay valid until bh_destroy; element types follow
 * the context's precision (double2/double or float2/float).  bh_device_state exposes the state
 * arrays as the device holds them: in exact mode that is the caller's order; in fp32 / mixed mode
 * the engine re-orders the bodies into sorted order every 16th tree build (so that its gathers
 * stay local) and swaps buffers when it does -- ask again after stepping, and use bh_download for
 * the caller's order. */
int bh_set_owned_fraction(bh_ctx *ctx, int32_t rank, int32_t world);
int bh_device_state(bh_ctx *ctx, void **pos, void **vel, void **mass, int64_t *n,
                    int32_t *elem_bytes);
int bh_owned_range(bh_ctx *ctx, int64_t *lo, int64_t *hi);
/* Sorted-order view used by the exchange: ONE buffer of {x, y, vx, vy} (4 floats) per sorted body.
 * After bh_step_local the owned slice holds the new state; after ONE all_gather of the slices
 * bh_scatter_sorted writes the whole buffer back to caller order. */
int bh_step_local(bh_ctx *ctx);
int bh_device_sorted(bh_ctx *ctx, void **sorted_state);
int bh_scatter_sorted(bh_ctx *ctx);
/* Distributed step with locally-essential trees (LET).  Unlike the replicated scheme above, a
 * context in LET mode holds ONLY ITS OWN bodies (bh_upload its subset; a contiguous range of a
 * space-filling-curve order of the bodies keeps the exchanged trees small).  Per step:
 *   bh_let_bounds   -> B = boxes_per_rank bounding boxes of consecutive slices of the local bodies
 *                      in a device buffer (B x 4 doubles: xmin, xmax, ymin, ymax; unpadded)
 *   [host: all_gather the W x B x 4 doubles into the all_bounds buffer]
 *   bh_let_build    -> global root box, local tree under it, and for every peer a compact LET
 *                      (the quads that some body inside one of the peer's boxes can open) packed
 *                      in the send buffer, W fixed-size blocks of let_cap quads, child links
 *                      already expressed in the receiver's index space
 *   [host: all_to_all of the blocks, send buffer -> recv buffer]
 *   bh_let_walk     -> every local body walks its own tree and the W-1 received LETs, integrate
 * bh_let_pointers exposes the device buffers for the two collectives; block_bytes is the size of
 * one per-peer block (the pointers change when let_cap does).  bh_let_counts waits for the stream and
 * returns, per peer, the LARGEST LET of any build since the previous bh_let_counts (or
 * bh_let_configure), and whether any of those builds overflowed: with overflow == NULL it fails with
 * BH_ERR_CAPACITY if one exceeded let_cap, otherwise it reports that in *overflow and returns BH_OK.
 * The flag is sticky over the interval (a check every N steps sees an overflow of ANY step in
 * between) and is also raised when the LOCAL tree outgrew node_capacity, because the send blocks are
 * then left as they were; reading the counters starts a new interval.  An overflowing LET is
 * truncated safely (links past the block are cut), so the step completes but its forces are wrong:
 * check the counts before trusting a run.  bh_let_configure may be called again with the same
 * rank/world and a new let_cap (size the blocks from measured counts).  bh_let_forces =
 * bh_let_walk without the integration.  fp32 and mixed precision. */
/* forest_base: where the received blocks start in a context's quad array.  A sender writes the child
 * links of a LET in the RECEIVER's index space, so this must be ONE number on all ranks: the largest
 * bh_let_local_quads of any rank (all_reduce MAX it once; contexts of equal capacity agree anyway). */
int bh_let_local_quads(bh_ctx *ctx, int64_t *local_quads);
int bh_let_configure(bh_ctx *ctx, int32_t rank, int32_t world, int64_t let_cap, int64_t forest_base);
int bh_let_bounds(bh_ctx *ctx);
int bh_let_pointers(bh_ctx *ctx, void **lbounds, void **all_bounds, void **send, void **recv,
                    int64_t *block_bytes, int32_t *boxes_per_rank);
int bh_let_build(bh_ctx *ctx);
int bh_let_walk(bh_ctx *ctx);
int bh_let_forces(bh_ctx *ctx);
/* bh_let_walk in two launches, so that the all_to_all can overlap the first: bh_let_walk_local walks
 * the local tree only (needs bh_let_build, not the received blocks); bh_let_walk_remote adds the
 * received LETs and finishes the step (integrate != 0) or only the forces.  Same sums in the same
 * order as bh_let_walk when the launch runs one wavefront per 64 bodies. */
int bh_let_walk_local(bh_ctx *ctx);
int bh_let_walk_remote(bh_ctx *ctx, int32_t integrate);
int bh_let_counts(bh_ctx *ctx, uint32_t *counts, int32_t *overflow);
/* --- device-side body migration and re-balancing for the LET scheme -------
 * The reference is single-GPU (project.cu:918-1024 keeps every body on one device); this is new design.
 * Ownership is defined by an orthogonal-recursive-bisection cut tree over the global root box: the node
 * that splits the ranks [r0, r0 + nr) (nr > 1) into nl = nr / 2 and nr - nl sends a body with
 * coordinate[axis] < value to the left.  Cuts are stored in pre-order: the left subtree's cuts follow
 * their parent directly (nl - 1 of them), the right subtree's come after those.
 *   bh_orb_histogram : for every region of depth `level` of the cut tree (cuts above it already fixed),
 *                      a histogram of this rank's bodies over BH_ORB_BINS bins across the ROOT box along
 *                      the region's axis (bin edges = depth-12 lines of the tree grid, so a cut taken
 *                      from it is a grid line), each body weighted by the cost of its 64-body group in the
 *                      last walk (1 before the first walk).  *hist = device pointer to n_cuts x
 *                      BH_ORB_BINS uint64 (row k = the region whose cut is k): the caller all-reduces it.
 *   bh_migrate_pack  : classify every local body by the cut tree, group the bodies by destination rank
 *                      (stable) into the send buffer -- 6 doubles per body: x, y, vx, vy, mass, id -- and
 *                      return the W counts (host; waits for the stream).
 *   [host: all_to_all of the counts, then ONE all_to_all of the records on the device pointers of
 *    bh_migrate_pointers with those splits; a rank's own group travels with the rest]
 *   bh_migrate_unpack: the received records become the local state (n_new bodies, arrival order =
 *                      caller order from here on); the tree is invalid until the next build.
 * bh_set_ids / bh_get_ids: a 64-bit identifier per body in caller order (default: the upload index);
 * it travels with the body. */
#define BH_ORB_BINS 4096
#define BH_ORB_MAX_CUTS 63
typedef struct bh_orb_cuts {
    int32_t world, n_cuts;           /* n_cuts = world - 1                                       */
    double  box[4];                  /* global root box: xmin, xmax, ymin, ymax                  */
    int32_t axis[BH_ORB_MAX_CUTS];   /* 0 = x, 1 = y                                             */
    int32_t pad;
    double  value[BH_ORB_MAX_CUTS];
} bh_orb_cuts;
int bh_set_ids(bh_ctx *ctx, const int64_t *ids);
int bh_get_ids(bh_ctx *ctx, int64_t *ids);
int bh_orb_histogram(bh_ctx *ctx, const bh_orb_cuts *cuts, int32_t level, void **hist, int64_t *n_words);
int bh_migrate_pack(bh_ctx *ctx, const bh_orb_cuts *cuts, int64_t *send_counts);
int bh_migrate_pointers(bh_ctx *ctx, void **send, void **recv, int64_t *capacity_records);
int bh_migrate_unpack(bh_ctx *ctx, int64_t n_new);
/* Run on an external HIP stream (e.g. torch's current stream), passed as void*. */
int bh_set_stream(bh_ctx *ctx, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BHGPU_H */
