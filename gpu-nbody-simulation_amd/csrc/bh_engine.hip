// bh_engine.hip -- libbhgpu.so: context, per-step pipeline and the C-ABI of include/bhgpu.h.
// Compiled with -ffp-contract=off (tree build + exact walk must not fuse multiply-adds; the fp32
// walk lives in bh_walk_fast.hip, compiled separately).  gfx950 only, no CPU fallback.
#include "../../include/bhgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "bh_tree.hpp"
#include "bh_walk_exact.hpp"
#include "bh_walk_f64.hpp"
#include "bh_init.hpp"
#include "bh_let.hpp"
#include "bh_migrate.hpp"
#include "bh_walk_fast.h"

using namespace bh;

// The one place the product translation unit knows about A/B builds: -DBHGPU_EXPERIMENTS (scripts/build_variants.sh) swaps the
// empty hook object for scripts/experiments/bh_engine_hooks.hpp (the per-wave timeline and clock stamps of the walk).
#ifdef BHGPU_EXPERIMENTS
#include "../../scripts/experiments/bh_engine_hooks.hpp"
#else
struct ExpHooks {
    hipError_t create(int64_t) { return hipSuccess; }
    void walk_args(WalkFastArgs &) const {}
    void destroy(int64_t) {}
};
#endif

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct bh_ctx {
    bh_config cfg{};
    int Dm = 0;
    bool exact = true, compat = true;
    bool exact_thr = false;        // BH_PRECISION_F64_EXACT: nodes carry exact d2 thresholds for the walk's acceptance test (off: BH_FLAG_WALK_PORTABLE)
    bool fast64 = false;           // BH_PRECISION_F64: the exact mode's tree and state, the throughput walk of bh_walk_f64.hpp
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0;
    bool uploaded = false, tree_valid = false;
    int64_t internal_cap = 0, node_cap = 0;
    int sort_passes = 0;
    bool state64 = false;          // fp64 state arrays: exact and mixed precision
    int64_t bfs_max = 12288;       // bit-exact walk: one wavefront per body (walk_exact_bfs_kernel) for launches up to here (BH_EXACT_BFS_MAX;
                                   // walk ms against the cooperative walk: 0.021 / 0.072 at N = 1,024, 0.042 / 0.108 at 4,096, 0.126 / 0.223
                                   // at 8,192, 0.180 / 0.221 at 12,288, 0.241 / 0.212 at 16,384)
    int exact_bpw = 0;             // BH_EXACT_BPW: bodies per wavefront in the fp64 walks (0 = by launch size; 64 = rounds 1-3)
    int walk_split = 0;            // 0 = automatic
    bool walk_asm = true;          // BH_WALK_ASM=0: the C++ loop everywhere (A/B)
    int sort_bucket = 1;           // 1: bucket sort when the previous build's sorted positions are this body set's
                                   // (BH_SORT_BUCKET=0: always the LSD passes; 2: always the bucket sort, tests)
    bool last_sort_bucket = false, last_sort_packed = false;   // what the last build's sort was (bh_stats bytes)
    int64_t samples_n = -1;        // spos holds the sorted positions of a build of this many bodies (-1: none)
    uint64_t *splitters = nullptr;
    uint16_t *sort_dig = nullptr;  // bucket of every key (written by the histogram, read by the scatter)
    int build_items = 0;           // 0 = automatic, else keys per thread in the sort / scan kernels (2, 4, 8)
    bool hilbert = false;                // fp32 mode: Hilbert-ordered keys (BH_HILBERT=0 disables, A/B)
    int partial_count = 0;         // > 0: partial[] holds per-workgroup min/max of the current positions
    double *bslots = nullptr;      // kBoundSlots running bounds records (bh_bounds.hpp)
    bool slots_valid = false;      // the last full-range fp32 walk folded the bounds of the current positions into bslots
    bool slots_dirty = true;       // bslots may hold something else than +-inf (written, not yet consumed by keys_kernel)

    // state (double2/double or float2/float).  Exact mode: caller order.  fp32 / mixed: DEVICE order --
    // every reorder_every-th build the bodies are physically permuted into that build's sorted order
    // (the gathers through `perm` in prep and in the walk's epilogue then touch neighbouring lines),
    // and orig[slot] remembers the caller's index; every host-facing call translates back.
    void *pos = nullptr, *vel = nullptr, *mass = nullptr, *force = nullptr;
    void *pos2 = nullptr, *vel2 = nullptr, *mass2 = nullptr, *force2 = nullptr;
    uint32_t *orig = nullptr, *orig2 = nullptr;
    bool orig_identity = true;
    int reorder_every = 16;        // BH_REORDER_EVERY; 0 = never
    int64_t builds = 0;            // builds since the last upload
    // sorted-order copies (fp32 mode)
    float2 *spos = nullptr;
    float4 *sstate = nullptr;      // sorted-order {x, y, vx, vy}: the replicated scheme's one exchange buffer
    float *smass = nullptr;
    // sort
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    uint64_t *keys_sorted = nullptr;
    uint32_t *perm = nullptr;
    uint32_t *radix_counts = nullptr, *bsum_sort = nullptr, *bsum_u32 = nullptr, *cnt = nullptr;
    uint64_t *coarse = nullptr;          // fp32: every 256th sorted key
    uint32_t *cell_first = nullptr;      // fp32: rank of a subdivided cell -> its first sorted body
    d3 *terms = nullptr, *bsum_d3 = nullptr;
    double *partial = nullptr, *box = nullptr;
    // tree
    NodeD *gd = nullptr;
    LinkD *ld = nullptr;
    QuadF *qf = nullptr;
    NodeAux *aux = nullptr;
    int32_t *self_node = nullptr, *cell_depth = nullptr;
    uint32_t *com_pending = nullptr;   // exact mode: subdivided children per cell still to be summed
    TreeCounters *ctr = nullptr;

    // ownership (multi-GPU): sorted range [lo, hi) = rank's share
    int rank = 0, world = 1;
    // distributed step with locally-essential trees (bh_let_*): this context holds only its own bodies
    bool let_mode = false, external_box = false;
    int64_t let_cap = 0, quads_local = 0;
    int64_t forest_base = 0;       // first quad of the received blocks: THE SAME ON EVERY RANK (>= quads_local)
    uint64_t *needmask = nullptr;
    uint32_t *let_tsum = nullptr, *let_outidx = nullptr;
    QuadF *let_send = nullptr;
    double *lbounds = nullptr, *all_bounds = nullptr;
    float2 *acc_part = nullptr;    // LET mode: raw sums of the local-tree walk (bh_let_walk_local)
    LetCounters *let_ctr = nullptr;
    // migration / re-balancing (bh_migrate.hpp)
    int64_t *gid = nullptr;             // 64-bit id per body, caller order
    uint32_t *group_cost = nullptr;     // cost of every 64-body group in the last walk (sorted order)
    const void **walk_consts = nullptr; // device block {aux, spos, smass, 0} for the assembly walk's bucket path
    bool aux_full = false;              // aux[] holds every node's record (after an export), not only the buckets'
    bool group_cost_valid = false;
    int64_t group_cost_n = 0;           // number of groups group_cost describes
    uint32_t *body_counts = nullptr;    // BH_FLAG_WALK_STATS: accepted force evaluations per body (device slot order)
    bool sort_pack = true;              // BH_SORT_PACK=0: separate key and index arrays in every pass (A/B)
    unsigned long long *orb_hist = nullptr;
    double *mig_send = nullptr, *mig_recv = nullptr;
    ExpHooks exp;                       // A/B builds only (-DBHGPU_EXPERIMENTS); an empty struct in the product

    // measurement
    std::vector<hipEvent_t> ev;        // pairs around the walk kernel, one pair per step
    hipEvent_t ev_step[2] = {nullptr, nullptr}, ev_build[2] = {nullptr, nullptr};
    hipEvent_t ev_grp[3] = {nullptr, nullptr, nullptr};   // after keys / sort / scan of the last timed build
    hipEvent_t ev_let[3] = {nullptr, nullptr, nullptr};   // bh_let_build: start, local tree built, LETs packed
    bool let_timed = false;
    bool time_groups = false;      // set by bh_step around its last build
    int64_t steps_done = 0;
    int32_t last_nsteps = 0;
    int timed_pairs = 0;
    int64_t walk_launches = 0;     // walk kernel launches of the last enqueue_walk
    bool step_timed = false;

    std::vector<void *> allocs;
    uint64_t device_bytes = 0;
    std::string err;
};

namespace {

int fail(bh_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define BH_HIP(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), BH_ERR_DEVICE,                                                     \
                        std::string(#call) + ": " + hipGetErrorString(e_));                     \
    } while (0)

template <typename T>
int dev_alloc(bh_ctx *c, T **out, size_t count)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail(c, BH_ERR_DEVICE, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    c->allocs.push_back(p);
    c->device_bytes += bytes;
    *out = static_cast<T *>(p);
    return BH_OK;
}

inline void dev_free(bh_ctx *c, void *p)
{
    if (!p) return;
    auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
    if (it != c->allocs.end()) { c->allocs.erase(it); (void)hipFree(p); }
}

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

void owned_range(const bh_ctx *c, int64_t *lo, int64_t *hi)
{
    if (c->let_mode) { *lo = 0; *hi = c->n; return; }     // a LET context holds only its own bodies
    // equal chunks of ceil(n/world) sorted slots rounded up to the workgroup size (the last ranks
    // may own fewer, or none): every rank then forms exactly the wavefronts the single-GPU run
    // forms, so per-body summation order -- and therefore every bit of the result -- does not
    // depend on the number of GPUs; and the all_gather moves one fixed-size block per rank
    const int64_t per = (c->n + c->world - 1) / c->world;
    const int64_t chunk = (per + kBlock - 1) / kBlock * kBlock;
    *lo = std::min<int64_t>(c->n, chunk * c->rank);
    *hi = std::min<int64_t>(c->n, chunk * (c->rank + 1));
}

constexpr int64_t kComClimbMaxBodies = 32768;     // exact mode: single-launch bottom-up mass pass up to here
constexpr int64_t kSmallBuildBodies = 327680;    // up to here: tiles of 512 instead of 2,048 (measured)
constexpr int64_t kMediumBuildBodies = 786432;   // up to here: tiles of 1,024
constexpr int kSmallItems = 2;

// Physically permute the state into the sorted order of the build that has just produced `perm`
// (sorted index -> slot); afterwards slot == sorted index, so perm becomes the identity.
template <typename Real2, typename Real>
__global__ __launch_bounds__(kBlock) void reorder_state_kernel(uint32_t *__restrict__ perm, const Real2 *__restrict__ pos,
                                                                const Real2 *__restrict__ vel, const Real *__restrict__ mass,
                                                                const float2 *__restrict__ acc, const uint32_t *__restrict__ orig,
                                                                Real2 *__restrict__ pos2, Real2 *__restrict__ vel2,
                                                                Real *__restrict__ mass2, float2 *__restrict__ acc2,
                                                                uint32_t *__restrict__ orig2, int64_t n)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t b = perm[s];
    pos2[s] = pos[b]; vel2[s] = vel[b]; mass2[s] = mass[b]; acc2[s] = acc[b];
    orig2[s] = orig ? orig[b] : b;                            // orig == nullptr: slots still are caller indices
    perm[s] = (uint32_t)s;
}

// fp32 node kernel.  full_aux: also the {first body, count} record of every node (the host export needs them;
// a step only needs those of bucket leaves).  Inputs are what the build left on the device, so it can be run
// again after the build (export_tree_host).
static void launch_nodes_fast(bh_ctx *c, bool full_aux, hipStream_t st)
{
    const int64_t n = c->n;
    const int Dm = c->Dm;
    // one thread per subdivided cell; I <= (n-1)*Dm and <= internal_cap
    const int64_t span = std::max<int64_t>(1, std::min<int64_t>(c->internal_cap, std::max<int64_t>(n - 1, 0) * (int64_t)std::max(1, Dm)));
    const unsigned nbc = blocks_for(span, kBlock);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(nbc), dim3(kBlock), 0, st, c->keys_sorted, c->coarse, c->cnt, c->cell_first,
                           c->spos, c->smass, c->box, c->terms, n, Dm, c->cfg.theta, c->internal_cap, c->qf, c->aux,
                           c->ctr);
    };
    if (c->compat) { if (full_aux) go(nodes_fast_kernel<true, true>); else go(nodes_fast_kernel<true, false>); }
    else { if (full_aux) go(nodes_fast_kernel<false, true>); else go(nodes_fast_kernel<false, false>); }
    c->aux_full = full_aux;
}

template <bool EXACT, bool STATE64 = EXACT, int ITEMS = kItems>
int enqueue_build_t(bh_ctx *c)
{
    constexpr int TILE = kBlock * ITEMS;
    static_assert(!EXACT || STATE64, "exact mode keeps its state in fp64");
    using Real2 = typename std::conditional<STATE64, double2, float2>::type;   // the state
    using Real = typename std::conditional<STATE64, double, float>::type;
    const int64_t n = c->n;
    const int Dm = c->Dm;
    hipStream_t st = c->stream;
    const Real2 *pos = static_cast<const Real2 *>(c->pos);     // (re-read after a reorder)
    const Real *mass = static_cast<const Real *>(c->mass);

    // 1. root box (ComputeRootBounds, project.cu:536-573); the per-workgroup partials usually
    //    come from the previous step's walk epilogue
    const bool from_slots = !c->external_box && c->slots_valid && n >= 2;
    double *slots = from_slots ? c->bslots : nullptr;
    if (from_slots) {
        c->partial_count = 0; c->slots_valid = false; c->slots_dirty = false;   // (keys_kernel below consumes and resets them)
    } else if (!c->external_box) {
        const bool from_walk = c->partial_count > 0;
        if (!from_walk) {
            const unsigned nbb = std::max(1u, std::min(1024u, blocks_for(n, kBlock)));
            hipLaunchKernelGGL((bounds_partial<Real2>), dim3(nbb), dim3(kBlock), 0, st, pos, n, c->partial);
            c->partial_count = (int)nbb;
        }
        hipLaunchKernelGGL(bounds_final, dim3(1), dim3(kBlock), 0, st, c->partial, c->partial_count, c->box, c->ctr, Dm, from_walk ? 1 : 0);
        c->partial_count = 0; c->slots_valid = false;
    }   // else: let_box_kernel has set the global box and cleared the counters

    if (n > 0) {
        // 2. keys by fp64 bisection, 3. stable radix sort
        // keys of <= 40 bits for <= 2^24 bodies carry the body index in the key word through the sort
        const bool pack = Dm >= 1 && 2 * Dm <= kPackShift && n <= ((int64_t)1 << (64 - kPackShift))
                          && c->sort_pack;
        // bucket sort (bh_sort.hpp): one counting pass by splitters from the previous build + one in-LDS sort per bucket
        // (exact modes and BH_HILBERT=0 too since round 3: the splitters only have to be sorted, whatever curve the keys follow)
        // a launch of up to 4,096 keys is ONE bucket: no splitters, no counting pass (three launches of ~5 us less), the first build
        // included.  (One workgroup's LDS holds 12,288 keys, but at twelve keys per thread its chain of passes is longer than the
        // counting pass it saves: N = 12,288 measured 0.100 ms per build this way against 0.090 at N = 16,384 the other way.
        // BH_SORT_BUCKET=2 keeps the splitter path at every size, for the tests that compare it with the LSD passes.)
        const bool single = pack && n >= 2 && n <= 4 * kBsThreads && c->sort_bucket == 1;
        const bool bucket = !single && pack && n >= 2 && n <= kBucketMaxNBig && (c->sort_bucket == 2 || (c->sort_bucket == 1 && c->samples_n == n));
        const int nb = (n <= kBucketMaxN) ? kBuckets : kBucketsBig;
        // sample positions behind the splitters: two per bucket where the key workgroups run long enough to hide the
        // sample workgroups (256 buckets: above 262k bodies; 1,024 buckets: above 3M), one per bucket otherwise
        const int ns = (nb == kBuckets) ? ((n > (int64_t)1 << 18) ? 2 * kBuckets : kBuckets)
                                        : ((n > (int64_t)3 << 20) ? kMaxSplitSamples : kBucketsBig);
        static_assert(kBucketMaxN == (int64_t)1 << 20 && kBucketMaxNBig == (int64_t)1 << 22, "BASELINE configs 3 and 4 fit");
        c->last_sort_bucket = bucket || single; c->last_sort_packed = pack;
        {
            const unsigned nkb = blocks_for(n, kBlock);
            auto keys_launch = [&](auto hil, auto pk, auto fs) {
                constexpr bool H = decltype(hil)::value, P = decltype(pk)::value, F = decltype(fs)::value;
                const bool smp = P && bucket;                       // (the splitter workgroups go with the packed keys)
                // samples: the previous build's sorted positions -- the fp32 walk's copy, or through the previous perm in the exact modes
                hipLaunchKernelGGL((keys_kernel<Real2, H, P, F>), dim3(nkb + (smp ? ns / kWave : 0)), dim3(kBlock), 0, st, pos, c->box,
                                   c->keys[0], c->vals[0], n, Dm, (smp && !EXACT) ? (const float2 *)c->spos : nullptr, c->splitters, nb, ns,
                                   slots, c->ctr, smp ? c->bsum_sort : nullptr, smp ? nb : 0, (smp && EXACT) ? c->perm : nullptr);
            };
            using T = std::true_type; using Fz = std::false_type;
            // (exact mode and BH_HILBERT=0: child-index keys, packed all the same)
            if (from_slots) {
                if (pack && c->hilbert) keys_launch(T{}, T{}, T{}); else if (pack) keys_launch(Fz{}, T{}, T{});
                else if (c->hilbert) keys_launch(T{}, Fz{}, T{}); else keys_launch(Fz{}, Fz{}, T{});
            } else {
                if (pack && c->hilbert) keys_launch(T{}, T{}, Fz{}); else if (pack) keys_launch(Fz{}, T{}, Fz{});
                else if (c->hilbert) keys_launch(T{}, Fz{}, Fz{}); else keys_launch(Fz{}, Fz{}, Fz{});
            }
        }
        if (c->time_groups) (void)hipEventRecord(c->ev_grp[0], st);
        const unsigned nbl = blocks_for(n, ITEMS == kItems ? kSortTile : TILE);
        int cur = 0;
        if (single) {
            hipLaunchKernelGGL(bucket_sort_kernel, dim3(1), dim3(kBsThreads), 0, st, c->keys[0], c->keys[1], c->vals[1],
                               nullptr, nullptr, &c->ctr->sort_spills, &c->ctr->sort_reruns, (int)n);
            cur = 1;
        } else if (bucket) {
            constexpr int SI = (ITEMS == kItems ? kSortItems : ITEMS);
            auto pass = [&](auto bits_tag) {
                constexpr int NBITS = decltype(bits_tag)::value;
                if constexpr (EXACT) {                               // (state in caller order: the pass with its row scan, see radix_scatter_w)
                    hipLaunchKernelGGL((radix_hist<SI, NBITS, true>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->radix_counts, n,
                                       0, (int)nbl, c->splitters, c->sort_dig, nullptr);
                    hipLaunchKernelGGL(radix_rowscan, dim3(1 << NBITS), dim3(kBlock), 0, st, c->radix_counts, c->bsum_sort, (int)nbl);
                    hipLaunchKernelGGL((radix_scatter_w<SI, NBITS, 1, true, false>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->vals[0],
                                       c->keys[1], c->vals[1], c->radix_counts, c->bsum_sort, n, 0, (int)nbl, c->sort_dig,
                                       c->bsum_sort + kBucketStartOffset);
                } else {
                    hipLaunchKernelGGL((radix_hist<SI, NBITS, true>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->radix_counts, n,
                                       0, (int)nbl, c->splitters, c->sort_dig, c->bsum_sort);
                    hipLaunchKernelGGL((radix_scatter_w<SI, NBITS, 1, true>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->vals[0],
                                       c->keys[1], c->vals[1], c->radix_counts, c->bsum_sort, n, 0, (int)nbl, c->sort_dig,
                                       c->bsum_sort + kBucketStartOffset);
                }
            };
            if (nb == kBuckets) pass(std::integral_constant<int, 8>{}); else pass(std::integral_constant<int, 10>{});
            hipLaunchKernelGGL(bucket_sort_kernel, dim3(nb), dim3(kBsThreads), 0, st, c->keys[1], c->keys[0], c->vals[0],
                               c->bsum_sort, c->bsum_sort + kBucketStartOffset, &c->ctr->sort_spills, &c->ctr->sort_reruns, 0);
            cur = 0;
        } else {
            // kSortBits-wide digits, wave-private ranking, digit-sorted write-out
            constexpr int SB = kSortBits, SR = 1 << SB, SI = (ITEMS == kItems ? kSortItems : ITEMS);
            const int passes = (2 * Dm + SB - 1) / SB;
            for (int p = 0; p < passes; ++p) {
                const int shift = p * SB;
                hipLaunchKernelGGL((radix_hist<SI, SB>), dim3(nbl), dim3(kBlock), 0, st, c->keys[cur], c->radix_counts, n,
                                   shift, (int)nbl);
                hipLaunchKernelGGL(radix_rowscan, dim3(SR), dim3(kBlock), 0, st, c->radix_counts, c->bsum_sort, (int)nbl);
                if (pack && p + 1 < passes)
                    hipLaunchKernelGGL((radix_scatter_w<SI, SB, 1>), dim3(nbl), dim3(kBlock), 0, st, c->keys[cur], c->vals[cur],
                                       c->keys[cur ^ 1], c->vals[cur ^ 1], c->radix_counts, c->bsum_sort, n, shift, (int)nbl);
                else if (pack)
                    hipLaunchKernelGGL((radix_scatter_w<SI, SB, 2>), dim3(nbl), dim3(kBlock), 0, st, c->keys[cur], c->vals[cur],
                                       c->keys[cur ^ 1], c->vals[cur ^ 1], c->radix_counts, c->bsum_sort, n, shift, (int)nbl);
                else
                    hipLaunchKernelGGL((radix_scatter_w<SI, SB>), dim3(nbl), dim3(kBlock), 0, st, c->keys[cur], c->vals[cur],
                                       c->keys[cur ^ 1], c->vals[cur ^ 1], c->radix_counts, c->bsum_sort, n, shift,
                                       (int)nbl);
                cur ^= 1;
            }
        }
        c->keys_sorted = c->keys[cur];
        c->perm = c->vals[cur];
        if constexpr (!EXACT) {
            if (c->reorder_every > 0 && c->builds % c->reorder_every == 0) {
                hipLaunchKernelGGL((reorder_state_kernel<Real2, Real>), dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, st,
                                   c->perm, pos, static_cast<const Real2 *>(c->vel), mass,
                                   static_cast<const float2 *>(c->force), c->orig_identity ? nullptr : c->orig,
                                   static_cast<Real2 *>(c->pos2), static_cast<Real2 *>(c->vel2),
                                   static_cast<Real *>(c->mass2), static_cast<float2 *>(c->force2), c->orig2, n);
                std::swap(c->pos, c->pos2); std::swap(c->vel, c->vel2); std::swap(c->mass, c->mass2);
                std::swap(c->force, c->force2); std::swap(c->orig, c->orig2);
                c->orig_identity = false;
                pos = static_cast<const Real2 *>(c->pos);
                mass = static_cast<const Real *>(c->mass);
            }
        }
        c->builds += 1;
        if (c->time_groups) (void)hipEventRecord(c->ev_grp[1], st);

        // 4. cells owned by each sorted neighbour pair (+ fp32: sorted copies and prefix-sum terms),
        // 5. their ranks / the prefix sums
        using SReal2 = typename std::conditional<EXACT, double2, float2>::type;    // what the walk reads
        using SReal = typename std::conditional<EXACT, double, float>::type;
        // elements per thread in prep / scan_apply2: the sort's tile size, except that between 768k and
        // 2M bodies tiles of 1,024 are still the better choice for these two (a workgroup's rows are a
        // chain of load -> scan -> store; measured at N = 1M: 42.5 -> 33 us for the pair)
        auto scan_part = [&](auto si_tag) {
            constexpr int SI = decltype(si_tag)::value;
            const unsigned nbs = blocks_for(n + 1, kBlock * SI);
            hipLaunchKernelGGL((prep_kernel<EXACT, SI, Real2, Real, SReal2, SReal>), dim3(nbs), dim3(kBlock), 0, st,
                               c->keys_sorted, c->perm, pos, mass, c->cnt, c->bsum_u32, (SReal2 *)c->spos,
                               (SReal *)c->smass, c->terms, c->bsum_d3, c->coarse, n, Dm, slots);
            constexpr bool TSRC = !EXACT && std::is_same<Real2, SReal2>::value;   // fp32 state: terms from the sorted copies
            if (nbs <= 8u * kBlock) {
                // few tiles: every workgroup sums the tile totals before it itself (no scan_top2 launch)
                hipLaunchKernelGGL((scan_apply2<EXACT, SI, true, TSRC>), dim3(nbs), dim3(kBlock), 0, st, c->cnt, c->bsum_u32,
                                   c->terms, c->bsum_d3, (int)nbs, n, c->cell_first, c->internal_cap, c->ctr,
                                   (const float2 *)c->spos, (const float *)c->smass);
            } else {
                hipLaunchKernelGGL(scan_top2, dim3(EXACT ? 1 : 2), dim3(kBlock), 0, st, c->bsum_u32, c->bsum_d3,
                                   (int)nbs, c->ctr);
                hipLaunchKernelGGL((scan_apply2<EXACT, SI, false, TSRC>), dim3(nbs), dim3(kBlock), 0, st, c->cnt, c->bsum_u32,
                                   c->terms, c->bsum_d3, (int)nbs, n, c->cell_first, c->internal_cap, c->ctr,
                                   (const float2 *)c->spos, (const float *)c->smass);
            }
        };
        if (ITEMS == kItems && n <= (int64_t)1 << 21) scan_part(std::integral_constant<int, 4>{});
        else scan_part(std::integral_constant<int, ITEMS>{});
        c->samples_n = n;                                        // spos (exact modes: perm): this build's sorted order
        if (c->time_groups) (void)hipEventRecord(c->ev_grp[2], st);
    } else {
        c->keys_sorted = c->keys[0];
        c->perm = c->vals[0];
        if (c->time_groups) for (auto e : c->ev_grp) (void)hipEventRecord(e, st);
    }

    // 6. nodes (thread 0 writes the root when nothing is subdivided)
    if constexpr (EXACT) {
        // one thread per subdivided cell; I <= (n-1)*Dm and <= internal_cap (at least one thread: the root-only case)
        const int64_t span = std::max<int64_t>(1, std::min<int64_t>(c->internal_cap, std::max<int64_t>(n - 1, 0) * (int64_t)std::max(1, Dm)));
        hipLaunchKernelGGL(nodes_exact_kernel, dim3(blocks_for(span, kBlock)), dim3(kBlock), 0, st, c->keys_sorted, c->perm,
                           c->cnt, c->cell_first, pos, mass, c->box, n, Dm, c->internal_cap, c->gd, c->ld, c->self_node,
                           c->cell_depth, c->com_pending, c->ctr, (c->fast64 || c->exact_thr) ? c->cfg.theta : 0.0,
                           c->exact_thr ? 1 : 0);
    } else {
        launch_nodes_fast(c, false, st);
    }
    // 7. exact bottom-up mass pass (ComputeMass, project.cu:473-502): ONE launch for small trees (the climb of
    //    com_up_kernel), one launch per depth for large ones, where the climb's coherence traffic costs more
    //    than the launches (measured cross-over ~65k bodies; see com_level_kernel).  Bitwise the same sums.
    if (EXACT && n > 1 && c->internal_cap > 0) {
        const int64_t span = std::min<int64_t>(c->internal_cap, std::max<int64_t>(1, (n - 1) * (int64_t)std::max(1, Dm)));
        if (n <= kComClimbMaxBodies) {
            hipLaunchKernelGGL(com_up_kernel, dim3(blocks_for(span, kBlock)), dim3(kBlock), 0, st, c->gd, c->ld,
                               c->self_node, c->com_pending, c->ctr, c->internal_cap);
        } else {
            for (int d = Dm - 1; d >= 0; --d)
                hipLaunchKernelGGL(com_level_kernel, dim3(blocks_for(span, kBlock)), dim3(kBlock), 0, st, c->gd,
                                   c->self_node, c->cell_depth, c->ctr, c->internal_cap, d);
        }
    }
    BH_HIP(c, hipGetLastError());
    c->tree_valid = true;
    return BH_OK;
}

template <int ITEMS>
static int enqueue_build_items(bh_ctx *c)
{
    if (c->exact) return enqueue_build_t<true, true, ITEMS>(c);
    return c->state64 ? enqueue_build_t<false, true, ITEMS>(c) : enqueue_build_t<false, false, ITEMS>(c);
}

int enqueue_build(bh_ctx *c)
{
    // keys per thread in the sort / scan kernels: launches of few bodies take smaller tiles (more
    // workgroups, fewer sequential rounds in each); BH_BUILD_ITEMS = 2, 4 or 8 overrides
    int items = c->build_items;
    if (items == 0) items = c->n <= kSmallBuildBodies ? 2 : c->n <= kMediumBuildBodies ? 4 : 8;
    if (c->n > (1 << 22)) items = 8;         // (scratch for small tiles is sized for 4M bodies)
    switch (items) {
    case 2: return enqueue_build_items<2>(c);
    case 4: return enqueue_build_items<4>(c);
    default: return enqueue_build_items<kItems>(c);
    }
}

// fp64 walks (exact and throughput): bodies per wavefront for a launch of `cnt` bodies.  A wave's walk is one dependent chain
// over the union of its bodies' walks (~900 node visits for 64 bodies, ~150 for one), and up to ~130k bodies the launch cannot
// fill the GPU's 8,192 wave slots with 64-body waves anyway.  Measured (scripts/bpw_ab.py, profiles/r04_f64/bpw_sweep.txt): the
// best number of waves is ~4,096 for the throughput walk: the smallest power of two that stays below that.  The bit-exact walk
// (round 4's assembly loop, ~50 vector instructions per visit; scripts/bpw_ab.py -> profiles/r04_exact/bpw_sweep.txt): one body
// per wave up to 4,096 bodies, from there 16 bodies per wave or what keeps the launch within ~2,048 waves (more waves than that
// cost more vector work than the shorter chains return).  Always within what `partial` holds (one record per workgroup).  The
// bit-exact mode's results do not depend on it, bit for bit.
static int exact_bodies_per_wave(const bh_ctx *c, int64_t cnt)
{
    // (the throughput walk adds a lane's terms in the order its WAVE meets them: its last bits depend on who shares the wave, like
    // the fp32 split walk's -- BH_FLAG_WALK_NO_SPLIT pins 64 bodies per wave for callers that need launch-size independence)
    if (c->fast64 && (c->cfg.flags & BH_FLAG_WALK_NO_SPLIT)) return kWave;
    int b = 1;
    if (c->exact_bpw > 0) b = c->exact_bpw;
    else if (c->fast64) {
        while (b < kWave && (int64_t)b * 4096 < cnt) b <<= 1;
    } else if (cnt > 4096) {
        b = 16;
        while (b < kWave && (int64_t)b * 2048 < cnt) b <<= 1;
    }
    const int64_t room = std::max<int64_t>(1024, (c->cfg.capacity + kWave - 1) / kWave);      // records in `partial`
    while (b < kWave && (cnt + (int64_t)kWavesPerBlock * b - 1) / ((int64_t)kWavesPerBlock * b) > room) b <<= 1;
    return b;
}

int enqueue_walk(bh_ctx *c, bool integrate, bool to_sorted, int part = 0)
{
    int64_t lo, hi;
    owned_range(c, &lo, &hi);
    if (hi <= lo) return BH_OK;
    const bool stats = (c->cfg.flags & BH_FLAG_WALK_STATS) != 0;
    if (part != 2) c->walk_launches = 0;
    if (integrate) c->slots_valid = false;                         // (the positions change; the fp32 branch may set it again)
    // a full-range integrating walk also leaves the min/max of the NEW positions per workgroup
    const bool want_partial = integrate && !to_sorted && lo == 0 && hi == c->n;
    double *partial = want_partial ? c->partial : nullptr;
    int per_partial = kBlock;
    int partial_records = -1;                                      // (a walk whose workgroups do not take a fixed number of bodies says so itself)
    // N_THREADS (project.cu:5-7, 703: `body_i += N_THREADS`): at most that many bodies are walked at a time -- the
    // range is taken in passes of n_threads bodies, rounded up to whole 256-thread workgroups, one launch after the
    // other on the stream, as the reference's threads take their bodies one after the other.  0 (the default):
    // one pass.  It gives the thread axis of the reference's first scaling experiment (first_scaling_script.sh:
    // 17-36) a meaning on this hardware: n_threads = 1 is one workgroup at a time.
    const int64_t pass = c->cfg.n_threads > 0 ? ((int64_t)c->cfg.n_threads + kBlock - 1) / kBlock * kBlock : hi - lo;
    // one launch over all bodies that integrates: the workgroups also fold their bounds into the slot records the
    // next keys_kernel reduces (bh_bounds.hpp) -- that build then needs no bounds_final launch
    const bool want_slots = want_partial && pass == hi - lo && part == 0 && !c->let_mode && !c->external_box && c->n >= 2;
    double *slots = nullptr;
    if (want_slots) {
        if (c->slots_dirty)
            hipLaunchKernelGGL(bounds_slots_reset, dim3(1), dim3(kWave), 0, c->stream, c->bslots);
        slots = c->bslots; c->slots_dirty = true;
    }
    if (integrate) c->slots_valid = want_slots;
    if (c->exact && c->fast64) {
        if (stats) {
            if (!c->body_counts) { int rc = dev_alloc(c, &c->body_counts, (size_t)std::max<int64_t>(c->cfg.capacity, 1)); if (rc) return rc; }
            // (a launch writes the slots of the bodies it walks: an owned range smaller than n, or a walk that returned
            // early on an overflowed tree, must not leave the others uninitialised)
            BH_HIP(c, hipMemsetAsync(c->body_counts, 0, (size_t)std::max<int64_t>(c->n, 1) * sizeof(uint32_t), c->stream));
        }
        // hand-written loop (walk64_asm): 32-bit byte offsets into the node array; the counting variant and
        // BH_FLAG_WALK_PORTABLE run the C++ statement of the same loop (same operations, same order, same bits)
        const bool use_asm = c->walk_asm && !stats && !(c->cfg.flags & BH_FLAG_WALK_PORTABLE) &&
                             c->node_cap * (int64_t)sizeof(NodeD) < (1ll << 32);
        const bool deep = 3 * c->Dm + 1 > kWave;                 // (deeper than 21 levels: the two-tier stack)
        const int bpw = exact_bodies_per_wave(c, std::min(pass, hi - lo));
        const int per_block = (kF64Block / kWave) * bpw;         // bodies per workgroup
        for (int64_t plo = lo; plo < hi; plo += pass) {
            const int64_t phi = std::min(hi, plo + pass);
            double *pp = partial ? partial + 4 * ((plo - lo) / per_block) : nullptr;
            WalkF64Args wa{};
            wa.gd = c->gd; wa.ld = c->ld; wa.perm = c->perm; wa.pos = (double2 *)c->pos; wa.vel = (double2 *)c->vel;
            wa.mass = (const double *)c->mass; wa.force_out = (double2 *)c->force; wa.lo = plo; wa.hi = phi;
            wa.G = c->cfg.G; wa.dt = c->cfg.dt; wa.integrate = integrate ? 1 : 0; wa.ctr = c->ctr; wa.partial = pp;
            wa.body_counts = stats ? c->body_counts : nullptr; wa.slots = slots; wa.bpw = bpw;
            auto args = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(blocks_for(phi - plo, per_block)), dim3(kF64Block), 0, c->stream, wa);
                c->walk_launches += 1;
            };
            auto pick = [&](auto compat_tag, auto deep_tag) {
                constexpr bool CP = decltype(compat_tag)::value, DP = decltype(deep_tag)::value;
                if (stats) args(walk_f64_kernel<CP, true, DP, false>);
                else if (use_asm) args(walk_f64_kernel<CP, false, DP, true>);
                else args(walk_f64_kernel<CP, false, DP, false>);
            };
            using T = std::true_type; using Fz = std::false_type;
            if (c->compat) { if (deep) pick(T{}, T{}); else pick(T{}, Fz{}); }
            else           { if (deep) pick(Fz{}, T{}); else pick(Fz{}, Fz{}); }
        }
        per_partial = per_block;
        BH_HIP(c, hipGetLastError());
    } else if (c->exact && c->exact_thr && !stats && c->exact_bpw == 0 && pass == hi - lo && hi - lo <= c->bfs_max &&
               c->node_cap * (int64_t)sizeof(NodeD) < (1ll << 32)) {
        // launches of a few thousand bodies: one wavefront per BODY, its tree breadth-first (walk_exact_bfs_kernel) -- the same
        // bits as the cooperative walk below, which an explicit BH_EXACT_BPW, the counting variant and the portable walk keep using
        // a workgroup = four wavefronts, each taking bodies (turn * grid + workgroup) * 4 + wave one after the other; as many
        // workgroups as `partial` has records (at least 1,024), every wave at most 64 bodies
        const int64_t cnt = std::min(pass, hi - lo);
        const int64_t room = std::max<int64_t>(1024, (c->cfg.capacity + kWave - 1) / kWave);
        const int64_t grid = std::max<int64_t>(std::min<int64_t>(blocks_for(cnt, kWavesPerBlock), room),
                                               blocks_for(cnt, kWavesPerBlock * kBfsBodiesPerWave));
        {
            auto args = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), 0, c->stream, c->gd, c->ld, c->perm, (double2 *)c->pos,
                                   (double2 *)c->vel, (const double *)c->mass, (double2 *)c->force, lo, hi, c->cfg.G, c->cfg.dt,
                                   integrate ? 1 : 0, c->ctr, partial, slots, c->box);
                c->walk_launches += 1;
            };
            const bool big = cnt > 4096;                         // (more interactions per body in larger trees: 384 terms per walk -- 52 KB of LDS per workgroup, three per CU; 512 would leave two)
            if (c->compat) { if (big) args(walk_exact_bfs_kernel<true, 384>); else args(walk_exact_bfs_kernel<true, 256>); }
            else           { if (big) args(walk_exact_bfs_kernel<false, 384>); else args(walk_exact_bfs_kernel<false, 256>); }
        }
        partial_records = (int)grid;
        BH_HIP(c, hipGetLastError());
    } else if (c->exact) {
        const int bpw = exact_bodies_per_wave(c, std::min(pass, hi - lo));
        const int per_block = kWavesPerBlock * bpw;              // bodies per workgroup
        for (int64_t plo = lo; plo < hi; plo += pass) {
            const int64_t phi = std::min(hi, plo + pass);
            double *pp = partial ? partial + 4 * ((plo - lo) / per_block) : nullptr;
            auto args = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(blocks_for(phi - plo, per_block)), dim3(kBlock), 0, c->stream, c->gd, c->ld, c->perm,
                                   (double2 *)c->pos, (double2 *)c->vel, (const double *)c->mass,
                                   (double2 *)c->force, plo, phi, c->cfg.theta, c->cfg.G, c->cfg.dt,
                                   integrate ? 1 : 0, c->ctr, pp, slots, bpw, c->box);
                c->walk_launches += 1;
            };
            // (the node kernel stored what this walk reads in the size slot: exact thresholds, or the sizes for the portable
            // walk; the hand-written loop uses 32-bit byte offsets into the node array and carries no counters)
            const bool use_asm = c->exact_thr && !stats && c->node_cap * (int64_t)sizeof(NodeD) < (1ll << 32);
            auto pick = [&](auto compat_tag) {
                constexpr bool CP = decltype(compat_tag)::value;
                if (!c->exact_thr) { if (stats) args(walk_exact_kernel<CP, true, false, false>); else args(walk_exact_kernel<CP, false, false, false>); }
                else if (stats) args(walk_exact_kernel<CP, true, true, false>);
                else if (use_asm) args(walk_exact_kernel<CP, false, true, true>);
                else args(walk_exact_kernel<CP, false, true, false>);
            };
            if (c->compat) pick(std::true_type{}); else pick(std::false_type{});
        }
        per_partial = per_block;
        BH_HIP(c, hipGetLastError());
    } else {
        WalkFastArgs a{};
        a.quads = c->qf; a.aux = c->aux; a.partial = partial; a.spos = c->spos; a.smass = c->smass; a.perm = c->perm;
        a.pos = (float2 *)c->pos; a.vel = (float2 *)c->vel;
        a.state64 = c->state64 ? 1 : 0;               // mixed precision: pos/vel point at double2 arrays
        a.sstate = c->sstate;
        a.acc_out = (float2 *)c->force; a.ctr = c->ctr;
        a.lo = lo; a.hi = hi; a.G = (float)c->cfg.G; a.dt = (float)c->cfg.dt;
        a.integrate = integrate ? 1 : 0; a.to_sorted = to_sorted ? 1 : 0;
        a.n_trees = c->let_mode ? c->world : 0; a.self_rank = c->let_mode ? c->rank : -1;
        a.part = part; a.acc_part = c->acc_part;
        a.forest_base = c->forest_base; a.let_cap = c->let_cap;
        c->exp.walk_args(a);
        a.group_cost = (lo == 0 && hi == c->n) ? c->group_cost : nullptr;
        a.bucket_consts = c->walk_consts;
        a.body_counts = nullptr;
        a.slots = slots;
        if (stats) {
            if (!c->body_counts) { int rc = dev_alloc(c, &c->body_counts, (size_t)std::max<int64_t>(c->cfg.capacity, 1)); if (rc) return rc; }
            if (part != 2) BH_HIP(c, hipMemsetAsync(c->body_counts, 0, (size_t)std::max<int64_t>(c->n, 1) * sizeof(uint32_t), c->stream));
            a.body_counts = c->body_counts;
        }
        // the register-lane stack holds 128 entries and pairs entries only while the bound of
        // walk_tree_asm allows it, so it serves every max_depth <= 32; the LDS stack is the flag's variant
        const bool lds = (c->cfg.flags & BH_FLAG_LDS_STACK) != 0;
        a.pair_limit = std::max(0, 116 - 3 * c->Dm);   // the stack bound of walk_tree_asm2 (bh_walk_fast.hip)
        // few bodies: several waves per 64-body group (bh_walk_fast.hip).  Measured best factor
        // (scripts/split_ab.sh): 8 up to 32k bodies per launch, 4 up to ~100k, one wave
        // per group -- the hand-scheduled loop with two quads in flight -- beyond (round 1's compiled loop
        // lost to the split walk up to 192k; at 131k: 0.097 against 0.102 ms).  BH_WALK_SPLIT overrides (1 = off).
        int split = (c->cfg.flags & BH_FLAG_WALK_NO_SPLIT) || c->cfg.n_threads > 0 ? 1 : c->walk_split;   // (n_threads: one thread per body)
        if (split <= 0) {
            const int64_t groups = (hi - lo + kWave - 1) / kWave;
            // (a forest walk keeps the split longer: the level-synchronous walk seeds its first frontier with
            // all the roots, the one-wave loop walks tree after tree -- 8 ranks x 135k bodies: 0.270 vs 0.293 ms;
            // 4 ranks x 268k: 0.346 vs 0.298)
            // (measured with the hand-scheduled chunk loop, Plummer, walk ms for split 1 / 2 / 4 / 8 --
            //  768 groups: .089 .061 .042 .040; 1,536: .094 .073 .059 .074; 2,048: .098 .074 .078 .100;
            //  3,072: .108 .115 .102 .136; 3,584: .115 .127 .118 .157; profiles/r02_final/split_sweep.txt)
            split = groups <= (c->let_mode ? 512 : 768) ? 8 : groups <= 3072 ? 4 : 1;
        }
        if (3 * c->Dm + 2 > kWave) split = 1;        // the level-synchronous walk's depth-first fallback has 64 entries
        // hand-scheduled loop: byte offsets into the quad array and the sorted bodies are 32-bit there, and the SGPR
        // offset of s_load is an UNSIGNED 32-bit value on gfx950 (scripts/calib/soffset_calib.hip, profiles/r03_final/
        // soffset_calib.txt: offsets up to 0xf0000100 read base + offset): 4 GiB of quads = 53.6 M.  (Round 2 stopped at
        // 2 GiB, so BASELINE config 5 -- capacity 33.5 M quads -- ran the C++ loop.)
        const int64_t forest_quads = c->let_mode ? c->forest_base + (int64_t)c->world * c->let_cap : c->internal_cap + 1;
        const bool use_asm = c->walk_asm && !(c->cfg.flags & BH_FLAG_WALK_PORTABLE) &&
                             forest_quads * (int64_t)sizeof(QuadF) < (1ll << 32) && c->n < (1ll << 28);
        if (walk_fast_split_effective(a, lds, split)) per_partial = kWave;
        for (int64_t plo = lo; plo < hi; plo += pass) {
            a.lo = plo; a.hi = std::min(hi, plo + pass);
            a.partial = partial ? partial + 4 * ((plo - lo) / per_partial) : nullptr;
            BH_HIP(c, launch_walk_fast(a, lds, stats, split, use_asm, c->stream));
            c->walk_launches += 1;
        }
    }
    if (want_partial) c->partial_count = partial_records >= 0 ? partial_records : (int)blocks_for(hi - lo, per_partial);
    if (!c->exact && lo == 0 && hi == c->n) c->group_cost_valid = true;
    return BH_OK;
}

int check_overflow(bh_ctx *c)
{
    TreeCounters h{};
    BH_HIP(c, hipMemcpyAsync(&h, c->ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    if (h.overflow || (int64_t)h.n_internal > c->internal_cap)
        return fail(c, BH_ERR_CAPACITY, "tree needs " + std::to_string(1 + 4 * (int64_t)h.n_internal) +
                                        " nodes, node_capacity is " + std::to_string(c->node_cap));
    return BH_OK;
}

}  // namespace

// ================================================================================================
static int download_pairs(bh_ctx *c, const void *dev, double *host, int64_t count, bool is64);
static int to_caller_order(bh_ctx *c, double *host, int per);

extern "C" {

int bh_abi_version(void) { return BHGPU_ABI_VERSION; }

#ifndef BHGPU_BUILD_INFO
#define BHGPU_BUILD_INFO "digest=unknown flags=unknown"
#endif
const char *bh_build_info(void) { return BHGPU_BUILD_INFO; }

const char *bh_last_error(const bh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int bh_create(const bh_config *cfg, bh_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, BH_ERR_ARG, "bh_create: null argument");
    *out = nullptr;
    if (cfg->capacity < 0 || cfg->capacity > 0x7fffffffLL - 8)
        return fail(nullptr, BH_ERR_ARG, "bh_create: capacity out of range");
    if (cfg->max_depth < 1 || cfg->max_depth > 32)
        return fail(nullptr, BH_ERR_ARG, "bh_create: max_depth must be 1..32");
    if (!(cfg->theta > 0.0)) return fail(nullptr, BH_ERR_ARG, "bh_create: theta must be > 0");
    if (cfg->precision != BH_PRECISION_F64_EXACT && cfg->precision != BH_PRECISION_F32 &&
        cfg->precision != BH_PRECISION_MIXED && cfg->precision != BH_PRECISION_F64)
        return fail(nullptr, BH_ERR_ARG, "bh_create: unknown precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, BH_ERR_NO_DEVICE, "bh_create: no HIP device (libbhgpu has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, BH_ERR_ARG, "bh_create: device ordinal out of range");

    bh_ctx *c = new bh_ctx();
    c->cfg = *cfg;
    c->Dm = cfg->max_depth - 1;
    c->fast64 = cfg->precision == BH_PRECISION_F64;
    c->exact = cfg->precision == BH_PRECISION_F64_EXACT || c->fast64;   // (same state, same build)
    c->state64 = c->exact || cfg->precision == BH_PRECISION_MIXED;
    c->compat = cfg->reference_compat != 0;
    c->device = cfg->device;
    c->sort_passes = (2 * c->Dm + kRadixBits - 1) / kRadixBits;
    if (const char *e = std::getenv("BH_WALK_SPLIT")) c->walk_split = std::atoi(e);
    if (const char *e = std::getenv("BH_EXACT_BPW")) { const int b = std::atoi(e); c->exact_bpw = (b >= 1 && b <= kWave && (b & (b - 1)) == 0) ? b : 0; }
    if (const char *e = std::getenv("BH_WALK_ASM")) c->walk_asm = std::atoi(e) != 0;
    if (const char *e = std::getenv("BH_EXACT_BFS_MAX")) c->bfs_max = std::max(0, std::atoi(e));
    c->exact_thr = c->exact && !c->fast64 && c->walk_asm && !(cfg->flags & BH_FLAG_WALK_PORTABLE);
    if (const char *e = std::getenv("BH_SORT_PACK")) c->sort_pack = std::atoi(e) != 0;
    if (const char *e = std::getenv("BH_SORT_BUCKET")) c->sort_bucket = std::atoi(e);
    if (const char *e = std::getenv("BH_BUILD_ITEMS")) c->build_items = std::atoi(e);
    if (const char *e = std::getenv("BH_REORDER_EVERY")) c->reorder_every = std::max(0, std::atoi(e));
    c->hilbert = !c->exact;
    if (const char *e = std::getenv("BH_HILBERT")) c->hilbert = !c->exact && std::atoi(e) != 0;
    auto bail = [&](int rc) { g_create_error = c->err; bh_destroy(c); return rc; };

    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(BH_ERR_DEVICE); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        c->err = "hipStreamCreate failed"; return bail(BH_ERR_DEVICE);
    }
    c->own_stream = true;

    const int64_t cap = std::max<int64_t>(cfg->capacity, 1);
    c->node_cap = cfg->node_capacity > 0 ? cfg->node_capacity : 8 * cap + 1024;
    // a depth-capped tree can never exceed the full tree (QUADTREE_MAX_SIZE, project.cu:62)
    if (c->Dm < 15) {
        int64_t full = 0, wdt = 1;
        for (int l = 0; l <= c->Dm; ++l) { full += wdt; wdt *= 4; }
        c->node_cap = std::min(c->node_cap, std::max<int64_t>(full, 1));
    }
    c->internal_cap = (c->node_cap - 1) / 4;

    const size_t rs = c->state64 ? sizeof(double) : sizeof(float);
    int rc = 0;
    auto A = [&](auto **pp, size_t count) { if (!rc) rc = dev_alloc(c, pp, count); };
    { char *t; A(&t, cap * 2 * rs); c->pos = t; }
    { char *t; A(&t, cap * 2 * rs); c->vel = t; }
    { char *t; A(&t, cap * rs); c->mass = t; }
    { char *t; A(&t, cap * 2 * (c->exact ? sizeof(double) : sizeof(float))); c->force = t; }
    if (!c->exact) {
        { char *t; A(&t, cap * 2 * rs); c->pos2 = t; }
        { char *t; A(&t, cap * 2 * rs); c->vel2 = t; }
        { char *t; A(&t, cap * rs); c->mass2 = t; }
        { char *t; A(&t, cap * 2 * sizeof(float)); c->force2 = t; }
        A(&c->orig, cap); A(&c->orig2, cap);
    }
    A(&c->keys[0], cap); A(&c->keys[1], cap); A(&c->vals[0], cap); A(&c->vals[1], cap);
    A(&c->cnt, cap + 1);
    { const size_t nbl = std::max<size_t>(blocks_for(cap, kSortTile), blocks_for(std::min<int64_t>(cap, 1 << 22), kBlock * kSmallItems));
      // (the bucket pass of launches above 1M bodies counts 1,024 buckets per tile; sized for the SMALLEST tile that
      // can reach it -- BH_BUILD_ITEMS=2 is honoured up to 4M bodies: 2 words per body, 32 MB at 4M -- ADVICE r2:
      // sized for kSortTile only, the 512-key tiles of that override wrote past the end between 1M and 4M bodies)
      const size_t big = (cap > kBucketMaxN) ? (size_t)kBucketsBig * blocks_for(std::min<int64_t>(cap, kBucketMaxNBig), kBlock * kSmallItems) : 0;
      A(&c->radix_counts, std::max<size_t>((size_t)(1 << kSortBits) * nbl, big));
      A(&c->bsum_sort, 2 * kBucketStartOffset);               // bucket totals, then bucket starts
      A(&c->splitters, kBucketsBig);
      A(&c->sort_dig, (size_t)std::min<int64_t>(cap, kBucketMaxNBig) + 16);
    }
    A(&c->bsum_u32, std::max<size_t>(blocks_for(cap + 1, kTile), blocks_for(std::min<int64_t>(cap, 1 << 22) + 1, kBlock * kSmallItems)) + 8);
    A(&c->partial, 4 * (std::max<size_t>(1024, blocks_for(cap, kWave)) + 2)); A(&c->box, 8); A(&c->bslots, 4 * kBoundSlots);
    A(&c->ctr, 1);
    if (c->exact) {
        // (node 0 is the root, the four children of cell r are nodes 1 + 4 r ..: the arrays start three records into their
        // allocations, so that a sibling quad -- 128 bytes of NodeD, 32 of LinkD, what one visit of the fp64 walks loads -- is ONE
        // aligned 128-byte line and half a 64-byte line instead of straddling two and two: profiles/r04_f64/walk_traffic.json
        // read 1.85x the algorithmic bytes before)
        { NodeD *g0 = nullptr; LinkD *l0 = nullptr; A(&g0, c->node_cap + 4); A(&l0, c->node_cap + 4);
          if (!rc) { c->gd = g0 + 3; c->ld = l0 + 3; } }
        A(&c->self_node, c->internal_cap + 1); A(&c->com_pending, c->internal_cap + 1); A(&c->cell_depth, c->internal_cap + 1);
        A(&c->cell_first, c->internal_cap + 1);
    } else {
        A(&c->qf, c->internal_cap + 1); A(&c->aux, 4 * (c->internal_cap + 1));
        A(&c->cell_first, c->internal_cap + 1);
        A(&c->gid, cap); A(&c->group_cost, cap / kWave + 2); A(&c->walk_consts, 4);
        A(&c->coarse, cap / 256 + 2);
        A(&c->spos, cap); A(&c->sstate, cap + 64 * kBlock + 1024); A(&c->smass, cap);
        A(&c->terms, cap + 1); A(&c->bsum_d3, std::max<size_t>(blocks_for(cap + 1, kTile), blocks_for(std::min<int64_t>(cap, 1 << 22) + 1, kBlock * kSmallItems)) + 8);
    }
    if (rc) return bail(rc);
    if (hipMemset(c->ctr, 0, sizeof(TreeCounters)) != hipSuccess) { c->err = "hipMemset failed"; return bail(BH_ERR_DEVICE); }
    if (!c->exact) {
        const void *wc[4] = {c->aux, c->spos, c->smass, nullptr};
        if (hipMemcpy(c->walk_consts, wc, sizeof(wc), hipMemcpyHostToDevice) != hipSuccess) { c->err = "hipMemcpy failed"; return bail(BH_ERR_DEVICE); }
    }
    if (!c->exact && c->exp.create(cap) != hipSuccess) { c->err = "experiment hooks: allocation failed"; return bail(BH_ERR_DEVICE); }
    for (auto &e : c->ev_step) if (hipEventCreate(&e) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(BH_ERR_DEVICE); }
    for (auto &e : c->ev_build) if (hipEventCreate(&e) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(BH_ERR_DEVICE); }
    for (auto &e : c->ev_grp) if (hipEventCreate(&e) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(BH_ERR_DEVICE); }
    for (auto &e : c->ev_let) if (hipEventCreate(&e) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(BH_ERR_DEVICE); }
    *out = c;
    return BH_OK;
}

void bh_destroy(bh_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->exp.destroy(c->n);
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_step) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_build) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_grp) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_let) if (e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bh_set_stream(bh_ctx *c, void *hip_stream)
{
    if (!c) return BH_ERR_ARG;
    BH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    return BH_OK;
}

int bh_upload(bh_ctx *c, const double *pos, const double *vel, const double *mass, int64_t n)
{
    if (!c) return BH_ERR_ARG;
    if (n < 0 || (n > 0 && (!pos || !vel || !mass))) return fail(c, BH_ERR_ARG, "bh_upload: null array");
    if (n > c->cfg.capacity)
        return fail(c, BH_ERR_ARG, "Requested number of bodies exceeds N_BODIES.");   // project.cu:110-112
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->state64) {
        BH_HIP(c, hipMemcpy(c->pos, pos, n * 2 * sizeof(double), hipMemcpyHostToDevice));
        BH_HIP(c, hipMemcpy(c->vel, vel, n * 2 * sizeof(double), hipMemcpyHostToDevice));
        BH_HIP(c, hipMemcpy(c->mass, mass, n * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> t(std::max<int64_t>(2 * n, 1));
        for (int64_t i = 0; i < 2 * n; ++i) t[i] = (float)pos[i];
        BH_HIP(c, hipMemcpy(c->pos, t.data(), n * 2 * sizeof(float), hipMemcpyHostToDevice));
        for (int64_t i = 0; i < 2 * n; ++i) t[i] = (float)vel[i];
        BH_HIP(c, hipMemcpy(c->vel, t.data(), n * 2 * sizeof(float), hipMemcpyHostToDevice));
        for (int64_t i = 0; i < n; ++i) t[i] = (float)mass[i];
        BH_HIP(c, hipMemcpy(c->mass, t.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    BH_HIP(c, hipMemset(c->force, 0, std::max<int64_t>(n, 1) * 2 * (c->exact ? sizeof(double) : sizeof(float))));
    c->n = n;
    c->partial_count = 0; c->slots_valid = false;
    c->samples_n = -1;                                        // new bodies: the next build sorts with the LSD passes
    c->uploaded = true;
    c->tree_valid = false;
    c->steps_done = 0;
    c->orig_identity = true;
    c->builds = 0;
    c->group_cost_valid = false;
    if (c->gid && n > 0) {
        hipLaunchKernelGGL(iota_i64_kernel, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, c->gid, n);
        BH_HIP(c, hipGetLastError());
    }
    return BH_OK;
}

static int download_pairs(bh_ctx *c, const void *dev, double *host, int64_t count, bool is64)
{
    if (is64) {
        BH_HIP(c, hipMemcpy(host, dev, count * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        std::vector<float> t(std::max<int64_t>(count, 1));
        BH_HIP(c, hipMemcpy(t.data(), dev, count * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < count; ++i) host[i] = (double)t[i];
    }
    return BH_OK;
}

// host array with `per` doubles per body, read from the device in DEVICE order -> caller order
static int to_caller_order(bh_ctx *c, double *host, int per)
{
    if (c->exact || c->orig_identity || c->n == 0) return BH_OK;
    const int64_t n = c->n;
    std::vector<uint32_t> o(n);
    BH_HIP(c, hipMemcpy(o.data(), c->orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<double> t(host, host + (size_t)per * n);
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < per; ++k) host[(size_t)per * o[i] + k] = t[(size_t)per * i + k];
    return BH_OK;
}

int bh_initialize(bh_ctx *c, int64_t n, uint64_t seed, int32_t kind, double lower_m, double higher_m,
                  double lower_p, double higher_p, double lower_v, double higher_v)
{
    if (!c) return BH_ERR_ARG;
    if (n < 0 || n > c->cfg.capacity)
        return fail(c, BH_ERR_ARG, "Requested number of bodies exceeds N_BODIES.");
    if (kind != 0 && kind != 1) return fail(c, BH_ERR_ARG, "bh_initialize: kind must be 0 (box) or 1 (plummer)");
    BH_HIP(c, hipSetDevice(c->device));
    if (n > 0) {
        const unsigned g = blocks_for(n, kBlock);
        if (c->state64)
            hipLaunchKernelGGL((init_bodies_kernel<double2, double>), dim3(g), dim3(kBlock), 0, c->stream,
                               (double2 *)c->pos, (double2 *)c->vel, (double *)c->mass, n, seed, kind, lower_m,
                               higher_m, lower_p, higher_p, lower_v, higher_v);
        else
            hipLaunchKernelGGL((init_bodies_kernel<float2, float>), dim3(g), dim3(kBlock), 0, c->stream,
                               (float2 *)c->pos, (float2 *)c->vel, (float *)c->mass, n, seed, kind, lower_m,
                               higher_m, lower_p, higher_p, lower_v, higher_v);
        BH_HIP(c, hipGetLastError());
    }
    BH_HIP(c, hipMemsetAsync(c->force, 0, std::max<int64_t>(n, 1) * 2 * (c->exact ? sizeof(double) : sizeof(float)), c->stream));
    c->n = n;
    c->partial_count = 0; c->slots_valid = false;
    c->samples_n = -1;                                        // new bodies: the next build sorts with the LSD passes
    c->uploaded = true;
    c->tree_valid = false;
    c->steps_done = 0;
    c->orig_identity = true;
    c->builds = 0;
    c->group_cost_valid = false;
    if (c->gid && n > 0) {
        hipLaunchKernelGGL(iota_i64_kernel, dim3(blocks_for(n, kBlock)), dim3(kBlock), 0, c->stream, c->gid, n);
        BH_HIP(c, hipGetLastError());
    }
    return BH_OK;
}

int bh_download_masses(bh_ctx *c, double *mass)
{
    if (!c || !mass) return fail(c, BH_ERR_ARG, "bh_download_masses: null array");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_download_masses before bh_upload/bh_initialize");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    int rc = download_pairs(c, c->mass, mass, c->n, c->state64);
    return rc ? rc : to_caller_order(c, mass, 1);
}

int bh_sync(bh_ctx *c)
{
    if (!c) return BH_ERR_ARG;
    BH_HIP(c, hipSetDevice(c->device));
    // also reports a tree that outgrew node_capacity during bh_step (the walk of such a step does
    // nothing, so the state is the last good one)
    if (c->tree_valid) return check_overflow(c);
    BH_HIP(c, hipStreamSynchronize(c->stream));
    return BH_OK;
}

int bh_download(bh_ctx *c, double *pos, double *vel)
{
    if (!c || !pos) return fail(c, BH_ERR_ARG, "bh_download: null array");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_download before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = c->tree_valid ? check_overflow(c) : BH_OK;
    if (rc) return rc;
    BH_HIP(c, hipStreamSynchronize(c->stream));
    rc = download_pairs(c, c->pos, pos, 2 * c->n, c->state64);
    if (!rc) rc = to_caller_order(c, pos, 2);
    if (rc) return rc;
    if (vel) {
        rc = download_pairs(c, c->vel, vel, 2 * c->n, c->state64);
        if (!rc) rc = to_caller_order(c, vel, 2);
    }
    return rc;
}

int bh_build_tree(bh_ctx *c)
{
    if (!c) return BH_ERR_ARG;
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_build_tree before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_build(c);
    if (rc) return rc;
    return check_overflow(c);
}

int bh_compute_forces(bh_ctx *c)
{
    if (!c) return BH_ERR_ARG;
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_compute_forces before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_build(c);
    if (rc) return rc;
    rc = enqueue_walk(c, false, false);
    if (rc) return rc;
    return check_overflow(c);
}

int bh_step(bh_ctx *c, int32_t nsteps)
{
    if (!c || nsteps < 0) return BH_ERR_ARG;
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_step before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    // one event pair per step around the walk kernel (bounded pool)
    const int want = std::min<int>(nsteps, 4096);
    while ((int)c->ev.size() < 2 * want) {
        hipEvent_t e;
        BH_HIP(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }
    BH_HIP(c, hipEventRecord(c->ev_step[0], c->stream));
    for (int s = 0; s < nsteps; ++s) {
        if (s == nsteps - 1) BH_HIP(c, hipEventRecord(c->ev_build[0], c->stream));
        c->time_groups = (s == nsteps - 1);
        int rc = enqueue_build(c);
        c->time_groups = false;
        if (rc) return rc;
        if (s == nsteps - 1) BH_HIP(c, hipEventRecord(c->ev_build[1], c->stream));
        if (s < want) BH_HIP(c, hipEventRecord(c->ev[2 * s], c->stream));
        rc = enqueue_walk(c, true, false);
        if (rc) return rc;
        if (s < want) BH_HIP(c, hipEventRecord(c->ev[2 * s + 1], c->stream));
    }
    BH_HIP(c, hipEventRecord(c->ev_step[1], c->stream));
    c->steps_done += nsteps;
    c->last_nsteps = nsteps;
    c->timed_pairs = want;
    c->step_timed = nsteps > 0;
    return BH_OK;
}

int bh_get_forces(bh_ctx *c, double *out)
{
    if (!c || !out) return fail(c, BH_ERR_ARG, "bh_get_forces: null array");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    int rc = download_pairs(c, c->force, out, 2 * c->n, c->exact);
    if (rc) return rc;
    if (!c->exact) {   // fp32 / mixed mode store accelerations; force = a * m_i
        std::vector<double> m(std::max<int64_t>(c->n, 1));
        rc = download_pairs(c, c->mass, m.data(), c->n, c->state64);
        if (rc) return rc;
        for (int64_t i = 0; i < c->n; ++i) { out[2 * i] *= m[i]; out[2 * i + 1] *= m[i]; }
    }
    return to_caller_order(c, out, 2);
}

int bh_get_accel(bh_ctx *c, double *out)
{
    if (!c || !out) return fail(c, BH_ERR_ARG, "bh_get_accel: null array");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    int rc = download_pairs(c, c->force, out, 2 * c->n, c->exact);
    if (rc) return rc;
    if (c->exact) {    // exact mode stores forces; a = F / m_i (updateAccelerations, project.cu:795-801)
        std::vector<double> m(std::max<int64_t>(c->n, 1));
        BH_HIP(c, hipMemcpy(m.data(), c->mass, c->n * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < c->n; ++i) { out[2 * i] /= m[i]; out[2 * i + 1] /= m[i]; }
    }
    return to_caller_order(c, out, 2);
}

int bh_get_interaction_counts(bh_ctx *c, uint32_t *out)
{
    if (!c || !out) return fail(c, BH_ERR_ARG, "bh_get_interaction_counts: null array");
    if ((c->exact && !c->fast64) || !(c->cfg.flags & BH_FLAG_WALK_STATS) || !c->body_counts)
        return fail(c, BH_ERR_STATE, "bh_get_interaction_counts: fp32 / mixed / BH_PRECISION_F64 with BH_FLAG_WALK_STATS, after a walk");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    const int64_t n = c->n;
    if (n == 0) return BH_OK;
    std::vector<uint32_t> t(n);
    BH_HIP(c, hipMemcpy(t.data(), c->body_counts, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (c->orig_identity) { std::memcpy(out, t.data(), n * sizeof(uint32_t)); return BH_OK; }
    std::vector<uint32_t> o(n);
    BH_HIP(c, hipMemcpy(o.data(), c->orig, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) out[o[i]] = t[i];
    return BH_OK;
}

// ---- tree export: DFS pre-order, children in index order (TraverseTreeToFile, project.cu:504-534)
static int export_tree_host(bh_ctx *c, std::vector<bh_tree_node> &out, std::vector<int32_t> &depth)
{
    if (!c->tree_valid) return fail(c, BH_ERR_STATE, "no tree built yet");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    TreeCounters h{};
    BH_HIP(c, hipMemcpy(&h, c->ctr, sizeof(h), hipMemcpyDeviceToHost));
    if (h.overflow || (int64_t)h.n_internal > c->internal_cap) return fail(c, BH_ERR_CAPACITY, "tree overflowed node_capacity");
    const int64_t nn = 1 + 4 * (int64_t)h.n_internal;
    double box[4];
    BH_HIP(c, hipMemcpy(box, c->box, sizeof(box), hipMemcpyDeviceToHost));
    std::vector<NodeD> gd;
    std::vector<LinkD> ld;
    std::vector<QuadF> qf;
    std::vector<NodeAux> aux;
    std::vector<uint32_t> perm(std::max<int64_t>(c->n, 1));
    if (c->n > 0) BH_HIP(c, hipMemcpy(perm.data(), c->perm, c->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (!c->exact && !c->orig_identity && c->n > 0) {          // slots -> the caller's body indices
        std::vector<uint32_t> o(c->n);
        BH_HIP(c, hipMemcpy(o.data(), c->orig, c->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (auto &b : perm) b = o[b];
    }
    if (c->exact) {
        gd.resize(nn); ld.resize(nn);
        BH_HIP(c, hipMemcpy(gd.data(), c->gd, nn * sizeof(NodeD), hipMemcpyDeviceToHost));
        BH_HIP(c, hipMemcpy(ld.data(), c->ld, nn * sizeof(LinkD), hipMemcpyDeviceToHost));
    } else {
        const int64_t nq = (int64_t)h.n_internal + 1;      // quad 0 = root
        if (!c->aux_full && c->n > 0) {                    // the step's node kernel writes bucket records only
            launch_nodes_fast(c, true, c->stream);
            BH_HIP(c, hipGetLastError());
            BH_HIP(c, hipStreamSynchronize(c->stream));
        }
        qf.resize(nq); aux.resize(4 * nq);
        BH_HIP(c, hipMemcpy(qf.data(), c->qf, nq * sizeof(QuadF), hipMemcpyDeviceToHost));
        BH_HIP(c, hipMemcpy(aux.data(), c->aux, 4 * nq * sizeof(NodeAux), hipMemcpyDeviceToHost));
    }
    struct Item { int32_t node; int32_t depth; int64_t parent_out; int slot; double x0, x1, y0, y1; int state; };
    std::vector<Item> stack;
    stack.push_back({0, 0, -1, 0, box[0], box[1], box[2], box[3], 0});
    out.clear(); depth.clear();
    out.reserve(nn); depth.reserve(nn);
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const int64_t node_limit = c->exact ? nn : 4 * ((int64_t)h.n_internal + 1);
        if (it.node < 0 || it.node >= node_limit) return fail(c, BH_ERR_DEVICE, "corrupt child link in device tree");
        bh_tree_node q{};
        int32_t child;       // id of child 0, or -1
        if (c->exact) {
            q.comx = gd[it.node].cx; q.comy = gd[it.node].cy; q.mass = gd[it.node].m;
            q.particle = (double)ld[it.node].occ;
            child = ld[it.node].child;
        } else {
            const QuadF &f = qf[it.node >> 2];
            const int sl = it.node & 3;
            const NodeAux &ax = aux[it.node];
            q.comx = f.xy[2 * sl]; q.comy = f.xy[2 * sl + 1]; q.mass = f.m[sl];
            child = f.child[sl] >= 1 ? 4 * f.child[sl] : -1;
            if (child < 0 && ax.count == 1) {      // single occupant: the reference's PARTICLE_INDEX
                const int64_t body = perm[ax.first];
                q.particle = (it.depth == c->Dm) ? (double)(-body - 2) : (double)body;
            } else q.particle = -1.0;
        }
        q.xmin = it.x0; q.xmax = it.x1; q.ymin = it.y0; q.ymax = it.y1;
        for (int k = 0; k < 4; ++k) q.child[k] = -1.0;
        const int64_t me = (int64_t)out.size();
        if (it.parent_out >= 0) out[it.parent_out].child[it.slot] = (double)me;
        out.push_back(q);
        depth.push_back(it.depth);
        if (child >= 0) {
            const double mx = (it.x0 + it.x1) / 2.0, my = (it.y0 + it.y1) / 2.0;
            for (int k = 3; k >= 0; --k) {          // push reversed: child 0 is visited first
                // k is the GEOMETRIC child index (the reference's order); with Hilbert keys the
                // sibling sits in slot H[state][k] of the quad
                const int slot = c->hilbert ? hilbert_digit(it.state, k) : k;
                Item ch{child + slot, it.depth + 1, me, k,
                        (k & 1) ? mx : it.x0, (k & 1) ? it.x1 : mx,
                        (k & 2) ? my : it.y0, (k & 2) ? it.y1 : my,
                        c->hilbert ? hilbert_next(it.state, k) : 0};
                stack.push_back(ch);
            }
        }
    }
    return BH_OK;
}

int bh_export_tree(bh_ctx *c, bh_tree_node *nodes, int32_t *depth, int64_t cap, int64_t *n_nodes)
{
    if (!c || !n_nodes) return fail(c, BH_ERR_ARG, "bh_export_tree: null argument");
    std::vector<bh_tree_node> out;
    std::vector<int32_t> dep;
    int rc = export_tree_host(c, out, dep);
    if (rc) return rc;
    *n_nodes = (int64_t)out.size();
    if (!nodes || cap < (int64_t)out.size()) return fail(c, BH_ERR_CAPACITY, "bh_export_tree: buffer too small");
    std::memcpy(nodes, out.data(), out.size() * sizeof(bh_tree_node));
    if (depth) std::memcpy(depth, dep.data(), dep.size() * sizeof(int32_t));
    return BH_OK;
}

int bh_write_quadtree_file(bh_ctx *c, const char *path)
{
    if (!c || !path) return fail(c, BH_ERR_ARG, "bh_write_quadtree_file: null argument");
    std::vector<bh_tree_node> out;
    std::vector<int32_t> dep;
    int rc = export_tree_host(c, out, dep);
    if (rc) return rc;
    std::vector<double> pos(std::max<int64_t>(2 * c->n, 2));
    rc = download_pairs(c, c->pos, pos.data(), 2 * c->n, c->state64);
    if (!rc) rc = to_caller_order(c, pos.data(), 2);
    if (rc) return rc;
    FILE *fp = std::fopen(path, "w");
    if (!fp) return fail(c, BH_ERR_IO, std::string("cannot open ") + path);
    for (size_t i = 0; i < out.size(); ++i) {
        const bh_tree_node &q = out[i];
        // `file << depth << " " << xmin ...` with the default ostream precision == "%g"
        std::fprintf(fp, "%d %g %g %g %g %g", dep[i], q.xmin, q.xmax, q.ymin, q.ymax, q.mass);
        const long long occ = (long long)q.particle;
        if (occ != -1) {
            const long long b = occ >= 0 ? occ : -(occ + 2);   // the reference reads out of bounds here
            std::fprintf(fp, " occupantIndex=%lld occupantPos=(%g,%g)", occ, pos[2 * b], pos[2 * b + 1]);
        } else if (q.mass > 0) {
            std::fprintf(fp, " occupantIndex=%lld occupantPos=(%g,%g)", occ, q.comx, q.comy);
        }
        std::fputc('\n', fp);
    }
    std::fclose(fp);
    return BH_OK;
}

int bh_stats(bh_ctx *c, bh_stats_t *out)
{
    if (!c || !out) return fail(c, BH_ERR_ARG, "bh_stats: null argument");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    std::memset(out, 0, sizeof(*out));
    out->n_bodies = c->n;
    out->steps_done = c->steps_done;
    out->walk_launches = (uint64_t)c->walk_launches;
    out->device_bytes = c->device_bytes;
    {
        TreeCounters h{};
        BH_HIP(c, hipMemcpy(&h, c->ctr, sizeof(h), hipMemcpyDeviceToHost));
        out->sort_spill_buckets = h.sort_spills;
        out->sort_rerun_buckets = h.sort_reruns;
    }
    if (c->tree_valid) {
        TreeCounters h{};
        BH_HIP(c, hipMemcpy(&h, c->ctr, sizeof(h), hipMemcpyDeviceToHost));
        out->n_internal = h.n_internal;
        out->n_nodes = 1 + 4 * (int64_t)h.n_internal;
        out->visits = h.visits;
        out->interactions = h.interactions;
        out->wave_nodes = h.wave_nodes;
        out->wave_quads = h.wave_quads;
        out->wave_accepts = h.wave_accepts;
    }
    if (c->let_timed) {
        float ms = 0.f;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_let[0], c->ev_let[1])); out->let_tree_ms = ms;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_let[1], c->ev_let[2])); out->let_pack_ms = ms;
    }
    if (c->step_timed) {
        float ms = 0.f;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_step[0], c->ev_step[1]));
        out->last_step_ms = (double)ms / std::max(1, c->last_nsteps);
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_build[0], c->ev_build[1]));
        out->build_ms = ms;
        double acc = 0.0;
        for (int s = 0; s < c->timed_pairs; ++s) {
            BH_HIP(c, hipEventElapsedTime(&ms, c->ev[2 * s], c->ev[2 * s + 1]));
            acc += ms;
        }
        out->walk_ms = c->timed_pairs ? acc / c->timed_pairs : 0.0;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_build[0], c->ev_grp[0])); out->keys_ms = ms;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_grp[0], c->ev_grp[1])); out->sort_ms = ms;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_grp[1], c->ev_grp[2])); out->scan_ms = ms;
        BH_HIP(c, hipEventElapsedTime(&ms, c->ev_grp[2], c->ev_build[1])); out->nodes_ms = ms;
        // algorithmic bytes per body of one build.  keys: position in, key (+ index) out;
        // sort: per LSD pass 8 (histogram) + 16 (scatter of packed keys; 24 with a separate index array), or the
        // bucket sort's 9 + 17 + 20 (histogram + bucket byte, scatter, in-LDS sort with the unpacked write-out);
        // prep: key, index, position and mass in, count and sorted copies out (+ 24 of prefix-sum terms when the
        // state is fp64); scans: counts in and out, cell starts, terms -> prefix sums; nodes: ~0.72 cells per body,
        // each reading its key window share, ranks and five prefix sums and writing an 80-byte quad
        const int passes = (2 * c->Dm + kSortBits - 1) / kSortBits;
        const uint64_t keys_b = (c->state64 ? 24u : 16u) + (c->last_sort_packed ? 0u : 4u);
        const uint64_t sort_b = c->last_sort_bucket ? 46u : (uint64_t)passes * (c->last_sort_packed ? 24u : 32u);
        const uint64_t prep_b = c->state64 ? 76u : 40u, scan_b = c->state64 ? 59u : 47u, nodes_b = c->exact ? 140u : 117u;
        out->build_bytes = (uint64_t)c->n * (keys_b + sort_b + prep_b + scan_b + nodes_b);
        out->walk_bytes = out->wave_nodes ? (uint64_t)c->n * 44u + out->wave_nodes * 20u : 0u;
    }
    return BH_OK;
}

int bh_step_times(bh_ctx *c, double *step_ms, double *walk_ms, int32_t cap, int32_t *n_out)
{
    if (!c || !n_out) return fail(c, BH_ERR_ARG, "bh_step_times: null argument");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    const int n = c->step_timed ? c->timed_pairs : 0;
    *n_out = n;
    if (cap < n) return (step_ms || walk_ms) ? fail(c, BH_ERR_CAPACITY, "bh_step_times: buffer too small") : BH_OK;
    for (int s = 0; s < n; ++s) {
        float ms = 0.f;
        if (walk_ms) { BH_HIP(c, hipEventElapsedTime(&ms, c->ev[2 * s], c->ev[2 * s + 1])); walk_ms[s] = ms; }
        // a step ends where its walk ends; the first one starts where bh_step started
        if (step_ms) { BH_HIP(c, hipEventElapsedTime(&ms, s ? c->ev[2 * s - 1] : c->ev_step[0], c->ev[2 * s + 1])); step_ms[s] = ms; }
    }
    return BH_OK;
}

// ---- multi-GPU plumbing -------------------------------------------------------------------------
int bh_set_owned_fraction(bh_ctx *c, int32_t rank, int32_t world)
{
    if (!c || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, BH_ERR_ARG, "bh_set_owned_fraction: bad rank/world (world <= 64)");
    c->rank = rank;
    c->world = world;
    return BH_OK;
}

int bh_owned_range(bh_ctx *c, int64_t *lo, int64_t *hi)
{
    if (!c || !lo || !hi) return BH_ERR_ARG;
    owned_range(c, lo, hi);
    return BH_OK;
}

int bh_device_state(bh_ctx *c, void **pos, void **vel, void **mass, int64_t *n, int32_t *elem_bytes)
{
    if (!c) return BH_ERR_ARG;
    if (pos) *pos = c->pos;
    if (vel) *vel = c->vel;
    if (mass) *mass = c->mass;
    if (n) *n = c->n;
    if (elem_bytes) *elem_bytes = c->state64 ? 8 : 4;
    return BH_OK;
}

int bh_device_sorted(bh_ctx *c, void **sorted_state)
{
    if (!c) return BH_ERR_ARG;
    if (c->state64) return fail(c, BH_ERR_STATE, "the sorted exchange buffer exists in fp32 mode only");
    if (sorted_state) *sorted_state = c->sstate;
    return BH_OK;
}

}  // extern "C"

namespace bh {
__global__ __launch_bounds__(kBlock) void scatter_sorted_kernel(const uint32_t *__restrict__ perm,
                                                                 const float4 *__restrict__ sstate,
                                                                 float2 *__restrict__ pos,
                                                                 float2 *__restrict__ vel, int64_t n)
{
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n) return;
    const uint32_t b = perm[s];
    const float4 t = sstate[s];
    pos[b] = float2{t.x, t.y};
    vel[b] = float2{t.z, t.w};
}
}  // namespace bh

extern "C" {

int bh_step_local(bh_ctx *c)
{
    if (!c) return BH_ERR_ARG;
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_step_local before bh_upload");
    if (c->state64) return fail(c, BH_ERR_STATE, "bh_step_local: fp32 mode only");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_build(c);
    if (rc) return rc;
    return enqueue_walk(c, true, true);
}

int bh_scatter_sorted(bh_ctx *c)
{
    if (!c) return BH_ERR_ARG;
    if (c->state64) return fail(c, BH_ERR_STATE, "bh_scatter_sorted: fp32 mode only");
    BH_HIP(c, hipSetDevice(c->device));
    if (c->n > 0) {
        hipLaunchKernelGGL(scatter_sorted_kernel, dim3(blocks_for(c->n, kBlock)), dim3(kBlock), 0, c->stream,
                           c->perm, c->sstate, (float2 *)c->pos, (float2 *)c->vel, c->n);
        BH_HIP(c, hipGetLastError());
    }
    c->partial_count = 0; c->slots_valid = false;
    c->steps_done += 1;
    return BH_OK;
}

// ---- distributed step with locally-essential trees --------------------------------------------------
int bh_let_local_quads(bh_ctx *c, int64_t *local_quads)
{
    if (!c || !local_quads) return fail(c, BH_ERR_ARG, "bh_let_local_quads: null argument");
    *local_quads = c->internal_cap + 1;
    return BH_OK;
}

int bh_let_configure(bh_ctx *c, int32_t rank, int32_t world, int64_t let_cap, int64_t forest_base)
{
    if (!c || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || let_cap < 1)
        return fail(c, BH_ERR_ARG, "bh_let_configure: bad rank/world/let_cap (world <= 64)");
    if (forest_base < c->internal_cap + 1)
        return fail(c, BH_ERR_ARG, "bh_let_configure: forest_base must be the LARGEST bh_let_local_quads of all ranks "
                                   "(a sender writes child links in the receiver's index space)");
    if (forest_base + (int64_t)world * let_cap > 0x7fffffffLL)
        return fail(c, BH_ERR_ARG, "bh_let_configure: forest too large for 32-bit quad indices");
    if (c->exact) return fail(c, BH_ERR_STATE, "bh_let_configure: fp32 and mixed precision only");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    int rc = 0;
    auto A = [&](auto **pp, size_t count) { if (!rc) rc = dev_alloc(c, pp, count); };
    if (c->let_mode) {
        // a second call may only change let_cap (LetStepper.autotune: size the blocks from measured counts)
        if (rank != c->rank || world != c->world || forest_base != c->forest_base)
            return fail(c, BH_ERR_STATE, "bh_let_configure: rank/world/forest_base cannot change once configured");
        if (let_cap == c->let_cap) return BH_OK;
        dev_free(c, c->qf); dev_free(c, c->let_send);
        c->qf = nullptr; c->let_send = nullptr;
    } else {
        c->rank = rank; c->world = world;
        c->quads_local = c->internal_cap + 1;
        dev_free(c, c->qf);              // the single-tree node array: the forest below replaces it
        c->qf = nullptr;
        A(&c->needmask, (size_t)c->quads_local);
        A(&c->let_tsum, (size_t)kMaxWorld);            // per-peer slot counters of the LET extraction
        A(&c->let_outidx, (size_t)world * c->quads_local);
        A(&c->lbounds, 4 * kLetBoxes); A(&c->all_bounds, 4 * kLetBoxes * (size_t)world);
        A(&c->acc_part, (size_t)std::max<int64_t>(c->cfg.capacity, 1));
        A(&c->let_ctr, 1);
    }
    c->let_cap = let_cap;
    c->forest_base = forest_base;
    A(&c->qf, (size_t)(forest_base + (int64_t)world * let_cap));
    A(&c->let_send, (size_t)world * let_cap);
    if (rc) return rc;
    BH_HIP(c, hipMemset(c->let_ctr, 0, sizeof(LetCounters)));
    BH_HIP(c, hipMemset(c->let_tsum, 0, kMaxWorld * sizeof(uint32_t)));
    c->let_mode = true;
    c->external_box = true;
    c->tree_valid = false;
    return BH_OK;
}

int bh_let_bounds(bh_ctx *c)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_bounds: call bh_let_configure first");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_let_bounds before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    if (c->partial_count <= 0) {
        const unsigned nbb = kLetBoxes * kLetBoxParts;
        if (c->state64)
            hipLaunchKernelGGL((let_slice_bounds_kernel<double2>), dim3(nbb), dim3(kBlock), 0, c->stream,
                               (const double2 *)c->pos, c->n, c->partial);
        else
            hipLaunchKernelGGL((let_slice_bounds_kernel<float2>), dim3(nbb), dim3(kBlock), 0, c->stream,
                               (const float2 *)c->pos, c->n, c->partial);
        c->partial_count = (int)nbb;
    }
    hipLaunchKernelGGL(let_local_bounds_kernel, dim3(kLetBoxes), dim3(kWave), 0, c->stream, c->partial,
                       c->partial_count, c->lbounds);
    c->partial_count = 0; c->slots_valid = false;
    BH_HIP(c, hipGetLastError());
    return BH_OK;
}

int bh_let_pointers(bh_ctx *c, void **lbounds, void **all_bounds, void **send, void **recv, int64_t *block_bytes,
                    int32_t *boxes_per_rank)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_pointers: call bh_let_configure first");
    if (lbounds) *lbounds = c->lbounds;
    if (all_bounds) *all_bounds = c->all_bounds;
    if (send) *send = c->let_send;
    if (recv) *recv = c->qf + c->forest_base;
    if (block_bytes) *block_bytes = c->let_cap * (int64_t)sizeof(QuadF);
    if (boxes_per_rank) *boxes_per_rank = kLetBoxes;
    return BH_OK;
}

int bh_let_build(bh_ctx *c)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_build: call bh_let_configure first");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_let_build before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    (void)hipEventRecord(c->ev_let[0], st);
    hipLaunchKernelGGL(let_box_kernel, dim3(1), dim3(64), 0, st, c->all_bounds, c->world * kLetBoxes, c->box, c->ctr,
                       c->let_ctr, c->Dm);
    int rc = enqueue_build(c);
    if (rc) return rc;
    (void)hipEventRecord(c->ev_let[1], st);
    const int64_t nq = c->quads_local;
    hipLaunchKernelGGL(let_mark_alloc_kernel, dim3(blocks_for(4 * nq, kBlock)), dim3(kBlock), 0, st, c->qf, c->all_bounds,
                       c->world, c->rank, c->ctr, c->internal_cap, c->needmask, c->let_tsum, c->let_outidx, nq);
    hipLaunchKernelGGL(let_pack_kernel, dim3(blocks_for(nq, kBlock)), dim3(kBlock), 0, st, c->qf, c->needmask,
                       c->let_outidx, nq, c->world, c->rank, c->ctr, c->internal_cap, c->let_send,
                       (uint32_t)c->let_cap, c->forest_base + (int64_t)c->rank * c->let_cap, c->let_tsum, c->let_ctr);
    (void)hipEventRecord(c->ev_let[2], st);
    c->let_timed = true;
    BH_HIP(c, hipGetLastError());
    return BH_OK;
}

int bh_let_walk(bh_ctx *c)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_walk: call bh_let_configure first");
    if (!c->tree_valid) return fail(c, BH_ERR_STATE, "bh_let_walk before bh_let_build");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_walk(c, true, false);
    if (rc) return rc;
    c->steps_done += 1;
    return BH_OK;
}

// The forest walk in two launches, so that the all_to_all of the LETs can run under the first:
// bh_let_walk_local needs only bh_let_build's local tree; bh_let_walk_remote needs the received
// blocks, adds their contribution and finishes the step (integrate != 0) or just the forces.
int bh_let_walk_local(bh_ctx *c)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_walk_local: call bh_let_configure first");
    if (!c->tree_valid) return fail(c, BH_ERR_STATE, "bh_let_walk_local before bh_let_build");
    BH_HIP(c, hipSetDevice(c->device));
    return enqueue_walk(c, false, false, 1);
}

int bh_let_walk_remote(bh_ctx *c, int32_t integrate)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_walk_remote: call bh_let_configure first");
    if (!c->tree_valid) return fail(c, BH_ERR_STATE, "bh_let_walk_remote before bh_let_build");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = enqueue_walk(c, integrate != 0, false, 2);
    if (rc) return rc;
    if (integrate) c->steps_done += 1;
    return BH_OK;
}

int bh_let_forces(bh_ctx *c)
{
    if (!c || !c->let_mode) return fail(c, BH_ERR_STATE, "bh_let_forces: call bh_let_configure first");
    if (!c->tree_valid) return fail(c, BH_ERR_STATE, "bh_let_forces before bh_let_build");
    BH_HIP(c, hipSetDevice(c->device));
    return enqueue_walk(c, false, false);
}

// ---- device-side migration and re-balancing (bh_migrate.hpp) ----------------------------------------
int bh_set_ids(bh_ctx *c, const int64_t *ids)
{
    if (!c || !ids) return fail(c, BH_ERR_ARG, "bh_set_ids: null argument");
    if (!c->gid) return fail(c, BH_ERR_STATE, "bh_set_ids: fp32 and mixed precision only");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, "bh_set_ids before bh_upload");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    BH_HIP(c, hipMemcpy(c->gid, ids, c->n * sizeof(int64_t), hipMemcpyHostToDevice));
    return BH_OK;
}

int bh_get_ids(bh_ctx *c, int64_t *ids)
{
    if (!c || !ids) return fail(c, BH_ERR_ARG, "bh_get_ids: null argument");
    if (!c->gid) return fail(c, BH_ERR_STATE, "bh_get_ids: fp32 and mixed precision only");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    BH_HIP(c, hipMemcpy(ids, c->gid, c->n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return BH_OK;
}

static int check_cuts(bh_ctx *c, const bh_orb_cuts *cuts, const char *who)
{
    if (!c || !cuts) return fail(c, BH_ERR_ARG, std::string(who) + ": null argument");
    if (c->exact) return fail(c, BH_ERR_STATE, std::string(who) + ": fp32 and mixed precision only");
    if (!c->uploaded) return fail(c, BH_ERR_STATE, std::string(who) + " before bh_upload");
    if (cuts->world < 1 || cuts->world > kMaxWorld || cuts->n_cuts != cuts->world - 1)
        return fail(c, BH_ERR_ARG, std::string(who) + ": world must be 1..64 and n_cuts = world - 1");
    return BH_OK;
}

int bh_orb_histogram(bh_ctx *c, const bh_orb_cuts *cuts, int32_t level, void **hist, int64_t *n_words)
{
    int rc = check_cuts(c, cuts, "bh_orb_histogram");
    if (rc) return rc;
    if (!hist || !n_words || level < 0) return fail(c, BH_ERR_ARG, "bh_orb_histogram: bad argument");
    BH_HIP(c, hipSetDevice(c->device));
    const size_t words = (size_t)std::max(cuts->n_cuts, 1) * BH_ORB_BINS;
    if (!c->orb_hist) {
        rc = dev_alloc(c, &c->orb_hist, (size_t)BH_ORB_MAX_CUTS * BH_ORB_BINS);
        if (rc) return rc;
    }
    BH_HIP(c, hipMemsetAsync(c->orb_hist, 0, words * sizeof(unsigned long long), c->stream));
    if (c->n > 0) {
        // weights: the cost of the body's 64-body group in the last full walk; they are indexed by sorted
        // position, so the bodies are visited through the last build's permutation
        const bool weighted = c->group_cost_valid && c->tree_valid;
        const uint32_t *perm = weighted ? c->perm : nullptr;
        const uint32_t *cost = weighted ? c->group_cost : nullptr;
        const unsigned g = blocks_for(c->n, kBlock);
        if (c->state64)
            hipLaunchKernelGGL((orb_hist_kernel<double2>), dim3(g), dim3(kBlock), 0, c->stream, (const double2 *)c->pos,
                               perm, cost, c->n, *cuts, (int)level, c->orb_hist);
        else
            hipLaunchKernelGGL((orb_hist_kernel<float2>), dim3(g), dim3(kBlock), 0, c->stream, (const float2 *)c->pos,
                               perm, cost, c->n, *cuts, (int)level, c->orb_hist);
        BH_HIP(c, hipGetLastError());
    }
    *hist = c->orb_hist;
    *n_words = (int64_t)words;
    return BH_OK;
}

static int migrate_buffers(bh_ctx *c)
{
    if (c->mig_send) return BH_OK;
    const size_t cap = (size_t)std::max<int64_t>(c->cfg.capacity, 1) * kMigrateRecord;
    int rc = dev_alloc(c, &c->mig_send, cap);
    if (!rc) rc = dev_alloc(c, &c->mig_recv, cap);
    return rc;
}

int bh_migrate_pack(bh_ctx *c, const bh_orb_cuts *cuts, int64_t *send_counts)
{
    int rc = check_cuts(c, cuts, "bh_migrate_pack");
    if (rc) return rc;
    if (!send_counts) return fail(c, BH_ERR_ARG, "bh_migrate_pack: null argument");
    BH_HIP(c, hipSetDevice(c->device));
    rc = migrate_buffers(c);
    if (rc) return rc;
    const int64_t n = c->n;
    const int W = cuts->world;
    for (int r = 0; r < W; ++r) send_counts[r] = 0;
    c->tree_valid = false;                                  // the sort buffers are reused from here on
    if (n == 0) { BH_HIP(c, hipStreamSynchronize(c->stream)); return BH_OK; }
    hipStream_t st = c->stream;
    const unsigned g = blocks_for(n, kBlock);
    if (c->state64)
        hipLaunchKernelGGL((migrate_classify_kernel<double2>), dim3(g), dim3(kBlock), 0, st, (const double2 *)c->pos, n,
                           *cuts, c->keys[0], c->vals[0]);
    else
        hipLaunchKernelGGL((migrate_classify_kernel<float2>), dim3(g), dim3(kBlock), 0, st, (const float2 *)c->pos, n,
                           *cuts, c->keys[0], c->vals[0]);
    // one stable radix pass on the destination rank (< 64 < 256): slots grouped by destination, slot order kept
    const unsigned nbl = blocks_for(n, kSortTile);
    hipLaunchKernelGGL((radix_hist<kSortItems, kSortBits>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->radix_counts, n,
                       0, (int)nbl);
    hipLaunchKernelGGL(radix_rowscan, dim3(1 << kSortBits), dim3(kBlock), 0, st, c->radix_counts, c->bsum_sort, (int)nbl);
    hipLaunchKernelGGL((radix_scatter_w<kSortItems, kSortBits>), dim3(nbl), dim3(kBlock), 0, st, c->keys[0], c->vals[0],
                       c->keys[1], c->vals[1], c->radix_counts, c->bsum_sort, n, 0, (int)nbl);
    const uint32_t *orig = c->orig_identity ? nullptr : c->orig;
    if (c->state64)
        hipLaunchKernelGGL((migrate_pack_kernel<double2, double>), dim3(g), dim3(kBlock), 0, st, c->vals[1],
                           (const double2 *)c->pos, (const double2 *)c->vel, (const double *)c->mass, orig, c->gid, n,
                           c->mig_send);
    else
        hipLaunchKernelGGL((migrate_pack_kernel<float2, float>), dim3(g), dim3(kBlock), 0, st, c->vals[1],
                           (const float2 *)c->pos, (const float2 *)c->vel, (const float *)c->mass, orig, c->gid, n,
                           c->mig_send);
    BH_HIP(c, hipGetLastError());
    uint32_t totals[kMaxWorld];
    BH_HIP(c, hipMemcpyAsync(totals, c->bsum_sort, W * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    BH_HIP(c, hipStreamSynchronize(st));
    int64_t sum = 0;
    for (int r = 0; r < W; ++r) { send_counts[r] = totals[r]; sum += totals[r]; }
    if (sum != n) return fail(c, BH_ERR_DEVICE, "bh_migrate_pack: destination counts do not add up");
    return BH_OK;
}

int bh_migrate_pointers(bh_ctx *c, void **send, void **recv, int64_t *capacity_records)
{
    if (!c) return BH_ERR_ARG;
    if (c->exact) return fail(c, BH_ERR_STATE, "bh_migrate_pointers: fp32 and mixed precision only");
    BH_HIP(c, hipSetDevice(c->device));
    int rc = migrate_buffers(c);
    if (rc) return rc;
    if (send) *send = c->mig_send;
    if (recv) *recv = c->mig_recv;
    if (capacity_records) *capacity_records = std::max<int64_t>(c->cfg.capacity, 1);
    return BH_OK;
}

int bh_migrate_unpack(bh_ctx *c, int64_t n_new)
{
    if (!c) return BH_ERR_ARG;
    if (c->exact || !c->mig_recv) return fail(c, BH_ERR_STATE, "bh_migrate_unpack before bh_migrate_pack");
    if (n_new < 0 || n_new > c->cfg.capacity)
        return fail(c, BH_ERR_CAPACITY, "bh_migrate_unpack: " + std::to_string(n_new) + " bodies arrive, capacity is " +
                                        std::to_string(c->cfg.capacity));
    BH_HIP(c, hipSetDevice(c->device));
    if (n_new > 0) {
        const unsigned g = blocks_for(n_new, kBlock);
        if (c->state64)
            hipLaunchKernelGGL((migrate_unpack_kernel<double2, double>), dim3(g), dim3(kBlock), 0, c->stream, c->mig_recv,
                               n_new, (double2 *)c->pos, (double2 *)c->vel, (double *)c->mass, (float2 *)c->force, c->gid);
        else
            hipLaunchKernelGGL((migrate_unpack_kernel<float2, float>), dim3(g), dim3(kBlock), 0, c->stream, c->mig_recv,
                               n_new, (float2 *)c->pos, (float2 *)c->vel, (float *)c->mass, (float2 *)c->force, c->gid);
        BH_HIP(c, hipGetLastError());
    }
    c->n = n_new;
    c->partial_count = 0; c->slots_valid = false;
    c->samples_n = -1;                                        // new bodies: the next build sorts with the LSD passes
    c->tree_valid = false;
    c->orig_identity = true;                                // arrival order is the caller order from here on
    c->builds = 0;
    c->group_cost_valid = false;
    return BH_OK;
}

int bh_let_counts(bh_ctx *c, uint32_t *counts, int32_t *overflow)
{
    if (!c || !c->let_mode || !counts) return fail(c, BH_ERR_STATE, "bh_let_counts: not in LET mode");
    BH_HIP(c, hipSetDevice(c->device));
    BH_HIP(c, hipStreamSynchronize(c->stream));
    LetCounters h{};
    BH_HIP(c, hipMemcpy(&h, c->let_ctr, sizeof(h), hipMemcpyDeviceToHost));
    BH_HIP(c, hipMemset(c->let_ctr, 0, sizeof(LetCounters)));        // reading starts a new observation interval
    for (int r = 0; r < c->world; ++r) counts[r] = h.count[r];
    if (overflow) { *overflow = (int32_t)h.overflow; return BH_OK; }   // the caller inspects the flag
    if (h.overflow) return fail(c, BH_ERR_CAPACITY, "a locally-essential tree exceeded let_cap");
    return BH_OK;
}

}  // extern "C"

