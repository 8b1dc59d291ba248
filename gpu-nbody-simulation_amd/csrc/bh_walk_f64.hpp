// bh_walk_f64.hpp -- fp64 theta-walk for THROUGHPUT (BH_PRECISION_F64): the reference's arithmetic type
// (project.cu:38-65) with the walk design of the fp32 kernel instead of the reference's visiting order.
// Replaces computeForcesGpu (project.cu:679-793) + updateAccVelPos (project.cu:819-836) like bh_walk_exact.hpp,
// and runs on exactly the tree the exact mode builds (keys by fp64 bisection, stable sort, cells, bottom-up
// centre-of-mass pass in the reference's child order): every node -- box, mass, centre of mass, occupant --
// is BITWISE the reference's.  What differs from BH_PRECISION_F64_EXACT is the walk only:
//   * free visiting order.  The four children of an opened cell are four consecutive NodeD / LinkD records
//     (node ids 1+4r .. 4+4r): one 128-byte + one 32-byte scalar load per opened cell instead of one 40-byte load
//     per visited node, and all four are evaluated before the next load is issued.  The first child some lane
//     opens stays in scalar registers and is the next cell taken (no push / pop); the others go to a
//     register-lane stack (entry k in lane k of three VGPRs, 128 entries: depth-first needs <= 3 * max_depth + 1).
//     A body's terms are therefore added in another order than the reference's: results agree with the oracle
//     to summation rounding (tests: <= 1e-12 relative), not bit for bit.
//   * the acceptance criterion on d^2 (round 4).  The reference accepts a cell when size / d < theta with
//     d = sqrt(d2) + 1e-15 (project.cu:634, 643), i.e. when d2 > (size / theta - 1e-15)^2: the build stores that
//     right-hand side in the node's `size` slot (f64_walk_threshold, bh_tree.hpp; only this walk reads the slot in this
//     precision), and ONE compare on d2 decides -- the reciprocal square root and everything after it are needed by
//     the accepting lanes only and are skipped for the cells every lane opens.  Decisions can differ from the
//     oracle's only where size / d is within a few 1e-16 of theta: the tests find identical per-body interaction
//     counts.
//   * 1/d by v_rsq_f64 and ONE Newton step instead of IEEE sqrt and three divisions per interaction
//     (project.cu:634, 651-655): v_rsq_f64 is good to ~2^-26, so with e = 1 - d2 y0^2 the step y = y0 + y0 e / 2
//     leaves 3 e^2 / 8 ~ 3e-16; the force is G m_i M d_vec / (d2 * d) with 1 / d2 = y^2 and
//     1 / d = 1 / (sqrt(d2) + 1e-15) = y - 1e-15 y^2 (second order: 1e-20 relative for d >= 1e-5).
//   * self skip, empty-node cut-off, depth-cap aggregation: the reference's rules unchanged (occupant index,
//     `mass <= 1e-15`, project.cu:617, 646), so reference_compat means what it means in the exact mode.
//   * the traversal loop is hand-written gfx950 assembly (walk64_asm; round 4), the C++ loop beside it states the same
//     abstract machine with the same operations in the same order and gives the same bits
//     (BH_FLAG_WALK_PORTABLE / BH_FLAG_WALK_STATS select it; tests/test_gpu_f64.py).
// Explicit fma() throughout: this header is compiled with -ffp-contract=off like the rest of the engine unit.
#pragma once

#include "bh_tree.hpp"

namespace bh {

#define BH64_CONSTANT __attribute__((address_space(4)))
extern "C" __device__ int bh64_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

typedef int32_t w64_v16i __attribute__((ext_vector_type(16)));
typedef int32_t w64_v8i __attribute__((ext_vector_type(8)));

struct Quad64 {            // four sibling nodes as the scalar loads deliver them
    w64_v16i a, b;         // NodeD x 4: {cx, cy, m, thr} each, 8 dwords per node
    w64_v8i l;             // LinkD x 4: {child, occ}
};

__device__ __forceinline__ double w64_f64(int32_t lo, int32_t hi)
{
    return __hiloint2double(hi, lo);
}

// Four wavefronts per workgroup.  (Round 3's compiler-scheduled loop preferred one -- a workgroup's slot is only re-used when
// its LAST wave has finished, and its waves' walks take 0.3x .. 3x the mean: 1.088 -> 1.067 ms.  With the assembly loop at
// 8 resident waves it is the other way round, 0.682 -> 0.673 ms at N = 1M: profiles/r04_f64/walk_ab.txt.)
#ifndef BH_F64_BLOCK
#define BH_F64_BLOCK 256
#endif
constexpr int kF64Block = BH_F64_BLOCK;

// ---- the hand-written traversal loop ------------------------------------------------------------------------------
// What bounds this walk: every fp64 vector instruction costs a SIMD ~5.3 cycles per wave, v_rsq_f64 17
// (scripts/calib/f64_issue_calib.hip), so an evaluated node is ~26 vector cycles up to the compare and ~75 more where
// some lane accepts -- several times the scalar work around it, which other waves' vector instructions overlap.  Round 3's
// kernel wrapped hand-written arithmetic in compiler-scheduled control flow (21 scalar instructions and 8 branches per
// node; lane masks and the hand-off slot as booleans in SGPR pairs; the vector pipe 69 % busy).  Here, per child:
//   s_cmp + branch   empty-node cut-off on the mass's high dword (the exact `m <= 1e-15` out of line)
//   4 x fp64         dx, dy, d2
//   s_cmp + branch   leaf?  (a leaf narrows EXEC by the occupant test, v_cmpx_ne_u32, out of line)
//   v_cmpx_lt_f64    thr < d2: EXEC := the accepting lanes, vcc likewise
//   s_andn2 (SCC)    open = mask & ~vcc; nobody opens -> straight to the force
//   hand-off / push  the first opened child of the quad stays in s72 / s[68:69], the others are pushed (v_writelane x 3)
//   s_cbranch_execz  no lane accepts: the twelve instructions of the force are skipped
//   force            v_rsq_f64, Newton step (4), 1/d2, 1/d, weight (5), two v_fma_f64 into the sums -- under EXEC
//   s_mov exec       back to the quad's mask
// One quad per iteration (a quad is 40 SGPRs: two in flight do not fit the 102 a wave can have -- and they are not needed:
// a quad is ~300-400 vector cycles of work against a ~600-cycle round trip, three resident waves keep the pipe busy).
// Fixed SGPRs: s[24:55] the four nodes {cx, cy, m, thr} x 4, s[56:63] the links {child, occ} x 4, s[64:65] the quad's lane
// mask, s66 its index, s67 offsets, s[68:69] / s72 mask and index of the handed-over child (s72 == -1: none), s[70:71] open
// mask.  Fixed VGPRs: v[20:23] the body's position, v[24:27] dx dy, v[28:29] d2, v[30:31] y, v[32:35] scratch, v[36:39]
// the sums, v40 the body's index, v41 the index the reference's depth-cap occupant code compares equal to
// (reference_compat), v42..v44 / v45..v47 the stack (entry k in lane k & 63 of the first / second triple).
// Hazards the assembler does not handle inside inline assembly (gfx940 family): one wait state between v_rsq_f64 and its
// use; four between a VALU write of EXEC (v_cmpx) and v_writelane / v_readfirstlane -- the push path has s_andn2, two
// branches, an s_cmp and an s_nop in between, a pop is reached through the loop head; m0 is written at least one
// instruction before a lane select uses it.
// -DBH_ASM_GUARD=1 (first run of a rewritten loop on hardware): a wave leaves after 2^22 iterations whatever its
// stack says.
#if defined(BH_ASM_GUARD) && BH_ASM_GUARD
#define BH64_GUARD_INIT "s_mov_b32 s73, 0\n"
#define BH64_GUARD "s_add_u32 s73, s73, 1\n s_cmp_gt_u32 s73, 0x400000\n s_cbranch_scc1 Ldone_%=\n"
#define BH64_GUARD_CLOBBER "s73",
#else
#define BH64_GUARD_INIT ""
#define BH64_GUARD ""
#define BH64_GUARD_CLOBBER
#endif
// A/B hooks (scripts/f64_variants.sh; the product builds with neither): -DBH64_SETPRIO=1 raises a wave's priority from an
// iteration's start to its s_waitcnt as the fp32 loop does; -DBH64_LDS_PAD=bytes pads the workgroup's LDS to limit residency.
#if defined(BH64_SETPRIO) && BH64_SETPRIO
#define BH64_PRIO_UP "s_setprio 2\n"
#define BH64_PRIO_DOWN "s_setprio 0\n"
#else
#define BH64_PRIO_UP ""
#define BH64_PRIO_DOWN ""
#endif
#define BH64_FORCE(M)                                                                               \
    "v_rsq_f64_e32 v[30:31], v[28:29]\n"                                                            \
    "s_nop 0\n"                                                                                     \
    "v_mul_f64 v[32:33], v[28:29], v[30:31]\n"              /* t = d2 y0                */         \
    "v_fma_f64 v[34:35], -v[32:33], v[30:31], 1.0\n"        /* e = 1 - d2 y0^2          */         \
    "v_mul_f64 v[32:33], v[30:31], v[34:35]\n"              /* h = y0 e                 */         \
    "v_fma_f64 v[30:31], v[32:33], 0.5, v[30:31]\n"         /* y = y0 + h / 2           */         \
    "v_mul_f64 v[34:35], v[30:31], v[30:31]\n"              /* a = 1 / d2               */         \
    "v_fma_f64 v[32:33], v[34:35], %[ntiny], v[30:31]\n"    /* b = y - 1e-15 y^2 = 1/d  */         \
    "v_mul_f64 v[34:35], v[34:35], " M "\n"                 /* a M                      */         \
    "v_mul_f64 v[34:35], v[34:35], v[32:33]\n"              /* w = M / (d2 d)           */         \
    "v_fma_f64 v[36:37], v[34:35], v[24:25], v[36:37]\n"                                            \
    "v_fma_f64 v[38:39], v[34:35], v[26:27], v[38:39]\n"
#define BH64_CHILD(CX, CY, M, MHI, THR, CS, TAG)                                                    \
    "s_cmp_gt_i32 " MHI ", 0x3cd203af\n"                    /* m > 1e-15 for sure */                \
    "s_cbranch_scc0 Lrare" TAG "_%=\n"                                                              \
    "Lcont" TAG "_%=:\n"                                                                            \
    "v_add_f64 v[24:25], " CX ", -v[20:21]\n"                                                       \
    "v_add_f64 v[26:27], " CY ", -v[22:23]\n"                                                       \
    "s_cmp_lt_i32 " CS ", 0\n"                                                                      \
    "v_mul_f64 v[28:29], v[26:27], v[26:27]\n"                                                      \
    "v_fma_f64 v[28:29], v[24:25], v[24:25], v[28:29]\n"                                            \
    "s_cbranch_scc1 Lleaf" TAG "_%=\n"                                                              \
    "v_cmpx_lt_f64_e32 vcc, " THR ", v[28:29]\n"                                                    \
    "s_andn2_b64 s[70:71], s[64:65], vcc\n"                                                         \
    "s_cbranch_scc0 Lforce" TAG "_%=\n"                     /* nobody opens */                      \
    "s_cmp_gt_i32 s72, -1\n"                                                                        \
    "s_cbranch_scc1 Lpush" TAG "_%=\n"                                                              \
    "s_mov_b32 s72, " CS "\n"                               /* handed over in registers */          \
    "s_mov_b64 s[68:69], s[70:71]\n"                                                                \
    "Lforce" TAG "_%=:\n"                                                                           \
    "s_cbranch_execz Lskip" TAG "_%=\n"                                                             \
    BH64_FORCE(M)                                                                                   \
    "Lskip" TAG "_%=:\n"                                                                            \
    "s_mov_b64 exec, s[64:65]\n"                                                                    \
    "Lnext" TAG "_%=:\n"
// out of line: the exact empty test (bits of 1e-15: 0x3CD203AF'9EE75616; for doubles that are not NaN a <= b is the same
// on their sign-magnitude integers, and a node's mass is a sum of the bodies' masses or +0.0), the occupant test of a
// leaf (project.cu:646), the push
#define BH64_STUBS(MLO, MHI, CS, OS, TAG, COMPAT_CMP, PUSHCHK)                                      \
    "Lrare" TAG "_%=:\n"                                                                            \
    "s_cmp_lt_i32 " MHI ", 0x3cd203af\n"                                                            \
    "s_cbranch_scc1 Lnext" TAG "_%=\n"                                                              \
    "s_cmp_le_u32 " MLO ", 0x9ee75616\n"                                                            \
    "s_cbranch_scc1 Lnext" TAG "_%=\n"                                                              \
    "s_branch Lcont" TAG "_%=\n"                                                                    \
    "Lleaf" TAG "_%=:\n"                                                                            \
    "v_cmpx_ne_u32_e32 vcc, " OS ", v40\n"                                                          \
    COMPAT_CMP(OS)                                                                                  \
    "s_branch Lforce" TAG "_%=\n"                                                                   \
    "Lpush" TAG "_%=:\n"                                                                            \
    "s_nop 0\n"                                                                                     \
    PUSHCHK(TAG)                                                                                    \
    "v_writelane_b32 v42, " CS ", m0\n"                                                             \
    "v_writelane_b32 v43, s70, m0\n"                                                                \
    "v_writelane_b32 v44, s71, m0\n"                                                                \
    "LpushBack" TAG "_%=:\n"                                                                        \
    "s_add_u32 m0, m0, 1\n"                                                                         \
    "s_branch Lforce" TAG "_%=\n"
#define BH64_COMPAT_ON(OS) "v_cmpx_ne_u32_e32 vcc, " OS ", v41\n"
#define BH64_COMPAT_OFF(OS) ""
// (the lane select of v_writelane and the shift count of s_lshl_b64 use m0[5:0]: entry k sits in lane k & 63)
#define BH64_PUSHCHK(TAG) "s_bitcmp1_b32 m0, 6\n s_cbranch_scc1 LpushHi" TAG "_%=\n"
#define BH64_NOCHK(TAG) ""
#define BH64_PUSH_HI(CS, TAG)                                                                       \
    "LpushHi" TAG "_%=:\n"                                                                          \
    "v_writelane_b32 v45, " CS ", m0\n"                                                             \
    "v_writelane_b32 v46, s70, m0\n"                                                                \
    "v_writelane_b32 v47, s71, m0\n"                                                                \
    "s_branch LpushBack" TAG "_%=\n"
#define BH64_POP_FAST                                                                               \
    "s_lshl_b64 exec, 1, m0\n"                                                                      \
    "v_readfirstlane_b32 s66, v42\n v_readfirstlane_b32 s64, v43\n v_readfirstlane_b32 s65, v44\n"
#define BH64_POP_DEEP                                                                               \
    "s_lshl_b64 exec, 1, m0\n"                                                                      \
    "s_bitcmp1_b32 m0, 6\n"                                                                         \
    "s_cbranch_scc1 LpopHi_%=\n"                                                                    \
    "v_readfirstlane_b32 s66, v42\n v_readfirstlane_b32 s64, v43\n v_readfirstlane_b32 s65, v44\n"  \
    "s_branch Lload_%=\n"                                                                           \
    "LpopHi_%=:\n"                                                                                  \
    "v_readfirstlane_b32 s66, v45\n v_readfirstlane_b32 s64, v46\n v_readfirstlane_b32 s65, v47\n"
#define BH64_LOOP(POP, COMPAT_CMP, PUSHCHK, HI_STUBS)                                               \
    "v_mov_b32_e32 v20, %[pxl]\n v_mov_b32_e32 v21, %[pxh]\n"                                       \
    "v_mov_b32_e32 v22, %[pyl]\n v_mov_b32_e32 v23, %[pyh]\n"                                       \
    "v_mov_b32_e32 v36, %[sxl]\n v_mov_b32_e32 v37, %[sxh]\n"                                       \
    "v_mov_b32_e32 v38, %[syl]\n v_mov_b32_e32 v39, %[syh]\n"                                       \
    "v_mov_b32_e32 v40, %[body]\n"                                                                  \
    "v_sub_u32_e32 v41, -2, v40\n"                          /* occ + 2 == -body, project.cu:646 */  \
    "s_mov_b32 m0, 0\n"                                                                             \
    BH64_GUARD_INIT                                                                                 \
    "Lloop_%=:\n"                                                                                   \
    BH64_GUARD                                                                                      \
    BH64_PRIO_UP                                                                                    \
    "s_cmp_gt_i32 s72, -1\n"                                                                        \
    "s_cbranch_scc1 Lhave_%=\n"                                                                     \
    "s_sub_u32 m0, m0, 1\n"                                 /* SCC = borrow: the stack was empty */ \
    "s_cbranch_scc1 Ldone_%=\n"                                                                     \
    POP                                                                                             \
    "s_branch Lload_%=\n"                                                                           \
    "Lhave_%=:\n"                                                                                   \
    "s_mov_b32 s66, s72\n"                                                                          \
    "s_mov_b64 s[64:65], s[68:69]\n"                                                                \
    "Lload_%=:\n"                                                                                   \
    "s_lshl_b32 s67, s66, 5\n"                                                                      \
    "s_load_dwordx16 s[24:39], %[gd], s67\n"                                                        \
    "s_load_dwordx16 s[40:55], %[gd], s67 offset:0x40\n"                                            \
    "s_lshl_b32 s67, s66, 3\n"                                                                      \
    "s_load_dwordx8 s[56:63], %[ld], s67\n"                                                         \
    "s_mov_b32 s72, -1\n"                                                                           \
    "s_mov_b64 exec, s[64:65]\n"                                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                        \
    BH64_PRIO_DOWN                                                                                  \
    BH64_CHILD("s[24:25]", "s[26:27]", "s[28:29]", "s29", "s[30:31]", "s56", "0")                   \
    BH64_CHILD("s[32:33]", "s[34:35]", "s[36:37]", "s37", "s[38:39]", "s58", "1")                   \
    BH64_CHILD("s[40:41]", "s[42:43]", "s[44:45]", "s45", "s[46:47]", "s60", "2")                   \
    BH64_CHILD("s[48:49]", "s[50:51]", "s[52:53]", "s53", "s[54:55]", "s62", "3")                   \
    "s_branch Lloop_%=\n"                                                                           \
    BH64_STUBS("s28", "s29", "s56", "s57", "0", COMPAT_CMP, PUSHCHK)                                \
    BH64_STUBS("s36", "s37", "s58", "s59", "1", COMPAT_CMP, PUSHCHK)                                \
    BH64_STUBS("s44", "s45", "s60", "s61", "2", COMPAT_CMP, PUSHCHK)                                \
    BH64_STUBS("s52", "s53", "s62", "s63", "3", COMPAT_CMP, PUSHCHK)                                \
    HI_STUBS                                                                                        \
    "Ldone_%=:\n"                                                                                   \
    BH64_PRIO_DOWN                                                                                  \
    "s_mov_b64 exec, -1\n"                                  /* (the kernel runs the traversal with all lanes enabled) */ \
    "v_mov_b32_e32 %[sxl], v36\n v_mov_b32_e32 %[sxh], v37\n"                                       \
    "v_mov_b32_e32 %[syl], v38\n v_mov_b32_e32 %[syh], v39\n"
#define BH64_HI_STUBS BH64_PUSH_HI("s56", "0") BH64_PUSH_HI("s58", "1") BH64_PUSH_HI("s60", "2") BH64_PUSH_HI("s62", "3")
#define BH64_OPERANDS                                                                               \
    : [sxl] "+v"(sxl), [sxh] "+v"(sxh), [syl] "+v"(syl), [syh] "+v"(syh), "+{s72}"(first), "+{s[68:69]}"(mask)         \
    : [gd] "s"(gd), [ld] "s"(ld), [ntiny] "s"(ntiny), [pxl] "v"(pxl), [pxh] "v"(pxh), [pyl] "v"(pyl), [pyh] "v"(pyh),  \
      [body] "v"(body)                                                                              \
    : BH64_GUARD_CLOBBER                                                                            \
      "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",   \
      "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",   \
      "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s70", "s71",                 \
      "m0", "vcc", "scc", "memory",                                                                 \
      "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",   \
      "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"

// Walks the subtree below quad `first` (node id of its first sibling) for the lanes in `mask`; the sums continue from
// (sx, sy).  gd / ld: the node and link arrays; byte offsets are 32-bit (the caller checks node_cap * 32 < 4 GiB).
template <bool COMPAT, bool DEEP>
__device__ __forceinline__ void walk64_asm(const char BH64_CONSTANT *gd, const char BH64_CONSTANT *ld, int32_t first, uint64_t mask,
                                           double px, double py, int32_t body, double &sx, double &sy)
{
    const double ntiny = -1e-15;
    int32_t pxl = __double2loint(px), pxh = __double2hiint(px), pyl = __double2loint(py), pyh = __double2hiint(py);
    int32_t sxl = __double2loint(sx), sxh = __double2hiint(sx), syl = __double2loint(sy), syh = __double2hiint(sy);
    if constexpr (DEEP) {
        if constexpr (COMPAT) asm volatile(BH64_LOOP(BH64_POP_DEEP, BH64_COMPAT_ON, BH64_PUSHCHK, BH64_HI_STUBS) BH64_OPERANDS);
        else asm volatile(BH64_LOOP(BH64_POP_DEEP, BH64_COMPAT_OFF, BH64_PUSHCHK, BH64_HI_STUBS) BH64_OPERANDS);
    } else {
        if constexpr (COMPAT) asm volatile(BH64_LOOP(BH64_POP_FAST, BH64_COMPAT_ON, BH64_NOCHK, "") BH64_OPERANDS);
        else asm volatile(BH64_LOOP(BH64_POP_FAST, BH64_COMPAT_OFF, BH64_NOCHK, "") BH64_OPERANDS);
    }
    sx = w64_f64(sxl, sxh);
    sy = w64_f64(syl, syh);
}

// DEEP: trees deeper than 21 levels need more than 64 stack entries (3 * (max_depth - 1) + 1): a second register-lane
// tier, and a test on every push and pop that the usual depth does without.
// ASM: the hand-written loop (no counters); otherwise the C++ statement of the same machine.
struct WalkF64Args {
    const NodeD *gd;
    const LinkD *ld;
    const uint32_t *perm;
    double2 *pos, *vel;
    const double *mass;
    double2 *force_out;
    int64_t lo, hi;
    double G, dt;
    int32_t integrate, pad0;
    TreeCounters *ctr;
    double *partial;               // per-workgroup min/max of the new positions (next root box), may be null
    uint32_t *body_counts;         // counting variant: accepted force evaluations per body, may be null
    double *slots;                 // bh_bounds.hpp: running bounds records, may be null
    int32_t bpw, pad1;             // bodies per wavefront, a power of two <= 64 (see walk_exact_kernel): few bodies, short chains
};

template <bool COMPAT, bool STATS, bool DEEP = false, bool ASM = false>
__global__ __launch_bounds__(kF64Block) void walk_f64_kernel(WalkF64Args a)
{
    static_assert(!ASM || !STATS, "the assembly loop carries no counters");
#if defined(BH64_LDS_PAD) && BH64_LDS_PAD
    __shared__ int s_pad[BH64_LDS_PAD / 4];
    if (a.dt == -12345.0) s_pad[threadIdx.x] = 1;                // (never true: keeps the array alive)
    asm volatile("" ::"v"(s_pad[0]));
#endif
    if (a.ctr->overflow) return;
    const int lane = lane_id();
    const int64_t s = a.lo + ((int64_t)blockIdx.x * (kF64Block / kWave) + wave_id()) * a.bpw + lane;
    const bool valid = lane < a.bpw && s < a.hi;
    const int64_t body = valid ? (int64_t)a.perm[s] : -1;
    const double2 p = valid ? a.pos[body] : double2{0.0, 0.0};
    const double mi = valid ? a.mass[body] : 0.0;
    const NodeD *gd = a.gd;
    const LinkD *ld = a.ld;
    double sx = 0.0, sy = 0.0;                       // sum of M * d_vec / (d2 * d)
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0, n_quad = 0, n_acc = 0;
    uint32_t my_int = 0;

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const char BH64_CONSTANT *cg = (const char BH64_CONSTANT *)gd;
    const char BH64_CONSTANT *cl = (const char BH64_CONSTANT *)ld;
#pragma clang diagnostic pop
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

    // one node for the lanes in `mask` (all arguments but the position wave-uniform): the statement of BH64_CHILD
    const int32_t body32 = (int32_t)body;                         // (perm is 32-bit; -1 on padding lanes)
    const int32_t compat32 = -2 - body32;                         // occ + 2 == -body, project.cu:646
    auto eval = [&](double cx, double cy, double m, double thr, int32_t child, int32_t occ, uint64_t mask) {
        // m <= 1e-15 (project.cu:617) on the bit pattern, with scalar integer compares
        {
            const int32_t mh = __double2hiint(m);
            if (__builtin_expect(mh <= 0x3CD203AF, 0)) {
                if (mh < 0x3CD203AF) return;
                if ((uint32_t)__double2loint(m) <= 0x9EE75616u) return;
            }
        }
        const double dx = cx - p.x, dy = cy - p.y;
        const double d2 = fma(dx, dx, dy * dy);
        uint64_t takem, open;
        if (child < 0) {                                          // a leaf: everybody but its occupant (project.cu:623-626, 646)
            uint64_t self = __builtin_amdgcn_ballot_w64(occ == body32);
            if (COMPAT) self |= __builtin_amdgcn_ballot_w64(occ == compat32);
            takem = mask & ~self;
            open = 0;
        } else {                                                  // size / d < theta (project.cu:643) as thr < d2, see the header
            const uint64_t acc = __builtin_amdgcn_ballot_w64(thr < d2);
            takem = mask & acc;
            open = mask & ~acc;
        }
        if (open != 0) {
            if (h_idx < 0) { h_idx = child; h_mask = open; }
            else push(child, open);
        }
        if (takem != 0) {
            // M / (d2 * d):  1 / d2 = y * y,  1 / d = 1 / (sqrt(d2) + 1e-15) = y - 1e-15 y^2 to second order; added for the
            // accepting lanes only (the others may hold inf / NaN here: a body's own leaf has d2 = 0)
            const double y0 = __builtin_amdgcn_rsq(d2);
            const double t = d2 * y0;
            const double e = fma(-t, y0, 1.0);
            const double h = y0 * e;
            const double y = fma(h, 0.5, y0);
            const double a = y * y;
            const double b = fma(a, -1e-15, y);
            const double w = (a * m) * b;
            if ((takem >> lane) & 1ull) { sx = fma(w, dx, sx); sy = fma(w, dy, sy); }
        }
        if (STATS) { n_vis += __popcll(mask); ++n_wave; n_int += __popcll(takem); n_acc += takem != 0; my_int += (uint32_t)((takem >> lane) & 1ull); }
    };
    auto node_of = [&](const Quad64 &q, int k, double &cx, double &cy, double &m, double &thr) {
        const w64_v16i &t = (k < 2) ? q.a : q.b;
        const int o = (k & 1) * 8;
        cx = w64_f64(t[o + 0], t[o + 1]); cy = w64_f64(t[o + 2], t[o + 3]);
        m = w64_f64(t[o + 4], t[o + 5]); thr = w64_f64(t[o + 6], t[o + 7]);
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
    if (ASM) {
        if (na >= 0) walk64_asm<COMPAT, DEEP>(cg, cl, na, nam, p.x, p.y, body32, sx, sy);
    } else {
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
                double cx, cy, m, thr;
                node_of(q, k, cx, cy, m, thr);
                eval(cx, cy, m, thr, q.l[2 * k], q.l[2 * k + 1], mask);
            }
            na = h_idx; nam = h_mask;
        }
    }

    // Epilogue.  Its arguments are read AGAIN from the kernarg segment through a laundered pointer: the compiler otherwise
    // keeps the ones used here alive in SGPRs across the traversal loop -- 82 SGPRs, 7 resident waves per SIMD; at most 80
    // is 8, and this walk answers to residency (measured, profiles/r04_f64/walk_ab.txt: 8 / 7 / 5 / 4 / 3 waves).
    const WalkF64Args BH64_CONSTANT *ka;
    {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        ka = (const WalkF64Args BH64_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
    }
    asm volatile("" : "+s"(ka));
    const WalkF64Args BH64_CONSTANT &e = *ka;
    double2 np = p;
    if (valid) {
        const double gm = e.G * mi;                               // (G * masses[i]) * nodeMass / d2 * d_vec / d, project.cu:651-658
        const double fx = gm * sx, fy = gm * sy;
        e.force_out[body] = double2{fx, fy};
        if (e.integrate) {
            const double ax = e.G * sx, ay = e.G * sy;            // F / m_i (updateAccVelPos, project.cu:827-834)
            double2 v = e.vel[body];
            v.x = fma(ax, e.dt, v.x);  v.y = fma(ay, e.dt, v.y);
            e.vel[body] = v;
            np.x = fma(v.x, e.dt, np.x);  np.y = fma(v.y, e.dt, np.y);
            e.pos[body] = np;
        }
        if (STATS && e.body_counts) e.body_counts[body] = my_int;
    }
    if (e.partial) {                                              // min/max of the new positions per workgroup (next root box)
        if (kF64Block == kWave) {
            const double xlo = wave_min(valid ? np.x : (double)INFINITY), xhi = wave_max(valid ? np.x : -(double)INFINITY);
            const double ylo = wave_min(valid ? np.y : (double)INFINITY), yhi = wave_max(valid ? np.y : -(double)INFINITY);
            if (lane == 0) {
                double *o = e.partial + 4 * (size_t)blockIdx.x;
                o[0] = xlo; o[1] = xhi; o[2] = ylo; o[3] = yhi;
                if (e.slots) bounds_to_slot(xlo, xhi, ylo, yhi, e.slots, blockIdx.x);
            }
        } else {
            block_bounds_to_partial(valid, np.x, np.y, e.partial + 4 * (size_t)blockIdx.x, e.slots);
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&e.ctr->visits, n_vis);
        atomicAdd(&e.ctr->interactions, n_int);
        atomicAdd(&e.ctr->wave_nodes, n_wave);
        atomicAdd(&e.ctr->wave_quads, n_quad);
        atomicAdd(&e.ctr->wave_accepts, n_acc);
    }
}

}  // namespace bh
