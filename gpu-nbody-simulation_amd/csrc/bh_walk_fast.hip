// bh_walk_fast.hip -- fp32 theta-walk + integrator, the throughput path (BASELINE configs
// "fp32").  Replaces computeForcesGpu (project.cu:679-793) and updateAccVelPos
// (project.cu:819-836); designed for CDNA4 wave64, not translated from them.
//
//   * one wavefront = 64 curve-adjacent (Hilbert order) bodies, one per lane; the traversal state is
//     wave-uniform, so node reads are scalar loads (through the scalar data cache) broadcast to all
//     lanes for free -- the reference's per-thread walk re-reads each 96-byte node once per body
//     (project.cu:726), this reads 20 bytes per node once per wave;
//   * the four children of a subdivided cell are one 80-byte structure-of-arrays record and are
//     evaluated together: one memory round trip per opened cell instead of one per node, and that
//     round trip is overlapped with the evaluation of the previous quad (see the kernel);
//   * a stack entry is {child quad, 64-bit lane mask of the bodies that opened the parent}.  The
//     default stack lives in three VGPRs addressed by lane (v_writelane/v_readlane): entry k sits
//     in lane k, so push/pop are single VALU instructions with no LDS round trip.  The LDS
//     variant (BH_FLAG_LDS_STACK, and automatically for max_depth > 21 where 64 entries do not
//     suffice) keeps the same entries in LDS; DESIGN.md quotes the measured difference.
//   * MAC per body exactly as the reference's (size/dist < theta, evaluated per lane), in the
//     algebraically equal form d2 > (size/theta)^2 with the right side precomputed per node; the
//     same compare doubles as the self test for leaves (thr = 0) -- see eval().
//   * force per accepted node: G*M*d/(|d|^3) through v_rsq_f32; the reference's 1e-15 offset on
//     dist (project.cu:634) is below fp32 resolution and omitted; a node at distance exactly 0
//     (the body itself, or an exactly coincident body) contributes nothing.
//   * epilogue: a = G*sum, v += a*dt, p += v*dt written back in caller order (scatter through
//     perm), or into the sorted arrays for the multi-GPU exchange; plus the per-workgroup min/max of
//     the new positions for the next step's root box.
//   * SPLIT > 1 (few bodies: N <= 192k on one GPU, or one rank's share of a multi-GPU run).  The
//     walk of a 64-body group is a dependent chain of ~200 quad visits; a lone wave spends ~370
//     cycles waiting for each quad and ~1000 issuing its evaluation (measured with s_memtime),
//     ~0.1 ms in all however empty the GPU is, while a SIMD with 16 resident waves retires a quad
//     every ~270 cycles.  With few groups a workgroup of SPLIT waves therefore shares ONE group and
//     walks the forest LEVEL-SYNCHRONOUSLY: the frontier (quads some lane opened) of the current
//     level sits in LDS, every wave takes an equal contiguous chunk of it, evaluates it with the
//     usual per-body MAC (pushes go to its register-lane stack), and the waves' pushes are
//     concatenated in wave order into the next level's frontier.  Chunks, offsets and the final
//     wave-order sum of the partial accelerations are all fixed by the data, so results are
//     reproducible run to run (they differ from the one-wave walk by fp32 summation order only).
#include "bh_prims.hpp"
#include "bh_nodes.hpp"
#include "bh_bounds.hpp"
#include "bh_walk_fast.h"
#include <cstdlib>

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

// one sibling quad (80 B, structure of arrays) through the scalar data cache: s_load_dwordx16 +
// s_load_dwordx4 at a wave-uniform address
typedef int32_t v16i __attribute__((ext_vector_type(16)));
typedef int32_t v4i __attribute__((ext_vector_type(4)));
typedef int32_t v2i __attribute__((ext_vector_type(2)));
struct QuadRegs { v16i g; v4i c; };

__device__ __forceinline__ QuadRegs load_quad(const QuadF BH_CONSTANT *q)
{
    QuadRegs r;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    r.g = *(const v16i BH_CONSTANT *)q;
    r.c = *(const v4i BH_CONSTANT *)((const char BH_CONSTANT *)q + 64);
#pragma clang diagnostic pop
    return r;
}

// ---- hand-scheduled traversal of ONE tree (the default path: register-lane stack, no counters) -------
// Why assembly, and what bounds it.  Measured on MI355X (scripts/calib/issue_calib.hip, 8 waves per SIMD) a SIMD
// spends ~4.2 cycles per SCALAR instruction, ~2.2 per plain fp32 VALU instruction, ~4.2 per VALU instruction with an
// SGPR operand or an SGPR-pair result (v_cmp), ~8.3 per v_rsq_f32 / v_readlane_b32; the scalar and the vector
// stream of different waves overlap.  PMC of the loop (profiles/r03_*): the vector pipe is busy ~91 % and the
// scalar side ~80-90 % of the kernel's cycles -- BOTH issue streams are nearly full, so an instruction removed
// from one of them returns about a third of its cost, and an instruction added to the fuller one costs all of
// it (round 3: hoisting the four children's distance / rsqrt math in front of the per-child logic for ILP spent
// +12 % vector instructions on empty children and took +12 % time; a register hand-off that removed 6 % of the
// vector instructions for 6 % more scalar ones returned 2 %).  The loop is therefore written for the fewest
// instructions of BOTH kinds:
//   * EXEC holds the entry's lane mask for the whole quad and v_cmpx narrows it to the accepting lanes: no
//     v_cndmask, no s_and of the ballot with the mask, the force math runs under EXEC;
//   * `open = mask & ~vcc` sets SCC, which is the push decision: 1 SALU + 1 branch for the ~70 % of the
//     children nobody opens; the leaf test (child == -1) is only reached by the rest;
//   * the stack pointer lives in m0 (the lane select of v_writelane, the shift of s_lshl_b64 exec), the quad
//     address is one s_mul_i32 feeding the SGPR-offset form of s_load;
//   * REGISTER HAND-OFF (round 3).  Every quad but the root used to be pushed once (three v_writelane) and popped
//     once (three v_readfirstlane) although its entry sits in SGPRs when it is made.  Now the FIRST child of A
//     that some lane opens stays in scalar registers and is the next iteration's A (NA: index s72, lane mask
//     s[68:69], moved to s[44:45] when A's own mask dies); the first opened child of B becomes the next B (NB:
//     index s70, mask s[68:69] -> s[46:47]).  Only the other opened children go through the VGPR stack.  Which
//     child opens first is known at run time only, so each child block exists in two flavours -- "slot free"
//     (BH_CHILD_F: the open mask is computed straight into the hand-off pair; a taken child costs one s_mov and
//     leaves through BH_TAKE, which carries a copy of the force math and continues in the other chain) and "slot
//     taken" (BH_CHILD_T: push) -- and the program counter remembers the state;
//   * s_setprio 2 between an iteration's start and its s_waitcnt: the waves that are about to issue their loads go
//     first, the others are evaluating (-1.3 %; priorities 1..3 measure the same).
// Abstract machine (the C++ loop in the kernel implements the same one, bit for bit --
// tests/test_gpu_fp32.py::test_asm_walk_equals_the_portable_walk):
//   A := NA, else the stack's top (a bucket reference is served on the spot and the iteration ends), else NB, else done;
//   B := NB, else -- if the stack is not empty and holds at most pair_limit entries -- its top (a bucket reference
//        is left there), else none;            [two quads in flight per wave: one s_waitcnt serves both]
//   evaluate A: the first opened child that is a quad becomes NA, other opened children are pushed;
//   evaluate B: likewise with NB, but only if the stack holds at most pair_limit entries when B's evaluation starts.
// Stack bound.  Entries live in six VGPRs addressed by lane: 128 entries.  An iteration with a B is entered with at
// most pair_limit + 3 entries (an NB was taken at <= pair_limit entries and at most 3 pushes followed; a popped B
// leaves <= pair_limit - 1) and pushes at most 8; from S entries a run of single-quad iterations (depth first:
// three waiting siblings per level, four at the last) never holds more than S + 3 * Dm + 1; with pair_limit =
// 116 - 3 * Dm (56 at max_depth 21, 23 at 32) the 128 entries always suffice -- the engine passes it in (0 = never
// pair).  An iteration that starts with at most 56 entries touches only the first register triple and runs a
// copy of the loop without the "which triple?" tests.
// Hazards handled by instruction order (the assembler inserts no wait states into inline assembly): the consumer
// of v_rsq_f32 is separated from it by the scalar push logic; the consumer of v_pk_add_f32 by an s_nop;
// m0 is written at least one instruction before a lane select uses it.
// Fixed SGPRs: s[24:43] quad A, s[44:45] its lane mask, s[46:47] B's mask (0: no B in flight), s[48:67] quad B,
// s68 / s69 A's quad index and byte offset and, once the loads are issued, the hand-off pair (also the open-mask
// scratch of the "free" flavour), s70 / s71 B's index and offset, then the open-mask scratch of the "taken" flavour
// while A is evaluated (s[24:25] while B is: quad A is dead by then), s72 NA's index (-1: none), s70 NB's index
// between iterations (-1: none).  The block sits right above the ~20 SGPRs the compiler keeps live across the
// loop: 79 in all, and a wave's allocation is its count + 16 rounded up to 16 out of 800 per SIMD -- 80 is the
// last value that leaves 8 resident waves (tests/test_kernel_resources_cpu.py).
// Fixed VGPRs (the two-dword operands of v_pk_add_f32 need named halves): v[20:21] body position, v[22:23] dx,dy,
// v24 d2 then w, v25 1/d, v26 scratch, v[28:29] acceleration sums, v30..v32 / v33..v35 the stack.
// -DBH_ASM_GUARD=1 (first run of a rewritten loop on hardware): a wave leaves after 2^20 iterations whatever its
// stack says.
#if defined(BH_ASM_GUARD) && BH_ASM_GUARD
#define BH_LOOP_GUARD "s_cmp_gt_u32 %[cost], 0x100000\n s_cbranch_scc1 Ldone_%=\n"
#else
#define BH_LOOP_GUARD ""
#endif
#define BH_FORCE_MATH(MS, MASK)                                                                     \
    "v_mul_f32_e32 v26, " MS ", v25\n"                                                              \
    "v_mul_f32_e32 v26, v25, v26\n"                                                                 \
    "v_mul_f32_e32 v24, v25, v26\n"                                                                 \
    "v_fmac_f32_e32 v28, v24, v22\n"                                                                \
    "v_fmac_f32_e32 v29, v24, v23\n"                                                                \
    "s_mov_b64 exec, " MASK "\n"
// The empty-cell test comes FIRST (end of round 3): one child in six of an evaluated quad is empty (17.7 M child slots,
// 14.7 M non-empty per launch at N = 1M), and the loop is made of vector cycles -- a v_pk_add_f32 per empty child was
// 1.6 % of them.  The wait state a packed result needs before it is read is an s_nop now (it sat in the two scalar
// instructions of the test): scalar issue, which other waves' vector instructions overlap.  Walk 0.3130 -> 0.3088 ms.
#define BH_CHILD_HEAD(XY, MS, TS, MASK, SPAIR, TAG)                                                 \
    "s_cmp_eq_u32 " MS ", 0\n"                                                                      \
    "s_cbranch_scc1 Lnext" TAG "_%=\n"                                                              \
    "v_pk_add_f32 v[22:23], " XY ", v[20:21] neg_lo:[0,1] neg_hi:[0,1]\n"                           \
    "s_nop 0\n"                                                                                     \
    "v_mul_f32_e32 v24, v23, v23\n"                                                                 \
    "v_fmac_f32_e32 v24, v22, v22\n"                                                                \
    "v_cmpx_lt_f32_e32 vcc, " TS ", v24\n"                                                          \
    "v_rsq_f32_e32 v25, v24\n"                                                                      \
    "s_andn2_b64 " SPAIR ", " MASK ", vcc\n"                                                        \
    "s_cbranch_scc0 Lforce" TAG "_%=\n"
// slot taken (and the level-synchronous list walk): opened children -- quads and bucket references -- are pushed.
// PUSHCHK: "" where every entry touched sits in lanes of v30..v32, BH_PUSHCHK(TAG) otherwise (entries 64..127
// live in v33..v35)
#define BH_CHILD_T(XY, MS, TS, CS, MASK, SPAIR, SLO, SHI, TAG, PUSHCHK)                             \
    BH_CHILD_HEAD(XY, MS, TS, MASK, SPAIR, TAG)                                                     \
    "s_cmp_eq_u32 " CS ", -1\n"                                                                     \
    "s_cbranch_scc1 Lforce" TAG "_%=\n"                                                             \
    PUSHCHK                                                                                         \
    "v_writelane_b32 v30, " CS ", m0\n"                                                             \
    "v_writelane_b32 v31, " SLO ", m0\n"                                                            \
    "v_writelane_b32 v32, " SHI ", m0\n"                                                            \
    "LpushBack" TAG "_%=:\n"                                                                        \
    "s_add_u32 m0, m0, 1\n"                                                                         \
    "Lforce" TAG "_%=:\n"                                                                           \
    BH_FORCE_MATH(MS, MASK)                                                                         \
    "Lnext" TAG "_%=:\n"
// slot free: the open mask lands in the hand-off pair s[68:69]; an opened quad is taken (BH_TAKE, out of line),
// an opened bucket reference is pushed, an "opened" leaf (the body itself, d2 == 0) is nothing
#define BH_CHILD_F(XY, MS, TS, CS, MASK, TAG, PUSHCHK)                                              \
    BH_CHILD_HEAD(XY, MS, TS, MASK, "s[68:69]", TAG)                                                \
    "s_cmp_gt_i32 " CS ", 0\n"                                                                      \
    "s_cbranch_scc1 Ltake" TAG "_%=\n"                                                              \
    "s_cmp_eq_u32 " CS ", -1\n"                                                                     \
    "s_cbranch_scc1 Lforce" TAG "_%=\n"                                                             \
    PUSHCHK                                                                                         \
    "v_writelane_b32 v30, " CS ", m0\n"                                                             \
    "v_writelane_b32 v31, s68, m0\n"                                                                \
    "v_writelane_b32 v32, s69, m0\n"                                                                \
    "LpushBack" TAG "_%=:\n"                                                                        \
    "s_add_u32 m0, m0, 1\n"                                                                         \
    "Lforce" TAG "_%=:\n"                                                                           \
    BH_FORCE_MATH(MS, MASK)                                                                         \
    "Lnext" TAG "_%=:\n"
#define BH_TAKE(MS, CS, MASK, NIDX, TAG, NEXT)                                                      \
    "Ltake" TAG "_%=:\n"                                                                            \
    "s_mov_b32 " NIDX ", " CS "\n"                                                                  \
    BH_FORCE_MATH(MS, MASK)                                                                         \
    "s_branch " NEXT "_%=\n"
// (the lane select of v_writelane and the shift count of s_lshl_b64 use m0[5:0]: entry k sits in lane k & 63)
#define BH_PUSHCHK(TAG) "s_bitcmp1_b32 m0, 6\n s_cbranch_scc1 LpushHi" TAG "_%=\n"
#define BH_NOCHK(TAG) ""
#define BH_PUSH_HI(CS, SLO, SHI, TAG)                                                               \
    "LpushHi" TAG "_%=:\n"                                                                          \
    "v_writelane_b32 v33, " CS ", m0\n"                                                             \
    "v_writelane_b32 v34, " SLO ", m0\n"                                                            \
    "v_writelane_b32 v35, " SHI ", m0\n"                                                            \
    "s_branch LpushBack" TAG "_%=\n"
// pop: EXEC = the entry's lane, v_readfirstlane (4.1 cycles each; v_readlane: 8.3)
#define BH_POP_FAST(IDX, LO, HI)                                                                    \
    "s_lshl_b64 exec, 1, m0\n"                                                                      \
    "v_readfirstlane_b32 " IDX ", v30\n v_readfirstlane_b32 " LO ", v31\n v_readfirstlane_b32 " HI ", v32\n"
#define BH_POP(IDX, LO, HI, TAG)                                                                    \
    "s_lshl_b64 exec, 1, m0\n"                                                                      \
    "s_bitcmp1_b32 m0, 6\n"                                                                         \
    "s_cbranch_scc1 LpopHi" TAG "_%=\n"                                                             \
    "v_readfirstlane_b32 " IDX ", v30\n v_readfirstlane_b32 " LO ", v31\n v_readfirstlane_b32 " HI ", v32\n" \
    "LpopBack" TAG "_%=:\n"
#define BH_POP_HI(IDX, LO, HI, TAG)                                                                 \
    "LpopHi" TAG "_%=:\n"                                                                           \
    "v_readfirstlane_b32 " IDX ", v33\n v_readfirstlane_b32 " LO ", v34\n v_readfirstlane_b32 " HI ", v35\n" \
    "s_branch LpopBack" TAG "_%=\n"
#define BH_ITERATION(SFX, POPA, POPB, CHK)                                                              \
    /* ---- A: the handed-over child, else the stack's top, else the handed-over B */               \
    "s_cmp_gt_i32 s72, -1\n"                                                                        \
    "s_cbranch_scc1 LAn" SFX "_%=\n"                                                                \
    "s_sub_u32 m0, m0, 1\n"                              /* SCC = borrow: the stack was empty */    \
    "s_cbranch_scc1 LAe" SFX "_%=\n"                                                                \
    POPA                                                                                            \
    "s_cmp_lt_i32 s68, 0\n"                                                                         \
    "s_cbranch_scc1 Lspecial_%=\n"                                                                  \
    "s_branch LAr" SFX "_%=\n"                                                                      \
    "LAe" SFX "_%=:\n"                                                                              \
    "s_mov_b32 m0, 0\n"                                                                             \
    "s_cmp_lt_i32 s70, 0\n"                                                                         \
    "s_cbranch_scc1 Ldone_%=\n"                                                                     \
    "s_mov_b32 s68, s70\n"                                                                          \
    "s_mov_b64 s[44:45], s[46:47]\n"                                                                \
    "s_mov_b32 s70, -1\n"                                                                           \
    "s_branch LAr" SFX "_%=\n"                                                                      \
    "LAn" SFX "_%=:\n"                                                                              \
    "s_mov_b32 s68, s72\n"                               /* (its mask is in s[44:45] already) */    \
    "LAr" SFX "_%=:\n"                                   /* s68 = quad index >= 0, s[44:45] = mask */ \
    "s_mul_i32 s69, s68, 0x50\n"                                                                    \
    "s_load_dwordx16 s[24:39], %[quads], s69\n"                                                     \
    "s_load_dwordx4 s[40:43], %[quads], s69 offset:0x40\n"                                          \
    /* ---- B: the handed-over child, else the stack's top while the stack bound allows pairs */    \
    "s_cmp_gt_i32 s70, -1\n"                                                                        \
    "s_cbranch_scc1 LBn" SFX "_%=\n"                                                                \
    "s_mov_b64 s[46:47], 0\n"                            /* == 0: no B in flight */                 \
    "s_cmp_eq_u32 m0, 0\n"                                                                          \
    "s_cbranch_scc1 LW" SFX "_%=\n"                                                                 \
    "s_cmp_gt_u32 m0, %[plim]\n"                                                                    \
    "s_cbranch_scc1 LW" SFX "_%=\n"                                                                 \
    "s_sub_u32 m0, m0, 1\n"                                                                         \
    POPB                                                                                            \
    "s_cmp_lt_i32 s70, 0\n"                                                                         \
    "s_cbranch_scc1 Lunpop" SFX "_%=\n"                                                             \
    "LBn" SFX "_%=:\n"                                                                              \
    "s_mul_i32 s71, s70, 0x50\n"                                                                    \
    "s_load_dwordx16 s[48:63], %[quads], s71\n"                                                     \
    "s_load_dwordx4 s[64:67], %[quads], s71 offset:0x40\n"                                          \
    "LW" SFX "_%=:\n"                                                                               \
    "s_mov_b64 exec, s[44:45]\n"                                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                                    \
    "s_setprio 0\n"                                                                                 \
    /* ---- A's children, hand-off slot free */                                                     \
    BH_CHILD_F("s[24:25]", "s32", "s36", "s40", "s[44:45]", SFX "A0f", CHK(SFX "A0f"))              \
    BH_CHILD_F("s[26:27]", "s33", "s37", "s41", "s[44:45]", SFX "A1f", CHK(SFX "A1f"))              \
    BH_CHILD_F("s[28:29]", "s34", "s38", "s42", "s[44:45]", SFX "A2f", CHK(SFX "A2f"))              \
    BH_CHILD_F("s[30:31]", "s35", "s39", "s43", "s[44:45]", SFX "A3f", CHK(SFX "A3f"))              \
    "s_mov_b32 s72, -1\n"                                /* nothing handed over */                  \
    "s_branch LphB" SFX "_%=\n"                                                                     \
    /* ---- A's children after one was handed over: the rest is pushed (open masks in s[70:71]) */  \
    "LA1t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[26:27]", "s33", "s37", "s41", "s[44:45]", "s[70:71]", "s70", "s71", SFX "A1t", CHK(SFX "A1t")) \
    "LA2t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[28:29]", "s34", "s38", "s42", "s[44:45]", "s[70:71]", "s70", "s71", SFX "A2t", CHK(SFX "A2t")) \
    "LA3t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[30:31]", "s35", "s39", "s43", "s[44:45]", "s[70:71]", "s70", "s71", SFX "A3t", CHK(SFX "A3t")) \
    "LAet" SFX "_%=:\n"                                                                             \
    "s_mov_b64 s[44:45], s[68:69]\n"                     /* NA's mask to its place (A's own is dead) */ \
    "s_mov_b32 s70, -1\n"                                /* (s70 was scratch; NB is decided below) */ \
    "LphB" SFX "_%=:\n"                                                                             \
    "s_cmp_eq_u64 s[46:47], 0\n"                                                                    \
    "s_cbranch_scc1 Lloop_%=\n"                          /* no B: s70 == -1 here */                 \
    "s_mov_b64 exec, s[46:47]\n"                                                                    \
    "s_cmp_gt_u32 m0, %[plim]\n"                         /* too deep for another pair: push everything */ \
    "s_cbranch_scc1 LBover" SFX "_%=\n"                                                             \
    BH_CHILD_F("s[48:49]", "s56", "s60", "s64", "s[46:47]", SFX "B0f", CHK(SFX "B0f"))              \
    BH_CHILD_F("s[50:51]", "s57", "s61", "s65", "s[46:47]", SFX "B1f", CHK(SFX "B1f"))              \
    BH_CHILD_F("s[52:53]", "s58", "s62", "s66", "s[46:47]", SFX "B2f", CHK(SFX "B2f"))              \
    BH_CHILD_F("s[54:55]", "s59", "s63", "s67", "s[46:47]", SFX "B3f", CHK(SFX "B3f"))              \
    "s_mov_b32 s70, -1\n"                                                                           \
    "s_branch Lloop_%=\n"                                                                           \
    "LB0t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[48:49]", "s56", "s60", "s64", "s[46:47]", "s[24:25]", "s24", "s25", SFX "B0t", CHK(SFX "B0t")) \
    "LB1t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[50:51]", "s57", "s61", "s65", "s[46:47]", "s[24:25]", "s24", "s25", SFX "B1t", CHK(SFX "B1t")) \
    "LB2t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[52:53]", "s58", "s62", "s66", "s[46:47]", "s[24:25]", "s24", "s25", SFX "B2t", CHK(SFX "B2t")) \
    "LB3t" SFX "_%=:\n"                                                                             \
    BH_CHILD_T("s[54:55]", "s59", "s63", "s67", "s[46:47]", "s[24:25]", "s24", "s25", SFX "B3t", CHK(SFX "B3t")) \
    "LBet" SFX "_%=:\n"                                                                             \
    "s_mov_b64 s[46:47], s[68:69]\n"                     /* NB's mask to its place (garbage if s70 == -1) */ \
    "s_branch Lloop_%=\n"                                                                           \
    "Lunpop" SFX "_%=:\n"                                /* B is a bucket reference: leave it there */ \
    "s_add_u32 m0, m0, 1\n"                                                                         \
    "s_mov_b32 s70, -1\n"                                                                           \
    "s_mov_b64 s[46:47], 0\n"                                                                       \
    "s_branch LW" SFX "_%=\n"                                                                       \
    "LBover" SFX "_%=:\n"                                /* (s70 may still hold B's own index) */   \
    "s_mov_b32 s70, -1\n"                                                                           \
    "s_branch LB0t" SFX "_%=\n"                                                                     \
    BH_TAKE("s32", "s40", "s[44:45]", "s72", SFX "A0f", "LA1t" SFX)                                 \
    BH_TAKE("s33", "s41", "s[44:45]", "s72", SFX "A1f", "LA2t" SFX)                                 \
    BH_TAKE("s34", "s42", "s[44:45]", "s72", SFX "A2f", "LA3t" SFX)                                 \
    BH_TAKE("s35", "s43", "s[44:45]", "s72", SFX "A3f", "LAet" SFX)                                 \
    BH_TAKE("s56", "s64", "s[46:47]", "s70", SFX "B0f", "LB1t" SFX)                                 \
    BH_TAKE("s57", "s65", "s[46:47]", "s70", SFX "B1f", "LB2t" SFX)                                 \
    BH_TAKE("s58", "s66", "s[46:47]", "s70", SFX "B2f", "LB3t" SFX)                                 \
    BH_TAKE("s59", "s67", "s[46:47]", "s70", SFX "B3f", "LBet" SFX)
#define BH_HI_STUBS(SFX)                                                                            \
    BH_PUSH_HI("s40", "s68", "s69", SFX "A0f") BH_PUSH_HI("s41", "s68", "s69", SFX "A1f")          \
    BH_PUSH_HI("s42", "s68", "s69", SFX "A2f") BH_PUSH_HI("s43", "s68", "s69", SFX "A3f")          \
    BH_PUSH_HI("s41", "s70", "s71", SFX "A1t") BH_PUSH_HI("s42", "s70", "s71", SFX "A2t")          \
    BH_PUSH_HI("s43", "s70", "s71", SFX "A3t")                                                     \
    BH_PUSH_HI("s64", "s68", "s69", SFX "B0f") BH_PUSH_HI("s65", "s68", "s69", SFX "B1f")          \
    BH_PUSH_HI("s66", "s68", "s69", SFX "B2f") BH_PUSH_HI("s67", "s68", "s69", SFX "B3f")          \
    BH_PUSH_HI("s64", "s24", "s25", SFX "B0t") BH_PUSH_HI("s65", "s24", "s25", SFX "B1t")          \
    BH_PUSH_HI("s66", "s24", "s25", SFX "B2t") BH_PUSH_HI("s67", "s24", "s25", SFX "B3t")

// (the loop proper, shared by walk_tree_asm and walk_tree_asm_y: LIMITCHK is empty, or the iteration limit of the latter)
#define BH_TREE_LOOP(LIMITCHK) \
    /* ---------------------------------------------------------------- next entries */ \
    "Lloop_%=:\n" \
    "s_add_u32 %[cost], %[cost], 1\n" \
    BH_LOOP_GUARD LIMITCHK \
    "s_setprio 2\n" \
    "s_cmp_gt_u32 m0, 56\n" \
    "s_cbranch_scc1 LloopChk_%=\n" \
    BH_ITERATION("F", BH_POP_FAST("s68", "s44", "s45"), BH_POP_FAST("s70", "s46", "s47"), BH_NOCHK) \
    "LloopChk_%=:\n" /* more than 56 entries: pushes / pops pick their VGPRs */ \
    BH_ITERATION("C", BH_POP("s68", "s44", "s45", "A"), BH_POP("s70", "s46", "s47", "B"), BH_PUSHCHK) \
    BH_HI_STUBS("C") \
    BH_POP_HI("s68", "s44", "s45", "A") \
    BH_POP_HI("s70", "s46", "s47", "B") \
    /* ---- bucket reference -(node id) - 2: the cell's bodies one by one for the lanes that reached it */ \
    /* (self and exactly coincident bodies contribute nothing: d2 > 0 fails); -1 is dropped. */ \
    /* NB (s70, s[46:47]) is preserved; s72 is -1 here. */ \
    "Lspecial_%=:\n" \
    "s_cmp_eq_u32 s68, -1\n" \
    "s_cbranch_scc1 Lloop_%=\n" \
    "s_load_dwordx8 s[56:63], %[consts], 0x0\n" /* {aux, sorted positions, sorted masses}: this path only */ \
    "s_sub_i32 s68, -2, s68\n" \
    "s_lshl_b32 s69, s68, 3\n" \
    "s_mov_b64 exec, s[44:45]\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "s_load_dwordx2 s[48:49], s[56:57], s69\n" /* {first sorted body, count} */ \
    "s_waitcnt lgkmcnt(0)\n" \
    "s_cmp_lt_i32 s49, 1\n" \
    "s_cbranch_scc1 Lloop_%=\n" \
    "s_add_u32 s49, s48, s49\n" \
    "Lbody_%=:\n" \
    "s_lshl_b32 s69, s48, 3\n" \
    "s_load_dwordx2 s[50:51], s[58:59], s69\n" \
    "s_lshl_b32 s69, s48, 2\n" \
    "s_load_dword s52, s[60:61], s69\n" \
    "s_add_u32 s48, s48, 1\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_pk_add_f32 v[22:23], s[50:51], v[20:21] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "s_cmp_lt_u32 s48, s49\n" /* loop condition (and the packed result's wait state) */ \
    "v_mul_f32_e32 v24, v23, v23\n" \
    "v_fmac_f32_e32 v24, v22, v22\n" \
    "v_cmpx_lt_f32_e32 vcc, 0, v24\n" \
    "v_rsq_f32_e32 v25, v24\n" \
    "s_nop 0\n" /* wait state between v_rsq and its use */ \
    "v_mul_f32_e32 v26, s52, v25\n" \
    "v_mul_f32_e32 v26, v25, v26\n" \
    "v_mul_f32_e32 v24, v25, v26\n" \
    "v_fmac_f32_e32 v28, v24, v22\n" \
    "v_fmac_f32_e32 v29, v24, v23\n" \
    "s_mov_b64 exec, s[44:45]\n" \
    "s_cbranch_scc1 Lbody_%=\n" \
    "s_branch Lloop_%=\n"

__device__ __forceinline__ uint32_t walk_tree_asm(const QuadF BH_CONSTANT *quads, const void BH_CONSTANT *consts,
                                                   int32_t root, uint64_t everyone, int32_t pair_limit, float px,
                                                   float py, float &ax, float &ay)
{
    uint32_t cost;
    asm volatile(
        "v_mov_b32_e32 v20, %[px]\n"
        "v_mov_b32_e32 v21, %[py]\n"
        "v_mov_b32_e32 v28, %[ax]\n"
        "v_mov_b32_e32 v29, %[ay]\n"
        "s_mov_b32 m0, 0\n"                                     // (s68 = root quad, s[44:45] = lane mask: bound operands)
        "s_mov_b32 s70, -1\n"                                   // no NB
        "s_mov_b32 s72, -1\n"                                   // no NA
        "s_mov_b32 %[cost], 1\n"                                // loop iterations: the group's cost (re-balancing weight)
        "s_branch LArF_%=\n"                                    // the root quad alone
        BH_TREE_LOOP("")
        "Ldone_%=:\n"
        "s_setprio 0\n"
        "s_mov_b64 exec, -1\n"                                  // (the kernel runs the traversal with all lanes enabled)
        "v_mov_b32_e32 %[ax], v28\n"
        "v_mov_b32_e32 %[ay], v29\n"
        : [ax] "+v"(ax), [ay] "+v"(ay), [cost] "=&s"(cost), "+{s68}"(root), "+{s[44:45]}"(everyone)
        : [quads] "s"(quads), [consts] "s"(consts), [px] "v"(px), [py] "v"(py), [plim] "s"(pair_limit)
        : "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",
          "s40", "s41", "s42", "s43", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
          "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s69", "s70", "s71", "s72",
          "m0", "vcc", "scc", "memory",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
    return cost;
}
// The same child blocks driven by a LIST of entries instead of the stack: the level-synchronous walk of small
// launches (SPLIT > 1) hands every wave a chunk of at most 16 frontier entries, lane j of (in_base, in_lo,
// in_hi) holding entry j.  They are taken in order, two at a time
// (a bucket reference is served alone), and the children some lane opens are pushed to a fresh stack that
// starts at entry 0: at most 64 pushes, all in the first register triple.  Returns the number of pushes; the
// stack itself comes back in (out_base, out_lo, out_hi).  Same operations in the same order as the C++ loop
// it replaces (bit-identical: test_asm_walk_equals_the_portable_walk, split cases).
__device__ __forceinline__ int32_t walk_list_asm(const QuadF BH_CONSTANT *quads, const void BH_CONSTANT *consts,
                                                 int32_t in_base, int32_t in_lo, int32_t in_hi, int32_t mine,
                                                 float px, float py, float &ax, float &ay, int32_t &out_base,
                                                 int32_t &out_lo, int32_t &out_hi)
{
    int32_t sp;
    asm volatile(
        "v_mov_b32_e32 v20, %[px]\n"
        "v_mov_b32_e32 v21, %[py]\n"
        "v_mov_b32_e32 v28, %[ax]\n"
        "v_mov_b32_e32 v29, %[ay]\n"
        "v_mov_b32_e32 v33, %[ib]\n"
        "v_mov_b32_e32 v34, %[il]\n"
        "v_mov_b32_e32 v35, %[ih]\n"
        "s_mov_b32 m0, 0\n"                                     // pushes so far
        "s_mov_b32 s72, 0\n"                                    // j: next list entry
        "Lloop_%=:\n"
        "s_cmp_ge_u32 s72, %[mine]\n"
        "s_cbranch_scc1 Ldone_%=\n"
        "v_readlane_b32 s68, v33, s72\n v_readlane_b32 s44, v34, s72\n v_readlane_b32 s45, v35, s72\n"
        "s_add_u32 s72, s72, 1\n"
        "s_cmp_lt_i32 s68, 0\n"
        "s_cbranch_scc1 Lspecial_%=\n"
        "s_mov_b32 s71, 0\n"                                    // s71 != 0: a second quad (B) is in flight
        "s_cmp_ge_u32 s72, %[mine]\n"
        "s_cbranch_scc1 LloadA_%=\n"
        "v_readlane_b32 s70, v33, s72\n v_readlane_b32 s46, v34, s72\n v_readlane_b32 s47, v35, s72\n"
        "s_cmp_lt_i32 s70, 0\n"
        "s_cbranch_scc1 LloadA_%=\n"                            // a bucket reference: it is served in its own turn
        "s_add_u32 s72, s72, 1\n"
        "s_mul_i32 s71, s70, 0x50\n"
        "s_load_dwordx16 s[48:63], %[quads], s71\n"
        "s_load_dwordx4 s[64:67], %[quads], s71 offset:0x40\n"
        "LloadA_%=:\n"
        "s_mul_i32 s69, s68, 0x50\n"
        "s_load_dwordx16 s[24:39], %[quads], s69\n"
        "s_load_dwordx4 s[40:43], %[quads], s69 offset:0x40\n"
        "s_mov_b64 exec, s[44:45]\n"
        "s_waitcnt lgkmcnt(0)\n"
        BH_CHILD_T("s[24:25]", "s32", "s36", "s40", "s[44:45]", "s[68:69]", "s68", "s69", "LA0", "")
        BH_CHILD_T("s[26:27]", "s33", "s37", "s41", "s[44:45]", "s[68:69]", "s68", "s69", "LA1", "")
        BH_CHILD_T("s[28:29]", "s34", "s38", "s42", "s[44:45]", "s[68:69]", "s68", "s69", "LA2", "")
        BH_CHILD_T("s[30:31]", "s35", "s39", "s43", "s[44:45]", "s[68:69]", "s68", "s69", "LA3", "")
        "s_cmp_eq_u32 s71, 0\n"
        "s_cbranch_scc1 Lloop_%=\n"
        "s_mov_b64 exec, s[46:47]\n"
        BH_CHILD_T("s[48:49]", "s56", "s60", "s64", "s[46:47]", "s[68:69]", "s68", "s69", "LB0", "")
        BH_CHILD_T("s[50:51]", "s57", "s61", "s65", "s[46:47]", "s[68:69]", "s68", "s69", "LB1", "")
        BH_CHILD_T("s[52:53]", "s58", "s62", "s66", "s[46:47]", "s[68:69]", "s68", "s69", "LB2", "")
        BH_CHILD_T("s[54:55]", "s59", "s63", "s67", "s[46:47]", "s[68:69]", "s68", "s69", "LB3", "")
        "s_branch Lloop_%=\n"
        // ---- bucket reference -(node id) - 2 (see walk_tree_asm); -1 is dropped
        "Lspecial_%=:\n"
        "s_cmp_eq_u32 s68, -1\n"
        "s_cbranch_scc1 Lloop_%=\n"
        "s_load_dwordx8 s[56:63], %[consts], 0x0\n"
        "s_sub_i32 s68, -2, s68\n"
        "s_lshl_b32 s69, s68, 3\n"
        "s_mov_b64 exec, s[44:45]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_load_dwordx2 s[48:49], s[56:57], s69\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_cmp_lt_i32 s49, 1\n"
        "s_cbranch_scc1 Lloop_%=\n"
        "s_add_u32 s49, s48, s49\n"
        "Lbody_%=:\n"
        "s_lshl_b32 s69, s48, 3\n"
        "s_load_dwordx2 s[50:51], s[58:59], s69\n"
        "s_lshl_b32 s69, s48, 2\n"
        "s_load_dword s52, s[60:61], s69\n"
        "s_add_u32 s48, s48, 1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_pk_add_f32 v[22:23], s[50:51], v[20:21] neg_lo:[0,1] neg_hi:[0,1]\n"
        "s_cmp_lt_u32 s48, s49\n"
        "v_mul_f32_e32 v24, v23, v23\n"
        "v_fmac_f32_e32 v24, v22, v22\n"
        "v_cmpx_lt_f32_e32 vcc, 0, v24\n"
        "v_rsq_f32_e32 v25, v24\n"
        "s_nop 0\n"
        "v_mul_f32_e32 v26, s52, v25\n"
        "v_mul_f32_e32 v26, v25, v26\n"
        "v_mul_f32_e32 v24, v25, v26\n"
        "v_fmac_f32_e32 v28, v24, v22\n"
        "v_fmac_f32_e32 v29, v24, v23\n"
        "s_mov_b64 exec, s[44:45]\n"
        "s_cbranch_scc1 Lbody_%=\n"
        "s_branch Lloop_%=\n"
        "Ldone_%=:\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b32 %[sp], m0\n"
        "v_mov_b32_e32 %[ax], v28\n"
        "v_mov_b32_e32 %[ay], v29\n"
        "v_mov_b32_e32 %[ob], v30\n"
        "v_mov_b32_e32 %[ol], v31\n"
        "v_mov_b32_e32 %[oh], v32\n"
        : [ax] "+v"(ax), [ay] "+v"(ay), [sp] "=&s"(sp), [ob] "=&v"(out_base), [ol] "=&v"(out_lo), [oh] "=&v"(out_hi)
        : [quads] "s"(quads), [consts] "s"(consts), [px] "v"(px), [py] "v"(py), [ib] "v"(in_base), [il] "v"(in_lo),
          [ih] "v"(in_hi), [mine] "s"(mine)
        : "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
          "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72",
          "m0", "vcc", "scc", "memory",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
    return sp;
}
#undef BH_CHILD_T
#undef BH_CHILD_F
#undef BH_ITERATION

constexpr int kLdsStackDepth = 128;   // 3*31+4 entries worst case
constexpr int kSplitFrontier = 512;   // split walk: frontier entries per level kept in LDS (12 B each, x2)
constexpr int kSplitRound = 16;       // split walk: quads per wave per round (4 pushes each fill the 64-lane stack)

// Per child: one v_cmp decides accept/open/self (see eval); lane masks stay in SGPR pairs; the
// three-register stack write happens only for nodes that some lane opens (uniform branches).
//
// One wave per 64 bodies takes TWO stack entries per iteration where it can (see walk_tree_asm):
// fp32 mode does not need the reference's visiting order, so taking an entry off the stack before the
// quad above it has pushed its children is allowed.
template <bool LDS_STACK, bool STATS, int SPLIT, bool ASM = false>
__global__ __launch_bounds__(SPLIT > 1 ? kWave * SPLIT : kBlock) void walk_fast_kernel(WalkFastArgs a)
{
    static_assert(SPLIT == 1 || !LDS_STACK, "the split walk uses the register-lane stack");
    static_assert(!ASM || (!LDS_STACK && !STATS), "the assembly loops serve the default configuration");
    __shared__ int32_t s_base[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];
    __shared__ uint64_t s_mask[LDS_STACK ? kWavesPerBlock : 1][LDS_STACK ? kLdsStackDepth : 1];
    // split walk: two frontiers (current / next level), the waves' push counts, the partial sums
    constexpr int FCAP = SPLIT > 1 ? kSplitFrontier : 1;
    __shared__ int32_t fr_base[2][FCAP], fr_lo[2][FCAP], fr_hi[2][FCAP];
    __shared__ int32_t f_cnt[SPLIT > 1 ? SPLIT : 1];
    __shared__ float2 f_red[SPLIT > 1 ? SPLIT : 1][SPLIT > 1 ? kWave : 1];

    if (a.ctr->overflow) return;
#ifdef BHGPU_EXPERIMENTS
    const uint64_t dbg_t0 = a.timeline ? __builtin_amdgcn_s_memrealtime() : 0;     // 100 MHz wall clock
    const uint64_t dbg_c0 = a.timeline ? __builtin_amdgcn_s_memtime() : 0;         // shader clock cycles
#endif
    // Workgroup -> group of bodies: dispatch order.  (Measured and rejected, rounds 1-3: an XCD-contiguous placement --
    // XCD x takes the x-th contiguous eighth of the sorted order -- halves the L2 misses of the launch, 1.61 M -> 0.81 M,
    // and changes nothing: the waves that wait less for memory wait for an issue slot instead; reversed, strided and
    // heaviest-first orders: nothing either.  DESIGN.md section 4.)
    const uint32_t lb = blockIdx.x;
    if (lb >= a.nblocks) return;
    const int lane = lane_id(), w = wave_id();
    // SPLIT > 1: every wave of the workgroup holds the SAME 64 bodies
    const int64_t s = SPLIT > 1 ? a.lo + (int64_t)lb * kWave + lane : a.lo + (int64_t)lb * kBlock + threadIdx.x;
    bool valid = s < a.hi;
    const float2 p = valid ? a.spos[s] : float2{0.f, 0.f};
    asm volatile("" ::"v"(p.x), "v"(p.y));                // take the one-time vmcnt wait here, not per child
    float ax = 0.f, ay = 0.f;
    if (a.part == 2 && valid && (SPLIT == 1 || w == 0)) { const float2 t = a.acc_part[s]; ax = t.x; ay = t.y; }
    asm volatile("" : "+v"(ax), "+v"(ay));                // (same for this load: no s_waitcnt vmcnt in the loop)
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0, n_quad = 0;
    uint32_t my_int = 0;                                     // counting variant: this lane's accepted force evaluations
    uint32_t cost = 0;                                       // loop iterations of this group's walk (re-balancing weight)

    const QuadF BH_CONSTANT *quads = as_constant(a.quads);
    const NodeAux BH_CONSTANT *aux = as_constant(a.aux);
    const float2 BH_CONSTANT *cpos = as_constant(a.spos);
    const float BH_CONSTANT *cmass = as_constant(a.smass);

    int32_t v_base = 0, v_lo = 0, v_hi = 0;      // register-lane stack: entry k lives in lane k & 63 of the
    int32_t v_base2 = 0, v_lo2 = 0, v_hi2 = 0;   // first (k < 64) or second triple: 128 entries (walk_tree_asm)
    int sp = 0;                                   // wave-uniform

    // hand-off slot of the quad being evaluated (walk_tree_asm): the first opened child that is a quad lands here
    // instead of on the stack; h_free == false: everything is pushed
    bool h_free = false;
    int32_t h_idx = -1;
    uint64_t h_mask = 0;

    auto eval = [&](const float cx, const float cy, const int32_t mbits, const float thr, const int32_t child,
                    const uint64_t mask) {
        if (mbits == 0) return;                             // empty cell (project.cu:617): scalar int test
        const float m = __int_as_float(mbits);
        const float dx = cx - p.x, dy = cy - p.y;
        const float d2 = fmaf(dx, dx, dy * dy);
        // One compare decides everything (a v_cmp result IS its ballot, so the rest is SALU):
        //   subdivided cell: thr = (size/theta)^2  -> the reference's MAC, per body (project.cu:643)
        //   leaf:            thr = 0               -> accepted unless d2 == 0, i.e. unless it is the
        //                                             body itself (the self skip, project.cu:646) or an
        //                                             exactly coincident body, where the reference divides
        //                                             by zero (inf*0 -> NaN, project.cu:651-658); fp32
        //                                             positions are quantised, that case is reachable, and
        //                                             one NaN would poison the root box of every later step
        //   bucket:          thr = +inf            -> accepted by nobody, opened by everybody
        const uint64_t farm = __ballot(d2 > thr);
        const uint64_t accm = mask & farm;
        // (measured: a uniform `if (accm != 0)` around the force math -- skipping it for cells that
        // every lane opens -- costs more in branches than it saves: 0.482 vs 0.466 ms)
        const float ri = __builtin_amdgcn_rsqf(d2);
        const float wgt = __builtin_amdgcn_inverse_ballot_w64(accm) ? m * ri * ri * ri : 0.f;
        ax = fmaf(wgt, dx, ax);
        ay = fmaf(wgt, dy, ay);
        if (STATS) { n_vis += __popcll(mask); ++n_wave; n_int += __popcll(accm); my_int += (uint32_t)((accm >> lane) & 1ull); }
        if (child != -1) {                                  // subdivided cell or bucket reference
            const uint64_t open = mask & ~farm;
            if (open != 0 && h_free && child > 0) {         // handed over in registers
                h_idx = child; h_mask = open; h_free = false;
            } else if (open != 0) {                         // ~30 % of the evaluated nodes
                if (LDS_STACK) {
                    if (lane == 0) { s_base[w][sp] = child; s_mask[w][sp] = open; }
                } else if (sp < kWave) {
                    v_base = bh_writelane_i32(child, sp, v_base);
                    v_lo = bh_writelane_i32((int32_t)(uint32_t)open, sp, v_lo);
                    v_hi = bh_writelane_i32((int32_t)(uint32_t)(open >> 32), sp, v_hi);
                } else {
                    v_base2 = bh_writelane_i32(child, sp - kWave, v_base2);
                    v_lo2 = bh_writelane_i32((int32_t)(uint32_t)open, sp - kWave, v_lo2);
                    v_hi2 = bh_writelane_i32((int32_t)(uint32_t)(open >> 32), sp - kWave, v_hi2);
                }
                ++sp;
            }
        }
    };

    auto eval_quad = [&](const QuadRegs &q, const uint64_t mask) {
        if (STATS) ++n_quad;
        eval(__int_as_float(q.g[0]), __int_as_float(q.g[1]), q.g[8], __int_as_float(q.g[12]), q.c[0], mask);
        eval(__int_as_float(q.g[2]), __int_as_float(q.g[3]), q.g[9], __int_as_float(q.g[13]), q.c[1], mask);
        eval(__int_as_float(q.g[4]), __int_as_float(q.g[5]), q.g[10], __int_as_float(q.g[14]), q.c[2], mask);
        eval(__int_as_float(q.g[6]), __int_as_float(q.g[7]), q.g[11], __int_as_float(q.g[15]), q.c[3], mask);
    };

    // depth-cap cell holding several bodies (compat off): summed body by body for the lanes that
    // reached it; self and exactly coincident bodies contribute nothing (d2 == 0)
    auto bucket = [&](const int32_t node, const uint64_t mask) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        const v2i rng = *(const v2i BH_CONSTANT *)(aux + node);
        for (int32_t j = rng[0]; j < rng[0] + rng[1]; ++j) {
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
            if (STATS) { n_int += __popcll(okm); my_int += (uint32_t)((okm >> lane) & 1ull); }
        }
    };

    auto pop_raw = [&](int32_t &base, uint64_t &mask) {         // sp > 0
        --sp;
        if (LDS_STACK) {
            base = __builtin_amdgcn_readfirstlane(s_base[w][sp]);
            const uint64_t m = s_mask[w][sp];
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(m >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)m);
        } else if (sp < kWave) {
            base = __builtin_amdgcn_readlane(v_base, sp);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi, sp) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(v_lo, sp);
        } else {
            base = __builtin_amdgcn_readlane(v_base2, sp - kWave);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_hi2, sp - kWave) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(v_lo2, sp - kWave);
        }
    };
    // take the next quad entry off the stack; bucket references (-(node id) - 2) are served on the
    // way; returns false when the stack is empty
    auto pop_quad = [&](int32_t &base, uint64_t &mask) -> bool {
        while (sp > 0) {
            --sp;
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
            if (base >= 0) return true;
            if (base <= -2) bucket(-base - 2, mask);       // base == -1 (a leaf opened by a NaN) is dropped
        }
        return false;
    };

    if (SPLIT > 1) {
        const uint64_t everyone = __ballot(valid);
        // level 0: the root quad of the local tree and of every received LET (at most 57 entries)
        const int n_remote = (a.n_trees > 0 && a.part != 1) ? a.n_trees - 1 : 0;
        const int n_local = (a.part != 2) ? 1 : 0;
        int F = n_local + n_remote;
        if (w == 0 && lane < F) {
            int32_t base = 0;
            if (lane >= n_local) {
                int32_t t = lane - n_local;
                if (t >= a.self_rank) ++t;                      // the peers in rank order, self skipped
                base = (int32_t)(a.forest_base + (int64_t)t * a.let_cap);
            }
            fr_base[0][lane] = base;
            fr_lo[0][lane] = (int32_t)(uint32_t)everyone;
            fr_hi[0][lane] = (int32_t)(uint32_t)(everyone >> 32);
        }
        __syncthreads();
        auto lane_entry = [&](int32_t vb, int32_t vl, int32_t vh, int j, int32_t &base, uint64_t &mask) {
            base = __builtin_amdgcn_readlane(vb, j);
            mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(vh, j) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane(vl, j);
        };
        int cur = 0;
        while (F > 0) {                                         // one iteration per tree level
            cost += (uint32_t)F;
            int produced = 0;
            for (int r0 = 0; r0 < F; r0 += SPLIT * kSplitRound) {
                const int rem = (F - r0 < SPLIT * kSplitRound) ? F - r0 : SPLIT * kSplitRound;
                const int chunk = (rem + SPLIT - 1) / SPLIT;    // equal contiguous chunks, <= kSplitRound
                const int first = r0 + w * chunk;
                int mine = r0 + rem - first;
                mine = (mine < 0) ? 0 : (mine > chunk ? chunk : mine);
                int32_t in_base = 0, in_lo = 0, in_hi = 0;      // lane j holds this wave's j-th entry
                if (lane < mine) {
                    in_base = fr_base[cur][first + lane]; in_lo = fr_lo[cur][first + lane]; in_hi = fr_hi[cur][first + lane];
                }
                sp = 0;
                // All of this wave's quads are known before the first is evaluated, so the scalar loads
                // of entry j+1 are issued before entry j is evaluated (~1000 cycles for a lone wave against
                // a ~370-cycle load): the depth-first loop cannot do this, its next address is the
                // result of the evaluation.  (Bucket references and lanes past `mine` read quad 0.)
                auto quad_of = [&](int j) {
                    const int32_t b = __builtin_amdgcn_readlane(in_base, j & (kWave - 1));
                    return load_quad(quads + (b < 0 ? 0 : b));
                };
                if (ASM) {
                    sp = walk_list_asm(quads, as_constant(a.bucket_consts), in_base, in_lo, in_hi, __builtin_amdgcn_readfirstlane(mine), p.x, p.y, ax, ay,
                                       v_base, v_lo, v_hi);
                } else {
                    QuadRegs qn = quad_of(0);
                    for (int j = 0; j < mine; ++j) {
                        int32_t base; uint64_t mask;
                        lane_entry(in_base, in_lo, in_hi, j, base, mask);
                        const QuadRegs q = qn;
                        qn = quad_of(j + 1);
                        if (base <= -2) { bucket(-base - 2, mask); continue; }
                        if (base < 0) continue;
                        eval_quad(q, mask);                     // opened children -> private stack, sp <= 64
                    }
                }
                if (lane == 0) f_cnt[w] = sp;
                __syncthreads();
                int off = produced, total = 0;
#pragma unroll
                for (int k = 0; k < SPLIT; ++k) {
                    const int ck = __builtin_amdgcn_readfirstlane(f_cnt[k]);
                    off += (k < w) ? ck : 0;
                    total += ck;
                }
                if (produced + total <= FCAP) {                 // uniform over the workgroup
                    if (lane < sp) {
                        fr_base[cur ^ 1][off + lane] = v_base; fr_lo[cur ^ 1][off + lane] = v_lo; fr_hi[cur ^ 1][off + lane] = v_hi;
                    }
                    produced += total;
                } else {
                    // next frontier full (never seen in practice): every wave finishes the subtrees it
                    // has just opened depth-first, one at a time so the 64-entry stack bound holds
                    in_base = v_base; in_lo = v_lo; in_hi = v_hi;
                    const int todo = sp;
                    for (int j = 0; j < todo; ++j) {
                        int32_t base; uint64_t mask;
                        lane_entry(in_base, in_lo, in_hi, j, base, mask);
                        sp = 0;
                        if (base <= -2) { bucket(-base - 2, mask); continue; }
                        if (base < 0) continue;
                        do {
                            const QuadRegs q = load_quad(quads + base);
                            eval_quad(q, mask);
                        } while (pop_quad(base, mask));
                    }
                }
                __syncthreads();
            }
            cur ^= 1;
            F = produced;
        }
        // ---- partial sums back to wave 0, added in wave order
        f_red[w][lane] = float2{ax, ay};
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int k = 1; k < SPLIT; ++k) { ax += f_red[k][lane].x; ay += f_red[k][lane].y; }
        }
        valid = valid && (w == 0);
    } else {
        // the local tree, then (distributed step) the locally-essential tree of every peer: one
        // traversal per tree, so the stack never holds more than one tree's entries
        const uint64_t everyone = __ballot(valid);
        const int32_t t_first = (a.part == 2) ? 0 : -1, t_end = (a.part == 1) ? 0 : a.n_trees;
        for (int32_t t = t_first; t < t_end; ++t) {
            if (t >= 0 && t == a.self_rank) continue;
            int32_t base = (t < 0) ? 0 : (int32_t)(a.forest_base + (int64_t)t * a.let_cap);
            if (ASM) {
                cost += walk_tree_asm(quads, as_constant(a.bucket_consts), base, everyone, a.pair_limit, p.x, p.y, ax, ay);
                continue;
            }
            {
                // the C++ statement of walk_tree_asm's abstract machine: same order, same operations
                int32_t na = -1, nb = -1;                       // handed-over children (quad index, -1: none) ...
                uint64_t nam = 0, nbm = 0;                      // ... and the lanes that opened them
                bool first = true;
                for (;;) {
                    int32_t bA, bB = -1;
                    uint64_t mA, mB = 0;
                    ++cost;
                    if (first) { bA = base; mA = everyone; first = false; }
                    else {
                        if (na >= 0) { bA = na; mA = nam; }
                        else if (sp > 0) {
                            pop_raw(bA, mA);
                            if (bA < 0) {
                                if (bA <= -2) bucket(-bA - 2, mA);  // -1 (a leaf opened by a NaN) is dropped
                                continue;
                            }
                        } else if (nb >= 0) { bA = nb; mA = nbm; nb = -1; }
                        else break;
                        if (nb >= 0) { bB = nb; mB = nbm; }
                        else if (sp > 0 && sp <= a.pair_limit) {
                            pop_raw(bB, mB);
                            if (bB < 0) { ++sp; bB = -1; }          // a bucket reference: leave it on the stack
                        }
                    }
                    const QuadRegs A = load_quad(quads + bA);
                    QuadRegs B = A;
                    if (bB >= 0) B = load_quad(quads + bB);         // (both in flight before A is evaluated)
                    h_free = true; h_idx = -1;
                    eval_quad(A, mA);
                    na = h_idx; nam = h_mask;
                    nb = -1;
                    if (bB >= 0) {
                        h_free = sp <= a.pair_limit; h_idx = -1;    // too deep for another pair: push everything
                        eval_quad(B, mB);
                        nb = h_idx; nbm = h_mask;
                    }
                    h_free = false;
                }
            }
        }
    }

    // Epilogue.  Its arguments are read AGAIN from the kernarg segment through a laundered pointer: the
    // compiler otherwise loads all ~50 argument dwords up front and keeps the ones used here alive across
    // the traversal loop, which pushed the kernel to 106 SGPRs = 6 resident waves per SIMD instead of 8
    // (measured: 6,144 resident waves; the walk is latency-bound, waves are what hides the latency).
    const WalkFastArgs BH_CONSTANT *ka;
    {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        ka = (const WalkFastArgs BH_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
    }
    asm volatile("" : "+s"(ka));
    const WalkFastArgs BH_CONSTANT &e = *ka;
    float2 np = p;
    double2 np64{0.0, 0.0};
    if (e.part == 1) {
        if (valid) e.acc_part[s] = float2{ax, ay};              // raw sums; part 2 carries on from here
    } else if (valid) {
        const float gx = e.G * ax, gy = e.G * ay;
        const uint32_t body = e.perm[s];
        if (e.acc_out) e.acc_out[body] = float2{gx, gy};
        if (e.integrate && e.state64) {
            // mixed precision: the fp32 acceleration advances the fp64 state (updateAccVelPos,
            // project.cu:819-836, in the state's precision)
            double2 *pos64 = reinterpret_cast<double2 *>(e.pos), *vel64 = reinterpret_cast<double2 *>(e.vel);
            double2 v = vel64[body];
            const double2 q = pos64[body];
            v.x = fma((double)gx, (double)e.dt, v.x);
            v.y = fma((double)gy, (double)e.dt, v.y);
            np64 = double2{fma(v.x, (double)e.dt, q.x), fma(v.y, (double)e.dt, q.y)};
            vel64[body] = v;
            pos64[body] = np64;
        } else if (e.integrate) {
            float2 v = e.vel[body];
            v.x = fmaf(gx, e.dt, v.x);
            v.y = fmaf(gy, e.dt, v.y);
            np = float2{fmaf(v.x, e.dt, p.x), fmaf(v.y, e.dt, p.y)};
            if (e.to_sorted) {
                e.sstate[s] = float4{np.x, np.y, v.x, v.y};
            } else {
                e.vel[body] = v;
                e.pos[body] = np;
            }
        }
    }
    // min/max of the new positions per workgroup: the next step's root box needs no body pass
    const double bx = e.state64 ? np64.x : (double)np.x, by = e.state64 ? np64.y : (double)np.y;
    if (SPLIT > 1) {
        if (e.partial && w == 0) {                          // one partial per 64-body group
            const double xlo = wave_min(valid ? bx : (double)INFINITY), xhi = wave_max(valid ? bx : -(double)INFINITY);
            const double ylo = wave_min(valid ? by : (double)INFINITY), yhi = wave_max(valid ? by : -(double)INFINITY);
            if (lane == 0) {
                double *o = e.partial + 4 * (size_t)lb;
                o[0] = xlo; o[1] = xhi; o[2] = ylo; o[3] = yhi;
                if (e.slots) bounds_to_slot(xlo, xhi, ylo, yhi, e.slots, (uint32_t)lb);
            }
        }
    } else if (e.partial) block_bounds_to_partial(valid, bx, by, e.partial + 4 * (size_t)lb, e.slots);
#ifdef BHGPU_EXPERIMENTS
    if (e.timeline && lane == 0) {                              // per wave: start, end (10 ns ticks), hardware id, cost, clock stamps
        const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;
        const uint64_t c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
        e.timeline[6 * wv + 0] = dbg_t0;
        e.timeline[6 * wv + 1] = t1;
        e.timeline[6 * wv + 2] = (uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        e.timeline[6 * wv + 3] = cost;
        e.timeline[6 * wv + 4] = dbg_c0;
        e.timeline[6 * wv + 5] = c1;
    }
#endif
    if (e.group_cost && lane == 0 && (SPLIT == 1 || w == 0)) {
        const int64_t g = (SPLIT > 1 ? e.lo + (int64_t)lb * kWave : e.lo + (int64_t)lb * kBlock + (int64_t)w * kWave) >> 6;
        if (e.part == 2) e.group_cost[g] += cost;               // the second launch of a split forest walk adds its share
        else e.group_cost[g] = cost;
    }
    if (STATS && e.body_counts && s < e.hi && my_int) atomicAdd(&e.body_counts[e.perm[s]], my_int);   // (every wave of a split group adds its share)
    if (STATS && lane == 0) {
        atomicAdd(&e.ctr->visits, n_vis);
        atomicAdd(&e.ctr->interactions, n_int);
        atomicAdd(&e.ctr->wave_nodes, n_wave);
        atomicAdd(&e.ctr->wave_quads, n_quad);
    }
}

template <bool L, bool S, int SPLIT = 1, bool ASM = false>
static hipError_t launch(WalkFastArgs a, hipStream_t st)
{
    const int64_t cnt = a.hi - a.lo;
    if (cnt <= 0) return hipSuccess;
    constexpr int per_group = SPLIT > 1 ? kWave : kBlock;
    a.nblocks = (uint32_t)((cnt + per_group - 1) / per_group);
    hipLaunchKernelGGL((walk_fast_kernel<L, S, SPLIT, ASM>), dim3(a.nblocks), dim3(SPLIT > 1 ? kWave * SPLIT : kBlock), 0, st, a);
    return hipGetLastError();
}

template <bool S, bool ASM = false>
static hipError_t launch_split(const WalkFastArgs &a, int split, hipStream_t st)
{
    // (16 waves per group was instantiated through round 2: never the measured best at any size, and its
    // code object spilled 15 SGPRs into VGPR lanes around the assembly blocks; requests above 8 get 8)
    switch (split) {
    case 2: return launch<false, S, 2, ASM>(a, st);
    case 4: return launch<false, S, 4, ASM>(a, st);
    default: return launch<false, S, 8, ASM>(a, st);
    }
}

hipError_t launch_walk_fast(const WalkFastArgs &a, bool lds_stack, bool stats, int split, bool use_asm, hipStream_t st)
{
    // the hand-scheduled loop serves the default configuration: one wave per 64 bodies, register-lane
    // stack, no counters; every other variant runs the C++ loops (same operations, same order)
    if (use_asm && !lds_stack && !stats && !walk_fast_split_effective(a, lds_stack, split))
        return launch<false, false, 1, true>(a, st);
    // the split walk exists for the register-lane stack only; its queue holds 56 roots
    if (walk_fast_split_effective(a, lds_stack, split))
        return stats ? launch_split<true>(a, split, st)
                     : (use_asm ? launch_split<false, true>(a, split, st) : launch_split<false>(a, split, st));
    if (lds_stack) return stats ? launch<true, true>(a, st) : launch<true, false>(a, st);
    return stats ? launch<false, true>(a, st) : launch<false, false>(a, st);
}

}  // namespace bh
