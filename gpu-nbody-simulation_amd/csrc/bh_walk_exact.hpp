// bh_walk_exact.hpp -- fp64 theta-walk that reproduces the reference's forces BIT FOR BIT, fused
// with the integrator.  Replaces computeForces / computeForcesGpu (project.cu:593-675, 679-793)
// and updateAccVelPos (project.cu:819-836).  Compiled with -ffp-contract=off.
//
// One wavefront walks the tree for 64 Morton-adjacent bodies (one body per lane).  The traversal
// state is wave-uniform: the level being walked = {first child of the quad, 64-bit mask of the
// lanes that opened the parent, next child to visit} in scalar registers, the levels above it in
// one lane each of four vector registers.  Children are visited
// 3,2,1,0 -- the reference's pop order (it pushes 0..3 on a LIFO, project.cu:662-668) -- and a
// child's whole subtree is finished before its sibling starts.  The wave therefore visits the
// UNION of its lanes' private walks in the reference's DFS order, and each lane, masked to the
// nodes its own walk would pop, adds its terms in exactly the reference's order: same fp64
// operations, same order, same bits.  Node loads are wave-uniform (scalar loads), so a visit
// costs 40 bytes per wave instead of 64 divergent gathers.
// Four statements of the walk, bit-identical on every input (tests/test_gpu_exact.py): the hand-written gfx950 loop
// (walk_exact_asm: the product from ~12k bodies up), the C++ loop with the same arithmetic (the counting variant), the walk
// written as the reference writes it -- sqrt, size / d < theta, three divisions (BH_FLAG_WALK_PORTABLE) --, and for small
// launches one wavefront per BODY, breadth-first, with the terms added in DFS-key order (walk_exact_bfs_kernel, at the end).
#pragma once

#include "bh_tree.hpp"

namespace bh {

constexpr int kExactLevels = 34;   // max_depth <= 32 -> at most 32 stacked levels
extern "C" __device__ int exact_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// ---- the traversal as one block of gfx950 assembly ---------------------------------------------------------------------
// What the counters said of the C++ loop around a per-node block (profiles/r04_exact/): per visited node 47 vector
// instructions, but also 55 scalar ones and 14 branches -- the scalar unit as busy as the vector pipe.  Here, per node:
//   s_cmp + branch       empty node, from the high word of its mass (exact test out of line for the one word in between)
//   v_add x2, v_mul x2   dx, dy, dx^2, dy^2          s_sub + s_cmp + branch: mass inside the range of the short sequences
//   v_add                d2                           s_cmp + branch: leaf (occupant test by v_cmpx_ne_u32, out of line)
//   v_cmpx_le_f64        thr <= d2: EXEC := the lanes that accept; s_andn2: the lanes that open
//   s_cbranch_execz      nobody takes the node
//   v_cmp x2, s_and x2   operand ranges of the takers (dx^2, dy^2 >= 2^-400, G m_i in 2^+-150; d2 <= 2^400 follows from the root
//                        box, checked once per launch: lanes_safe is empty when an extent exceeds 2^199); s_andn2 + branch
//   36 x fp64            sqrt(d2) (v_rsq_f64, one coupled Newton step, two residual corrections), 1 / d2 and 1 / d
//                        (v_rcp_f64 + two Newton steps each), three quotients q + r (a - b q), two products, two sums --
//                        the instruction sequences the compiler expands sqrt() and `/` to, the three chains interleaved
//   s_cmp_eq_u64 + branch  somebody opens: the level's state goes to lane m0 of four VGPRs (unless it is exhausted)
// Children 3,2 and 1,0 of a quad are the halves of one 128-byte line: a pair is ONE s_load_dwordx16 (+ one x4 of links), and
// the second child of a pair waits for nothing.  Out of line, bit for bit the compiler's full expansions (v_div_scale,
// v_div_fmas, v_div_fixup; the 2^256 scaling and the class test of sqrt): any taker outside the ranges sends the wave
// there for that node.  Fixed SGPRs: s[24:31] / s[32:39] the lower / upper node of the pair {cx, cy, m, thr}, s[40:43] their
// links {child, occ}, s[44:49] scratch, s[64:65] the level's lanes, s66 its quad, s67 the next child, s[70:71] the lanes
// that open; m0 = stack entries.  Fixed VGPRs: v[32:37] dx dy d2, v[38:51] scratch, v52..v55 the stack.
// Hazards handled by hand (gfx940 family): an independent instruction or s_nop between a transcendental and its use; four
// wait states between a vector write of EXEC and v_writelane; two between a vector write of VCC and v_cndmask; four
// between v_div_scale's VCC and v_div_fmas; m0 written at least one instruction before a lane select uses it.
#if defined(BH_ASM_GUARD) && BH_ASM_GUARD
#define BHX_GUARD_INIT "s_mov_b32 s73, 0\n"
#define BHX_GUARD "s_add_u32 s73, s73, 1\n s_cmp_gt_u32 s73, 0x400000\n s_cbranch_scc1 Ldone_%=\n"
#define BHX_GUARD_CLOBBER "s73",
#else
#define BHX_GUARD_INIT ""
#define BHX_GUARD ""
#define BHX_GUARD_CLOBBER
#endif
#define BHX_DX "v[32:33]"
#define BHX_DY "v[34:35]"
#define BHX_D2 "v[36:37]"
#define BHX_TA "v[38:39]"
#define BHX_TB "v[40:41]"
#define BHX_TC "v[42:43]"
#define BHX_TD "v[44:45]"
#define BHX_TE "v[46:47]"
#define BHX_TF "v[48:49]"
#define BHX_TG "v[50:51]"
// the accepted term on checked operand ranges (M: the node's mass)
#define BHX_FAST(M)                                                                                 \
    "v_rsq_f64 " BHX_TA ", " BHX_D2 "\n"                                                            \
    "v_rcp_f64 " BHX_TB ", " BHX_D2 "\n"                                                            \
    "v_mul_f64 " BHX_TC ", %[gm], " M "\n"                          /* G m_i m                   */ \
    "v_mul_f64 " BHX_TD ", " BHX_D2 ", " BHX_TA "\n"                /* g = d2 y                  */ \
    "v_mul_f64 " BHX_TE ", " BHX_TA ", 0.5\n"                       /* h = y / 2                 */ \
    "v_fma_f64 " BHX_TA ", -" BHX_D2 ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TF ", -" BHX_TE ", " BHX_TD ", 0.5\n"          /* r = 1/2 - h g             */ \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TA ", " BHX_TB "\n"                                    \
    "v_fma_f64 " BHX_TD ", " BHX_TD ", " BHX_TF ", " BHX_TD "\n"                                    \
    "v_fma_f64 " BHX_TE ", " BHX_TE ", " BHX_TF ", " BHX_TE "\n"                                    \
    "v_fma_f64 " BHX_TA ", -" BHX_D2 ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TF ", -" BHX_TD ", " BHX_TD ", " BHX_D2 "\n"                                   \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TA ", " BHX_TB "\n"    /* 1 / d2                    */ \
    "v_fma_f64 " BHX_TD ", " BHX_TF ", " BHX_TE ", " BHX_TD "\n"                                    \
    "v_mul_f64 " BHX_TA ", " BHX_TC ", " BHX_TB "\n"                                                \
    "v_fma_f64 " BHX_TF ", -" BHX_TD ", " BHX_TD ", " BHX_D2 "\n"                                   \
    "v_fma_f64 " BHX_TG ", -" BHX_D2 ", " BHX_TA ", " BHX_TC "\n"                                   \
    "v_fma_f64 " BHX_TD ", " BHX_TF ", " BHX_TE ", " BHX_TD "\n"    /* sqrt(d2)                  */ \
    "v_fma_f64 " BHX_TA ", " BHX_TG ", " BHX_TB ", " BHX_TA "\n"    /* (G m_i m) / d2, :651      */ \
    "v_add_f64 " BHX_TD ", " BHX_TD ", %[eps]\n"                    /* d = sqrt(d2) + 1e-15, :634 */ \
    "v_rcp_f64 " BHX_TB ", " BHX_TD "\n"                                                            \
    "s_nop 0\n"                                                                                     \
    "v_fma_f64 " BHX_TC ", -" BHX_TD ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TC ", " BHX_TB "\n"                                    \
    "v_fma_f64 " BHX_TC ", -" BHX_TD ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TC ", " BHX_TB "\n"    /* 1 / d                     */ \
    "v_mul_f64 " BHX_TC ", " BHX_DX ", " BHX_TB "\n"                                                \
    "v_mul_f64 " BHX_TE ", " BHX_DY ", " BHX_TB "\n"                                                \
    "v_fma_f64 " BHX_TF ", -" BHX_TD ", " BHX_TC ", " BHX_DX "\n"                                   \
    "v_fma_f64 " BHX_TG ", -" BHX_TD ", " BHX_TE ", " BHX_DY "\n"                                   \
    "v_fma_f64 " BHX_TC ", " BHX_TF ", " BHX_TB ", " BHX_TC "\n"    /* dx / d, :654              */ \
    "v_fma_f64 " BHX_TE ", " BHX_TG ", " BHX_TB ", " BHX_TE "\n"    /* dy / d, :655              */ \
    "v_mul_f64 " BHX_TC ", " BHX_TA ", " BHX_TC "\n"                                                \
    "v_mul_f64 " BHX_TE ", " BHX_TA ", " BHX_TE "\n"                                                \
    "v_add_f64 %[fx], %[fx], " BHX_TC "\n"                                                          \
    "v_add_f64 %[fy], %[fy], " BHX_TE "\n"
// NUM / DEN -> RES, the compiler's expansion of an IEEE fp64 division (RES may be NUM); scratch: ta tb te tf tg
#define BHX_DIV(RES, NUM, DEN)                                                                      \
    "v_div_scale_f64 " BHX_TA ", s[48:49], " DEN ", " DEN ", " NUM "\n"                             \
    "v_rcp_f64 " BHX_TB ", " BHX_TA "\n"                                                            \
    "v_div_scale_f64 " BHX_TE ", vcc, " NUM ", " DEN ", " NUM "\n"                                  \
    "v_fma_f64 " BHX_TF ", -" BHX_TA ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TF ", " BHX_TB "\n"                                    \
    "v_fma_f64 " BHX_TF ", -" BHX_TA ", " BHX_TB ", 1.0\n"                                          \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TF ", " BHX_TB "\n"                                    \
    "v_mul_f64 " BHX_TF ", " BHX_TE ", " BHX_TB "\n"                                                \
    "v_fma_f64 " BHX_TG ", -" BHX_TA ", " BHX_TF ", " BHX_TE "\n"                                   \
    "s_nop 1\n"                                                                                     \
    "v_div_fmas_f64 " BHX_TG ", " BHX_TG ", " BHX_TB ", " BHX_TF "\n"                               \
    "v_div_fixup_f64 " RES ", " BHX_TG ", " DEN ", " NUM "\n"
// the accepted term through the full expansions (EXEC = the lanes that take the node; dx dy d2 as left by BHX_EVAL)
#define BHX_GENERIC(M)                                                                              \
    "v_mul_f64 " BHX_TC ", %[gm], " M "\n"                                                          \
    "s_mov_b32 s48, 0\n s_brev_b32 s49, 8\n"                        /* 2^-767 */                    \
    "v_mov_b32 v50, 0x100\n"                                                                        \
    "v_cmp_gt_f64 vcc, s[48:49], " BHX_D2 "\n"                      /* sqrt: tiny arguments scaled by 2^256 */ \
    "v_mov_b32 v51, 0xffffff80\n"                                                                   \
    "s_nop 1\n"                                                                                     \
    "v_cndmask_b32 v50, 0, v50, vcc\n"                                                              \
    "v_cndmask_b32 v51, 0, v51, vcc\n"                                                              \
    "v_ldexp_f64 " BHX_TA ", " BHX_D2 ", v50\n"                                                     \
    "v_rsq_f64 " BHX_TB ", " BHX_TA "\n"                                                            \
    "v_mov_b32 v50, 0x260\n"                                        /* class: -0, +0, +inf */       \
    "v_mul_f64 " BHX_TD ", " BHX_TA ", " BHX_TB "\n"                                                \
    "v_mul_f64 " BHX_TB ", " BHX_TB ", 0.5\n"                                                       \
    "v_fma_f64 " BHX_TE ", -" BHX_TB ", " BHX_TD ", 0.5\n"                                          \
    "v_fma_f64 " BHX_TD ", " BHX_TD ", " BHX_TE ", " BHX_TD "\n"                                    \
    "v_fma_f64 " BHX_TB ", " BHX_TB ", " BHX_TE ", " BHX_TB "\n"                                    \
    "v_fma_f64 " BHX_TE ", -" BHX_TD ", " BHX_TD ", " BHX_TA "\n"                                   \
    "v_fma_f64 " BHX_TD ", " BHX_TE ", " BHX_TB ", " BHX_TD "\n"                                    \
    "v_fma_f64 " BHX_TE ", -" BHX_TD ", " BHX_TD ", " BHX_TA "\n"                                   \
    "v_fma_f64 " BHX_TD ", " BHX_TE ", " BHX_TB ", " BHX_TD "\n"                                    \
    "v_ldexp_f64 " BHX_TD ", " BHX_TD ", v51\n"                                                     \
    "v_cmp_class_f64 vcc, " BHX_TA ", v50\n"                                                        \
    "s_nop 1\n"                                                                                     \
    "v_cndmask_b32 v44, v44, v38, vcc\n"                                                            \
    "v_cndmask_b32 v45, v45, v39, vcc\n"                                                            \
    "v_add_f64 " BHX_TD ", " BHX_TD ", %[eps]\n"                    /* d, project.cu:634 */         \
    BHX_DIV(BHX_TC, BHX_TC, BHX_D2)                                 /* project.cu:651 */            \
    BHX_DIV(BHX_DX, BHX_DX, BHX_TD)                                 /* project.cu:654 */            \
    BHX_DIV(BHX_DY, BHX_DY, BHX_TD)                                 /* project.cu:655 */            \
    "v_mul_f64 " BHX_TA ", " BHX_TC ", " BHX_DX "\n"                                                \
    "v_mul_f64 " BHX_TB ", " BHX_TC ", " BHX_DY "\n"                                                \
    "v_add_f64 %[fx], %[fx], " BHX_TA "\n"                                                          \
    "v_add_f64 %[fy], %[fy], " BHX_TB "\n"
// one node: EXEC = the level's lanes on entry; s[70:71] = the lanes that open it on exit (EXEC is left narrowed)
#define BHX_EVAL(CX, CY, M, MLO, MHI, THR, CS, OS, T, COMPAT_CMP)                                   \
    "s_mov_b64 s[70:71], 0\n"                                                                       \
    "s_cmp_lt_u32 " MHI ", 0x3cd203af\n"                            /* m < 1e-15 for sure: empty, project.cu:617 */ \
    "s_cbranch_scc1 Lend" T "_%=\n"                                                                 \
    "v_add_f64 " BHX_DX ", " CX ", -%[px]\n"                                                        \
    "v_add_f64 " BHX_DY ", " CY ", -%[py]\n"                                                        \
    "s_sub_u32 s46, " MHI ", 0x3cd203b0\n"                                                          \
    "v_mul_f64 " BHX_TA ", " BHX_DX ", " BHX_DX "\n"                                                \
    "v_mul_f64 " BHX_TB ", " BHX_DY ", " BHX_DY "\n"                                                \
    "s_cmp_lt_u32 s46, 0xc7dfc50\n"                                 /* 1e-15 < m < 2^150 for sure */ \
    "s_cbranch_scc0 Lodd" T "_%=\n"                                                                 \
    "v_add_f64 " BHX_D2 ", " BHX_TA ", " BHX_TB "\n"                                                \
    "s_cmp_lt_i32 " CS ", 0\n"                                                                      \
    "s_cbranch_scc1 Lleaf" T "_%=\n"                                                                \
    "v_cmpx_le_f64_e32 vcc, " THR ", " BHX_D2 "\n"                  /* project.cu:634, 643 */       \
    "s_andn2_b64 s[70:71], s[64:65], vcc\n"                                                         \
    "Ltake" T "_%=:\n"                                                                              \
    "s_cbranch_execz Lend" T "_%=\n"                                                                \
    "v_cmp_ge_f64 vcc, " BHX_TA ", %[tiny2]\n"                                                      \
    "v_cmp_ge_f64 s[44:45], " BHX_TB ", %[tiny2]\n"                                                 \
    "s_and_b64 vcc, vcc, s[44:45]\n"                                                                \
    "s_and_b64 vcc, vcc, %[lsafe]\n"                                                                \
    "s_andn2_b64 s[44:45], exec, vcc\n"                                                             \
    "s_cbranch_scc1 Lgen" T "_%=\n"                                                                 \
    BHX_FAST(M)                                                                                     \
    "Lend" T "_%=:\n"
#define BHX_STUBS(M, MLO, MHI, THR, CS, OS, T, COMPAT_CMP)                                          \
    "Lleaf" T "_%=:\n"                                              /* project.cu:623-626, 646 */   \
    "v_cmpx_ne_u32_e32 vcc, " OS ", %[body]\n"                                                      \
    COMPAT_CMP(OS)                                                                                  \
    "s_branch Ltake" T "_%=\n"                                                                      \
    "Lodd" T "_%=:\n"                                               /* the exact empty test, then the node as above */ \
    "v_mov_b32 v50, " MLO "\n v_mov_b32 v51, " MHI "\n"                                             \
    "v_cmp_le_f64 vcc, " BHX_TG ", %[eps]\n"                                                        \
    "s_nop 1\n"                                                                                     \
    "s_cbranch_vccnz Lend" T "_%=\n"                                                                \
    "v_add_f64 " BHX_D2 ", " BHX_TA ", " BHX_TB "\n"                                                \
    "s_cmp_lt_i32 " CS ", 0\n"                                                                      \
    "s_cbranch_scc1 LleafG" T "_%=\n"                                                               \
    "v_cmpx_le_f64_e32 vcc, " THR ", " BHX_D2 "\n"                                                  \
    "s_andn2_b64 s[70:71], s[64:65], vcc\n"                                                         \
    "s_branch LgenE" T "_%=\n"                                                                      \
    "LleafG" T "_%=:\n"                                                                             \
    "v_cmpx_ne_u32_e32 vcc, " OS ", %[body]\n"                                                      \
    COMPAT_CMP(OS)                                                                                  \
    "LgenE" T "_%=:\n"                                                                              \
    "s_cbranch_execz Lend" T "_%=\n"                                                                \
    "Lgen" T "_%=:\n"                                                                               \
    BHX_GENERIC(M)                                                                                  \
    "s_branch Lend" T "_%=\n"
#define BHX_COMPAT_ON(OS) "v_cmpx_ne_u32_e32 vcc, " OS ", %[alt]\n"
#define BHX_COMPAT_OFF(OS) ""
#define BHX_PUSH                                                                                    \
    "s_nop 3\n"                                                                                     \
    "v_writelane_b32 v52, s66, m0\n v_writelane_b32 v53, s67, m0\n"                                 \
    "v_writelane_b32 v54, s64, m0\n v_writelane_b32 v55, s65, m0\n"                                 \
    "s_add_u32 m0, m0, 1\n"
#define BHX_LOOP(COMPAT_CMP)                                                                        \
    "s_mov_b32 m0, 0\n"                                                                             \
    "s_mov_b32 s66, %[quad]\n s_mov_b32 s67, 3\n s_mov_b64 s[64:65], %[live]\n"                     \
    BHX_GUARD_INIT                                                                                  \
    "Lloop_%=:\n"                                                   /* the pair that holds child s67 of quad s66 */ \
    BHX_GUARD                                                                                       \
    "s_and_b32 s46, s67, -2\n"                                                                      \
    "s_add_u32 s46, s66, s46\n"                                                                     \
    "s_lshl_b32 s47, s46, 5\n"                                                                      \
    "s_load_dwordx16 s[24:39], %[gd], s47\n"                                                        \
    "s_lshl_b32 s47, s46, 3\n"                                                                      \
    "s_load_dwordx4 s[40:43], %[ld], s47\n"                                                         \
    "s_mov_b64 exec, s[64:65]\n"                                                                    \
    "s_bitcmp1_b32 s67, 0\n"                                                                        \
    "s_waitcnt lgkmcnt(0)\n"                                                                        \
    "s_cbranch_scc0 Llo_%=\n"                                                                       \
    "s_sub_u32 s67, s67, 1\n"                                                                       \
    BHX_EVAL("s[32:33]", "s[34:35]", "s[36:37]", "s36", "s37", "s[38:39]", "s42", "s43", "H", COMPAT_CMP) \
    "s_cmp_eq_u64 s[70:71], 0\n"                                                                    \
    "s_cbranch_scc1 LloE_%=\n"                                                                      \
    BHX_PUSH                                                        /* the level continues at its even child */ \
    "s_mov_b32 s66, s42\n s_mov_b32 s67, 3\n s_mov_b64 s[64:65], s[70:71]\n"                        \
    "s_branch Lloop_%=\n"                                                                           \
    "LloE_%=:\n"                                                                                    \
    "s_mov_b64 exec, s[64:65]\n"                                                                    \
    "Llo_%=:\n"                                                                                     \
    "s_sub_u32 s67, s67, 1\n"                                                                       \
    BHX_EVAL("s[24:25]", "s[26:27]", "s[28:29]", "s28", "s29", "s[30:31]", "s40", "s41", "L", COMPAT_CMP) \
    "s_cmp_eq_u64 s[70:71], 0\n"                                                                    \
    "s_cbranch_scc1 Lnext_%=\n"                                                                     \
    "s_cmp_lt_i32 s67, 0\n"                                                                         \
    "s_cbranch_scc1 Ltail_%=\n"                                     /* an exhausted level is not stacked */ \
    BHX_PUSH                                                                                        \
    "Ltail_%=:\n"                                                                                   \
    "s_mov_b32 s66, s40\n s_mov_b32 s67, 3\n s_mov_b64 s[64:65], s[70:71]\n"                        \
    "s_branch Lloop_%=\n"                                                                           \
    "Lnext_%=:\n"                                                                                   \
    "s_cmp_gt_i32 s67, -1\n"                                                                        \
    "s_cbranch_scc1 Lloop_%=\n"                                                                     \
    "s_sub_u32 m0, m0, 1\n"                                         /* SCC = borrow: the stack was empty */ \
    "s_cbranch_scc1 Ldone_%=\n"                                                                     \
    "s_lshl_b64 exec, 1, m0\n"                                                                      \
    "v_readfirstlane_b32 s66, v52\n v_readfirstlane_b32 s67, v53\n"                                 \
    "v_readfirstlane_b32 s64, v54\n v_readfirstlane_b32 s65, v55\n"                                 \
    "s_branch Lloop_%=\n"                                                                           \
    BHX_STUBS("s[36:37]", "s36", "s37", "s[38:39]", "s42", "s43", "H", COMPAT_CMP)                  \
    BHX_STUBS("s[28:29]", "s28", "s29", "s[30:31]", "s40", "s41", "L", COMPAT_CMP)                  \
    "Ldone_%=:\n"                                                                                   \
    "s_mov_b64 exec, -1\n"                                          /* (the kernel runs the traversal with all lanes enabled) */
#define BHX_OPERANDS                                                                                \
    : [fx] "+v"(fx), [fy] "+v"(fy)                                                                  \
    : [px] "v"(px), [py] "v"(py), [gm] "v"(gm), [body] "v"(body), [alt] "v"(alt), [gd] "s"(gd), [ld] "s"(ld),           \
      [tiny2] "s"(tiny2), [eps] "s"(eps), [lsafe] "s"(lanes_safe), [quad] "s"(quad), [live] "s"(live)                    \
    : BHX_GUARD_CLOBBER                                                                             \
      "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39",   \
      "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s64", "s65", "s66", "s67", "s70", "s71",   \
      "m0", "vcc", "scc", "memory",                                                                 \
      "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",   \
      "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

// Walks the subtree below quad `quad` (node id of its first sibling) for the lanes in `live`, visiting children 3,2,1,0 and
// every subtree before the next sibling: each lane's terms are added in the reference's order (project.cu:662-668).
// gd / ld: the node and link arrays; byte offsets are 32-bit (the caller checks node_cap * 32 < 4 GiB).
// lanes_safe: the lanes whose G m_i is inside 2^+-150 -- and NONE unless both extents of the root box are <= 2^199 (every body and
// every centre of mass lies inside it, so dx^2 + dy^2 <= 2^399: the one range test this loop leaves to its caller).
template <bool COMPAT>
__device__ __forceinline__ void walk_exact_asm(const char __attribute__((address_space(4))) *gd,
                                               const char __attribute__((address_space(4))) *ld, int32_t quad, uint64_t live,
                                               uint64_t lanes_safe, double px, double py, double gm, int32_t body, int32_t alt,
                                               double &fx, double &fy)
{
    const double tiny2 = 0x1p-400, eps = 1e-15;
    if constexpr (COMPAT) asm volatile(BHX_LOOP(BHX_COMPAT_ON) BHX_OPERANDS);
    else asm volatile(BHX_LOOP(BHX_COMPAT_OFF) BHX_OPERANDS);
}

// THR (the product): the node's `size` slot holds the EXACT d2 threshold of its acceptance test (exact_walk_threshold,
// bh_tree.hpp), so the test is one comparison instead of sqrt + add + divide -- the same decisions bit for bit -- and the
// square root and the three divisions of an accepted term run only on nodes some lane accepts, through the SAME
// instruction sequences the compiler expands sqrt() and `/` to, minus their range handling (v_div_scale / v_div_fmas /
// v_div_fixup, the 2^256 pre-scaling of sqrt), which is the identity on the operand ranges checked first; dx / d and
// dy / d share the refined reciprocal of d.  Any lane outside those ranges sends the wave through the plain expressions
// for that node.  THR = false (BH_FLAG_WALK_PORTABLE; the node kernel then stores the size) is the walk written as the
// reference writes it: tests/test_gpu_exact.py holds the two bit-identical on every input, extreme ranges included.
// ASM (the product, with THR): the traversal below the root is walk_exact_asm above; THR without ASM is the C++ statement
// of the same operations (the counting variant runs it), and the three are compared bit for bit.
template <bool COMPAT, bool STATS, bool THR = true, bool ASM = false>
__global__ __launch_bounds__(kBlock) void walk_exact_kernel(
    const NodeD *__restrict__ gd, const LinkD *__restrict__ ld, const uint32_t *__restrict__ perm,
    double2 *__restrict__ pos, double2 *__restrict__ vel, const double *__restrict__ mass,
    double2 *__restrict__ force_out, int64_t lo, int64_t hi, double theta, double G, double dt,
    int integrate, TreeCounters *ctr, double *__restrict__ partial, double *slots, int bpw, const double *__restrict__ box)
{
    // bpw: bodies per wavefront, a power of two <= 64 (lanes bpw .. 63 idle).  A wave's walk is ONE dependent chain over the
    // union of its lanes' walks; a launch of few bodies leaves the GPU empty however it is cut, so the engine gives every
    // wave fewer bodies -- 1 at N <= 4,096 -- and the chain shrinks to one body's walk (config 1, N = 1,024: 0.30 -> see
    // DESIGN.md section 4).  Every lane still adds its own terms in the reference's order: same bits.
    // The traversal stack -- one entry per tree level: {quad of the level, next child to visit, lanes that walk it} -- lives
    // in four VGPRs, entry k in lane k (round 3; it was three LDS arrays written by lane 0 and read back through
    // v_readfirstlane: a round trip through LDS on every visited node of a walk that is one long dependent chain).
    int32_t v_quad = 0, v_next = 0, v_mlo = 0, v_mhi = 0;

    if (ctr->overflow) return;
    const int lane = lane_id();
    const int64_t s = lo + ((int64_t)blockIdx.x * kWavesPerBlock + wave_id()) * bpw + lane;
    const bool valid = lane < bpw && s < hi;
    const int64_t body = valid ? (int64_t)perm[s] : -1;
    const double2 p = valid ? pos[body] : double2{0.0, 0.0};
    const double mi = valid ? mass[body] : 0.0;
    const double Gm = G * mi;                      // (G * masses[i]) * nodeMass, project.cu:651
    // operand ranges on which the range handling of IEEE division and sqrt is the identity (v_div_scale acts when the
    // exponents of numerator and denominator differ by >= 768 or either is within ~2^53 of the ends of the format)
    constexpr double kLaneLo = 0x1p-150, kLaneHi = 0x1p150, kTiny = 0x1p-200, kHuge = 0x1p400;
    const bool lane_safe = THR && fabs(Gm) >= kLaneLo && fabs(Gm) <= kLaneHi;
    // (the root box bounds every |dx|, |dy| of this launch: with extents <= 2^199 no d2 exceeds 2^399)
    const bool box_ok = ASM && (box[1] - box[0]) <= 0x1p199 && (box[3] - box[2]) <= 0x1p199;
    const uint64_t lanes_safe = __ballot(lane_safe && box_ok);
    const int32_t body_lo = (int32_t)body, body_alt = (int32_t)(-body - 2);     // (bodies < 2^31: bh_create)
    double fx = 0.0, fy = 0.0;
    unsigned long long n_vis = 0, n_int = 0, n_wave = 0, n_acc = 0;

    // evaluate one node -- its record `q` and links `k` in scalar registers -- for the lanes in `live`; returns the mask of
    // lanes that must open it
    auto eval = [&](const NodeD q, const LinkD k, uint64_t live, int32_t &child_out) -> uint64_t {
        child_out = k.child;
        if (q.m <= 1e-15) return 0;                // project.cu:617
        const bool mine = (live >> lane) & 1ull;
        const bool leaf = k.child < 0;             // all four children -1, project.cu:623-626
        const double dx = q.cx - p.x;
        const double dy = q.cy - p.y;
        const double d2 = dx * dx + dy * dy;
        bool accept;
        double d = 0.0;
        if (THR) {
            accept = leaf || (d2 >= q.size);       // q.size = the exact threshold: project.cu:634, 643 in one comparison
        } else {
            d = sqrt(d2) + 1e-15;                  // project.cu:634
            accept = leaf || (q.size / d < theta); // project.cu:643
        }
        bool self = false;
        if (leaf) {
            self = ((int64_t)k.occ == body);
            if (COMPAT) self = self || ((int64_t)k.occ + 2 == -body);   // project.cu:646
        }
        if (mine && accept && !self) {
            const double num = Gm * q.m;
            bool fast = false;
            if (THR) {
                const bool safe = lane_safe && q.m <= kLaneHi && fabs(dx) >= kTiny && fabs(dy) >= kTiny && d2 <= kHuge;
                fast = __ballot(!safe) == 0ull;    // (of the lanes that take this node)
            }
            if (fast) {
                // sqrt(d2): y ~ 1/sqrt, g ~ sqrt, h ~ 1/(2 sqrt); one coupled step, two residual corrections
                const double y = __builtin_amdgcn_rsq(d2);
                double g = d2 * y, h = y * 0.5;
                const double r = __builtin_fma(-h, g, 0.5);
                g = __builtin_fma(g, r, g);
                double e = __builtin_fma(-g, g, d2);
                h = __builtin_fma(h, r, h);
                g = __builtin_fma(e, h, g);
                e = __builtin_fma(-g, g, d2);
                g = __builtin_fma(e, h, g);
                const double dd = g + 1e-15;       // project.cu:634
                // a / b = fma(a - b q, r, q) with q = a r and r = 1/b after two Newton steps
                auto recip = [](double b) {
                    double r0 = __builtin_amdgcn_rcp(b);
                    double t = __builtin_fma(-b, r0, 1.0);
                    r0 = __builtin_fma(r0, t, r0);
                    t = __builtin_fma(-b, r0, 1.0);
                    return __builtin_fma(r0, t, r0);
                };
                auto quot = [](double a, double b, double rb) {
                    const double q0 = a * rb;
                    const double rem = __builtin_fma(-b, q0, a);
                    return __builtin_fma(rem, rb, q0);
                };
                const double r2 = recip(d2), rd = recip(dd);
                const double f = quot(num, d2, r2);                           // project.cu:651
                const double ux = quot(dx, dd, rd), uy = quot(dy, dd, rd);    // project.cu:654-655
                fx += f * ux;
                fy += f * uy;
            } else {
                if (THR) d = sqrt(d2) + 1e-15;
                const double f = num / d2;             // project.cu:651
                const double ux = dx / d, uy = dy / d; // project.cu:654-655
                fx += f * ux;
                fy += f * uy;
            }
        }
        if (STATS) {
            n_vis += __popcll(live);
            ++n_wave;
            const uint64_t takers = __ballot(mine && accept && !self);
            n_int += __popcll(takers);
            n_acc += takers != 0;
        }
        if (leaf) return 0;
        return __ballot(mine && !accept);
    };

    // The level being walked -- its quad, the next child to visit, the lanes that walk it -- is wave-uniform state in
    // scalar registers; the stack (one lane of four VGPRs per entry) holds the levels above it, written when a child is
    // opened and read when a level is exhausted.  A level whose last child is the one being opened is not stacked at all.
    auto push = [&](int at, int32_t quad, int32_t next, uint64_t mask) {
        v_quad = exact_writelane_i32(quad, at, v_quad);
        v_next = exact_writelane_i32(next, at, v_next);
        v_mlo = exact_writelane_i32((int32_t)(uint32_t)mask, at, v_mlo);
        v_mhi = exact_writelane_i32((int32_t)(uint32_t)(mask >> 32), at, v_mhi);
    };
    typedef int32_t v16i __attribute__((ext_vector_type(16)));
    typedef int32_t v8i __attribute__((ext_vector_type(8)));
    typedef int32_t v4i __attribute__((ext_vector_type(4)));
    typedef int32_t v2i __attribute__((ext_vector_type(2)));
    int sp = -1;
    int32_t cur_quad = 0, cur_c = -1;
    uint64_t cur_live = 0;
    {
        v8i qa; v2i ka;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\t"
                     "s_load_dwordx2 %1, %3, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(qa), "=&s"(ka) : "s"(gd), "s"(ld) : "memory");
        int32_t child;
        const uint64_t open = eval(NodeD{__hiloint2double(qa[1], qa[0]), __hiloint2double(qa[3], qa[2]),
                                         __hiloint2double(qa[5], qa[4]), __hiloint2double(qa[7], qa[6])},
                                   LinkD{ka[0], ka[1]}, __ballot(valid), child);
        if (open != 0 && child >= 0) { cur_quad = child; cur_c = 3; cur_live = open; }
    }
    // Children 3,2 and 1,0 of a quad are the two halves of one 128-byte line (bh_engine.hip allocates the node arrays so):
    // a pair comes in ONE scalar request, and its second child -- visited right after the first unless that one was
    // opened -- waits for nothing.  A level resumed after a descent reads its pair again.
    v16i qa = {};
    v4i ka = {};
    bool have_pair = false;
    bool more = cur_c >= 0;
    if constexpr (ASM) {
        if (more) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            walk_exact_asm<COMPAT>((const char __attribute__((address_space(4))) *)gd,
                                   (const char __attribute__((address_space(4))) *)ld, cur_quad, cur_live, lanes_safe, p.x, p.y,
                                   Gm, body_lo, COMPAT ? body_alt : body_lo, fx, fy);
#pragma clang diagnostic pop
        }
        more = false;
    }
    while (more) {
        if (!have_pair) {
            const int32_t base = cur_quad + (cur_c & ~1);
            const NodeD *pq = gd + base;
            const LinkD *pk = ld + base;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\t"
                         "s_load_dwordx4 %1, %3, 0x0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&s"(qa), "=&s"(ka) : "s"(pq), "s"(pk) : "memory");
        }
        const int32_t parity = cur_c;                            // odd: the upper half of the pair
        have_pair = (cur_c & 1) != 0;
        --cur_c;
        // (scalar selects, spelled out: the compiler turns a select between halves of the register tuple into indexed moves
        // through vector registers)
        NodeD q;
        LinkD k;
        asm("s_bitcmp1_b32 %[c], 0\n\t"
            "s_cselect_b64 %[cx], %[h0], %[l0]\n\t"
            "s_cselect_b64 %[cy], %[h1], %[l1]\n\t"
            "s_cselect_b64 %[m], %[h2], %[l2]\n\t"
            "s_cselect_b64 %[sz], %[h3], %[l3]\n\t"
            "s_cselect_b32 %[ch], %[hc], %[lc]\n\t"
            "s_cselect_b32 %[oc], %[ho], %[lo]"
            : [cx] "=&s"(q.cx), [cy] "=&s"(q.cy), [m] "=&s"(q.m), [sz] "=&s"(q.size), [ch] "=&s"(k.child), [oc] "=&s"(k.occ)
            : [c] "s"(parity), [h0] "s"(__hiloint2double(qa[9], qa[8])), [h1] "s"(__hiloint2double(qa[11], qa[10])),
              [h2] "s"(__hiloint2double(qa[13], qa[12])), [h3] "s"(__hiloint2double(qa[15], qa[14])),
              [l0] "s"(__hiloint2double(qa[1], qa[0])), [l1] "s"(__hiloint2double(qa[3], qa[2])),
              [l2] "s"(__hiloint2double(qa[5], qa[4])), [l3] "s"(__hiloint2double(qa[7], qa[6])),
              [hc] "s"(ka[2]), [ho] "s"(ka[3]), [lc] "s"(ka[0]), [lo] "s"(ka[1])
            : "scc");
        int32_t child;
        const uint64_t open = eval(q, k, cur_live, child);
        if (open != 0 && child >= 0 && (cur_c < 0 || sp + 1 < kExactLevels)) {      // (the bound cannot bind: max_depth <= 32)
            if (cur_c >= 0) {
                ++sp;
                push(sp, cur_quad, cur_c, cur_live);
            }
            cur_quad = child; cur_c = 3; cur_live = open;
            have_pair = false;
        } else if (cur_c < 0) {
            // level exhausted: back to the one above it
            if (sp < 0) more = false;
            else {
                cur_quad = __builtin_amdgcn_readlane(v_quad, sp);
                cur_c = __builtin_amdgcn_readlane(v_next, sp);
                cur_live = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v_mhi, sp) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane(v_mlo, sp);
                --sp;
                have_pair = false;
            }
        }
    }

    double2 np = p;
    if (valid) {
        force_out[body] = double2{fx, fy};
        if (integrate) {
            // updateAccVelPos, project.cu:827-834
            const double ax = fx / mi, ay = fy / mi;
            double2 v = vel[body];
            v.x += ax * dt;  v.y += ay * dt;
            vel[body] = v;
            np.x += v.x * dt;  np.y += v.y * dt;
            pos[body] = np;
        }
    }
    // min/max of the new positions per workgroup: the next step's root box needs no body pass
    if (partial) block_bounds_to_partial(valid, np.x, np.y, partial + 4 * (size_t)blockIdx.x, slots);
    if (STATS) {
        // one atomic per wave
        if (lane == 0) {
            atomicAdd(&ctr->visits, n_vis);
            atomicAdd(&ctr->interactions, n_int);
            atomicAdd(&ctr->wave_nodes, n_wave);
            atomicAdd(&ctr->wave_accepts, n_acc);
        }
    }
}

// ---- launches of a few thousand bodies: ONE WAVEFRONT PER BODY, the tree level by level ---------------------------------
// At the reference's own sizes (config 1: 1,024 bodies) the walk above is one body per wave and ~150 node visits one
// after the other, each a memory round trip and a 40-instruction dependent chain: 68 us for 0.1 ms of step.  But the terms
// a body adds are independent of each other -- only the ORDER of the additions is the reference's.  So here the 64 lanes of
// a wave walk ONE body's tree breadth-first: a queue of nodes in LDS, 64 nodes per round (vector loads), every lane decides
// its node by the exact threshold and computes its term; an opened node appends its four children.  Every queued node
// carries its DFS sort key -- two bits per level, child 3 first, as the reference pops them (project.cu:662-668) -- and the
// accepted nodes of a walk are an antichain of the tree, so sorting their terms by key IS the reference's order of additions:
// the terms are ranked by counting (<= 256 of them), permuted in LDS, and added one after the other from 0.0.  About ten
// rounds instead of ~150 visits.  A walk whose queue or term list would overflow (theta -> 0, adversarial trees) starts
// again through walk_exact_asm: same bits either way (tests/test_gpu_exact.py: every size from 1 to 4,096 against the
// reference's own vectors and against the cooperative walk).
constexpr int kBfsQueue = 256;                      // per wavefront; a ring.  Terms per walk: 256 or 384 (template)
constexpr int kBfsBodiesPerWave = 64;              // a wave takes up to this many bodies, one after the other

// one accepted term f * (dx, dy) / d (project.cu:634, 651-655): the short sequences where `safe` (the operand ranges of
// walk_exact_kernel), the plain expressions otherwise -- the same bits
__device__ __forceinline__ void exact_term(double dx, double dy, double d2, double num, bool safe, double &tx, double &ty)
{
    double f, ux, uy;
    if (safe) {
        const double y = __builtin_amdgcn_rsq(d2);
        double g = d2 * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        double e = __builtin_fma(-g, g, d2);
        h = __builtin_fma(h, r, h);
        g = __builtin_fma(e, h, g);
        e = __builtin_fma(-g, g, d2);
        g = __builtin_fma(e, h, g);
        const double dd = g + 1e-15;
        auto recip = [](double b) {
            double r0 = __builtin_amdgcn_rcp(b);
            double t = __builtin_fma(-b, r0, 1.0);
            r0 = __builtin_fma(r0, t, r0);
            t = __builtin_fma(-b, r0, 1.0);
            return __builtin_fma(r0, t, r0);
        };
        auto quot = [](double a, double b, double rb) {
            const double q0 = a * rb;
            const double rem = __builtin_fma(-b, q0, a);
            return __builtin_fma(rem, rb, q0);
        };
        const double r2 = recip(d2), rd = recip(dd);
        f = quot(num, d2, r2);
        ux = quot(dx, dd, rd); uy = quot(dy, dd, rd);
    } else {
        const double d = sqrt(d2) + 1e-15;
        f = num / d2;
        ux = dx / d; uy = dy / d;
    }
    tx = f * ux;
    ty = f * uy;
}

#define BH_WAVE_SYNC()                                                                              \
    do {                                                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                      \
        __builtin_amdgcn_wave_barrier();                                                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                      \
    } while (0)

template <bool COMPAT, int TERMS>
__global__ __launch_bounds__(kBlock) void walk_exact_bfs_kernel(
    const NodeD *__restrict__ gd, const LinkD *__restrict__ ld, const uint32_t *__restrict__ perm,
    double2 *__restrict__ pos, double2 *__restrict__ vel, const double *__restrict__ mass,
    double2 *__restrict__ force_out, int64_t lo, int64_t hi, double G, double dt, int integrate, TreeCounters *ctr,
    double *__restrict__ partial, double *slots, const double *__restrict__ box)
{
    __shared__ uint64_t s_meta[kWavesPerBlock][kBfsQueue];       // node id | depth << 32
    __shared__ uint64_t s_qkey[kWavesPerBlock][kBfsQueue];
    __shared__ uint64_t s_tkey[kWavesPerBlock][TERMS];
    __shared__ double2 s_term[kWavesPerBlock][TERMS];
    constexpr int kBfsTerms = TERMS;
    if (ctr->overflow) return;
    const int lane = lane_id(), w = wave_id();
    constexpr double kLaneLo = 0x1p-150, kLaneHi = 0x1p150, kTiny = 0x1p-200, kHuge = 0x1p400;
    const bool box_ok = (box[1] - box[0]) <= 0x1p199 && (box[3] - box[2]) <= 0x1p199;        // (walk_exact_asm's precondition)
    // body k of this wave: sorted index lo + (k * gridDim.x + blockIdx.x) * 4 + w; its new position stays in lane k for the
    // workgroup's bounds record
    double npx = 0.0, npy = 0.0;
    bool np_valid = false;
    for (int turn = 0; turn < kBfsBodiesPerWave; ++turn) {
    const int64_t s = lo + ((int64_t)turn * gridDim.x + blockIdx.x) * kWavesPerBlock + w;
    const bool have = s < hi;                                     // wave-uniform
    if (!have) break;
    const int64_t body = (int64_t)perm[s];
    const double2 p = pos[body];
    const double mi = mass[body];
    const double Gm = G * mi;
    const bool lane_safe = fabs(Gm) >= kLaneLo && fabs(Gm) <= kLaneHi;
    double fx = 0.0, fy = 0.0;

    {
        // one node for this lane: its term (take) or its children (open)
        auto decide = [&](int32_t node, bool active, bool &take, bool &open, int32_t &child, double &tx, double &ty) {
            take = false; open = false; child = -1; tx = 0.0; ty = 0.0;
            if (!active) return;
            const NodeD q = gd[node];
            const LinkD k = ld[node];
            if (q.m <= 1e-15) return;                             // project.cu:617
            const bool leaf = k.child < 0;
            const double dx = q.cx - p.x, dy = q.cy - p.y;
            const double d2 = dx * dx + dy * dy;
            const bool accept = leaf || (d2 >= q.size);           // the exact threshold: project.cu:634, 643
            bool self = false;
            if (leaf) {
                self = ((int64_t)k.occ == body);
                if (COMPAT) self = self || ((int64_t)k.occ + 2 == -body);   // project.cu:646
            }
            if (accept) {
                if (self) return;
                const bool safe = lane_safe && q.m <= kLaneHi && fabs(dx) >= kTiny && fabs(dy) >= kTiny && d2 <= kHuge;
                exact_term(dx, dy, d2, Gm * q.m, safe, tx, ty);
                take = true;
            } else {
                open = true; child = k.child;
            }
        };
        uint32_t head = 0, tail = 1, tcount = 0;
        bool spill = false;
        if (lane == 0) { s_meta[w][0] = 0ull; s_qkey[w][0] = 0ull; }
        BH_WAVE_SYNC();
        while (head != tail && !spill) {
            const uint32_t cnt = (tail - head < (uint32_t)kWave) ? tail - head : (uint32_t)kWave;
            const bool active = (uint32_t)lane < cnt;
            const uint32_t at = (head + (uint32_t)lane) & (kBfsQueue - 1);
            const uint64_t meta = active ? s_meta[w][at] : 0ull;
            const uint64_t key = active ? s_qkey[w][at] : 0ull;
            head += cnt;
            bool take, open;
            int32_t child;
            double tx, ty;
            decide((int32_t)(uint32_t)meta, active, take, open, child, tx, ty);
            const uint64_t tb = __ballot(take), ob = __ballot(open);
            const uint32_t nt = (uint32_t)__popcll(tb), no = (uint32_t)__popcll(ob);
            if (tcount + nt > (uint32_t)kBfsTerms || (tail - head) + 4u * no > (uint32_t)kBfsQueue) { spill = true; break; }
            const uint64_t below = (1ull << lane) - 1ull;
            if (take) {
                const uint32_t at_t = tcount + (uint32_t)__popcll(tb & below);
                s_tkey[w][at_t] = key;
                s_term[w][at_t] = double2{tx, ty};
            }
            if (open) {
                const uint32_t depth = (uint32_t)(meta >> 32) + 1u;               // the children's depth: 1 .. 31
                const uint32_t base = tail + 4u * (uint32_t)__popcll(ob & below);
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                    const uint32_t q_at = (base + c) & (kBfsQueue - 1);
                    s_meta[w][q_at] = (uint64_t)(uint32_t)(child + (int32_t)c) | ((uint64_t)depth << 32);
                    s_qkey[w][q_at] = key | ((uint64_t)(3u - c) << (62u - 2u * depth));   // child 3 is popped first
                }
            }
            tcount += nt;
            tail += 4u * no;
            BH_WAVE_SYNC();
        }
        if (!spill) {
            // the reference's order of additions = ascending key: rank by counting, permute, add one after the other
            // (64 terms at a time: most walks hold 100-200 terms, two to four rounds of the TERMS / 64 the list has room for)
            double2 mt[kBfsTerms / kWave];
            uint32_t rk[kBfsTerms / kWave];
            const uint32_t tpad = (tcount + 3u) & ~3u;
            if ((uint32_t)lane < tpad - tcount) s_tkey[w][tcount + (uint32_t)lane] = ~0ull;      // (pads the last group of four: never below a key)
            BH_WAVE_SYNC();
#pragma unroll
            for (int k = 0; k < kBfsTerms / kWave; ++k) {
                const uint32_t i = (uint32_t)lane + (uint32_t)(k * kWave);
                rk[k] = 0;
                mt[k] = double2{0.0, 0.0};
                if ((uint32_t)(k * kWave) < tcount) {                                             // wave-uniform
                    const uint64_t mine = (i < tcount) ? s_tkey[w][i] : 0ull;
                    mt[k] = (i < tcount) ? s_term[w][i] : double2{0.0, 0.0};
                    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                    for (uint32_t j = 0; j < tpad; j += 4) {
                        r0 += (s_tkey[w][j] < mine) ? 1u : 0u;
                        r1 += (s_tkey[w][j + 1] < mine) ? 1u : 0u;
                        r2 += (s_tkey[w][j + 2] < mine) ? 1u : 0u;
                        r3 += (s_tkey[w][j + 3] < mine) ? 1u : 0u;
                    }
                    rk[k] = (r0 + r1) + (r2 + r3);
                }
            }
            BH_WAVE_SYNC();
#pragma unroll
            for (int k = 0; k < kBfsTerms / kWave; ++k)
                if ((uint32_t)lane + (uint32_t)(k * kWave) < tcount) s_term[w][rk[k]] = mt[k];
            BH_WAVE_SYNC();
            for (uint32_t r = 0; r < tcount; ++r) {
                const double2 t = s_term[w][r];
                fx += t.x;
                fy += t.y;
            }
        } else {
            // the cooperative walk for this body alone (lane 0), from the root
            bool take, open;
            int32_t child;
            double tx, ty;
            decide(0, true, take, open, child, tx, ty);
            if (take) { fx += tx; fy += ty; }
            const uint64_t ob = __ballot(open && lane == 0);
            if (ob != 0ull) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
                walk_exact_asm<COMPAT>((const char __attribute__((address_space(4))) *)gd,
                                       (const char __attribute__((address_space(4))) *)ld,
                                       __builtin_amdgcn_readfirstlane(child), 1ull, __ballot(lane_safe && box_ok) & 1ull, p.x, p.y, Gm,
                                       (int32_t)body, COMPAT ? (int32_t)(-body - 2) : (int32_t)body, fx, fy);
#pragma clang diagnostic pop
            }
        }
    }

    // (every lane holds the same sums: every lane integrates, lane 0 stores)
    double2 np = p;
    if (lane == 0) force_out[body] = double2{fx, fy};
    if (integrate) {
        // updateAccVelPos, project.cu:827-834
        const double ax = fx / mi, ay = fy / mi;
        double2 v = vel[body];
        v.x += ax * dt;  v.y += ay * dt;
        np.x += v.x * dt;  np.y += v.y * dt;
        BH_WAVE_SYNC();                                           // (all lanes have read vel / pos of this body)
        if (lane == 0) { vel[body] = v; pos[body] = np; }
    }
    if (lane == turn) { npx = np.x; npy = np.y; np_valid = true; }
    }   // turn
    if (partial) block_bounds_to_partial(np_valid, npx, npy, partial + 4 * (size_t)blockIdx.x, slots);
}

}  // namespace bh
