#!/usr/bin/env python3
"""Writes valu_bank.hip: a VALU issue-rate microbenchmark with EXPLICIT register numbers.

Question it answers: which operand forms of the candidate inner-loop instructions issue at 2 cycles per
wave64 instruction on gfx950 and which at 4 -- by number of distinct VGPR sources, by the bank
(register number mod 4) those sources sit in, by SGPR / inline-constant operands, by encoding (VOP2 vs
VOP3), and whether the destination is also a source.  Every kernel is one asm block (so the compiler
cannot move registers): 8 independent chains v0..v7, 64 instructions per loop trip, operands in v8..v31
and s21.

    python3 gen_valu_bank.py > valu_bank.hip && hipcc -O3 --offload-arch=gfx950 valu_bank.hip -o valu_bank
"""

FORMS = []


def form(name, fn, insts=64):
    body = [fn(i % 8, i) for i in range(insts)]
    FORMS.append((name, body))


def forms_set1():
    # ---- fp32 add -----------------------------------------------------------------------------------
    form("add_f32 vN,vN,vN (vop2, 1 reg)", lambda n, i: f"v_add_f32 v{n}, v{n}, v{n}")
    form("add_f32 vN,v8,vN (fixed x)", lambda n, i: f"v_add_f32 v{n}, v8, v{n}")
    form("add_f32 vN,v[8+N],vN (same bank)", lambda n, i: f"v_add_f32 v{n}, v{8 + n}, v{n}")
    form("add_f32 vN,v[9+N],vN (bank+1)", lambda n, i: f"v_add_f32 v{n}, v{9 + n}, v{n}")
    form("add_f32 vN,v[10+N],vN (bank+2)", lambda n, i: f"v_add_f32 v{n}, v{10 + n}, v{n}")
    form("add_f32 vN,v8,v9 (dst not src)", lambda n, i: f"v_add_f32 v{n}, v8, v9")
    form("add_f32 vN,v8,v12 (dst not src, same bank)", lambda n, i: f"v_add_f32 v{n}, v8, v12")
    form("add_f32 vN,v8,v8 (dst not src, 1 reg)", lambda n, i: f"v_add_f32 v{n}, v8, v8")
    form("add_f32 vN,s21,vN (vop2 sgpr)", lambda n, i: f"v_add_f32 v{n}, s21, v{n}")
    form("add_f32_e64 vN,vN,s21 (vop3 sgpr)", lambda n, i: f"v_add_f32_e64 v{n}, v{n}, s21")
    form("add_f32 vN,1.0,vN (inline const)", lambda n, i: f"v_add_f32 v{n}, 1.0, v{n}")
    form("add_f32_e64 vN,vN,|vN| (vop3 1 reg)", lambda n, i: f"v_add_f32_e64 v{n}, v{n}, |v{n}|")
    form("mov_b32 vN,v8", lambda n, i: f"v_mov_b32 v{n}, v8")
    form("mov_b32 vN,s21", lambda n, i: f"v_mov_b32 v{n}, s21")
    # ---- integer two-operand ---------------------------------------------------------------------------
    form("add_u32 vN,vN,vN", lambda n, i: f"v_add_u32 v{n}, v{n}, v{n}")
    form("add_u32 vN,v[9+N],vN", lambda n, i: f"v_add_u32 v{n}, v{9 + n}, v{n}")
    form("min_u32 vN,v[9+N],vN", lambda n, i: f"v_min_u32 v{n}, v{9 + n}, v{n}")
    form("pk_add_u16 vN,vN,vN", lambda n, i: f"v_pk_add_u16 v{n}, v{n}, v{n}")
    form("pk_add_u16 vN,v[9+N],vN", lambda n, i: f"v_pk_add_u16 v{n}, v{9 + n}, v{n}")
    form("pk_sub_i16 vN,v[9+N],v[18+N] (dst not src)", lambda n, i: f"v_pk_sub_i16 v{n}, v{9 + n}, v{18 + n}")
    form("pk_max_i16 vN,vN,vN neg_lo/hi", lambda n, i: f"v_pk_max_i16 v{n}, v{n}, v{n}")
    # ---- three-operand ---------------------------------------------------------------------------------
    form("sad_u16 vN,vN,vN,vN (1 reg)", lambda n, i: f"v_sad_u16 v{n}, v{n}, v{n}, v{n}")
    form("sad_u16 vN,v8,v8,vN (2 regs)", lambda n, i: f"v_sad_u16 v{n}, v8, v8, v{n}")
    form("sad_u16 vN,v8,v9,vN (3 regs, fixed x y)", lambda n, i: f"v_sad_u16 v{n}, v8, v9, v{n}")
    form("sad_u16 vN,v[8+N],v[16+N],vN (all same bank)", lambda n, i: f"v_sad_u16 v{n}, v{8 + n}, v{16 + n}, v{n}")
    form("sad_u16 vN,v[9+N],v[18+N],vN (3 banks)", lambda n, i: f"v_sad_u16 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("sad_u16 vN,v[9+N],s21,vN (2 banks + sgpr)", lambda n, i: f"v_sad_u16 v{n}, v{9 + n}, s21, v{n}")
    form("sad_u16 vN,v[8+N],s21,vN (same bank + sgpr)", lambda n, i: f"v_sad_u16 v{n}, v{8 + n}, s21, v{n}")
    form("sad_u16 vN,vN,s21,vN (1 reg + sgpr)", lambda n, i: f"v_sad_u16 v{n}, v{n}, s21, v{n}")
    form("sad_u16 vN,vN,1,vN (1 reg + const)", lambda n, i: f"v_sad_u16 v{n}, v{n}, 1, v{n}")
    form("sad_u16 vN,v[9+N],1,vN (2 banks + const)", lambda n, i: f"v_sad_u16 v{n}, v{9 + n}, 1, v{n}")
    form("sad_u16 vN,v8,v9,v10 (dst not src)", lambda n, i: f"v_sad_u16 v{n}, v8, v9, v10")
    form("sad_u16 vN,v[9+N],v[18+N],0 (no acc)", lambda n, i: f"v_sad_u16 v{n}, v{9 + n}, v{18 + n}, 0")
    form("sad_u8 vN,v[9+N],v[18+N],vN (3 banks)", lambda n, i: f"v_sad_u8 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("sad_u8 vN,vN,vN,vN (1 reg)", lambda n, i: f"v_sad_u8 v{n}, v{n}, v{n}, v{n}")
    form("fma_f32 vN,vN,vN,vN (1 reg)", lambda n, i: f"v_fma_f32 v{n}, v{n}, v{n}, v{n}")
    form("fma_f32 vN,v[9+N],v[18+N],vN (3 banks)", lambda n, i: f"v_fma_f32 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("fma_f32 vN,v[9+N],s21,vN", lambda n, i: f"v_fma_f32 v{n}, v{9 + n}, s21, v{n}")
    form("fmac_f32 vN,v[9+N],v[18+N] (vop2)", lambda n, i: f"v_fmac_f32 v{n}, v{9 + n}, v{18 + n}")
    form("add3_u32 vN,vN,vN,vN", lambda n, i: f"v_add3_u32 v{n}, v{n}, v{n}, v{n}")
    form("add3_u32 vN,v[9+N],v[18+N],vN", lambda n, i: f"v_add3_u32 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("med3_u32 vN,v[9+N],v[18+N],vN", lambda n, i: f"v_med3_u32 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("lshl_add_u32 vN,vN,v[9+N],s21", lambda n, i: f"v_lshl_add_u32 v{n}, v{n}, v{9 + n}, s21")
    form("dot4_u32_u8 vN,v[9+N],v[18+N],vN", lambda n, i: f"v_dot4_u32_u8 v{n}, v{9 + n}, v{18 + n}, v{n}")
    form("dot2_u32_u16 vN,v[9+N],v[18+N],vN", lambda n, i: f"v_dot2_u32_u16 v{n}, v{9 + n}, v{18 + n}, v{n}")
    # ---- mixes: is the limit per instruction, or a shared port (SGPR reads, VGPR reads) over time? ------
    form("mix sad(v,s,acc) / add_f32 self", lambda n, i: (f"v_sad_u16 v{n}, v{9 + n}, s21, v{n}" if i % 2 == 0
                                                            else f"v_add_f32 v{n}, v{n}, v{n}"))
    form("mix sad(3 banks) / add_f32 self", lambda n, i: (f"v_sad_u16 v{n}, v{9 + n}, v{18 + n}, v{n}" if i % 2 == 0
                                                           else f"v_add_f32 v{n}, v{n}, v{n}"))
    form("mix sad(3 banks) x3 / add self x1", lambda n, i: (f"v_sad_u16 v{n}, v{9 + n}, v{18 + n}, v{n}" if i % 4 != 3
                                                              else f"v_add_f32 v{n}, v{n}, v{n}"))
    form("mix sub(v,s)->dst / add self (r01 pair)", lambda n, i: (f"v_sub_f32_e64 v{n}, v8, s21" if i % 2 == 0
                                                                   else f"v_add_f32_e64 v{n}, v{n}, |v{n}|"))
    form("sub_f32_e64 vN,v8,s21 (r01 'v_sub only')", lambda n, i: f"v_sub_f32_e64 v{n}, v8, s21")
    form("mix add(v8,v9)->dst / add self", lambda n, i: (f"v_add_f32 v{n}, v8, v9" if i % 2 == 0
                                                          else f"v_add_f32 v{n}, v{n}, v{n}"))
    # ---- packed / fp64 for scale --------------------------------------------------------------------------
    form("pk_add_f32 v[2N:2N+1] self", lambda n, i: f"v_pk_add_f32 v[{2 * n}:{2 * n + 1}], v[{2 * n}:{2 * n + 1}], v[{2 * n}:{2 * n + 1}]")
    form("add_f64 v[2N:2N+1] self", lambda n, i: f"v_add_f64 v[{2 * n}:{2 * n + 1}], v[{2 * n}:{2 * n + 1}], v[{2 * n}:{2 * n + 1}]")



def forms_set2():
    """Which single instructions issue at the full rate, and what co-issues with v_sad_u16."""
    a, b, c = "v{n}", "v{m}", "v{q}"

    def one(name, tmpl):
        form(name, lambda n, i: tmpl.format(n=n, m=9 + n, q=18 + n))

    one("min_f32 vN,vM,vN", "v_min_f32 v{n}, v{m}, v{n}")
    one("max_f32 vN,vM,vN", "v_max_f32 v{n}, v{m}, v{n}")
    one("med3_f32 vN,vM,vQ,vN", "v_med3_f32 v{n}, v{m}, v{q}, v{n}")
    one("min3_f32 vN,vM,vQ,vN", "v_min3_f32 v{n}, v{m}, v{q}, v{n}")
    one("max3_f32 vN,vM,vQ,vN", "v_max3_f32 v{n}, v{m}, v{q}, v{n}")
    one("mul_f32 vN,vM,vN", "v_mul_f32 v{n}, v{m}, v{n}")
    one("sub_f32 vN,vM,vN", "v_sub_f32 v{n}, v{m}, v{n}")
    one("sub_f32_e64 vN,vM,|vN|", "v_sub_f32_e64 v{n}, v{m}, |v{n}|")
    one("lshlrev_b32 vN,8,vN", "v_lshlrev_b32 v{n}, 8, v{n}")
    one("lshlrev_b32 vN,vM,vN", "v_lshlrev_b32 v{n}, v{m}, v{n}")
    one("lshrrev_b32 vN,8,vN", "v_lshrrev_b32 v{n}, 8, v{n}")
    one("or_b32 vN,vM,vN", "v_or_b32 v{n}, v{m}, v{n}")
    one("and_b32 vN,vM,vN", "v_and_b32 v{n}, v{m}, v{n}")
    one("xor_b32 vN,vM,vN", "v_xor_b32 v{n}, v{m}, v{n}")
    one("sub_u32 vN,vM,vN", "v_sub_u32 v{n}, v{m}, v{n}")
    one("max_u32 vN,vM,vN", "v_max_u32 v{n}, v{m}, v{n}")
    one("min_i32 vN,vM,vN", "v_min_i32 v{n}, v{m}, v{n}")
    one("cvt_f32_u32 vN,vM", "v_cvt_f32_u32 v{n}, v{m}")
    one("cvt_u32_f32 vN,vM", "v_cvt_u32_f32 v{n}, v{m}")
    one("cmp_lt_u32 vcc,vM,vN", "v_cmp_lt_u32 vcc, v{m}, v{n}")
    one("cmp_lt_f32 vcc,vM,vN", "v_cmp_lt_f32 vcc, v{m}, v{n}")
    one("cndmask_b32 vN,vM,vN,vcc", "v_cndmask_b32 v{n}, v{m}, v{n}, vcc")
    one("perm_b32 vN,vM,vQ,vN", "v_perm_b32 v{n}, v{m}, v{q}, v{n}")
    one("bfe_u32 vN,vM,8,8", "v_bfe_u32 v{n}, v{m}, 8, 8")
    one("lshl_or_b32 vN,vN,8,vM", "v_lshl_or_b32 v{n}, v{n}, 8, v{m}")
    one("and_or_b32 vN,vN,vM,vQ", "v_and_or_b32 v{n}, v{n}, v{m}, v{q}")
    one("lshl_add_u32 vN,vN,8,vM (no sgpr)", "v_lshl_add_u32 v{n}, v{n}, 8, v{m}")
    one("add_lshl_u32 vN,vN,vM,8", "v_add_lshl_u32 v{n}, v{n}, v{m}, 8")
    one("mad_u32_u24 vN,vM,vQ,vN", "v_mad_u32_u24 v{n}, v{m}, v{q}, v{n}")
    one("mul_u32_u24 vN,vM,vN", "v_mul_u32_u24 v{n}, v{m}, v{n}")
    one("mul_lo_u32 vN,vM,vN", "v_mul_lo_u32 v{n}, v{m}, v{n}")
    one("alignbit_b32 vN,vM,vN,8", "v_alignbit_b32 v{n}, v{m}, v{n}, 8")
    one("pk_add_f16 vN,vM,vN", "v_pk_add_f16 v{n}, v{m}, v{n}")
    one("pk_fma_f16 vN,vM,vQ,vN", "v_pk_fma_f16 v{n}, v{m}, v{q}, v{n}")
    one("pk_max_f16 vN,vM,vN", "v_pk_max_f16 v{n}, v{m}, v{n}")
    one("pk_mul_f16 vN,vM,vN", "v_pk_mul_f16 v{n}, v{m}, v{n}")
    one("add_f16 vN,vM,vN", "v_add_f16 v{n}, v{m}, v{n}")
    one("fma_f16 vN,vM,vQ,vN", "v_fma_f16 v{n}, v{m}, v{q}, v{n}")
    one("dot2_f32_f16 vN,vM,vQ,vN", "v_dot2_f32_f16 v{n}, v{m}, v{q}, v{n}")
    one("dot2c_f32_f16 vN,vM,vQ", "v_dot2c_f32_f16 v{n}, v{m}, v{q}")
    one("dot4c_i32_i8 vN,vM,vQ", "v_dot4c_i32_i8 v{n}, v{m}, v{q}")
    one("add_f32 vN,0x40490fdb,vN (literal)", "v_add_f32 v{n}, 0x40490fdb, v{n}")
    one("fmaak_f32 vN,vM,vN,0x40490fdb", "v_fmaak_f32 v{n}, v{m}, v{n}, 0x40490fdb")
    one("add_f32_dpp vN,vM,vN quad_perm", "v_add_f32_dpp v{n}, v{m}, v{n} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    one("add_f32_sdwa vN,vM,vN", "v_add_f32_sdwa v{n}, v{m}, v{n} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD")
    one("sad_u32 vN,vM,vQ,vN", "v_sad_u32 v{n}, v{m}, v{q}, v{n}")
    one("msad_u8 vN,vM,vQ,vN", "v_msad_u8 v{n}, v{m}, v{q}, v{n}")
    one("sad_hi_u8 vN,vM,vQ,vN", "v_sad_hi_u8 v{n}, v{m}, v{q}, v{n}")

    def mix(name, t_even, t_odd, period=2, odd_at=(1,)):
        form(name, lambda n, i: (t_odd if (i % period) in odd_at else t_even).format(n=n, m=9 + n, q=18 + n))

    SAD = "v_sad_u16 v{n}, v{m}, s21, v{n}"
    mix("mix sad / min_f32", SAD, "v_min_f32 v{n}, v{m}, v{n}")
    mix("mix sad / med3_f32", SAD, "v_med3_f32 v{n}, v{m}, v{q}, v{n}")
    mix("mix sad / min_u32", SAD, "v_min_u32 v{n}, v{m}, v{n}")
    mix("mix sad / med3_u32", SAD, "v_med3_u32 v{n}, v{m}, v{q}, v{n}")
    mix("mix sad / lshlrev 8", SAD, "v_lshlrev_b32 v{n}, 8, v{n}")
    mix("mix sad / or_b32", SAD, "v_or_b32 v{n}, v{m}, v{n}")
    mix("mix sad / add_u32", SAD, "v_add_u32 v{n}, v{m}, v{n}")
    mix("mix sad / lshl_add_u32 (no sgpr)", SAD, "v_lshl_add_u32 v{n}, v{n}, 8, v{m}")
    mix("mix sad / mov_b32 v,s", SAD, "v_mov_b32 v{n}, s21")
    mix("mix sad / add_f32 v,s", SAD, "v_add_f32 v{n}, s21, v{n}")
    mix("mix sad / cndmask", SAD, "v_cndmask_b32 v{n}, v{m}, v{n}, vcc")
    mix("mix sad / fma_f32", SAD, "v_fma_f32 v{n}, v{m}, v{q}, v{n}")
    mix("mix sad / pk_add_f16", SAD, "v_pk_add_f16 v{n}, v{m}, v{n}")
    mix("mix sad x1 / add_f32 x2", "v_add_f32 v{n}, v{m}, v{n}", SAD, period=3, odd_at=(0,))
    mix("mix sad x1 / add_f32 x3", "v_add_f32 v{n}, v{m}, v{n}", SAD, period=4, odd_at=(0,))
    mix("mix sad x2 / add_f32 x1", SAD, "v_add_f32 v{n}, v{m}, v{n}", period=3, odd_at=(2,))
    mix("mix min_u32 / add_f32", "v_min_u32 v{n}, v{m}, v{n}", "v_add_f32 v{n}, v{m}, v{n}")
    def pair_mix(name, op):
        def fn(n, i):
            k = (i // 2) % 4
            if i % 2 == 0:
                return f"{op} v[{2 * k}:{2 * k + 1}], v[{2 * k}:{2 * k + 1}], v[{2 * k}:{2 * k + 1}]"
            return f"v_add_f32 v{8 + (i // 2) % 8}, v{16 + (i // 2) % 8}, v{8 + (i // 2) % 8}"
        form(name, fn)

    pair_mix("mix pk_add_f32 / add_f32", "v_pk_add_f32")
    pair_mix("mix add_f64 / add_f32", "v_add_f64")



HEADER = r'''// GENERATED by gen_valu_bank.py -- VALU issue rate by operand form, explicit registers (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define CLOBBERS "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
    "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
    "s20","s21","scc","vcc"

#define INIT \
    "v_mov_b32 v0, %1\n\t" "v_add_u32 v1, 0x1234567, v0\n\t" "v_add_u32 v2, 0x2345671, v1\n\t" "v_add_u32 v3, 0x3456712, v2\n\t" \
    "v_add_u32 v4, 0x4567123, v3\n\t" "v_add_u32 v5, 0x5671234, v4\n\t" "v_add_u32 v6, 0x6712345, v5\n\t" "v_add_u32 v7, 0x7123456, v6\n\t" \
    "v_add_u32 v8, 0x1234567, v7\n\t" "v_add_u32 v9, 0x2345671, v8\n\t" "v_add_u32 v10, 0x3456712, v9\n\t" "v_add_u32 v11, 0x4567123, v10\n\t" \
    "v_add_u32 v12, 0x1234567, v11\n\t" "v_add_u32 v13, 0x2345671, v12\n\t" "v_add_u32 v14, 0x3456712, v13\n\t" "v_add_u32 v15, 0x4567123, v14\n\t" \
    "v_add_u32 v16, 0x1234567, v15\n\t" "v_add_u32 v17, 0x2345671, v16\n\t" "v_add_u32 v18, 0x3456712, v17\n\t" "v_add_u32 v19, 0x4567123, v18\n\t" \
    "v_add_u32 v20, 0x1234567, v19\n\t" "v_add_u32 v21, 0x2345671, v20\n\t" "v_add_u32 v22, 0x3456712, v21\n\t" "v_add_u32 v23, 0x4567123, v22\n\t" \
    "v_add_u32 v24, 0x1234567, v23\n\t" "v_add_u32 v25, 0x2345671, v24\n\t" "v_add_u32 v26, 0x3456712, v25\n\t" "v_add_u32 v27, 0x4567123, v26\n\t" \
    "v_add_u32 v28, 0x1234567, v27\n\t" "v_add_u32 v29, 0x2345671, v28\n\t" "v_add_u32 v30, 0x3456712, v29\n\t" "v_add_u32 v31, 0x4567123, v30\n\t" \
    "s_mov_b32 s20, %2\n\t" "s_mov_b32 s21, %3\n\t"

#define FINI \
    "s_sub_u32 s20, s20, 1\n\t" "s_cmp_lg_u32 s20, 0\n\t" "s_cbranch_scc1 1b\n\t" \
    "v_xor_b32 v0, v0, v1\n\t" "v_xor_b32 v0, v0, v2\n\t" "v_xor_b32 v0, v0, v3\n\t" "v_xor_b32 v0, v0, v4\n\t" \
    "v_xor_b32 v0, v0, v5\n\t" "v_xor_b32 v0, v0, v6\n\t" "v_xor_b32 v0, v0, v7\n\t" "v_mov_b32 %0, v0\n\t"

#define DEFINE_KERNEL(NAME, BODY)                                                                        \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, unsigned long long* clk, int iters, uint32_t sc_in) { \
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                                 \
    uint32_t x = tid * 2654435761u + 12345u;                                                             \
    uint32_t it = __builtin_amdgcn_readfirstlane((uint32_t)iters);                                        \
    uint32_t sc = __builtin_amdgcn_readfirstlane(sc_in);                                                  \
    uint32_t r;                                                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                             \
    asm volatile(INIT "1:\n\t" BODY FINI : "=v"(r) : "v"(x), "s"(it), "s"(sc) : CLOBBERS);              \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                             \
    if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }                                                 \
    if (r == 0x12345678u) out[tid] = r;                                                                   \
}
'''

MAIN = r'''
typedef void (*kern_t)(uint32_t*, unsigned long long*, int, uint32_t);
struct Form { const char* name; kern_t k; };

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    double clk_ghz = p.clockRate / 1e6;
    printf("device %s CUs %d nominal clock %.3f GHz\n", p.gcnArchName, cus, clk_ghz);
    std::vector<Form> forms = {
@FORMS@
    };
    uint32_t* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * cus * 8));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    printf("%-52s %6s %10s %12s %10s\n", "form", "w/SIMD", "ms", "shader GHz", "cyc/winst");
    for (auto& f : forms) {
        for (int wps : {1, 2, 4, 8}) {
            int grid = cus * wps;
            f.k<<<grid, 256>>>(out, clk, 64, 3);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                f.k<<<grid, 256>>>(out, clk, iters, 3);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) { best = ms; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); }
            }
            double winst = (double)iters * 64;
            double shader_ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;   // memrealtime ticks at 100 MHz
            // SIMD cycles per wave-instruction from the wall time: the launch's duration in shader cycles over the
            // wave-instructions one SIMD issued (wps waves of `winst` each).  (A wave's own cycle counter does not serve:
            // the oldest wave of a SIMD wins the arbitration and always sees its own, unshared rate.)
            double cyc = (double)best * 1e-3 * shader_ghz * 1e9 / (winst * wps);
            printf("%-52s %6d %10.4f %12.3f %10.3f\n", f.name, wps, best, shader_ghz, cyc);
        }
        fflush(stdout);
    }
    return 0;
}
'''


def main():
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "2":
        forms_set2()
    else:
        forms_set1()
    print(HEADER)
    rows = []
    for idx, (name, body) in enumerate(FORMS):
        s = " ".join('"%s\\n\\t"' % ins for ins in body)
        print(f"DEFINE_KERNEL(k_form{idx}, {s})")
        rows.append(f'        {{"{name}", k_form{idx}}},')
    print(MAIN.replace("@FORMS@", "\n".join(rows)))


if __name__ == "__main__":
    main()
