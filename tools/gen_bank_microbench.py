#!/usr/bin/env python3
"""Generates tools/bank_microbench.s: loops of 128 v_mad_u64_u32 / v_mad_i64_i32 with hand-picked operand registers, to
measure what the VGPR bank (register index mod 4) of the operands costs.  tools/bank_microbench.hip loads the code object
and prints cycles per wave instruction at 4 waves per SIMD."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "crypto_primitives_amd", "csrc", "asm"))
from f29asm import Asm, kernel_header, KERNEL_TAIL, metadata, FILE_HEAD  # noqa: E402

# name, list of (dst/src2 pair base, src0, src1) cycled over the 128 instructions
ACC = 8  # v[8:9]: banks 0, 1
PATTERNS = [
    ("vv_distinct", [(ACC, "v14", "v19")]),            # banks 2, 3 against the accumulator's 0, 1
    ("vv_same_bank", [(ACC, "v14", "v18")]),           # src0, src1 both bank 2
    ("vv_src0_acc_bank", [(ACC, "v12", "v19")]),       # src0 bank 0 = accumulator low
    ("vv_src1_acc_hi_bank", [(ACC, "v14", "v17")]),    # src1 bank 1 = accumulator high
    ("vv_both_acc_banks", [(ACC, "v12", "v17")]),
    ("vv_all_bank0", [(ACC, "v12", "v16")]),
    ("sv_clean", [(ACC, "v14", "s8")]),
    ("sv_conflict", [(ACC, "v12", "s8")]),
    ("vconst_clean", [(ACC, "v14", "1")]),
    ("vv_same_reg", [(ACC, "v14", "v14")]),            # a squaring term
    ("vv_rotating_operands", [(ACC, "v%d" % (12 + i), "v%d" % (32 + (7 * i) % 9)) for i in range(9)]),  # like a product column
    ("vv_two_accs_distinct", [(ACC, "v14", "v19"), (24, "v14", "v19")]),
    ("sub_only_e32", None),                            # v_sub_u32 for reference
]


def main():
    out = [FILE_HEAD]
    kernels = []
    for name, pat in PATTERNS:
        a = Asm()
        a.e("s_load_dword s6, s[0:1], 0x8")
        for r in range(8, 48):
            a.e("v_mov_b32_e32 v%d, %d" % (r, r))
        a.e("s_mov_b32 s8, 0x1234567")
        a.e("s_waitcnt lgkmcnt(0)")
        a.label(".L_%s" % name)
        for i in range(128):
            if pat is None:
                a.e("v_sub_u32_e32 v%d, v%d, v%d" % (8 + (i % 4), 14, 19))
            else:
                acc, x, y = pat[i % len(pat)]
                a.e("v_mad_u64_u32 v[%d:%d], vcc, %s, %s, v[%d:%d]" % (acc, acc + 1, x, y, acc, acc + 1))
        a.e("s_sub_u32 s6, s6, 1")
        a.e("s_cmp_lg_u32 s6, 0")
        a.e("s_cbranch_scc1 .L_%s" % name)
        out.append(kernel_header("bank_" + name) + a.text() + KERNEL_TAIL.format(name="bank_" + name, kernarg=12, vgprs=48, sgprs=24, accum=48))
        kernels.append(dict(name="bank_" + name, kernarg=12, vgprs=48, sgprs=24, args=[(0, 8, "global_buffer"), (8, 4, "by_value")]))
    out.append(metadata(kernels))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bank_microbench.s")
    open(path, "w").write("".join(out))
    open(os.path.join(os.path.dirname(path), "bank_microbench.names"), "w").write("\n".join("bank_" + n for n, _ in PATTERNS) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
