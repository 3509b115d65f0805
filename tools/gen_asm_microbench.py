#!/usr/bin/env python3
"""Generates tools/asm_microbench.s: loops of hand-allocated F29 mul / sqr / dot3 on internal limbs (12-dword padded
elements), for timing and bit-exact comparison with the C++ f29.hpp routines (tools/microbench.hip loads the code
object with hipModuleLoad)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "crypto_primitives_amd", "csrc", "asm"))
from f29asm import Asm, F29Ctx, v, vv, s, kernel_header, KERNEL_TAIL, metadata, FILE_HEAD  # noqa: E402

A, B, C3, OUT, M, A2, ACC = 10, 20, 60, 40, 30, 70, 50


def prologue(a, nload):
    a.e("s_load_dwordx2 s[4:5], s[0:1], 0x0")
    a.e("s_load_dword s6, s[0:1], 0x8")
    a.e("v_lshl_or_b32 v1, s2, 8, v0")       # global lane index
    a.e("v_mov_b32_e32 v3, 48")
    a.e("s_waitcnt lgkmcnt(0)")
    for n in range(nload):                    # element n of lane i is x[i ^ n]
        a.e("v_xor_b32_e32 v2, %d, v1" % n)
        a.e("v_mad_u64_u32 %s, vcc, v2, v3, s[4:5]" % vv(4 + 2 * n))
    for n, base in zip(range(nload), (A, B, C3)):
        addr = vv(4 + 2 * n)
        a.e("global_load_dwordx4 v[%d:%d], %s, off" % (base, base + 3, addr))
        a.e("global_load_dwordx4 v[%d:%d], %s, off offset:16" % (base + 4, base + 7, addr))
        a.e("global_load_dword v%d, %s, off offset:32" % (base + 8, addr))
    a.e("s_waitcnt vmcnt(0)")


def epilogue(a):
    a.e("global_store_dwordx4 v[4:5], v[%d:%d], off" % (A, A + 3))
    a.e("global_store_dwordx4 v[4:5], v[%d:%d], off offset:16" % (A + 4, A + 7))
    a.e("global_store_dword v[4:5], v%d, off offset:32" % (A + 8))


def loop_ctl(a, name):
    a.e("s_sub_u32 s6, s6, 1")
    a.e("s_cmp_lg_u32 s6, 0")
    a.e("s_cbranch_scc1 %s" % name)


def main():
    out = [FILE_HEAD]
    kernels = []
    for name in ("asm_f29_mul", "asm_f29_sqr", "asm_f29_dot3"):
        a = Asm()
        ctx = F29Ctx(a, 8, ACC, M)
        prologue(a, {"asm_f29_mul": 2, "asm_f29_sqr": 1, "asm_f29_dot3": 3}[name])
        ctx.load_p()
        a.label(".L_%s_loop" % name)
        n0 = a.count
        if name == "asm_f29_mul":
            ctx.mul(A, B, OUT)
            for i in range(9):
                a.e("v_mov_b32_e32 %s, %s" % (v(A + i), v(OUT + i)))
        elif name == "asm_f29_sqr":
            ctx.sqr(A, OUT, A2)
            for i in range(9):
                a.e("v_mov_b32_e32 %s, %s" % (v(A + i), v(OUT + i)))
        else:  # r = dot3(a, one, b, k_in, c, k_out) with constants re-used from b-limbs: use VGPR operands b, c, a rotated
            ctx.dot3(A, [v(B + i) for i in range(9)], B, [v(C3 + i) for i in range(9)], C3, [v(B + i) for i in range(9)], OUT)
            for i in range(9):  # c = b; b = a; a = r
                a.e("v_mov_b32_e32 %s, %s" % (v(C3 + i), v(B + i)))
                a.e("v_mov_b32_e32 %s, %s" % (v(B + i), v(A + i)))
                a.e("v_mov_b32_e32 %s, %s" % (v(A + i), v(OUT + i)))
        sys.stderr.write("%s: %d instructions per iteration\n" % (name, a.count - n0))
        loop_ctl(a, ".L_%s_loop" % name)
        epilogue(a)
        out.append(kernel_header(name) + a.text() + KERNEL_TAIL.format(name=name, kernarg=12, vgprs=80, sgprs=24, accum=80))
        kernels.append(dict(name=name, kernarg=12, vgprs=80, sgprs=24, args=[(0, 8, "global_buffer"), (8, 4, "by_value")]))
    out.append(metadata(kernels))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "asm_microbench.s")
    open(path, "w").write("".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
