#!/usr/bin/env python3
"""Generates f29_asm.inc: GCN inline-assembly versions of the unsigned radix-2^29 square and product of f29.hpp
(f29_sqr / f29_mul for FU), written as ONE accumulator chain per routine -- what the hand-allocated kernels of
f29asm.py showed to be ~5 % faster than the compiler's multi-chain schedule (profiles/r01_s7_microbench_asm_vs_cpp.txt).

Same arithmetic, column by column, as the C++ routines:
  column k: acc += sum a_i b_(k-i) (+ sum m_i p_(k-i));  k < 9: m_k = -acc mod 2^29, acc += m_k;  acc >>= 29
  (k >= 9: limb k-9 of the result = acc mod 2^29).
The quotient digits m_k live in the result registers (limb j is written at column 9 + j, m_j is last read at column
j + 8).  The 64-bit accumulator is pinned to v[ACC:ACC+1] so that its low half can be named; p[1..8] arrive in SGPRs.

    python3 gen_inline.py > f29_asm.inc
"""
MASK = "0x1fffffff"
ACC = 2  # v[2:3]


def emit_routine(name, kind, signed=False, terms=1, inplace=False):
    """kind: "sqr" | "mul" | "mulc" (terms = 1) | "dot" (terms = 2 or 3: sum of a_k * b_k with the b_k in SGPRs).
    inplace (sqr / mul): the result overwrites a -- limb j of a is last read at column j + 8 and limb j of the result is
    written at column 9 + j -- and the quotient digits get registers of their own; a loop `r = r * r` then needs no
    copies between iterations."""
    lines = []
    e = lines.append
    acc64, acclo = "v[%d:%d]" % (ACC, ACC + 1), "v%d" % ACC
    first = [True]
    madop = "v_mad_i64_i32" if signed else "v_mad_u64_u32"
    shr = "v_ashrrev_i64" if signed else "v_lshrrev_b64"

    def mad(x, y, unsigned_term=False):
        src2 = "0" if first[0] else acc64
        first[0] = False
        # quotient digits and p are non-negative and < 2^29: the signed multiply-add takes them as they are
        e("%s %s, vcc, %s, %s, %s" % (madop, acc64, x, y, src2))

    M = "m" if inplace else "t"  # quotient digits: own registers (in place) or the result registers
    O = "a" if inplace else "t"  # result limbs
    if kind == "sqr":
        for i in range(9):
            e("v_lshlrev_b32_e32 %%[d%d], 1, %%[a%d]" % (i, i))
    for k in range(17):
        for i in range(max(0, k - 8), min(k, 8) + 1):
            j = k - i
            if kind == "sqr":
                if i > j:
                    continue
                mad("%%[d%d]" % i if i < j else "%%[a%d]" % i, "%%[a%d]" % j)
            elif kind in ("mul", "mulc"):
                mad("%%[a%d]" % i, "%%[b%d]" % j)
            else:
                for term in range(terms):
                    mad("%%[a%d_%d]" % (term, i), "%%[b%d_%d]" % (term, j))
        for i in range(max(0, k - 8), min(k - 1, 8) + 1):  # quotient digits times p[k - i], k - i >= 1
            mad("%%[%s%d]" % (M, i), "%%[p%d]" % (k - i))  # signed flavour: [p.] holds -p[.]
        if k < 9:
            if signed:
                # subtractive reduction: m = acc mod 2^29, acc -= m * p  (p[0] = 1).  Subtracting m only clears the low
                # 29 bits, which the arithmetic shift drops anyway (floor), so the column step is and + shift: no negation,
                # no add.  The result is (ab - mp) / 2^261 in (-2.1p, 1.1p) instead of (ab + mp) / 2^261 in (-1.1p, 2.1p):
                # same magnitude bound
                e("v_and_b32_e32 %%[%s%d], %s, %s" % (M, k, MASK, acclo))
            else:
                e("v_sub_u32_e32 %%[%s%d], 0, %s" % (M, k, acclo))
                e("v_and_b32_e32 %%[%s%d], %s, %%[%s%d]" % (M, k, MASK, M, k))
                mad("%%[%s%d]" % (M, k), "1")
            e("%s %s, 29, %s" % (shr, acc64, acc64))
        else:
            e("v_and_b32_e32 %%[%s%d], %s, %s" % (O, k - 9, MASK, acclo))
            e("%s %s, 29, %s" % (shr, acc64, acc64))
    e("v_mov_b32_e32 %%[%s8], %s" % (O, acclo))
    n_instr = len(lines)
    body = "\n".join('        "%s\\n"' % l for l in lines)
    T = "FS" if signed else "FU"
    L = "int32_t" if signed else "u32"
    if inplace:
        outs = ", ".join('[a%d] "+v"(a.l[%d])' % (i, i) for i in range(9)) + ", " + ", ".join('[m%d] "=&v"(m%d)' % (i, i) for i in range(9))
    else:
        outs = ", ".join('[t%d] "=&v"(t.l[%d])' % (i, i) for i in range(9))
    if kind == "sqr":
        outs += ", " + ", ".join('[d%d] "=&v"(d%d)' % (i, i) for i in range(9))
    outs += ', [acc] "=&{v[%d:%d]}"(acc)' % (ACC, ACC + 1)
    if kind == "dot":
        ins = ", ".join('[a%d_%d] "v"(a%d.l[%d])' % (k, i, k, i) for k in range(terms) for i in range(9))
        ins += ", " + ", ".join('[b%d_%d] "s"(b%d.l[%d])' % (k, i, k, i) for k in range(terms) for i in range(9))
        args = ", ".join("const %s& a%d, const %s& b%d" % (T, k, T, k) for k in range(terms))
    else:
        ins = "" if inplace else ", ".join('[a%d] "v"(a.l[%d])' % (i, i) for i in range(9))
        if kind in ("mul", "mulc"):
            ins += (", " if ins else "") + ", ".join('[b%d] "%s"(b.l[%d])' % (i, "v" if kind == "mul" else "s", i) for i in range(9))
        a_arg = ("%s& a" if inplace else "const %s& a") % T
        args = a_arg if kind == "sqr" else a_arg + ", const %s& b" % T
    ins += (", " if ins else "") + ", ".join('[p%d] "s"(%sp29(%d))' % (i, "-(int32_t)" if signed else "", i) for i in range(1, 9))
    decl = ""
    if kind == "sqr":
        decl += "    %s " % L + ", ".join("d%d" % i for i in range(9)) + ";\n"
    if inplace:
        decl += "    u32 " + ", ".join("m%d" % i for i in range(9)) + ";\n"
        return ("// %d instructions, in place\n__device__ __forceinline__ void %s(%s) {\n    %s acc;\n%s    asm(\n%s\n        : %s\n        : %s\n        : \"vcc\");\n    (void)acc;\n}\n"
                % (n_instr, name, args, "int64_t" if signed else "u64", decl, body, outs, ins))
    return ("// %d instructions\n__device__ __forceinline__ %s %s(%s) {\n    %s t;\n    %s acc;\n%s    asm(\n%s\n        : %s\n        : %s\n        : \"vcc\");\n    (void)acc;\n    return t;\n}\n"
            % (n_instr, T, name, args, T, "int64_t" if signed else "u64", decl, body, outs, ins))


print("// GENERATED by asm/gen_inline.py -- do not edit.  Single-chain inline-assembly field routines (device only).")
print(emit_routine("f29_sqr_asm", "sqr"))
print(emit_routine("f29_mul_asm", "mul"))
print("// in-place forms for chains r = r * r, r = r * x (the S-box): no register copies between the steps")
print(emit_routine("f29_sqr_ip_asm", "sqr", inplace=True))
print(emit_routine("f29_mul_ip_asm", "mul", inplace=True))
print("// wave-uniform second operands (b, b_k) in SGPRs: products with constants, the linear layers of Poseidon")
print(emit_routine("f29_mulc_asm", "mulc"))
print(emit_routine("f29_dot2_asm", "dot", terms=2))
print(emit_routine("f29_dot3_asm", "dot", terms=3))
print("// signed flavour (Jubjub arithmetic): v_mad_i64_i32 and arithmetic shifts; the quotient digits stay non-negative")
print(emit_routine("f29_sqr_asm", "sqr", signed=True))
print(emit_routine("f29_mul_asm", "mul", signed=True))
print(emit_routine("f29_mulc_asm", "mulc", signed=True))
print("// signed in-place forms and dot products: the Poseidon kernels run in the signed flavour too (two instructions per")
print("// reduction column instead of four)")
print(emit_routine("f29_sqr_ip_asm", "sqr", signed=True, inplace=True))
print(emit_routine("f29_mul_ip_asm", "mul", signed=True, inplace=True))
print(emit_routine("f29_dot2_asm", "dot", signed=True, terms=2))
print(emit_routine("f29_dot3_asm", "dot", signed=True, terms=3))
print("// four and five terms under ONE reduction: legal in 64 signed bits because the constants are stored with BALANCED digits")
print("// (limbs in [-2^28, 2^28), f29_balance): 45 products of < 2^57 stay below 2^62.5")
print(emit_routine("f29_dot4_asm", "dot", signed=True, terms=4))
print(emit_routine("f29_dot5_asm", "dot", signed=True, terms=5))
