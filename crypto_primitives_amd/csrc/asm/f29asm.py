"""Tiny assembler DSL for hand-allocated gfx950 F29 arithmetic (see ../f29.hpp for the representation).

Why hand-written assembly: hipcc's best code for a 9x9 limb Montgomery product is 232-236 VALU instructions (it
re-associates the column sums into separate chains and pays ~15 64-bit adds + ~14 moves to join them); the
schedule below is the straight chain: 162 v_mad_u64_u32 + 43 others = 205, and on gfx950 cost == instruction count
(DESIGN.md section 3).  Everything here only GENERATES text; the output is assembled by clang into a code object.
"""

P29 = [0x00000001, 0x1ffffff8, 0x1f96ffbf, 0x1b4805ff, 0x1d80553b, 0x0c0404d0, 0x1520cce7, 0x0a6533af, 0x0073eda7]
MASK = 0x1fffffff


class Asm:
    def __init__(self):
        self.lines = []
        self.count = 0

    def e(self, s):
        self.lines.append("\t" + s)
        if not s.startswith((";", ".")):
            self.count += 1

    def label(self, name):
        self.lines.append(name + ":")

    def comment(self, s):
        self.lines.append("\t; " + s)

    def text(self):
        return "\n".join(self.lines) + "\n"


def v(i):
    return "v%d" % i


def vv(i):
    assert i % 2 == 0, "64-bit VGPR operands must be even-aligned on gfx90a+"
    return "v[%d:%d]" % (i, i + 1)


def s(i):
    return "s%d" % i


class F29Ctx:
    """register conventions shared by the generated routines:
       p_sgpr: SGPR index of p[1] (p[1..8] in 8 consecutive SGPRs), acc: even VGPR index of the 64-bit accumulator,
       m: VGPR index of 9 scratch registers for the quotient digits."""

    def __init__(self, asm, p_sgpr, acc, m):
        self.a, self.p, self.acc, self.m = asm, p_sgpr, acc, m

    def load_p(self):
        for i in range(1, 9):
            self.a.e("s_mov_b32 %s, 0x%08x" % (s(self.p + i - 1), P29[i]))

    def _mad(self, x, y, first=False):
        # acc = x * y + acc   (x: VGPR name, y: VGPR / SGPR name or inline constant)
        src2 = "0" if first else vv(self.acc)
        self.a.e("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (vv(self.acc), x, y, src2))

    def _mstep(self, k):
        acc, m = self.acc, self.m
        self.a.e("v_sub_u32_e32 %s, 0, %s" % (v(m + k), v(acc)))
        self.a.e("v_and_b32_e32 %s, 0x%08x, %s" % (v(m + k), MASK, v(m + k)))
        self._mad(v(m + k), "1")
        self.a.e("v_lshrrev_b64 %s, 29, %s" % (vv(acc), vv(acc)))

    def _out(self, dst):
        acc = self.acc
        self.a.e("v_and_b32_e32 %s, 0x%08x, %s" % (dst, MASK, v(acc)))
        self.a.e("v_lshrrev_b64 %s, 29, %s" % (vv(acc), vv(acc)))

    def _reduce_terms(self, k):
        lo = max(0, k - 8)
        hi = min(k - 1, 8)
        for i in range(lo, hi + 1):
            if k - i >= 1:
                self._mad(v(self.m + i), s(self.p + (k - i) - 1))

    def mul(self, a, b, out, b_names=None):
        """out[0..8] = a * b / 2^261.  a, out: VGPR base indices; b: VGPR base index, or b_names = list of 9 operand
        names (e.g. SGPRs holding a constant)."""
        bn = b_names or [v(b + i) for i in range(9)]
        first = True
        for k in range(17):
            for i in range(max(0, k - 8), min(k, 8) + 1):
                self._mad(v(a + i), bn[k - i], first)
                first = False
            self._reduce_terms(k)
            if k < 9:
                self._mstep(k)
            else:
                self._out(v(out + k - 9))
        self.a.e("v_mov_b32_e32 %s, %s" % (v(out + 8), v(self.acc)))

    def sqr(self, a, out, a2):
        """out = a^2 / 2^261; a2: 9 scratch VGPRs for the doubled operand."""
        for i in range(9):
            self.a.e("v_lshlrev_b32_e32 %s, 1, %s" % (v(a2 + i), v(a + i)))
        first = True
        for k in range(17):
            for i in range(max(0, k - 8), 9):
                j = k - i
                if j < 0 or j > 8 or i > j:
                    continue
                if i < j:
                    self._mad(v(a2 + i), v(a + j), first)
                else:
                    self._mad(v(a + i), v(a + i), first)
                first = False
            self._reduce_terms(k)
            if k < 9:
                self._mstep(k)
            else:
                self._out(v(out + k - 9))
        self.a.e("v_mov_b32_e32 %s, %s" % (v(out + 8), v(self.acc)))

    def dot3(self, a0, b0n, a1, b1n, a2, b2n, out):
        """out = (a0*b0 + a1*b1 + a2*b2) / 2^261; b*n: lists of 9 operand names (SGPR constants or VGPRs)."""
        first = True
        for k in range(17):
            for i in range(max(0, k - 8), min(k, 8) + 1):
                for (a, bn) in ((a0, b0n), (a1, b1n), (a2, b2n)):
                    self._mad(v(a + i), bn[k - i], first)
                    first = False
            self._reduce_terms(k)
            if k < 9:
                self._mstep(k)
            else:
                self._out(v(out + k - 9))
        self.a.e("v_mov_b32_e32 %s, %s" % (v(out + 8), v(self.acc)))


KERNEL_TAIL = """
	s_endpgm
	.section	.rodata,"a",@progbits
	.p2align	6, 0x0
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size 0
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size {kernarg}
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr {vgprs}
		.amdhsa_next_free_sgpr {sgprs}
		.amdhsa_accum_offset {accum}
		.amdhsa_reserve_vcc 1
		.amdhsa_ieee_mode 1
		.amdhsa_dx10_clamp 1
	.end_amdhsa_kernel
	.text
"""


def kernel_header(name):
    return ('\t.text\n\t.protected\t{n}\n\t.globl\t{n}\n\t.p2align\t8\n\t.type\t{n},@function\n{n}:\n').format(n=name)


def metadata(kernels):
    """kernels: list of dict(name, kernarg, vgprs, sgprs, args=[(offset, size, kind)])"""
    out = ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for k in kernels:
        out.append("  - .agpr_count:     0")
        out.append("    .args:")
        for (off, size, kind) in k["args"]:
            if kind == "global_buffer":
                out.append("      - .address_space:  global")
                out.append("        .offset:         %d" % off)
            else:
                out.append("      - .offset:         %d" % off)
            out.append("        .size:           %d" % size)
            out.append("        .value_kind:     %s" % kind)
        out += ["    .group_segment_fixed_size: 0", "    .kernarg_segment_align: 8", "    .kernarg_segment_size: %d" % k["kernarg"],
                "    .max_flat_workgroup_size: 256", "    .name:           %s" % k["name"], "    .private_segment_fixed_size: 0",
                "    .sgpr_count:     %d" % k["sgprs"], "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % k["name"],
                "    .uniform_work_group_size: 1", "    .uses_dynamic_stack: false", "    .vgpr_count:     %d" % k["vgprs"],
                "    .vgpr_spill_count: 0", "    .wavefront_size: 64"]
    out += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "", "\t.end_amdgpu_metadata"]
    return "\n".join(out) + "\n"


FILE_HEAD = '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.amdhsa_code_object_version 6\n'
