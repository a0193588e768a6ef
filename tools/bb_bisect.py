#!/usr/bin/env python
"""Build container: instruction-level bisection of the backbone-kernel hazard.

Takes the device assembly of dmpfold2_amd/csrc/coords.hip (tools/_bin/bbv/v0.s, written by
tools/bb_hazard_variants.sh), and writes variants of backbone_kernel in which packed VALU instructions
(v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32) are replaced by the equivalent pair of
one-float instructions (through two spare VGPRs, so overlapping operands are safe):

    all_scalar          every packed instruction replaced (must behave like the -fno-slp-vectorize build)
    only_NNN            every packed instruction replaced EXCEPT number NNN
    class_<name>        only the instructions of one operand shape keep their packed form

tools/bb_hazard then tells which single instructions / shapes are enough to corrupt lanes 48-63 beside
the f16 convolution.  Usage: python tools/bb_bisect.py [singles|classes|list]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BBV = os.path.join(ROOT, "tools", "_bin", "bbv")
LL = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN3dmp15backbone_kernelEPKfS1_iffPfS2_"
T0, T1 = 36, 37                       # spare VGPRs (the kernel uses v0..v35)

PK = re.compile(r"^\s*(v_pk_(?:mul_f32|add_f32|fma_f32|mov_b32))\s+(.*)$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")


def half(opnd, sel):
    """the 32-bit half `sel` of a packed source operand"""
    m = re.match(r"([vs])\[(\d+):(\d+)\]", opnd)
    if m:
        return "%s%d" % (m.group(1), int(m.group(2)) + sel)
    return opnd                        # inline constant: both halves read the constant


def scalarise(line):
    m = PK.match(line)
    op, rest = m.group(1), m.group(2)
    mods = dict((k, [int(x) for x in v.split(",")]) for k, v in MOD.findall(rest))
    opnds = [o.strip() for o in MOD.sub("", rest).strip().rstrip(",").split(",") if o.strip()]
    dst, srcs = opnds[0], opnds[1:]
    d = int(re.match(r"v\[(\d+):", dst).group(1))
    n = len(srcs)
    op_sel = mods.get("op_sel", [0] * n)
    op_sel_hi = mods.get("op_sel_hi", [1] * n)
    neg_lo = mods.get("neg_lo", [0] * n)
    neg_hi = mods.get("neg_hi", [0] * n)
    out = []
    if op == "v_pk_mov_b32":
        out.append("\tv_mov_b32_e32 v%d, %s" % (T0, half(srcs[0], op_sel[0])))
        out.append("\tv_mov_b32_e32 v%d, %s" % (T1, half(srcs[1], op_sel[1])))
    else:
        one = {"v_pk_mul_f32": "v_mul_f32_e64", "v_pk_add_f32": "v_add_f32_e64", "v_pk_fma_f32": "v_fma_f32"}[op]
        for t, sel, neg in ((T0, op_sel, neg_lo), (T1, op_sel_hi, neg_hi)):
            args = []
            for i, s in enumerate(srcs):
                h = half(s, sel[i])
                if neg[i]:
                    h = "-" + h if not h.startswith("-") else h[1:]
                args.append(h)
            out.append("\t%s v%d, %s" % (one, t, ", ".join(args)))
    out.append("\tv_mov_b32_e32 v%d, v%d" % (d, T0))
    out.append("\tv_mov_b32_e32 v%d, v%d" % (d + 1, T1))
    return out


def shape(line):
    m = PK.match(line)
    s = re.sub(r"v\[\d+:\d+\]", "V", m.group(2))
    s = re.sub(r"s\[\d+:\d+\]", "S", s)
    return (m.group(1) + " " + s).replace(" ", "_").replace(",", "").replace(":", "").replace("[", "").replace("]", "")


def load():
    lines = open(os.path.join(BBV, "v0.s")).read().split("\n")
    a = lines.index(KERNEL + ": ; @" + KERNEL) if (KERNEL + ": ; @" + KERNEL) in lines else \
        next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    k = next(i for i, l in enumerate(lines) if l.strip() == ".amdhsa_kernel " + KERNEL)
    for i in range(k, k + 40):
        if ".amdhsa_next_free_vgpr" in lines[i]:
            assert int(lines[i].split()[-1]) <= T0, lines[i]
            lines[i] = "\t\t.amdhsa_next_free_vgpr 40"
        if ".amdhsa_accum_offset" in lines[i]:
            lines[i] = "\t\t.amdhsa_accum_offset 40"
    pk = [i for i in range(a, b) if PK.match(lines[i])]
    return lines, pk


def write(name, lines, pk, keep):
    out = list(lines)
    for n, i in enumerate(pk):
        if n not in keep:
            out[i] = "\n".join(scalarise(lines[i]))
    s = os.path.join(BBV, name + ".s")
    open(s, "w").write("\n".join(out))
    o = s[:-2] + ".o"
    subprocess.check_call([LL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LL + "/ld.lld", "-shared", o, "-o", s[:-2] + ".hsaco"])
    os.remove(o)
    os.remove(s)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "classes"
    lines, pk = load()
    if what == "list":
        for n, i in enumerate(pk):
            print(n, lines[i].strip())
        return
    write("all_scalar", lines, pk, set())
    if what == "singles":
        for n in range(len(pk)):
            write("only_%03d" % n, lines, pk, {n})
        print("wrote all_scalar and %d single-instruction variants" % len(pk))
    elif what == "classes":
        classes = {}
        for n, i in enumerate(pk):
            classes.setdefault(shape(lines[i]), set()).add(n)
        for c, keep in sorted(classes.items()):
            write("class_" + c, lines, pk, keep)
            print("class_%s: %d instructions" % (c, len(keep)))
    else:
        keep = set(int(x) for x in what.split(","))
        write("keep_" + what.replace(",", "_"), lines, pk, keep)


if __name__ == "__main__":
    main()
