#!/usr/bin/env python
"""Build-time guard against the packed-f32 hazard of DESIGN section 6.

Measured on MI355X (tools/pk_hazard.hip, tools/bb_hazard.hip + tools/bb_bisect.py): the packed-f32 VALU
instructions v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose LOW result takes src1 from the HIGH register
of its pair while src0 comes from the low register, i.e. op_sel[0] = 0 and op_sel[1] = 1, e.g.

    v_pk_mul_f32 v[0:1], v[4:5], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]

return 0 for that product in lanes 48..63 in about 1 % of their executions while waves of an f16 / bf16
MFMA kernel (conv5x5_f16x3_kernel) share the SIMD.  Alone the instruction is always right (0 of 6.4e6),
so no test without a co-running convolution can see it.  Not affected (0 of 6.4e6 beside the convolution):
default selection, op_sel_hi:[1,0], cross selection on src0 (op_sel:[1,0] op_sel_hi:[0,1]) or on src2,
op_sel:[1,1], v_pk_mov_b32.

This script compiles every translation unit of the library to gfx950 assembly with the library's flags
and fails (exit 1) if any kernel contains a packed-f32 instruction with op_sel[1] = 1 (a superset of the
measured condition).  It is run by tests/test_host_cpu.py and by __graft_entry__.build().

Second rule (round 5).  MFMAs written as INLINE ASSEMBLY (the vertical GRU's weight-stationary kernels read their A
operand straight from AGPRs, which the builtins cannot express) are invisible to the compiler's hazard recogniser.
Found on the GPU: the compiler sank the zero-initialisation of an accumulator (v_mov_b64) to directly in front of
its first assembly MFMA, and that MFMA accumulated onto the stale register pair (vgru_f32.hip, first build: columns
16..31, rows 4g and 4g+1 wrong by 2e-2).  The kernels now pin their accumulators behind a statement with wait states;
this lint checks the result: no VALU instruction may write a VGPR that an assembly MFMA reads (srcA, srcB or srcC)
fewer than two wait states ahead of it.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PK = re.compile(r"^\s*(v_pk_\w+)\s+(.*)$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")


def hazardous(line):
    """True for v_pk_{mul,add,fma}_f32 whose low result reads the high half of src1."""
    m = PK.match(line)
    if not m or m.group(1) not in ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"):
        return False
    rest = m.group(2).split(";")[0]
    mods = dict((k, [int(x) for x in v.split(",")]) for k, v in MOD.findall(rest))
    op_sel = mods.get("op_sel", [0, 0, 0])
    return len(op_sel) > 1 and op_sel[1] == 1


def scan_asm(text):
    """-> list of (symbol, instruction)"""
    found, sym = [], "?"
    for line in text.split("\n"):
        m = LABEL.match(line)
        if m and not line.startswith(".L"):
            sym = m.group(1)
        if hazardous(line):
            found.append((sym, line.strip()))
    return found


REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")
INS = re.compile(r"^\s*([a-z_][a-z0-9_]*)\s*(.*)$")
NOT_VGPR_WRITERS = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_mfma", "v_smfmac", "v_accvgpr_read", "v_nop")


def vregs(text):
    out = set()
    for lo, hi, one in REG.findall(text):
        if one:
            out.add(int(one))
        else:
            out.update(range(int(lo), int(hi) + 1))
    return out


def scan_asm_mfma_hazards(text, need=2):
    """-> list of (symbol, VALU instruction, MFMA) where a VALU write of an operand of an inline-assembly MFMA is fewer than
    `need` wait states ahead of it"""
    found, sym, in_asm = [], "?", False
    hist = []                                            # (instruction text, VGPRs written by a VALU or None, wait states it provides)
    for raw in text.split("\n"):
        line = raw.split(";")[0] if not raw.strip().startswith(";;#") else raw
        m = LABEL.match(line)
        if m and not line.startswith(".L"):
            sym, hist = m.group(1), []
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        mi = INS.match(line)
        if not mi or line.strip().startswith(".") or line.strip().endswith(":"):
            continue
        op, rest = mi.group(1), mi.group(2)
        if op == "s_nop":
            hist.append((line.strip(), None, int(rest.strip() or 0) + 1))
            continue
        if in_asm and op.startswith("v_mfma"):
            ops = [o.strip() for o in rest.split(",")]
            reads = set()
            for o in ops[1:4]:
                reads |= vregs(o)
            waited = 0
            for txt, writes, ws in reversed(hist[-6:]):
                if waited >= need:
                    break
                if writes is not None and writes & reads:
                    found.append((sym, txt, line.strip()))
                    break
                waited += ws
        writes = None
        if op.startswith("v_") and not op.startswith(NOT_VGPR_WRITERS):
            writes = vregs(rest.split(",")[0])
        hist.append((line.strip(), writes, 1))
        del hist[:-8]
    return found


def device_asm(src, flags):
    from dmpfold2_amd import build as B
    cmd = [B._hipcc()] + [f for f in flags if f not in ("-fPIC", "-c")] + ["-S", "--cuda-device-only", src, "-o", "-"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
    return r.stdout


def main():
    from dmpfold2_amd import build as B
    from concurrent.futures import ThreadPoolExecutor

    def one(src):
        path = os.path.join(B.CSRC, src)
        # exactly the flags of the shipped objects, tuning extras included (build.py)
        extra = os.environ.get("DMP_EXTRA_HIPCC_FLAGS", "").split()
        text = device_asm(path, B.FLAGS + extra + B.per_file_flags(src))
        return src, scan_asm(text), scan_asm_mfma_hazards(text)

    bad = bad2 = 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        for src, found, found2 in ex.map(one, B.SOURCES):
            for sym, ins in found:
                print("%s: %s: %s" % (src, sym, ins))
            for sym, valu, mfma in found2:
                print("%s: %s: VALU write too close to an assembly MFMA that reads it: %s  ->  %s" % (src, sym, valu, mfma))
            bad += len(found)
            bad2 += len(found2)
    print("isa_lint: %d hazardous packed instruction(s)" % bad)
    print("isa_lint: %d VALU write(s) within two wait states of an inline-assembly MFMA that reads them" % bad2)
    return 1 if (bad or bad2) else 0


if __name__ == "__main__":
    sys.exit(main())
