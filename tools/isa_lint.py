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
        return src, scan_asm(device_asm(path, B.FLAGS + extra + B.per_file_flags(src)))

    bad = 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        for src, found in ex.map(one, B.SOURCES):
            for sym, ins in found:
                print("%s: %s: %s" % (src, sym, ins))
            bad += len(found)
    print("isa_lint: %d hazardous packed instruction(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
