#!/usr/bin/env python
"""Build-time lint for vgru_persist_kernel: the state loads are inline assembly the compiler does not track, so no
instruction may read or write their destination registers between the load and the counted wait that covers it (a
register copy scheduled in between would copy data that has not landed: timing-dependent garbage).

    python tools/check_inflight_regs.py file.s        (assembly of dmpfold2_amd/csrc/vgru.hip)
"""
import re
import sys

NAME = "_ZN3dmp19vgru_persist_kernelENS_7VStaticEPKNS_9VGroupRecEPNS_6VPSyncEPiiii"


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(path):
    s = open(path).read()
    i0 = s.index(NAME + ":")
    lines = s[i0:s.index(".Lfunc_end", i0)].split("\n")
    inflight, issues, in_asm = [], [], False
    for n, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if in_asm and t.startswith("global_load"):
            inflight.append(regs(t.split()[1].rstrip(",")))
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m and in_asm:
            k = int(m.group(1))
            inflight = inflight[len(inflight) - k:] if 0 < k < len(inflight) else ([] if k == 0 else inflight)
            continue
        used = set()
        for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
            used |= regs(tk)
        fl = set().union(*inflight) if inflight else set()
        if used & fl and not t.startswith("global_store"):
            issues.append((n, t, sorted(used & fl)[:4]))
    for n, t, r in issues[:12]:
        print("  line %d: %s   touches in-flight v%s" % (n, t, r))
    print("check_inflight_regs: %d instruction(s) touch registers of loads in flight" % len(issues))
    return 1 if issues else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
