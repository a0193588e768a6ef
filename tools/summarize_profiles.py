#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_bench, gpurun_out/pmc)
into the small tracked files under profiles/:

    python tools/summarize_profiles.py r01

  profiles/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1`
  profiles/<tag>_conv5x5_pmc.txt          PMC counters of conv5x5_f16x3_kernel (separate --pmc passes)
  profiles/conv5x5_pmc.json               HBM traffic per conv launch, read by bench.py ("roofline.traffic")

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled.  The
WRITE_SIZE unit was checked here on kernels with exactly known traffic (a 256 MiB fill reads back as
262144 KiB, a 46.08 MB randn as 45000 KiB).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

stats = os.path.join(G, "prof_bench", "bench_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
    log = os.path.join(G, "prof_bench", "bench_under_rocprof.log")
    for line in open(log):
        if line.startswith("{"):
            open(os.path.join(P, f"{tag}_bench_under_rocprof.json"), "w").write(line)


def counters(name):
    path = os.path.join(G, "pmc", name, f"{name}_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


lines = []
conv = {}
calib = {}
for name in ("sq1", "sq2", "fetch", "write"):
    for kern, cs in counters(name).items():
        for c, vals in cs.items():
            mean = sum(vals) / len(vals)
            if "conv5x5_f16x3_kernel" in kern:
                conv[c] = mean
                lines.append(f"{name:6s} {c:28s} n={len(vals):2d} mean={mean:.6g} min={min(vals):.6g} max={max(vals):.6g}")
            elif c in ("FETCH_SIZE", "WRITE_SIZE") and ("act_pad" in kern or "vectorized_elementwise" in kern
                                                       or "distribution" in kern):
                calib[f"{kern[:48]} {c}"] = mean
if conv:
    L = 300
    mfma = conv.get("SQ_INSTS_MFMA", 0.0)
    busy = conv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = conv.get("GRBM_GUI_ACTIVE", 0.0)
    fetch_kib, write_kib = conv.get("FETCH_SIZE", 0.0), conv.get("WRITE_SIZE", 0.0)
    hbm = (2.0 * fetch_kib + write_kib) * 1024.0
    algo = 4.0 * (128 * L * L + 512 * 128 * 25 + 128 * L * L)
    with open(os.path.join(P, f"{tag}_conv5x5_pmc.txt"), "w") as fh:
        fh.write("# conv5x5_f16x3_kernel (default conv path), L=300, one launch; rocprofv3 --pmc passes (tools/profile_bench.sh)\n")
        fh.write("\n".join(lines) + "\n")
        if gui and busy:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum over the 1024 SIMDs
            fh.write(f"# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8) = "
                     f"{busy / (1024.0 * gui / 8.0):.3f}\n")
        fh.write(f"# SQ_INSTS_MFMA = {mfma:.0f} (expected 1444 WG * 4 waves * 8 stages * 25 taps * 8 tiles * 3 products = {1444 * 4 * 8 * 25 * 8 * 3})\n")
        fh.write(f"# HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB = {hbm / 1e6:.1f} MB; algorithmic "
                 f"{algo / 1e6:.1f} MB (activation pieces 46.1 + weight pieces 6.6 + out 46.1)\n")
        for k, v in calib.items():
            fh.write(f"# calibration: {k} = {v:.6g} KiB\n")
    json.dump({"kernel": "conv5x5_f16x3_kernel", "L": L, "hbm_bytes_per_launch": hbm,
               "fetch_size_kib": fetch_kib, "write_size_kib": write_kib,
               "algorithmic_bytes_per_launch": algo,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE half-count correction",
               "source": f"profiles/{tag}_conv5x5_pmc.txt",
               # the kernel's source at the time of the PMC passes: bench.py reports whether the tree still has it
               "kernel_source": "dmpfold2_amd/csrc/conv_f16.h",
               "kernel_source_sha256": __import__("hashlib").sha256(
                   open(os.path.join(ROOT, "dmpfold2_amd", "csrc", "conv_f16.h"), "rb").read()).hexdigest()},
              open(os.path.join(P, "conv5x5_pmc.json"), "w"), indent=1)
print(open(os.path.join(P, f"{tag}_conv5x5_pmc.txt")).read() if conv else "no PMC data")
