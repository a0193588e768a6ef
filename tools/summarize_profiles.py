#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_r05.sh (gpurun_out/prof_bench, prof_bench_f32, pmc) into the small
tracked files under profiles/:

    python tools/summarize_profiles.py r05

  profiles/<tag>_bench_kernel_stats_f16x3.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1 --legs f16x3`
  profiles/<tag>_bench_kernel_stats_f32.csv     the same of `--legs f32` (option precision = 1: float32 convolutions AND float32
                                                vertical GRU; the script fails if an f16 / bf16 matrix-core kernel appears in it)
  profiles/<tag>_conv5x5_pmc.txt / _f32_pmc.txt PMC counters of conv5x5_f16x3_kernel / conv5x5_maxout_kernel (separate --pmc passes)
  profiles/conv5x5_pmc.json, conv5x5_f32_pmc.json   HBM traffic per conv launch, read by bench.py (roofline.traffic, roofline_f32.traffic)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled.  The WRITE_SIZE unit was checked here on
kernels with exactly known traffic (a 256 MiB fill reads back as 262144 KiB).
"""
import collections
import csv
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

for sub, leg in (("prof_bench_bf16x3", "bf16x3"), ("prof_bench_f32", "f32"), ("prof_bench_f16x2", "f16x2")):
    stats = os.path.join(G, sub, "bench_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats_{leg}.csv"))
        for line in open(os.path.join(G, sub, "bench_under_rocprof.log")):
            if line.startswith("{"):
                open(os.path.join(P, f"{tag}_bench_under_rocprof_{leg}.json"), "w").write(line)


# the float32 leg runs the reference's arithmetic end to end: none of the split-product kernels may appear in its table
f32_stats = os.path.join(P, f"{tag}_bench_kernel_stats_f32.csv")
if os.path.exists(f32_stats):
    bad = [r["Name"] for r in csv.DictReader(open(f32_stats))
           if any(k in r["Name"] for k in ("f16x3", "bf16x6", "vgru_persist_kernel(", "vgru2_step_kernel", "act_split_kernel"))]
    print("float32 leg: f16 / bf16 matrix-core kernels in its kernel table:", bad or "none")
    assert not bad, bad


# the headline leg (precision 2): full-width operands - no split-f16 kernel (22-bit operands) may appear in its table
hl_stats = os.path.join(P, f"{tag}_bench_kernel_stats_bf16x3.csv")
if os.path.exists(hl_stats):
    bad = [r["Name"] for r in csv.DictReader(open(hl_stats))
           if any(k in r["Name"] for k in ("f16x3", "vgru_persist_kernel(", "vgru2_step_kernel", "act_split_kernel<0>"))]
    print("headline leg (precision 2): split-f16 kernels in its kernel table:", bad or "none")
    assert not bad, bad


def counters(name):
    path = os.path.join(G, "pmc", name, f"{name}_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


L = 300
# PMC passes: tools/pmc_conv.sh <conv_mode> 300 m<conv_mode> writes gpurun_out/pmc/m<mode>_{sq1,sq2,fetch,write}
for sfx, kernel, src, out_txt, out_json, expect, abytes in (
        ("m2", "conv5x5_bf16x6_kernel", "dmpfold2_amd/csrc/conv_bf16.h", f"{tag}_conv5x5_bf16_pmc.txt", "conv5x5_bf16_pmc.json",
         1444 * 4 * 8 * 25 * 8 * 6, 6.0),
        ("m0", "conv5x5_f16x3_kernel", "dmpfold2_amd/csrc/conv_f16.h", f"{tag}_conv5x5_pmc.txt", "conv5x5_pmc.json",
         1444 * 4 * 8 * 25 * 8 * 3, 4.0),
        ("m1", "conv5x5_maxout_kernel", "dmpfold2_amd/csrc/trunk.hip", f"{tag}_conv5x5_f32_pmc.txt", "conv5x5_f32_pmc.json",
         1444 * 4 * 64 * 25 * 8, 4.0)):
    lines, conv, calib = [], {}, {}
    for name in ("sq1", "sq2", "fetch", "write"):
        for kern, cs in counters(sfx + "_" + name).items():
            for c, vals in cs.items():
                mean = sum(vals) / len(vals)
                if kernel in kern:
                    conv[c] = mean
                    lines.append(f"{name:6s} {c:28s} n={len(vals):2d} mean={mean:.6g} min={min(vals):.6g} max={max(vals):.6g}")
                elif c in ("FETCH_SIZE", "WRITE_SIZE") and ("vectorized_elementwise" in kern or "FillFunctor" in kern):
                    calib[f"{kern[:60]} {c}"] = mean
    if not conv:
        print("no PMC data for", kernel)
        continue
    mfma = conv.get("SQ_INSTS_MFMA", 0.0)
    busy = conv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = conv.get("GRBM_GUI_ACTIVE", 0.0)
    fetch_kib, write_kib = conv.get("FETCH_SIZE", 0.0), conv.get("WRITE_SIZE", 0.0)
    hbm = (2.0 * fetch_kib + write_kib) * 1024.0
    # algorithmic bytes per launch: input activations (bf16x6: three bf16 pieces = 6 B per value; f16x3: two f16 pieces = 4 B;
    # f32: 4 B) + weights in the same form + the float32 output
    algo = abytes * (128 * L * L + 512 * 128 * 25) + 4.0 * 128 * L * L
    with open(os.path.join(P, out_txt), "w") as fh:
        fh.write(f"# {kernel}, L=300, 16 launches of one trunk pass; rocprofv3 --pmc passes (tools/pmc_conv.sh)\n")
        fh.write("\n".join(lines) + "\n")
        if gui and busy:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum over the 1024 SIMDs
            fh.write(f"# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8) = "
                     f"{busy / (1024.0 * gui / 8.0):.3f}\n")
        fh.write(f"# SQ_INSTS_MFMA = {mfma:.0f} per launch (expected {expect})\n")
        fh.write(f"# HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB = {hbm / 1e6:.1f} MB; algorithmic {algo / 1e6:.1f} MB\n")
        for k, v in calib.items():
            fh.write(f"# calibration: {k} = {v:.6g} KiB\n")
    json.dump({"kernel": kernel, "L": L, "hbm_bytes_per_launch": hbm, "fetch_size_kib": fetch_kib,
               "write_size_kib": write_kib, "algorithmic_bytes_per_launch": algo,
               "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE half-count correction",
               "source": f"profiles/{out_txt}",
               # the kernel's source at the time of the PMC passes: bench.py reports whether the tree still has it
               "kernel_source": src,
               "kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, src), "rb").read()).hexdigest()},
              open(os.path.join(P, out_json), "w"), indent=1)
    print(open(os.path.join(P, out_txt)).read())
