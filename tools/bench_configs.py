#!/usr/bin/env python
"""GPU box: every configuration of BASELINE.json on one MI355X, each with a roofline fraction of
its dominant kernel - in the three arithmetic settings (option "precision" 2: exact 3 x bf16 pieces, the headline; 1: f32 MFMA; 0: split f16).

    python tools/bench_configs.py [--c4 256] [--skip-c5-f32]  > profiles/r05_configs.jsonl

C1  PF10963 (L=82, N=252), -n 0 -m 0             single target, latency
C2  L=200, N=1000, 10 + 100                       single target latency; 12 targets through the 4-engine scheduler
NS  L=300, N=2000, 10 + 100                       (bench.py's workload) single-target latency
C3  L=500, N=5000 -> 3000, 30 + 200               single target, latency
C4  256 targets, L uniform in [100, 300], N=2000  4-engine scheduler, structures/s
C5  L=1000, N=2000, 100 + 1000                    single target, latency
One JSON line per (configuration, precision).  Per line: seconds per structure; the convolution's chip time per launch
from the HIP events recorded on the launching stream around every launch (dmp_profile_*: union of the intervals /
launches, as bench.py); `frac` = 2*128*512*25*L^2 FLOP / that time against the f16 dense peak / 3 (precision 0: three
f16 MFMA products per float32 product) or the 157.3 TFLOP/s f32 MFMA peak (precision 1); the tile quantisation of the
launch - (ceil(L/16))^2 pixel tiles x 4 channel splits workgroups on 512 slots (2 per CU) - and the fraction of the
tile area that is real pixels.  Weights: synthetic seed 0; inputs: synth_msa.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see bench.py
from dmpfold2_amd import synth, _lib                 # noqa: E402
from dmpfold2_amd.predict import Engine, Pipeline, encode_aln, read_aln   # noqa: E402

PEAK = {0: 2500.0 / 3.0, 1: 157.3, 2: 2500.0 / 6.0}  # TFLOP/s the algorithmic float32 FLOPs are priced against
lib = _lib.load()


def union_ms(iv):
    tot, end = 0.0, -1e30
    for a, b in sorted(iv):
        if b > end:
            tot += b - max(a, end)
            end = b
    return tot


def conv_intervals(engines, cap):
    iv = []
    for e in engines:
        a, b, n = (C.c_float * cap)(), (C.c_float * cap)(), C.c_int()
        _lib.check(lib.dmp_profile_conv_intervals(e.ctx, engines[0].ctx, a, b, cap, C.byref(n)))
        iv += [(a[i], b[i]) for i in range(n.value)]
    return iv


def quantisation(L, prec):
    t = math.ceil(L / 16)
    # the split-product kernels use 8 x 16 pixel tiles where 16 x 16 would leave half the CUs idle (round 6: L <= 80)
    bands = 2 if (prec != 1 and 8 * t * t <= 256) else 1
    rows = math.ceil(L / (16 // bands))
    wgs = 4 * t * rows
    slots = 512 if prec != 0 or bands == 1 else 768          # workgroups resident per chip (f16 8 x 16: three per CU)
    return {"tile_rows": 16 // bands, "pixel_tiles": t * rows, "workgroups": wgs, "rounds_of_slots": wgs / float(slots),
            "real_pixels_per_tile_area": round(L * L / float(t * rows * 256 // bands), 3)}


def single(eng, msa, n, m, reps, warm=True):
    d = torch.from_numpy(msa).cuda()
    L = msa.shape[1]
    if warm:
        eng.predict_device(d, None, min(n, 1), min(m, 5))
        eng.sync_check()
    cap = 16 * (n + 1) * reps + 16
    _lib.check(lib.dmp_profile_enable(eng.ctx, 1, cap))
    t0 = time.perf_counter()
    for _ in range(reps):
        out = eng.predict_device(d, None, n, m)
    eng.sync_check()
    dt = (time.perf_counter() - t0) / reps
    iv = conv_intervals([eng], cap)
    _lib.check(lib.dmp_profile_enable(eng.ctx, 0, 0))
    assert bool(torch.isfinite(out[0]).all())
    ms = union_ms(iv) / max(1, len(iv))
    flop = 2.0 * 128 * 512 * 25 * L * L
    return dt, ms, flop / (ms * 1e-3) / 1e12 if ms else 0.0, len(iv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4", type=int, default=256)
    ap.add_argument("--skip-c5-f32", action="store_true", help="C5 in float32 takes about 40 s of GPU time")
    ap.add_argument("--scheduler-only", action="store_true", help="skip the single-target latencies")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}

    def report(name, **kw):
        print(json.dumps({"config": name, **kw}), flush=True)

    if not args.scheduler_only:
        eng = Engine(dev, 1000, 3000)
        eng.set_weights(sd)
        pf = encode_aln(read_aln(os.path.join(ROOT, "tests", "golden", "PF10963.aln")))
        cases = [("C1 PF10963 L=82 N=252 n=0 m=0", pf, 0, 0, 5),
                 ("C2 L=200 N=1000 10+100", encode_aln(synth.synth_msa(200, 1000, seed=1)), 10, 100, 3),
                 ("NS L=300 N=2000 10+100", encode_aln(synth.synth_msa(300, 2000, seed=1)), 10, 100, 3),
                 ("C3 L=500 N=5000->3000 30+200", encode_aln(synth.synth_msa(500, 3000, seed=1)), 30, 200, 2),
                 ("C5 L=1000 N=2000 100+1000", encode_aln(synth.synth_msa(1000, 2000, seed=1)), 100, 1000, 1)]
        for prec in (2, 1, 0):
            eng.set_option("precision", prec)
            for name, msa, n, m, reps in cases:
                if prec == 1 and name.startswith("C5") and args.skip_c5_f32:
                    continue
                L = msa.shape[1]
                dt, ms, tf, cnt = single(eng, msa, n, m, reps)
                report(name, precision=prec, seconds_per_structure=dt, structures_per_s_single_stream=1.0 / dt,
                       conv_chip_ms_per_launch=ms, conv_launches_timed=cnt, conv_tflops=tf, peak_tflops=PEAK[prec],
                       frac=tf / PEAK[prec], conv_share_of_time=ms * 1e-3 * 16 * (n + 1) / dt, **quantisation(L, prec))
        eng.close()

    # ---- through the 4-engine scheduler
    pipe = Pipeline(dev, 300, 2000, sd, streams=4)
    c2 = [torch.from_numpy(encode_aln(synth.synth_msa(200, 1000, seed=10 + i))).to(dev) for i in range(12)]
    rng = np.random.default_rng(0)
    lens = rng.integers(100, 301, size=args.c4)
    c4 = [torch.from_numpy(encode_aln(synth.synth_msa(int(L), 2000, seed=1000 + i))).to(dev) for i, L in enumerate(lens)]
    order = sorted(range(len(c4)), key=lambda i: -int(lens[i]))          # longest first, as shard.py deals them
    for prec in (2, 1, 0):
        pipe.set_option("precision", prec)
        for name, tg, flops in (("C2 x12 through the 4-engine scheduler", c2, [2.0 * 128 * 512 * 25 * 200 * 200] * 12),
                                (f"C4 {args.c4} targets L in [100,300] N=2000 10+100 through the 4-engine scheduler",
                                 [c4[i] for i in order], [2.0 * 128 * 512 * 25 * float(lens[i]) ** 2 for i in order])):
            if prec == 1 and name.startswith("C4"):
                tg, flops = tg[:64], flops[:64]          # (the 64 longest: a quarter of the job is enough for the rate)
            pipe.run(tg[:3], 10, 100)
            torch.cuda.synchronize()
            cap = 16 * 11 * (len(tg) // 4 + 8)
            for e in pipe.engines:
                _lib.check(lib.dmp_profile_enable(e.ctx, 1, cap))
            t0 = time.perf_counter()
            pipe.run(tg, 10, 100)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            iv = conv_intervals(pipe.engines, cap)
            for e in pipe.engines:
                _lib.check(lib.dmp_profile_enable(e.ctx, 0, 0))
            pipe.sync_check()
            u = union_ms(iv)
            tf = 176.0 * sum(flops) / (u * 1e-3) / 1e12 if u else 0.0
            report(name, precision=prec, targets=len(tg), seconds=dt, structures_per_s=len(tg) / dt,
                   mean_L=float(np.mean([t.shape[1] for t in tg])), conv_launches_timed=len(iv),
                   conv_busy_share_of_time=u * 1e-3 / dt, conv_tflops_while_busy=tf, peak_tflops=PEAK[prec], frac=tf / PEAK[prec])
    pipe.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
