#!/usr/bin/env python
"""GPU box: time the other configurations of BASELINE.json on one MI355X.

    python tools/bench_configs.py [--c4 256]

C1  PF10963 (L=82, N=252), -n 0 -m 0             single target, latency
C2  L=200, N=1000, 10 + 100                       single target, latency and 4-stream throughput
NS  L=300, N=2000, 10 + 100                       (bench.py's workload) single-target latency
C3  L=500, N=5000 -> 3000, 30 + 200               single target, latency
C4  256 targets, L uniform in [100, 300], N=2000  4-stream scheduler, structures/s
C5  L=1000, N=2000, 100 + 1000                    single target, latency
Prints one JSON line per configuration.  Weights: synthetic seed 0; inputs: synth_msa.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see bench.py
from dmpfold2_amd import synth                       # noqa: E402
from dmpfold2_amd.predict import Engine, Pipeline, encode_aln, read_aln   # noqa: E402


def single(eng, msa, n, m, reps=2):
    d = torch.from_numpy(msa).cuda()
    eng.predict_device(d, None, n, m)
    eng.sync_check()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = eng.predict_device(d, None, n, m)
    eng.sync_check()
    assert bool(torch.isfinite(out[0]).all())
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4", type=int, default=256)
    ap.add_argument("--scheduler-only", action="store_true", help="skip the single-target latencies")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}

    def report(name, **kw):
        print(json.dumps({"config": name, **kw}), flush=True)

    pf = os.path.join(ROOT, "tests", "golden", "PF10963.aln")
    if args.scheduler_only:
        return scheduler_configs(args, dev, sd, report)
    eng = Engine(dev, 1000, 3000)
    eng.set_weights(sd)
    t = single(eng, encode_aln(read_aln(pf)), 0, 0, reps=5)
    report("C1 PF10963 L=82 N=252 n=0 m=0", seconds_per_structure=t)
    for name, L, N, n, m, reps in [("C2 L=200 N=1000 10+100", 200, 1000, 10, 100, 3),
                                   ("NS L=300 N=2000 10+100", 300, 2000, 10, 100, 3),
                                   ("C3 L=500 N=5000->3000 30+200", 500, 3000, 30, 200, 2),
                                   ("C5 L=1000 N=2000 100+1000", 1000, 2000, 100, 1000, 1)]:
        msa = encode_aln(synth.synth_msa(L, N, seed=1))
        t = single(eng, msa, n, m, reps=reps)
        report(name, seconds_per_structure=t, structures_per_s_single_stream=1.0 / t)
    eng.close()
    return scheduler_configs(args, dev, sd, report)


def scheduler_configs(args, dev, sd, report):
    pipe = Pipeline(dev, 300, 2000, sd, streams=4)
    tg = [torch.from_numpy(encode_aln(synth.synth_msa(200, 1000, seed=10 + i))).to(dev) for i in range(12)]
    pipe.run(tg[:3], 10, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(tg, 10, 100)
    torch.cuda.synchronize()
    report("C2 x12 through the 4-stream scheduler", structures_per_s=12 / (time.perf_counter() - t0))

    rng = np.random.default_rng(0)
    lens = rng.integers(100, 301, size=args.c4)
    tg = [torch.from_numpy(encode_aln(synth.synth_msa(int(L), 2000, seed=1000 + i))).to(dev)
          for i, L in enumerate(lens)]
    order = sorted(range(len(tg)), key=lambda i: -int(lens[i]))          # longest first, as shard.py deals them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run([tg[i] for i in order], 10, 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.sync_check()
    report(f"C4 {args.c4} targets L in [100,300] N=2000 10+100, one GPU", seconds=dt,
           structures_per_s=args.c4 / dt, mean_L=float(lens.mean()))
    pipe.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
