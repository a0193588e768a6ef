#!/usr/bin/env python
"""GPU box: host time of the scheduler's dmp_predict_issue_unit calls by unit kind (DMP_PUMP_TIMING)."""
import os, sys, time
os.environ["DMP_PUMP_TIMING"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dmpfold2_amd import synth, predict
from dmpfold2_amd.predict import Pipeline, encode_aln
L, N = 300, 2000
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, seed=i))).to(dev) for i in range(8)]
pipe = Pipeline(dev, L, N, sd, streams=4)
pipe.run(msas[:4], 10, 100); torch.cuda.synchronize()
predict._PUMP_TIMING.clear()
t0 = time.perf_counter()
pipe.run(msas * 2, 10, 100); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{16 / dt:.2f} structures/s", file=sys.stderr)
predict.pump_timing_report()
pipe.close()
