#!/usr/bin/env python
"""GPU box: does the throughput of a Pipeline depend on how many were created before it in the process (which pool
streams / hardware queues its engines get)?  Creates pipelines one after the other and times 16 resident targets on each."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Pipeline, encode_aln
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
msas = [torch.from_numpy(encode_aln(synth.synth_msa(300, 2000, seed=50 + i))).to(dev) for i in range(16)]
keep = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    pipe = Pipeline(dev, 300, 2000, sd, streams=4)
    pipe.run(msas[:4], 10, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(msas, 10, 100)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pipeline #{k + 1}: 16 targets in {dt:.2f} s = {16 / dt:.2f} structures/s; engine streams "
          f"{[hex(e.stream().value if hasattr(e.stream(), 'value') else int(e.stream())) for e in pipe.engines]}", flush=True)
    if os.environ.get("KEEP") == "1":
        keep.append(pipe)
    else:
        pipe.close()
