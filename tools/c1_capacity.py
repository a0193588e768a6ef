import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Engine, encode_aln, read_aln
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
msa = torch.from_numpy(encode_aln(read_aln(os.path.join(R, "tests/golden/PF10963.aln")))).cuda()
for cap in ((82, 252), (300, 2000), (1000, 3000)):
    eng = Engine("cuda:0", *cap); eng.set_weights(sd)
    eng.predict_device(msa, None, 0, 0); eng.sync_check()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.predict_device(msa, None, 0, 0); eng.sync_check()
    a = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        out = eng.predict_device(msa, None, 0, 0)
    eng.sync_check()
    b = (time.perf_counter() - t0) / 5
    print(f"capacity {cap}: {a * 1e3:.2f} ms per prediction with a sync after each, {b * 1e3:.2f} ms back to back", flush=True)
    eng.close()
