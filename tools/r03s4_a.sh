#!/bin/bash
# GPU box, round 3 session 4, call A: GPU suite with the hardware-gate sequence GRU / LDS-resident inverse iteration /
# DMP_MAX_L = 2048, single-target latency, a first prediction above 1280 columns, eigen-gap screen at L = 1344
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s4a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/single_trace.py run 300 2000 10 100 5 > $O/single.txt 2>&1; tail -3 $O/single.txt
timeout 300 python tools/gpu_diag.py --time 300 2000 > $O/diag_time.txt 2>&1; grep -E "gru_bidir|eigh|coords_from|predict" $O/diag_time.txt
timeout 600 python tools/screen_eig_gaps.py --L 1344 --N 64 --seeds 0-9 --n 0 > $O/gaps_1344.txt 2> $O/gaps_1344.err; cut -c1-60 $O/gaps_1344.txt | head -3; tail -3 $O/gaps_1344.err
timeout 900 python - > $O/big.txt 2>&1 <<'PY'
import time, numpy as np, torch
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Engine, encode_aln
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
for L, N in ((1536, 64), (2048, 128)):
    eng = Engine("cuda:0", L, N); eng.set_weights(sd)
    msa = torch.from_numpy(encode_aln(synth.synth_msa(L, N, 0))).to("cuda:0")
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        c, f = eng.predict_device(msa, None, 1, 10); eng.sync_check()
        dt = time.perf_counter() - t
    ca = c[:, 1].cpu().numpy(); d = np.linalg.norm(ca[1:] - ca[:-1], axis=1)
    print(f"L={L} N={N} n=1 m=10: {dt*1e3:.1f} ms finite={bool(torch.isfinite(c).all() and torch.isfinite(f).all())} "
          f"conf[{float(f.min()):.3f},{float(f.max()):.3f}] bond mean {d.mean():.3f} free GB {torch.cuda.mem_get_info()[0]/1e9:.1f}", flush=True)
    eng.close(); del eng
PY
cat $O/big.txt | tail -4
