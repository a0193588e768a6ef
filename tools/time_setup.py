import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
t_imp = time.perf_counter() - t00
from dmpfold2_amd import synth
from dmpfold2_amd.predict import Engine, encode_aln
sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter(); e = Engine(torch.device("cuda:0"), 128, 3000); torch.cuda.synchronize(); t1 = time.perf_counter()
e.set_weights(sd); torch.cuda.synchronize(); t2 = time.perf_counter()
a = encode_aln(synth.synth_msa(82, 252, 1))
c, f = e.predict(a, None, 0, 0); torch.cuda.synchronize(); t3 = time.perf_counter()
c, f = e.predict(a, None, 0, 0); torch.cuda.synchronize(); t4 = time.perf_counter()
print(f"import {t_imp:.2f} s; ctx create {t1 - t0:.3f} s; set_weights (pack + upload) {t2 - t1:.3f} s; first predict {t3 - t2:.3f} s; second {t4 - t3:.3f} s")
e.close()
