#!/usr/bin/env python
"""GPU box: alignment FILES -> PDB FILES through the batch front end (dmpfold2_amd.batch.run_batch, one rank):
read + encode + H2D + prediction + D2H + PDB text + write, everything inside the clock except the weight load -
the whole unit of work of SURVEY 8d, next to bench.py's number whose inputs are resident in HBM.

    python tools/batch_throughput.py [targets=48] [L=300] [N=2000]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
tmp = tempfile.mkdtemp(prefix="dmp_batch_")
maker = ("import sys, multiprocessing as mp\n"
         f"sys.path.insert(0, {ROOT!r})\n"
         "from dmpfold2_amd import synth\n"
         "def one(a):\n"
         f"    synth.write_aln(a[0], synth.synth_msa({L}, {N}, seed=a[1]))\n"
         "if __name__ == '__main__':\n"
         f"    jobs = [({tmp!r} + '/t%03d.aln' % k, 5000 + k) for k in range({K})]\n"
         "    with mp.Pool(min(16, mp.cpu_count())) as pool:\n"
         "        pool.map(one, jobs, chunksize=2)\n")
open(os.path.join(tmp, "make.py"), "w").write(maker)
subprocess.run([sys.executable, os.path.join(tmp, "make.py")], check=True)

import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402
from dmpfold2_amd import synth                          # noqa: E402
from dmpfold2_amd.batch import run_batch, expand_inputs  # noqa: E402

sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
targets = expand_inputs([tmp])
from dmpfold2_amd.predict import Pipeline             # noqa: E402
t0 = time.perf_counter()
p0 = Pipeline(torch.device("cuda:0"), L, N, sd, streams=4)
torch.cuda.synchronize()
setup = time.perf_counter() - t0
p0.close()
print(json.dumps({"pipeline_setup_seconds": setup, "note": "4 contexts + weight packing, paid once per run_batch call"}), flush=True)
# warm-up job (context creation, weight packing, graph builds are per Pipeline: a second job shows the steady state
# of a long batch, the first one the cost of a short one)
for label, subset in (("first job (cold: contexts, weight packing, graphs)", targets[:8]), ("second job", targets)):
    t0 = time.perf_counter()
    n, secs, outs = run_batch(subset, os.path.join(tmp, "out_" + label.split()[0]), 10, 100, state_dict=sd,
                              streams=4, device="cuda:0")
    wall = time.perf_counter() - t0
    print(json.dumps({"job": label, "targets": n, "L": L, "N": N, "seconds_run_batch": secs, "seconds_wall": wall,
                      "structures_per_s_files_to_pdb": n / wall,
                      "structures_per_s_without_setup": n / max(wall - setup, 1e-9)}), flush=True)
