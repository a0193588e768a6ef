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
tmp = os.environ.get("DMP_BT_DIR") or tempfile.mkdtemp(prefix="dmp_batch_")
maker = ("import sys, multiprocessing as mp\n"
         f"sys.path.insert(0, {ROOT!r})\n"
         "from dmpfold2_amd import synth\n"
         "def one(a):\n"
         f"    synth.write_aln(a[0], synth.synth_msa({L}, {N}, seed=a[1]))\n"
         "if __name__ == '__main__':\n"
         f"    jobs = [({tmp!r} + '/t%03d.aln' % k, 5000 + k) for k in range({K})]\n"
         "    with mp.Pool(min(16, mp.cpu_count())) as pool:\n"
         "        pool.map(one, jobs, chunksize=2)\n")
if not os.environ.get("DMP_BT_MODE"):
    open(os.path.join(tmp, "make.py"), "w").write(maker)
    subprocess.run([sys.executable, os.path.join(tmp, "make.py")], check=True)

import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402
from dmpfold2_amd import synth                          # noqa: E402
from dmpfold2_amd.batch import run_batch, expand_inputs  # noqa: E402

sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
targets = expand_inputs([tmp])
# Every measurement in its OWN process: which pool streams (hardware queues) a pipeline's engines get depends on how
# many streams the process created before, and the second set of four is 15 % slower on this runtime
# (tools/pipeline_order.py) - a long-lived service would see the first set, as bench.py does.
mode = os.environ.get("DMP_BT_MODE")
if mode is None:
    for m in ("files", "resident"):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(K), str(L), str(N)],
                       env=dict(os.environ, DMP_BT_MODE=m, DMP_BT_DIR=tmp), check=True)
    sys.exit(0)
from dmpfold2_amd.predict import Pipeline, encode_aln, read_aln   # noqa: E402
if mode == "files":
    t0 = time.perf_counter()
    n, secs, outs = run_batch(targets, os.path.join(tmp, "out"), 10, 100, state_dict=sd, streams=4, device="cuda:0")
    wall = time.perf_counter() - t0
    print(json.dumps({"job": "alignment files -> PDB files, one run_batch call in a fresh process (pipeline set-up included)",
                      "targets": n, "L": L, "N": N, "seconds_wall": wall, "structures_per_s_files_to_pdb": n / wall}), flush=True)
else:
    msas = [torch.from_numpy(encode_aln(read_aln(a))).to("cuda:0") for a, _ in targets]
    t0 = time.perf_counter()
    pipe = Pipeline(torch.device("cuda:0"), L, N, sd, streams=4)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    outs = pipe.run(msas, 10, 100)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    pipe.close()
    print(json.dumps({"job": "same targets resident in HBM, Pipeline.run in a fresh process (cold pipeline)", "targets": len(msas),
                      "pipeline_setup_seconds": setup, "seconds_wall": wall, "structures_per_s": len(msas) / wall}), flush=True)
