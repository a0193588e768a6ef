"""Build libdmpfold_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m dmpfold2_amd.build [--force]

One object per .hip file (compiled in parallel), linked into an in-tree shared library so it
travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libdmpfold_hip.so")
SOURCES = ["api.hip", "pipeline.hip", "gemm.hip", "msa.hip", "dca.hip", "gru.hip", "vgru.hip", "vgru_f32.hip", "vgru_x3.hip", "trunk.hip", "train.hip", "mds.hip",
           "coords.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# coords.hip: no SLP vectoriser = no packed-f32 instructions.  The vectoriser turns the cross products of the
# backbone kernel into v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0], which returns 0 in lanes 48..63 when
# f16 / bf16 MFMA waves share the SIMD (tools/isa_lint.py guards every kernel against it).
# gru.hip: the same for the matrix-vector loop of seq_gru_kernel (v_pk_fma_f32 ... op_sel:[0,1,0] appeared when the gate
# evaluation was spread over the lanes).
PER_FILE_FLAGS = {"coords.hip": ["-fno-slp-vectorize"], "gru.hip": ["-fno-slp-vectorize"]}


def per_file_flags(src: str) -> list:
    """library flags of one source + DMP_FLAGS_<FILE> from the environment (tuning experiments)"""
    name = os.path.basename(src)
    env = os.environ.get("DMP_FLAGS_" + name.split(".")[0].upper(), "").split()
    return PER_FILE_FLAGS.get(name, []) + env


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "conv_bf16.h", "conv_f16.h", "vgru.h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "dmpfold_hip.h"))
    extra = os.environ.get("DMP_EXTRA_HIPCC_FLAGS", "").split()      # tuning experiments (-DVG_CH=1 ...)
    lib = os.environ.get("DMP_LIB_OUT", LIB)
    force = force or bool(extra)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s, __file__] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + extra + per_file_flags(s) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
