"""Host-side mirror of the reference's prediction interface on the HIP engine.

`aln_to_coords` and `run_dmpfold` keep the names, argument meaning, return types and
error behaviour of the reference (dmpfold/predict.py:74-158 and 160-208); all
arithmetic happens in libdmpfold_hip.so through the C ABI of include/dmpfold_hip.h.
PyTorch is used only for device memory and the current stream.

Differences from the reference, all additive or forced by the environment:
  * there is no CPU path: `device` must name a GPU ("cuda", "cuda:1", an int, or a
    torch.device).  The default is "cuda" (the reference defaults to "cpu");
  * `aln_to_coords` synchronises with the GPU before it returns and checks the engine's device-side
    fault word: a prediction whose activations left the range of the default split-f16 convolution is
    re-run with the range-free bf16 split (a note goes to stderr), any other fault raises;
  * packed weights are cached per (device, weights file) instead of being rebuilt on
    every call (the reference constructs and loads a fresh network each time);
  * eigenvector signs of the MDS step follow a fixed rule (see include/dmpfold_hip.h);
  * missing trained weights raise FileNotFoundError (no download: predict.py:64-71).
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch

from . import _lib

default_device = "cuda"
default_iterations = 10
default_minsteps = 100

MAX_SEQS = 3000     # predict.py:130-132
MAX_L = 2048        # include/dmpfold_hip.h DMP_MAX_L (the reference has no limit)

_RESNAMES = {0: "ALA", 1: "ARG", 2: "ASN", 3: "ASP", 4: "CYS", 5: "GLN", 6: "GLU", 7: "GLY",
             8: "HIS", 9: "ILE", 10: "LEU", 11: "LYS", 12: "MET", 13: "PHE", 14: "PRO",
             15: "SER", 16: "THR", 17: "TRP", 18: "TYR", 19: "VAL"}


# ---------------------------------------------------------------------------
# host-side parsing (pure Python / C-ABI host helper, no GPU needed)
# ---------------------------------------------------------------------------
def read_aln(input_file):
    """Lines not starting with '>', right-stripped (predict.py:100-104)."""
    rows = []
    with open(input_file, "r") as fh:
        for line in fh.readlines():
            if not line.startswith(">"):
                rows.append(line.rstrip())
    return rows


def read_a3m(input_file):
    """An .a3m alignment as .aln rows: the reference README's conversion
    `egrep -v "^>" x.a3m | sed 's/[a-z]//g' > x.aln` (drop header lines and the lower-case insert
    columns), done in memory."""
    rows = []
    with open(input_file, "r") as fh:
        for line in fh.readlines():
            if not line.startswith(">"):
                rows.append("".join(ch for ch in line.rstrip() if not ("a" <= ch <= "z")))
    return rows


def encode_aln(rows):
    """Residue letters -> uint8 codes (N, L), capped at 3000 rows (predict.py:124-132).
    Ragged input raises ValueError from the reshape, as in the reference.  A character outside the
    alignment alphabet (lower-case a3m inserts, digits ...) maps to a code above 21, which the
    reference's 22-row embedding rejects with IndexError (network.py:223): raised here, for the
    rows that survive the 3000-row cap, as the reference would."""
    nseqs = len(rows)
    length = len(rows[0])
    text = np.frombuffer("".join(rows).encode("latin-1"), dtype=np.uint8)
    codes = np.empty_like(text)
    lib = _lib.load()
    _lib.check(lib.dmp_msa_encode(text.ctypes.data, text.size, codes.ctypes.data))
    alnmat = codes.reshape(nseqs, length)
    if nseqs > MAX_SEQS:
        alnmat = alnmat[:MAX_SEQS]
    if alnmat.size and int(alnmat.max()) > 21:
        raise IndexError("index out of range in self")
    return alnmat


def read_template_ca(template):
    """CA atoms of ATOM records, fixed PDB columns (predict.py:106-117)."""
    xyz = []
    with open(template, "r") as fh:
        for line in fh:
            if line[:4] == "ATOM" and line[12:16] == " CA ":
                xyz.append(np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])],
                                    dtype=np.float32))
    return np.asarray(xyz, dtype=np.float32).reshape(-1, 3)


def _resolve_device(device):
    dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"dmpfold2_amd runs on AMD GPUs only (got device '{device}'); there is no CPU path")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible to PyTorch-ROCm; dmpfold2_amd has no CPU fallback")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


def default_weight_files():
    modeldir = os.path.join(os.path.dirname(os.path.realpath(__file__)), "trained_model")
    return [os.path.join(modeldir, f"FINAL_fullmap_e2e_model_part{p}.pt") for p in ("1", "2")]


def load_state_dict(weights_file=None):
    """The reference's weight files: one pickled state_dict, or the two-part default that is
    merged with dict.update (predict.py:81-96)."""
    if weights_file is None:
        parts = default_weight_files()
        if not os.path.isfile(parts[0]):
            raise FileNotFoundError(
                f"trained model not found at {parts[0]}; the reference would download it, this "
                "build does not: place the two FINAL_fullmap_e2e_model_part*.pt files there or "
                "pass weights_file=/-w")
        sd = torch.load(parts[0], map_location="cpu", weights_only=True)
        sd.update(torch.load(parts[1], map_location="cpu", weights_only=True))
    else:
        # weights_only: a user-supplied -w file is data (a tensor dict), never unpickled code
        sd = torch.load(weights_file, map_location="cpu", weights_only=True)
    return sd


# device-side fault bits (include/dmpfold_hip.h, DMP_FAULT_*)
FAULT_SEQ_HANDOFF, FAULT_F16_RANGE, FAULT_REFINE_HANDOFF, FAULT_BAD_CODE, FAULT_EIG_HANDOFF, FAULT_VGRU_HANDOFF = 1, 2, 4, 8, 16, 32


class DeviceFault(_lib.DmpError):
    """A device-side fault invalidated a prediction (its outputs are NaN)."""

    def __init__(self, bits):
        self.bits = int(bits)
        what = [txt for bit, txt in ((FAULT_SEQ_HANDOFF, "sequence-GRU workgroup hand-off timed out"),
                                     (FAULT_REFINE_HANDOFF, "minimiser workgroup hand-off timed out"),
                                     (FAULT_EIG_HANDOFF, "tridiagonalisation workgroup hand-off timed out"),
                                     (FAULT_VGRU_HANDOFF, "vertical-GRU row barrier timed out"),
                                     (FAULT_F16_RANGE, "an activation left the f16 range of the "
                                      "split-product convolution (conv_mode 2 has no range limit)"))
                if bits & bit]
        super().__init__("device-side fault (results invalid): " + "; ".join(what))


def raise_for_faults(bits):
    if bits & FAULT_BAD_CODE:
        raise IndexError("index out of range in self")      # the embedding lookup of network.py:223
    if bits:
        raise DeviceFault(bits)


# ---------------------------------------------------------------------------
# engine: one dmp_ctx per GPU
# ---------------------------------------------------------------------------
def _env_precision():
    """DMPFOLD_PRECISION selects the arithmetic (option "precision" of include/dmpfold_hip.h) for the drop-in entry points -
    aln_to_coords, the CLI, the batch front end - which have no argument for it:
      2  full-width operands on the 16-bit matrix cores: the float32 operands of the convolutions and of the vertical GRU
         as three exact bf16 pieces (24 significand bits, six products);
      1  the reference's instructions: float32 matrix-core convolutions and vertical GRU (about half the speed of 2);
      0  the fast mode: two f16 pieces per operand (22-23 significand bits), about 1.8 x the speed of 2.
    Unset = the library's default."""
    v = os.environ.get("DMPFOLD_PRECISION", "").strip()
    if v == "":
        return None
    if v not in ("0", "1", "2"):
        raise ValueError(f"DMPFOLD_PRECISION must be 0, 1 or 2, got {v!r}")
    return int(v)


# What the reference's own entry points compute in is float32 (predict.py:136 `.float()`, network.py:25-31), so a context
# - through the C ABI, an `Engine`, a `Pipeline`, aln_to_coords, the CLI, the batch front end - starts in the setting whose
# operands carry float32's 24 significand bits at the 16-bit matrix cores' rate (option "precision" = 2).
# DMPFOLD_PRECISION=0 selects the fast 22-23-bit mode (about 1.8 x the speed), 1 the f32 matrix-core instructions.
DROP_IN_PRECISION = 2


def drop_in_precision():
    v = _env_precision()
    return DROP_IN_PRECISION if v is None else v


class Engine:
    """Owns one `dmp_ctx` (device buffers + packed weights) on one GPU."""

    def __init__(self, device, max_L, max_N, stream=None, precision=None):
        self.lib = _lib.load()
        self.device = _resolve_device(device)
        self.max_L = int(max_L)
        self.max_N = int(min(max_N, MAX_SEQS))
        self._stream = stream          # optional torch.cuda.Stream owned by this engine
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dmp_ctx_create(self.device.index, self.max_L, self.max_N,
                                               C.byref(self._ctx)))
        self.weights_tag = None
        self.last_fallback = False     # the last predict_*_checked call fell back to conv_mode 2
        prec = precision if precision is not None else _env_precision()
        if prec is not None:
            self.set_option("precision", prec)

    def close(self):
        if self._ctx:
            self.lib.dmp_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    @property
    def device_bytes(self):
        return self.get_option("device_mib") << 20

    def stream(self):
        s = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def set_weights(self, state_dict, tag=None):
        """Strict load like load_state_dict (predict.py:98): unknown, missing or mis-shaped
        tensors raise RuntimeError."""
        for key, val in state_dict.items():
            arr = np.ascontiguousarray(
                val.detach().cpu().float().numpy() if isinstance(val, torch.Tensor) else val,
                dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self.lib.dmp_weights_set(self._ctx, key.encode(), arr.ctypes.data, shape,
                                                arr.ndim), RuntimeError)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dmp_weights_finalize(self._ctx), RuntimeError)
        self.weights_tag = tag

    def share_weights(self, other):
        """Use the packed weights of `other` (an engine on the same GPU) instead of packing a copy."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dmp_weights_share(self._ctx, other.ctx), RuntimeError)
        self.weights_tag = other.weights_tag

    def predict(self, alnmat, template_ca=None, iterations=default_iterations,
                minsteps=default_minsteps):
        """codes (N, L) uint8 -> (coords (L,5,3), confs (L,)) float32 tensors on the GPU."""
        alnmat = np.ascontiguousarray(alnmat, dtype=np.uint8)
        with torch.cuda.device(self.device):
            d_msa = torch.from_numpy(alnmat).to(self.device)
        return self.predict_device(d_msa, template_ca, iterations, minsteps)

    def predict_device(self, d_msa, template_ca=None, iterations=default_iterations,
                       minsteps=default_minsteps):
        """Same as `predict` for residue codes already resident on the GPU (uint8 (N, L))."""
        assert d_msa.dtype == torch.uint8 and d_msa.is_contiguous() and d_msa.device == self.device
        n, L = d_msa.shape
        if L < 8:
            raise RuntimeError(f"alignment has {L} columns; the network needs at least 8 "
                               "(MDS embedding width, reference network.py:250-253)")
        with torch.cuda.device(self.device):
            coords = torch.empty((L, 5, 3), dtype=torch.float32, device=self.device)
            confs = torch.empty((L,), dtype=torch.float32, device=self.device)
            d_tpl, lt = None, 0
            if template_ca is not None:
                d_tpl = torch.as_tensor(template_ca, dtype=torch.float32).reshape(-1, 3).to(self.device)
                lt = d_tpl.shape[0]
                if lt != L:
                    raise RuntimeError(f"Sizes of tensors must match: template has {lt} CA atoms, "
                                       f"alignment has {L} columns")
            if self._stream is not None:
                # an engine with its own stream: order it after the producer of the inputs and tell the
                # caching allocator that these blocks are in use there
                self._stream.wait_stream(torch.cuda.current_stream(self.device))
                for x in (coords, confs, d_msa, d_tpl):
                    if x is not None:
                        x.record_stream(self._stream)
            _lib.check(self.lib.dmp_predict(
                self._ctx, d_msa.data_ptr(), n, L,
                d_tpl.data_ptr() if d_tpl is not None else None, lt,
                int(max(iterations, 0)), int(max(minsteps, 0)),
                coords.data_ptr(), confs.data_ptr(), self.stream()))
            # d_msa / d_tpl are stream-ordered temporaries: keep them alive until the work is queued
            self._keep = (d_msa, d_tpl)
        return coords, confs

    def set_option(self, name, value):
        """Additive engine options, e.g. ("conv_f32_exact", 1); see include/dmpfold_hip.h."""
        _lib.check(self.lib.dmp_ctx_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name):
        """The option's current value, read back from the context (not from a Python-side mirror: the
        C API may have been used directly)."""
        v = C.c_int(0)
        _lib.check(self.lib.dmp_ctx_get_option(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def sync_faults(self):
        """Wait for the queued work; the DMP_FAULT_* bits recorded since the last report (reporting
        clears them).  Predictions that ran while a bit was raised returned NaN."""
        bits = C.c_int(0)
        _lib.check(self.lib.dmp_sync_faults(self._ctx, self.stream(), C.byref(bits)))
        return bits.value

    def sync_check(self):
        """Wait for the queued work and raise if a device-side fault was recorded: IndexError for a
        residue code above 21 (as the reference's embedding), DeviceFault otherwise."""
        raise_for_faults(self.sync_faults())

    def predict_checked(self, alnmat, template_ca=None, iterations=default_iterations,
                        minsteps=default_minsteps):
        """`predict`, synchronised and verified.  The default convolution multiplies f16 pieces of its
        operands and needs |activation| < 6e4; a prediction that leaves that range (never seen with
        InstanceNorm'd trunks, but the trained weights decide) is repeated with the 3-way bf16 split,
        which has float32's range, at about half the convolution rate."""
        alnmat = np.ascontiguousarray(alnmat, dtype=np.uint8)
        with torch.cuda.device(self.device):
            d_msa = torch.from_numpy(alnmat).to(self.device)
        return self.predict_device_checked(d_msa, template_ca, iterations, minsteps)

    def predict_device_checked(self, d_msa, template_ca=None, iterations=default_iterations,
                               minsteps=default_minsteps):
        """`predict_checked` for residue codes already resident on the GPU."""
        coords, confs = self.predict_device(d_msa, template_ca, iterations, minsteps)
        bits = self.sync_faults()
        self.last_fallback = False
        if bits & FAULT_VGRU_HANDOFF and self.get_option("vgru_persistent"):
            # the persistent chain's row barriers timed out (its workgroups were not all resident: another process
            # on this GPU holds CUs with a launch of the same kind): the launch-per-row form has no such requirement
            print("dmpfold2_amd: the persistent vertical-GRU launch could not get the whole GPU; re-running this "
                  "alignment (and every later one on this engine) with one launch per alignment row", file=sys.stderr)
            self.set_option("vgru_persistent", 0)
            coords, confs = self.predict_device(d_msa, template_ca, iterations, minsteps)
            bits = self.sync_faults()
        if bits == FAULT_F16_RANGE and self.get_option("conv_mode") == 0:
            print("dmpfold2_amd: activations left the f16 range of the split-product convolution; "
                  "re-running this alignment with conv_mode=2 (bf16 split, no range limit)",
                  file=sys.stderr)
            self.last_fallback = True
            self.set_option("conv_mode", 2)
            try:
                coords, confs = self.predict_device(d_msa, template_ca, iterations, minsteps)
                bits = self.sync_faults()
            finally:
                self.set_option("conv_mode", 0)
        raise_for_faults(bits)
        return coords, confs

    def fetch(self, name, numel):
        out = torch.empty((int(numel),), dtype=torch.float32, device=self.device)
        n = _lib.check(self.lib.dmp_debug_fetch(self._ctx, name.encode(), out.data_ptr(),
                                                int(numel), self.stream()))
        return out[:n]


class _PipelineEngine(Engine):
    """Engine i of a `Pipeline`: a view of the context and stream the C pipeline owns (options, introspection, the
    repeat of a faulted target on an idle pipeline); closing it does nothing."""

    def __init__(self, lib, device, ctx, stream_ptr, max_L, max_N):       # noqa: super().__init__ creates a context
        self.lib = lib
        self.device = device
        self.max_L, self.max_N = int(max_L), int(max_N)
        self._ctx = C.c_void_p(ctx)
        self._stream = torch.cuda.ExternalStream(stream_ptr, device=device)
        self.weights_tag = None
        self.last_fallback = False

    def close(self):
        self._ctx = C.c_void_p()

    def __del__(self):
        pass


# ticket states of the C pipeline (include/dmpfold_hip.h, DMP_TICKET_*)
_T_QUEUED, _T_RUNNING, _T_ISSUED, _T_DONE, _T_FAILED = 0, 1, 2, 3, 4


class Pipeline:
    """Throughput mode on one GPU: `streams` engines (each its own context and HIP stream) share a lane, so their
    machine-filling convolutions take turns while the latency-bound kernels of one target (eigensolver, sequence GRUs,
    minimiser) run under the convolutions of another; targets that start together run their vertical GRUs as one chain.

    Round 6: the scheduler lives behind the C ABI (csrc/pipeline.hip, dmp_pipeline_*: one host thread inside the library
    issues every unit); this class allocates the tensors, keeps them alive, and mirrors the interface the Python scheduler
    of rounds 1-5 had - submit / pump / drain / result, step / poll / peek for the streaming batch front end, collect for
    the repeat of faulted targets."""

    def __init__(self, device, max_L, max_N, state_dict, streams=2, precision=None, torch_streams=False):
        """`torch_streams`: the engines run on PyTorch pool streams handed to the library (dmp_pipeline_create_on) instead of
        the library's own - for a host that wants every stream to be one its allocator knows."""
        self.lib = _lib.load()
        self.device = _resolve_device(device)
        S = max(1, int(streams))
        self._p = C.c_void_p()
        max_N = int(min(max_N, MAX_SEQS))
        with torch.cuda.device(self.device):
            if torch_streams:
                self._torch_streams = [torch.cuda.Stream(device=self.device) for _ in range(S)]
                arr = (C.c_void_p * S)(*[st.cuda_stream for st in self._torch_streams])
                _lib.check(self.lib.dmp_pipeline_create_on(self.device.index, int(max_L), max_N, S, arr, C.byref(self._p)))
            else:
                _lib.check(self.lib.dmp_pipeline_create(self.device.index, int(max_L), max_N, S, C.byref(self._p)))
        self.engines = [_PipelineEngine(self.lib, self.device, self.lib.dmp_pipeline_ctx(self._p, i),
                                        self.lib.dmp_pipeline_stream(self._p, i), max_L, max_N) for i in range(S)]
        self.engines[0].set_weights(state_dict)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dmp_pipeline_weights_ready(self._p), RuntimeError)    # packed once per pipeline
        prec = precision if precision is not None else _env_precision()
        if prec is not None:
            self.set_option("precision", prec)
        self._jobs = {}               # ticket -> (d_msa, iterations, minsteps, d_tpl, coords, confs, ready event): kept alive
        self._handed = []             # tickets whose result was handed out before the GPU finished them: released later

    def set_option(self, name, value):
        """An engine option on EVERY engine of the (idle) pipeline: a group's vertical-GRU chain runs in its leader's
        arithmetic and serves all members, so the engines must agree on "precision" / "vgru_f32" / "vgru_persistent"."""
        _lib.check(self.lib.dmp_pipeline_set_option(self._p, name.encode(), int(value)))

    def close(self):
        if self._p:
            self.lib.dmp_pipeline_destroy(self._p)           # joins the scheduler thread, synchronises the streams
            self._p = C.c_void_p()
        for e in self.engines:
            e.close()
        self.engines = []
        self._jobs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- submission --------------------------------------------------------------------------
    def submit(self, d_msa, iterations=default_iterations, minsteps=default_minsteps, template_ca=None):
        """Queue one target (uint8 (N, L) tensor on the GPU, optional template CA trace (L, 3));
        returns a ticket for `result`."""
        assert d_msa.dtype == torch.uint8 and d_msa.is_contiguous() and d_msa.device == self.device
        n, L = d_msa.shape
        if L < 8:
            raise RuntimeError(f"alignment has {L} columns; the network needs at least 8")
        if L > self.engines[0].max_L or n > self.engines[0].max_N:
            raise RuntimeError(f"alignment {n} x {L} exceeds the pipeline capacity "
                               f"{self.engines[0].max_N} x {self.engines[0].max_L}")
        d_tpl = None
        with torch.cuda.device(self.device):
            if template_ca is not None:
                d_tpl = torch.as_tensor(template_ca, dtype=torch.float32).reshape(-1, 3).to(self.device).contiguous()
                if d_tpl.shape[0] != L:
                    raise RuntimeError(f"Sizes of tensors must match: template has {d_tpl.shape[0]} CA atoms, "
                                       f"alignment has {L} columns")
            coords = torch.empty((L, 5, 3), dtype=torch.float32, device=self.device)
            confs = torch.empty((L,), dtype=torch.float32, device=self.device)
            # the stream that is current NOW produced d_msa (the caller's copy stream, say); the engine that takes the
            # target orders itself behind this point
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        # Inputs and outputs are used on the stream of whichever engine takes the target (and, as a rider, on a group
        # leader's): the caching allocator has to know before it recycles their blocks.
        for e in self.engines:
            for x in (coords, confs, d_msa, d_tpl):
                if x is not None:
                    x.record_stream(e._stream)
        t = _lib.check(self.lib.dmp_pipeline_submit(
            self._p, d_msa.data_ptr(), n, L, d_tpl.data_ptr() if d_tpl is not None else None,
            int(max(iterations, 0)), int(max(minsteps, 0)), coords.data_ptr(), confs.data_ptr(),
            C.c_void_p(ready.cuda_event)))
        self._jobs[t] = (d_msa, int(max(iterations, 0)), int(max(minsteps, 0)), d_tpl, coords, confs, ready)
        self._reap()
        return t

    def _status(self, ticket):
        st, bits = C.c_int(0), C.c_int(0)
        rc = self.lib.dmp_pipeline_status(self._p, ticket, C.byref(st), C.byref(bits))
        return st.value, bits.value, rc

    def _reap(self):
        keep = []
        for t in self._handed:
            st, _, _ = self._status(t)
            if st in (_T_DONE, _T_FAILED):
                self.lib.dmp_pipeline_release(self._p, t)
            else:
                keep.append(t)
        self._handed = keep

    def _raise_failed(self):
        """A target whose units could not be issued (a HIP or capacity error inside the scheduler) fails its ticket; the
        Python scheduler of rounds 1-5 raised from pump() / drain() at that point, and so does this."""
        for t in list(self._jobs):
            st, _, rc = self._status(t)
            if st == _T_FAILED:
                _lib.check(rc if rc < 0 else -5)

    def pump(self):
        """Block until every queued target has been started on an engine."""
        _lib.check(self.lib.dmp_pipeline_wait(self._p, 0))
        self._raise_failed()

    def drain(self):
        """Block until every queued target is fully enqueued; the current stream then waits for
        the engines' streams (the host is not synchronised with the GPU)."""
        _lib.check(self.lib.dmp_pipeline_wait(self._p, 1))
        cur = torch.cuda.current_stream(self.device)
        for e in self.engines:
            cur.wait_stream(e._stream)
        self._raise_failed()

    def result(self, ticket):
        job = self._jobs.pop(ticket)
        self._handed.append(ticket)
        self._reap()
        return job[4], job[5]

    # ---- streaming use (dmpfold2_amd.batch): submit / step / poll, no barrier between targets ----------------
    def step(self, rounds=32):
        """The scheduler runs in its own thread inside the library: there is nothing to step.  Yields the core for a moment
        (the caller's loop does a piece of host work per call) and answers False - "nothing more for you to issue"."""
        time.sleep(2e-4)
        return False

    def backlog(self):
        """Targets queued but not yet started on an engine."""
        q, r = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.dmp_pipeline_backlog(self._p, C.byref(q), C.byref(r)))
        return q.value

    def busy(self):
        q, r = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.dmp_pipeline_backlog(self._p, C.byref(q), C.byref(r)))
        return q.value > 0 or r.value > 0

    def poll(self):
        """Tickets whose prediction has COMPLETED on the GPU since the last call (their tensors may be read from
        any stream); `peek` / `result` hand the tensors out."""
        buf, n = (C.c_int64 * 64)(), C.c_int(0)
        out = []
        while True:
            _lib.check(self.lib.dmp_pipeline_poll(self._p, buf, 64, C.byref(n)))
            out += [buf[i] for i in range(n.value)]
            if n.value < 64:
                return out

    def peek(self, ticket):
        job = self._jobs[ticket]
        return job[4], job[5]

    def stats(self):
        v = (C.c_longlong * 8)()
        _lib.check(self.lib.dmp_pipeline_stats(self._p, v, 8))
        return {"groups": v[0], "max_group": v[1], "rider_chains": v[2], "max_riders": v[3], "riders_left": v[4],
                "idle_rounds": v[5], "rounds": v[6], "scheduler_thread_cpu_s": v[7] * 1e-6}

    def submit_many(self, d_msas, iterations=default_iterations, minsteps=default_minsteps):
        """Submit a batch with the scheduler paused, so that the first vertical-GRU group is formed from the whole batch
        (the scheduler's thread would otherwise start whatever has arrived when it looks)."""
        _lib.check(self.lib.dmp_pipeline_pause(self._p, 1))
        try:
            return [self.submit(m, iterations, minsteps) for m in d_msas]
        finally:
            _lib.check(self.lib.dmp_pipeline_pause(self._p, 0))

    def collect(self, tickets):
        """drain + synchronise + verify.  Returns {ticket: (coords, confs) or Exception}: a target
        whose prediction recorded a device-side fault (its outputs are NaN) is repeated alone through
        `Engine.predict_device_checked` - which falls back to the range-free convolution where that
        is the cure - and only if that fails too its entry is the exception.  One bad target never
        costs the others their results.  Once one repeat needed the range-free convolution the
        remaining repeats run in it directly (the weights, not the alignment, put a trunk outside the
        f16 range: the others would only fault again first), with one note on stderr for all of them."""
        _lib.check(self.lib.dmp_pipeline_wait(self._p, 2))
        cur = torch.cuda.current_stream(self.device)
        for e in self.engines:
            cur.wait_stream(e._stream)
        for e in self.engines:
            e.sync_faults()                     # the engines' latched words: cleared, the per-ticket words below decide
        out = {}
        eng = self.engines[0]
        mode0 = eng.get_option("conv_mode")
        try:
            for t in tickets:
                st, bits, rc = self._status(t)
                job = self._jobs.get(t)
                coords, confs = self.result(t)
                if st == _T_FAILED:
                    out[t] = _lib.DmpError(_lib.load().dmp_last_error().decode("utf-8", "replace") or f"error {rc}")
                    continue
                if bits:
                    d_msa, nloops, minsteps, d_tpl = job[:4]
                    try:
                        coords, confs = eng.predict_device_checked(d_msa, d_tpl, nloops, minsteps)
                        if eng.last_fallback:
                            eng.set_option("conv_mode", 2)
                        if not eng.get_option("vgru_persistent"):
                            # the repeat fell back to one vertical-GRU launch per row (another process holds CUs of
                            # this GPU): the other engines - and a group chain led by one of them - would only
                            # fault again, target after target
                            self.set_option("vgru_persistent", 0)
                    except (IndexError, _lib.DmpError) as exc:
                        out[t] = exc
                        continue
                out[t] = (coords, confs)
        finally:
            eng.set_option("conv_mode", mode0)
        return out

    def run(self, d_msas, iterations=default_iterations, minsteps=default_minsteps):
        """Predict every target (uint8 (N, L) tensors on the GPU).  Returns [(coords, confs)] in
        input order, ordered on the current stream; the host is not synchronised with the tail."""
        tickets = self.submit_many(d_msas, iterations, minsteps)
        self.drain()
        return [self.result(t) for t in tickets]

    def sync_check(self):
        """Synchronise every engine and raise for the first recorded fault (see `collect` for the
        per-target form)."""
        bits = 0
        for e in self.engines:
            bits |= e.sync_faults()
        raise_for_faults(bits)


class _EngineCache(dict):
    """device index -> the engine used last there (what tests and diagnostics look at).  Behind it a small LRU of
    engines per device, one per weights file: the reference builds a fresh network on every call (predict.py:79), so
    callers may alternate between weight files - or call from several threads - without re-packing 140 MB each time."""
    MAX_PER_DEVICE = 2

    def __init__(self):
        super().__init__()
        self.lru = {}                 # device index -> [engine, ...], most recently used last

    def clear(self):
        for engines in self.lru.values():
            for e in engines:
                e.close()
        self.lru = {}
        super().clear()


_ENGINES = _EngineCache()
_DEVICE_LOCKS = {}
_LOCKS_GUARD = threading.Lock()


def device_lock(device):
    """One re-entrant lock per GPU for the drop-in entry points: `aln_to_coords` may be called from several threads (the
    reference's function is re-entrant - every call builds its own network, predict.py:79); here the calls of a device
    share cached engines whose buffers one prediction owns from its first kernel to its synchronisation, so they take
    turns.  (Throughput across targets is what `Pipeline` / the batch front end are for.)"""
    dev = _resolve_device(device)
    with _LOCKS_GUARD:
        return _DEVICE_LOCKS.setdefault(dev.index, threading.RLock())


def get_engine(device, L, N, weights_file=None, state_dict=None):
    """Cached engine for `device` and these weights, grown when an alignment exceeds its capacity; weights are packed
    once per (engine, weights file).  Call it - and use the engine - under `device_lock(device)` when other threads may
    do the same."""
    if L > MAX_L:
        raise RuntimeError(f"alignment has {L} columns; this build supports at most {MAX_L} "
                           "(include/dmpfold_hip.h DMP_MAX_L: the eigensolver's LDS image)")
    dev = _resolve_device(device)
    N = min(N, MAX_SEQS)
    if state_dict is not None:
        tag = None
    else:
        files = [weights_file] if weights_file is not None else default_weight_files()
        tag = tuple((f, os.path.getmtime(f)) for f in files if os.path.isfile(f))
    with device_lock(dev):
        lru = _ENGINES.lru.setdefault(dev.index, [])
        eng = next((e for e in lru if tag and e.weights_tag == tag), None)
        if eng is None and lru and (not tag or len(lru) >= _EngineCache.MAX_PER_DEVICE):
            eng = lru[0]                                  # recycle the least recently used one (new weights below)
        if eng is not None:
            lru.remove(eng)
        if eng is None or L > eng.max_L or N > eng.max_N:
            max_L = max(L, eng.max_L if eng else 0)
            max_N = max(N, eng.max_N if eng else 0)
            if eng is not None:
                eng.close()
            eng = Engine(dev, max_L, max_N)              # (weights_tag None: packed below)
        lru.append(eng)
        _ENGINES[dev.index] = eng
        want = drop_in_precision()
        if eng.get_option("precision") != want:
            eng.set_option("precision", want)
        if state_dict is not None:
            eng.set_weights(state_dict, tag=None)
        elif eng.weights_tag != tag or not tag:
            eng.set_weights(load_state_dict(weights_file), tag=tag)
    return eng


# ---------------------------------------------------------------------------
# the reference's public functions
# ---------------------------------------------------------------------------
def aln_to_coords(input_file, device=default_device, template=None, iterations=default_iterations,
                  minsteps=default_minsteps, weights_file=None, return_alnmat=False):
    """Alignment file -> (coords (L,5,3) [N, CA, C, O, CB], confs (L,)) on `device`,
    plus the uint8 alignment matrix when `return_alnmat` is set (predict.py:74-158)."""
    dev = _resolve_device(device)
    aln = read_aln(input_file)
    template_ca = read_template_ca(template) if template is not None else None
    alnmat = encode_aln(aln)
    nseqs, length = alnmat.shape
    with device_lock(dev):                  # re-entrant like the reference's function: callers of one GPU take turns
        eng = get_engine(dev, length, nseqs, weights_file=weights_file)
        coords, confs = eng.predict_checked(alnmat, template_ca, iterations, minsteps)
    if return_alnmat:
        return coords, confs, alnmat
    return coords, confs


def pdb_text(coords, confs, alnmat):
    """The PDB text of predict.py:195-208 from host copies of the outputs."""
    coords = coords.detach().cpu()
    confs = confs.detach().cpu()
    lines = ["REMARK  CONF:  " + repr(confs.mean().item())]
    atoms = (" N  ", " CA ", " C  ", " O  ", " CB ")
    xyz, cf = coords.tolist(), confs.tolist()      # Python floats of the float32 values, as .item() gives them
    atomnum = 1
    for ri in range(coords.size(0)):
        code = int(alnmat[0, ri])
        for ai, an in enumerate(atoms):
            if code != 7 or ai != 4:          # glycine has no CB
                x, y, z = xyz[ri][ai]
                lines.append("ATOM   %4d %s %s  %4d    %8.3f%8.3f%8.3f  1.00%6.2f" % (
                    atomnum, an, _RESNAMES[code], ri + 1, x, y, z, cf[ri]))
                atomnum += 1
    lines.append("END")
    return "\n".join(lines) + "\n"


def run_dmpfold(argv=None):
    """Command-line entry point with the reference's flags (predict.py:160-208)."""
    parser = argparse.ArgumentParser(description=(
        "DMPfold2 end-to-end structure prediction on AMD MI355X (HIP engine). "
        "Prints a PDB format model file."))
    parser.add_argument("-i", "--input_file", type=str, required=True,
                        help="input sequence alignment in aln format")
    parser.add_argument("-d", "--device", type=str, default=default_device, required=False,
                        help="device to run on (cuda, cuda:1, ...)")
    parser.add_argument("-t", "--template", type=str, required=False,
                        help="use a PDB file as a template")
    parser.add_argument("-n", "--iterations", type=int, default=default_iterations, required=False,
                        help="number of iteration cycles")
    parser.add_argument("-m", "--minsteps", type=int, default=default_minsteps, required=False,
                        help="number of minimization steps")
    parser.add_argument("-w", "--model_weights", type=str, required=False,
                        help="use a custom set of model weights")
    args = parser.parse_args(argv)
    coords, confs, alnmat = aln_to_coords(args.input_file, device=args.device,
                                          template=args.template, iterations=args.iterations,
                                          minsteps=args.minsteps, weights_file=args.model_weights,
                                          return_alnmat=True)
    sys.stdout.write(pdb_text(coords, confs, alnmat))
