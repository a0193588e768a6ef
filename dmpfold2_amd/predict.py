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
      2  full-width operands on the 16-bit matrix cores: the convolutions' float32 operands as three exact bf16 pieces (24
         significand bits, six products), float32 vertical GRU;
      1  the reference's instructions: float32 matrix-core convolutions and vertical GRU (about half the speed of 2);
      0  the fast mode: two f16 pieces per operand (22-23 significand bits), about 1.8 x the speed of 2.
    Unset = the library's default."""
    v = os.environ.get("DMPFOLD_PRECISION", "").strip()
    if v == "":
        return None
    if v not in ("0", "1", "2"):
        raise ValueError(f"DMPFOLD_PRECISION must be 0, 1 or 2, got {v!r}")
    return int(v)


# What the reference's own entry points compute in is float32 (predict.py:136 `.float()`, network.py:25-31), so the DROP-IN
# entry points of this package - aln_to_coords, the CLI, the batch front end - default to the setting whose operands carry
# float32's 24 significand bits at the 16-bit matrix cores' rate (option "precision" = 2); DMPFOLD_PRECISION=0 selects the
# fast 22-23-bit mode (about 1.8 x the speed), 1 the f32 matrix-core instructions.  A context made through the C ABI or
# an `Engine` made directly starts in the library's setting (precision 0) unless DMPFOLD_PRECISION says otherwise.
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


# Engine streams are reused from one pipeline of the process to the next: which hardware queues a stream gets depends
# on how many streams were created before it, and on this runtime the SECOND set of four costs the scheduler 15 %
# (tools/pipeline_order.py: 5.87 structures/s on pool streams 5-8 against 6.9-7.0 on every other set).
_STREAM_POOL = {}            # device index -> [[stream, in use]]


def _take_stream(device):
    pool = _STREAM_POOL.setdefault(device.index, [])
    for item in pool:
        if not item[1]:
            item[1] = True
            return item[0]
    with torch.cuda.device(device):
        st = torch.cuda.Stream(device=device)
    pool.append([st, True])
    return st


def _release_stream(device, st):
    for item in _STREAM_POOL.get(device.index, []):
        if item[0] is st:
            item[1] = False


class Pipeline:
    """Throughput mode on one GPU: `streams` engines (each its own context and HIP stream) share a
    lane, so their machine-filling convolutions take turns while the latency-bound kernels of one
    target (eigensolver, sequence GRUs, minimiser, the vertical GRU's per-row launches) run under
    the convolutions of another.

    One host thread schedules all engines unit by unit (include/dmpfold_hip.h,
    dmp_predict_issue_unit): light units are enqueued at once, a residual block - whose convolution
    takes the lane, in issue order - only when everything its engine was given has completed, so
    the lane is always handed to a convolution that can start immediately and never waits behind
    an engine that is still in its eigensolver or front end.  Targets are taken from one queue by
    whichever engine is free."""

    def __init__(self, device, max_L, max_N, state_dict, streams=2, precision=None):
        self.lib = _lib.load()
        self.device = _resolve_device(device)
        self.engines = []
        for _ in range(max(1, int(streams))):
            st = _take_stream(self.device)
            eng = Engine(self.device, max_L, max_N, stream=st, precision=precision)
            if self.engines:
                eng.share_weights(self.engines[0])       # packed once per pipeline, not once per engine
            else:
                eng.set_weights(state_dict)
            if streams > 1:
                if self.engines:
                    _lib.check(self.lib.dmp_ctx_share_lane(eng.ctx, self.engines[0].ctx))    # one lane for all of them
                # Several engines: the eigensolver's Householder steps as one launch each, not as the cluster kernel
                # (same bits).  The cluster is the faster form for ONE prediction (0.9 against 1.9 ms at L = 300), but
                # its 32 resident workgroups poll beside the other engines' convolutions for that long: measured
                # 7.39 / 7.32 structures/s with the launches against 7.29 / 7.30 (bench.py, alternating, one box).
                eng.set_option("tridiag_cluster", 0)
            self.engines.append(eng)
        S = len(self.engines)
        self._pending = []            # (ticket, d_msa, iterations, minsteps)
        self._slot = [None] * S       # per engine: (ticket, coords, confs) of the prediction in flight
        self._done = [0] * S          # residual blocks issued / in total for the prediction in flight
        self._total = [0] * S
        self._results = {}
        self._done_ev = {}            # ticket -> event recorded behind dmp_predict_end (poll)
        self._jobs = {}               # ticket -> job, kept until the result is handed out (retry of faults)
        self._tickets = 0
        # An engine issues its first residual block only when the engine that started before it is half a pass
        # (8 blocks) into its own.  The lane deals the blocks out in turn, so engines that leave their front
        # ends together would also reach the end of every pass together and sit in their pass tails
        # (eigensolver, coordinate GRU: 6 ms without a convolution) at the same time; half a pass apart the
        # tails interleave.  Measured on three boxes, alternating runs: +0.8 .. +1.5 % structures/s for 5-12
        # blocks against 0 (7.06-7.11 against 7.01; 7.00 against 6.92).  DMP_TAIL_STAGGER overrides (0 = off).
        self._tail_stagger = int(os.environ.get("DMP_TAIL_STAGGER", "8"))
        # Group start: predictions that begin together run their vertical GRUs (2001 dependent launches each at the
        # north-star size) as ONE launch chain that serves the columns of all of them (dmp_predict_group_vgru): four
        # targets take 52-55 ms in one chain against 110 ms as four chains side by side.  A free engine therefore
        # waits for the engines that are about to finish (at most DMP_GROUP_PATIENCE residual blocks left) and
        # starts together with them, up to DMP_VGRU_GROUP members (1 = every prediction runs its own chain).
        self._group_max = max(1, min(4, int(os.environ.get("DMP_VGRU_GROUP", "4"))))   # engines that start together (the state buffers hold 8 x max_L columns: members + riders)
        self._group_patience = int(os.environ.get("DMP_GROUP_PATIENCE", "40"))
        # Riders: a group's chain also serves the NEXT targets in the queue (dmp_predict_group_riders; members + riders
        # <= 8), whose results are handed over when those targets start (dmp_predict_set_vgru_result): a chain costs
        # 10.5 us per alignment row whatever it serves (launch boundary, cold L2s), so one chain of eight every second
        # round replaces two chains of four.  The chain still runs in a front-end phase - beside no convolution.
        # DMP_VGRU_RIDERS=0 switches it off.
        self._riders_max = max(0, min(7, int(os.environ.get("DMP_VGRU_RIDERS", "4")))) if S > 1 else 0
        self._riding = {}             # ticket -> True: a rider whose chain has not been issued to its end yet
        self._rider_wait = [None] * S  # per leading engine: (jobs, outs) of the riders in the chain it has yet to issue
        self._ahead = {}              # ticket -> (result tensor (L, 512), event recorded behind the chain it rode in)

    def set_option(self, name, value):
        """An engine option on EVERY engine of the pipeline (a group's vertical-GRU chain runs in its leader's arithmetic
        and serves all members: the engines must agree on "precision" / "vgru_f32" / "vgru_persistent")."""
        for e in self.engines:
            e.set_option(name, value)

    def close(self):
        for e in self.engines:
            e.close()
            if e._stream is not None:
                _release_stream(self.device, e._stream)
        self.engines = []
        self._ahead = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scheduler ---------------------------------------------------------------------------
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
        if template_ca is not None:
            d_tpl = torch.as_tensor(template_ca, dtype=torch.float32).reshape(-1, 3).to(self.device).contiguous()
            if d_tpl.shape[0] != L:
                raise RuntimeError(f"Sizes of tensors must match: template has {d_tpl.shape[0]} CA atoms, "
                                   f"alignment has {L} columns")
        t = self._tickets
        self._tickets += 1
        # the stream that is current NOW produced d_msa (the caller's copy stream, say); the engine that takes the
        # target later orders itself behind this point, whatever stream is current then
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        job = (t, d_msa, int(max(iterations, 0)), int(max(minsteps, 0)), d_tpl, ready)
        self._pending.append(job)
        self._jobs[t] = job
        return t

    def _begin_group(self, slots):
        """Start the next len(slots) queued targets on these free engines; those whose vertical GRU did not ride in an
        earlier chain form one vertical-GRU group (the first of them leads), and the chain takes the next queued
        targets along as riders."""
        grouped = []
        for s in slots:
            job = self._pending.pop(0)
            self._done[s] = 0
            self._total[s] = (job[2] + 1) * 16
            self._begin(s, job)
            ahead = self._ahead.pop(job[0], None)
            if ahead is not None:
                out, ev = ahead
                out.record_stream(self.engines[s]._stream)
                _lib.check(self.lib.dmp_predict_set_vgru_result(self.engines[s].ctx, out.data_ptr(),
                                                                C.c_void_p(ev.cuda_event)))
                self._slot[s] = self._slot[s] + (ahead,)          # keeps the tensor and the event alive
            else:
                grouped.append(s)
        slots = grouped
        if len(slots) > 1:
            lead = self.engines[slots[0]]
            for s in slots[1:]:                       # the leader's stream reads every member's alignment
                for x in self._slot[s][3]:
                    if x is not None:
                        x.record_stream(lead._stream)
            ctxs = (C.c_void_p * len(slots))(*[self.engines[s].ctx for s in slots])
            _lib.check(self.lib.dmp_predict_group_vgru(ctxs, len(slots)))
            if self._riders_max:
                room = min(self._riders_max, 8 - len(slots))
                jobs = [j for j in self._pending[:room] if j[0] not in self._ahead and j[0] not in self._riding]
                if jobs:
                    k = len(jobs)
                    with torch.cuda.device(self.device):
                        outs = [torch.empty((j[1].shape[1], 512), dtype=torch.float32, device=self.device) for j in jobs]
                    for j, o in zip(jobs, outs):
                        lead._stream.wait_event(j[5])             # the rider's alignment is ready
                        j[1].record_stream(lead._stream)
                        o.record_stream(lead._stream)
                    mp = (C.c_void_p * k)(*[j[1].data_ptr() for j in jobs])
                    op = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
                    Ns = (C.c_int * k)(*[j[1].shape[0] for j in jobs])
                    Ls = (C.c_int * k)(*[j[1].shape[1] for j in jobs])
                    _lib.check(self.lib.dmp_predict_group_riders(lead.ctx, k, mp, Ns, Ls, op))
                    self._rider_wait[slots[0]] = (jobs, outs)
                    for j in jobs:
                        self._riding[j[0]] = True

    def _begin(self, s, job):
        t, d_msa, nloops, minsteps, d_tpl, ready = job
        e = self.engines[s]
        cur = torch.cuda.current_stream(self.device)
        e._stream.wait_stream(cur)
        e._stream.wait_event(ready)
        n, L = d_msa.shape
        # The outputs belong to the caller's stream (drain() orders it after the engine); inputs and
        # outputs are used on the engine's stream, which the caching allocator has to know before it
        # recycles their blocks.
        with torch.cuda.device(self.device):
            coords = torch.empty((L, 5, 3), dtype=torch.float32, device=self.device)
            confs = torch.empty((L,), dtype=torch.float32, device=self.device)
        for x in (coords, confs, d_msa, d_tpl):
            if x is not None:
                x.record_stream(e._stream)
        _lib.check(self.lib.dmp_predict_begin_units(
            e.ctx, d_msa.data_ptr(), n, L, d_tpl.data_ptr() if d_tpl is not None else None,
            L if d_tpl is not None else 0, nloops, minsteps))
        self._slot[s] = (t, coords, confs, (d_msa, d_tpl))

    def _pump(self):
        """One scheduling round over the engines; True if anything was enqueued."""
        lib = self.lib
        gated = len(self.engines) > 1
        progressed = False
        # engines whose next unit is a residual block: when there is only one, nobody else can use the
        # lane, so it may queue its next block behind the running one instead of draining first
        n_conv = sum(1 for s, e in enumerate(self.engines)
                     if self._slot[s] is not None and lib.dmp_predict_next_unit(e.ctx) == 2) if gated else 0
        free = [s for s in range(len(self.engines)) if self._slot[s] is None]
        startable = free[:len(self._pending)]
        if self._riding and any(j[0] in self._riding for j in self._pending[:len(startable)]):
            startable = []            # the chain these targets ride in has not been issued to its end yet
        if startable:
            # engines about to finish: wait for them and start together (one vertical-GRU chain for the group)
            soon = [r for r in range(len(self.engines)) if self._slot[r] is not None
                    and self._total[r] - self._done[r] <= self._group_patience]
            soon_with_work = min(len(self._pending) - len(startable), len(soon))
            want = min(self._group_max, len(startable) + soon_with_work)
            if len(startable) >= want or not soon_with_work or self._group_max == 1:
                if self._group_max == 1:
                    for s in startable:
                        self._begin_group([s])
                else:
                    self._begin_group(startable[:max(want, 1)])
                progressed = True
        for s, e in enumerate(self.engines):
            if self._slot[s] is None:
                continue
            while True:
                kind = lib.dmp_predict_next_unit(e.ctx)
                if kind == 3:                         # waits for its group leader's vertical-GRU chain
                    break
                if kind == 0:
                    # final refinement + backbone; neither needs the lane
                    t, coords, confs = self._slot[s][:3]
                    _lib.check(lib.dmp_predict_end(e.ctx, coords.data_ptr(), confs.data_ptr(), e.stream()))
                    self._results[t] = (coords, confs)
                    ev = torch.cuda.Event()
                    ev.record(e._stream)
                    self._done_ev[t] = ev
                    self._slot[s] = None
                    progressed = True
                    break
                if self._tail_stagger and kind == 2 and self._done[s] == 0:
                    older = [r for r in range(len(self.engines)) if r != s and self._slot[r] is not None
                             and self._slot[r][0] < self._slot[s][0]]
                    if older:
                        prev = max(older, key=lambda r: self._slot[r][0])
                        # only behind an engine that is in (or at the door of) its trunk passes: one that is
                        # still in its front end must not hold the lane back
                        in_trunk = self._done[prev] > 0 or lib.dmp_predict_next_unit(self.engines[prev].ctx) == 2
                        if in_trunk and self._done[prev] < min(self._tail_stagger, self._total[prev]):
                            break
                if gated:
                    # a convolution is handed the lane only when it can start at once; light units
                    # are kept one deep so this loop returns to the other engines quickly
                    busy = _lib.check(lib.dmp_ctx_pending(e.ctx))
                    if busy > (0 if (kind == 2 and n_conv > 1) else 1):
                        break
                _lib.check(lib.dmp_predict_issue_unit(e.ctx, e.stream()))
                progressed = True
                if self._rider_wait[s] is not None and e.get_option("chain_issued"):
                    # the riders' results are behind this point of the leader's stream
                    jobs, outs = self._rider_wait[s]
                    self._rider_wait[s] = None
                    ev = torch.cuda.Event()
                    ev.record(e._stream)
                    for j, o in zip(jobs, outs):
                        self._ahead[j[0]] = (o, ev)
                        self._riding.pop(j[0], None)
                if kind == 2:
                    self._done[s] += 1
                if gated:
                    break
        return progressed

    def _idle(self):
        """Nothing could be issued: every engine waits for the GPU (or for an engine that does).  Rounds 1-3 spun on
        sched_yield here - one core per GPU at 100 %.  Now the thread sleeps.  "Until the oldest outstanding unit of an
        engine has completed" was tried first (blocking-sync events) and lost a fifth of the throughput: units complete
        out of order across the engines, and a thread blocked on one engine's 40 ms vertical-GRU chain serves nobody.
        A bounded 20 us sleep per idle round instead (round 3 measured: costs the throughput nothing; 100 us: 1.3 %).
        The lane keeps two convolutions queued, so the wake-up latency never leaves it empty."""
        time.sleep(2e-5)

    def pump(self):
        """Schedule until every queued target has been started on an engine."""
        with torch.cuda.device(self.device):          # launches and graph builds need the current device
            while self._pending:
                if not self._pump():
                    self._idle()

    def drain(self):
        """Schedule until every queued target is fully enqueued; the current stream then waits for
        the engines' streams (nothing is synchronised with the host)."""
        with torch.cuda.device(self.device):
            while self._pending or any(x is not None for x in self._slot):
                if not self._pump():
                    self._idle()
        cur = torch.cuda.current_stream(self.device)
        for e in self.engines:
            cur.wait_stream(e._stream)

    def result(self, ticket):
        self._jobs.pop(ticket, None)
        self._done_ev.pop(ticket, None)
        return self._results.pop(ticket)

    # ---- streaming use (dmpfold2_amd.batch): submit / step / poll, no barrier between targets ----------------
    def step(self, rounds=32):
        """Up to `rounds` scheduling rounds (fewer when a round finds nothing to issue: the core is yielded and the
        call returns, so that the caller can do a piece of host work while the GPU is busy).  True if the LAST round
        enqueued something - i.e. the scheduler may have more to issue right away."""
        with torch.cuda.device(self.device):
            for _ in range(max(1, int(rounds))):
                if not self._pump():
                    self._idle()
                    return False
        return True

    def backlog(self):
        """Targets queued but not yet started on an engine."""
        return len(self._pending)

    def busy(self):
        return bool(self._pending) or any(x is not None for x in self._slot)

    def poll(self):
        """Tickets whose prediction has COMPLETED on the GPU since the last call (their tensors may be read from
        any stream); `peek` / `result` hand the tensors out."""
        done = [t for t, ev in self._done_ev.items() if ev.query()]
        for t in done:
            del self._done_ev[t]
        return done

    def peek(self, ticket):
        return self._results[ticket]

    def collect(self, tickets):
        """drain + synchronise + verify.  Returns {ticket: (coords, confs) or Exception}: a target
        whose prediction recorded a device-side fault (its outputs are NaN) is repeated alone through
        `Engine.predict_device_checked` - which falls back to the range-free convolution where that
        is the cure - and only if that fails too its entry is the exception.  One bad target never
        costs the others their results.  Once one repeat needed the range-free convolution the
        remaining repeats run in it directly (the weights, not the alignment, put a trunk outside the
        f16 range: the others would only fault again first), with one note on stderr for all of them."""
        self.drain()
        bits = 0
        for e in self.engines:
            bits |= e.sync_faults()
        out = {}
        eng = self.engines[0]
        mode0 = eng.get_option("conv_mode")
        try:
            for t in tickets:
                job = self._jobs.get(t)
                coords, confs = self.result(t)
                if bits and bool(torch.isnan(confs[0])):
                    _, d_msa, nloops, minsteps, d_tpl = job[:5]
                    try:
                        coords, confs = eng.predict_device_checked(d_msa, d_tpl, nloops, minsteps)
                        if eng.last_fallback:
                            eng.set_option("conv_mode", 2)
                        if not eng.get_option("vgru_persistent"):
                            # the repeat fell back to one vertical-GRU launch per row (another process holds CUs of
                            # this GPU): the other engines - and a group chain led by one of them - would only
                            # fault again, target after target
                            for other in self.engines:
                                other.set_option("vgru_persistent", 0)
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
        tickets = [self.submit(m, iterations, minsteps) for m in d_msas]
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
