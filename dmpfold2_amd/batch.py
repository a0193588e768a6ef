"""Batch front end: many independent alignments -> one PDB file each, sharded over the GPUs of a node.

    python -m dmpfold2_amd.batch -l targets.txt -o out_dir [-n 10] [-m 100] [-w weights.pt] [--streams 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m dmpfold2_amd.batch -l targets.txt -o out_dir

    python -m dmpfold2_amd.batch -i msa_dir more.aln other.a3m -o out_dir --format ca

`targets.txt` holds one alignment path per line (.aln, or .a3m converted as the reference's README
describes), optionally followed by a template PDB path; `-i` takes alignment files and directories
(every *.aln / *.a3m inside, sorted).  `--format`: `pdb` (default) = the text of the single-target CLI
(N, CA, C, O, CB records, confidence in the B-factor column and the REMARK line, predict.py:195-208),
`ca` = the same records for the CA atoms only, `npz` = coords (L,5,3), confs (L,), alnmat as arrays.  The
reference has no batch mode (its CLI takes one alignment, predict.py:160-208); this is the
"independent alignments shard embarrassingly" axis of SURVEY.md section 8e: every rank (one process
per GPU) takes the targets `shard.partition_targets` assigns to it (longest first), runs them through
a `Pipeline` (several targets in flight per GPU) and writes `<out_dir>/<alignment basename>.pdb` with
the text of the single-target CLI.  No collective on the data path; `torch.distributed` only sums the
target count and takes the maximum elapsed time for the summary line.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import hashlib
import os
import sys
import time

import numpy as np
import torch

from . import shard
from .predict import (MAX_SEQS, Pipeline, drop_in_precision, default_iterations, default_minsteps, encode_aln, load_state_dict,
                      pdb_text, read_a3m, read_aln, read_template_ca)


class BatchFailures(RuntimeError):
    """Some targets of a shard failed; the PDB files of all the others were written."""

    def __init__(self, failed, n_done, elapsed, outputs):
        self.failed, self.n_done, self.elapsed, self.outputs = failed, n_done, elapsed, outputs
        super().__init__(f"{len(failed)} targets failed (the {len(outputs)} PDB files of the others "
                         "were written): " + ", ".join(f"{a} ({type(e).__name__}: {e})" for a, e in failed))


def read_target_list(path):
    """[(alignment path, template path or None)] from a text file (blank lines and # comments skipped)."""
    out = []
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            parts = line.split()
            out.append((parts[0], parts[1] if len(parts) > 1 else None))
    return out


def expand_inputs(inputs):
    """Files and directories -> [(alignment path, None)]; a directory contributes its *.aln and *.a3m
    files in sorted order.  The reference README's workflow converts x.a3m to x.aln in the same
    directory: where a directory holds both, only x.aln is taken (both would be written to x.pdb)."""
    out = []
    for item in inputs:
        if os.path.isdir(item):
            names = sorted(f for f in os.listdir(item) if f.endswith((".aln", ".a3m")))
            have_aln = {f[:-4] for f in names if f.endswith(".aln")}
            out += [(os.path.join(item, f), None) for f in names
                    if f.endswith(".aln") or f[:-4] not in have_aln]
        else:
            out.append((item, None))
    return out


def check_output_stems(targets):
    """Two targets with the same basename would be written to the same <out_dir>/<stem> file (and, sharded,
    by two ranks at once): refuse the job before anything is predicted."""
    seen = {}
    for aln_path, _ in targets:
        stem = os.path.splitext(os.path.basename(aln_path))[0]
        if stem in seen and seen[stem] != aln_path:
            raise ValueError(f"targets {seen[stem]} and {aln_path} would both be written to {stem}.*: "
                             "rename one of them or run them into different output directories")
        seen[stem] = aln_path


def scan_target(aln_path):
    """(L, N) of an alignment WITHOUT parsing it: the length of its first sequence line and the number of lines that
    do not start with '>' (the rows the reference keeps, predict.py:100-104) - counted on the raw bytes, so a missing
    final newline or trailing blanks on the first line do not change the count (a size / line-length estimate came
    out one or more rows short for such files, and the engines were then built too small for their deepest
    target).  Every rank scans every target so that all ranks compute the same partition; only the owner of a target
    decodes and encodes it.  An unreadable file scans as (0, 0): its owner reports the error when it reads it."""
    a3m = aln_path.endswith(".a3m")
    lines = headers = size = 0
    head, last, prev_nl = b"", b"", True                  # prev_nl: the byte before this chunk was a newline (or none)
    try:
        with open(aln_path, "rb") as fh:
            while True:                                  # fixed-size chunks: a rank never holds a whole multi-GB a3m set
                chunk = fh.read(1 << 20)
                if not chunk:
                    break
                if len(head) < (1 << 16):
                    head += chunk[:(1 << 16) - len(head)]
                size += len(chunk)
                lines += chunk.count(b"\n")
                headers += chunk.count(b"\n>") + (1 if prev_nl and chunk.startswith(b">") and (size > len(chunk)) else 0)
                prev_nl = chunk.endswith(b"\n")
                last = chunk[-1:]
    except OSError:
        return 0, 0
    if not size:
        return 0, 0
    lines += 0 if last == b"\n" else 1
    headers += 1 if head.startswith(b">") else 0
    L = 0
    for raw in head.split(b"\n", 64):                    # the first sequence line is within the first few lines
        if not raw.startswith(b">"):
            L = sum(1 for ch in raw.rstrip() if not (a3m and 97 <= ch <= 122))
            break
    return L, max(1, lines - headers)


def plan_shard(targets, iterations, rank, world):
    """Indices of the targets rank `rank` of `world` owns, longest first.  Deterministic and identical on every
    rank (it depends only on the header scans), every target exactly once over the ranks."""
    costs = [shard.estimate_cost(L, N, iterations) for L, N in (scan_target(a) for a, _ in targets)]
    return shard.partition_targets(costs, world)[rank]


def ca_only_text(coords, confs, alnmat):
    """The CLI's PDB text reduced to its CA records (atom serial numbers renumbered 1..L)."""
    keep = [ln for ln in pdb_text(coords, confs, alnmat).split("\n") if not ln.startswith("ATOM") or ln[12:16] == " CA "]
    out, k = [], 0
    for ln in keep:
        if ln.startswith("ATOM"):
            k += 1
            ln = "ATOM   %4d" % k + ln[11:]
        out.append(ln)
    return "\n".join(out)


def write_result(out_dir, aln_path, coords, confs, alnmat, fmt="pdb"):
    stem = os.path.join(out_dir, os.path.splitext(os.path.basename(aln_path))[0])
    if fmt == "npz":
        path = stem + ".npz"
        np.savez_compressed(path, coords=coords.detach().cpu().numpy(), confs=confs.detach().cpu().numpy(),
                            alnmat=alnmat)
        return path
    path = stem + ".pdb"
    with open(path, "w") as fh:
        fh.write(pdb_text(coords, confs, alnmat) if fmt == "pdb" else ca_only_text(coords, confs, alnmat))
    return path


class _StaticQueue:
    """This rank's share of a static partition (shard.partition_targets on the header scans), longest first."""

    def __init__(self, indices):
        self._it = iter(indices)

    def take(self):
        return next(self._it, None)


class _SharedQueue:
    """One queue for all ranks: the targets in descending order of estimated cost, handed out by an atomic counter
    in the job's torch.distributed key-value store (host side; no collective, nothing on the data path).  A rank
    that finishes early simply keeps taking targets: no tail of idle GPUs behind a mis-estimated partition."""

    _calls = {}          # per process: how many queues were made over each target list (ranks call run_batch in step)

    def __init__(self, order, store, tag=""):
        # one counter per JOB: a second run_batch over the same store must not continue the first one's counter (it
        # would find it exhausted and silently predict nothing), so the key carries the target list's digest and the
        # number of earlier queues this process made over that list - the same on every rank of an SPMD job
        nth = _SharedQueue._calls.get(tag, 0)
        _SharedQueue._calls[tag] = nth + 1
        self._order, self._store, self._key = list(order), store, f"dmpfold_batch_next/{tag}/{nth}"

    def take(self):
        # an exhausted counter is a normal state, also at a rank's FIRST take (more ranks than targets, or a rank that
        # starts after the others have taken everything): that rank has nothing to do and must still reach the job's
        # summary reduction, so this never raises
        k = int(self._store.add(self._key, 1)) - 1
        return self._order[k] if k < len(self._order) else None


def cost_order(targets, iterations):
    """(indices in descending order of estimated cost, scans): identical on every rank."""
    scans = [scan_target(a) for a, _ in targets]
    costs = [shard.estimate_cost(L, N, iterations) for L, N in scans]
    return sorted(range(len(targets)), key=lambda i: (-costs[i], i)), scans


def run_batch(targets, out_dir, iterations=default_iterations, minsteps=default_minsteps,
              weights_file=None, state_dict=None, streams=4, device=None, rank=0, world=1, fmt="pdb", store=None):
    """Predict this rank's targets; returns (number done, seconds, [output paths]).

    Which targets those are: with `store` (a torch.distributed key-value store shared by the ranks of the job) every
    rank takes the next most expensive target from ONE shared queue whenever it has room; without it the static
    partition of `plan_shard`.  Either way a rank reads and encodes only the targets it takes, reading / encoding,
    the GPU and the writing of finished structures overlap (targets are submitted while others run and results are
    written as they complete), and a failing target never costs the others their results."""
    if fmt not in ("pdb", "ca", "npz"):
        raise ValueError(f"unknown output format {fmt!r} (pdb, ca, npz)")
    check_output_stems(targets)
    os.makedirs(out_dir, exist_ok=True)
    # sharding needs only (L, N) estimates from a header scan: a rank parses its own targets and nobody else's
    order, scans = cost_order(targets, iterations)
    if store is not None and world > 1:
        digest = hashlib.sha1("\n".join(f"{a}\t{t}" for a, t in targets).encode()).hexdigest()[:16]
        queue = _SharedQueue(order, store, f"{digest}.{iterations}.{minsteps}")
        mine_scans = scans                              # any target may come this way
    else:
        owned = plan_shard(targets, iterations, rank, world)
        queue = _StaticQueue(owned)
        mine_scans = [scans[i] for i in owned]
    if not mine_scans:
        return 0, 0.0, []
    # capacity of the engines from the scans (L and N are exact: scan_target counts the rows on the raw bytes)
    any_a3m = any(a.endswith(".a3m") for a, _ in targets)
    max_L = max(8, max(L for L, _ in mine_scans))
    max_N = MAX_SEQS if any_a3m else max(1, min(MAX_SEQS, max(N for _, N in mine_scans)))
    t0 = time.perf_counter()
    pipe, dev, copy_stream = None, None, None
    failed, outputs, parsed, faulted = [], [], {}, []
    n_taken = 0

    def ensure_pipe():
        nonlocal pipe, dev, copy_stream
        if pipe is None:
            dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
            sd = state_dict if state_dict is not None else load_state_dict(weights_file)
            pipe = Pipeline(dev, max_L, max_N, sd, streams=streams, precision=drop_in_precision())
            if dev.type == "cuda":
                # every copy of this front end goes through its own (non-blocking) stream: nothing is ever enqueued on
                # the process's default stream while the engines run
                copy_stream = torch.cuda.Stream(device=dev)
        return pipe

    done = []                                           # completed on the GPU AND copied to the host, not yet written
    copying = []                                        # (ticket, host coords, host confs, event): D2H in flight

    # The scheduler's thread never blocks on the GPU: a synchronous copy on the default stream can queue behind an
    # engine's kernels (streams share hardware queues) and would stall the thread - and with it every engine.
    # Alignments go up from pinned memory and results come back into pinned memory, both asynchronously on the front
    # end's own stream; completion is polled.  (Measured at the north-star size: files -> PDB files 6.2 structures/s
    # including the pipeline's set-up against 6.7 for the same targets resident in HBM, tools/batch_throughput.py.)
    def start_copy_back(t):
        coords, confs = pipe.peek(t)
        if coords.is_cuda:
            hc = torch.empty(coords.shape, dtype=coords.dtype, pin_memory=True)
            hf = torch.empty(confs.shape, dtype=confs.dtype, pin_memory=True)
            with (torch.cuda.stream(copy_stream) if copy_stream is not None else contextlib.nullcontext()):
                hc.copy_(coords, non_blocking=True)   # the prediction has completed (polled): no ordering needed
                hf.copy_(confs, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
        else:
            hc, hf, ev = coords, confs, None
        copying.append((t, hc, hf, ev))

    def reap_copies():
        while copying and (copying[0][3] is None or copying[0][3].query()):
            done.append(copying.pop(0)[:3])

    def finish(item):
        t, coords, confs = item
        aln_path, alnmat, _ = parsed.pop(t)
        if bool(torch.isnan(confs[:1]).any()):          # a device-side fault poisoned it: repeated alone at the end
            parsed[t] = (aln_path, alnmat, None)
            faulted.append(t)
            return
        pipe.result(t)
        outputs.append(write_result(out_dir, aln_path, coords, confs, alnmat, fmt))

    exhausted = False
    cap = 2 * max(1, int(streams))                      # started + queued per rank
    backlog_cap = max(1, int(streams))                  # queued, not yet started

    def take_one():
        """Read, encode and submit the next target of the queue; False when the queue is empty."""
        nonlocal exhausted, n_taken
        i = queue.take()
        if i is None:
            exhausted = True
            return False
        n_taken += 1
        aln_path, tpl_path = targets[i]
        try:
            rows = read_a3m(aln_path) if aln_path.endswith(".a3m") else read_aln(aln_path)
            tpl = read_template_ca(tpl_path) if tpl_path else None
            alnmat = encode_aln(rows)
            p = ensure_pipe()
            h_msa = torch.from_numpy(np.ascontiguousarray(alnmat))
            with (torch.cuda.stream(copy_stream) if copy_stream is not None else contextlib.nullcontext()):
                if dev.type == "cuda":
                    h_msa = h_msa.pin_memory()
                    d_msa = h_msa.to(dev, non_blocking=True)
                else:
                    d_msa = h_msa
                # the pinned source stays alive in `parsed` until the target is written; the engine that takes the
                # target orders its stream behind the stream that is current at submission (the copy stream)
                parsed[p.submit(d_msa, iterations, minsteps, template_ca=tpl)] = (aln_path, alnmat, h_msa)
        except (IndexError, ValueError, OSError, UnicodeDecodeError, RuntimeError) as exc:
            # unknown residue letter / ragged rows / unreadable file / larger than the scan said: this target only
            failed.append((aln_path, exc))
        return True

    def room():
        return (not exhausted and (pipe is None or pipe.backlog() < backlog_cap)
                and len(parsed) - len(faulted) - len(done) - len(copying) < cap)

    # Host work - reading / encoding / uploading the next alignment, bringing a finished structure back, formatting
    # and writing it - is done ONE item per scheduling round in which nothing could be issued (the GPU has work
    # queued), never while units are waiting to be issued: the targets of a group finish and start together, and
    # 4 x (0.6 + 1.5) ms of host work in front of the next group's first launches is GPU idle time.
    while True:
        if pipe is None or not pipe.busy():
            # nothing left to issue: refill from the queue, else write what has completed, else wait for the GPU
            while room():
                if not take_one():
                    break
            if pipe is None or not pipe.busy():
                if pipe is not None:
                    for t in pipe.poll():
                        start_copy_back(t)
                    reap_copies()
                if done:
                    finish(done.pop(0))
                    continue
                if len(parsed) > len(faulted):          # issued to the end, not yet completed on the GPU
                    time.sleep(0.0002)
                    continue
                if exhausted or not room():
                    break
                continue
        progressed = pipe.step()
        if not progressed:
            for t in pipe.poll():
                start_copy_back(t)
            reap_copies()
            if room():
                take_one()
            elif done:
                finish(done.pop(0))
    if copying:
        if copying[-1][3] is not None:
            copying[-1][3].synchronize()
        reap_copies()
    for item in done:
        finish(item)
    if faulted:
        res = pipe.collect(faulted)
        for t in faulted:
            aln_path, alnmat, _ = parsed.pop(t)
            if isinstance(res[t], Exception):
                failed.append((aln_path, res[t]))
            else:
                outputs.append(write_result(out_dir, aln_path, res[t][0], res[t][1], alnmat, fmt))
    elapsed = time.perf_counter() - t0
    if pipe is not None:
        pipe.close()
    if failed:
        raise BatchFailures(failed, len(outputs), elapsed, outputs)
    return n_taken, elapsed, outputs


def main(argv=None):
    ap = argparse.ArgumentParser(description="DMPfold2 batch prediction on AMD MI355X (one process per GPU)")
    ap.add_argument("-l", "--list", default=None, help="text file: one alignment path (+ optional template) per line")
    ap.add_argument("-i", "--input", nargs="*", default=[],
                    help="alignment files (.aln / .a3m) and directories of them")
    ap.add_argument("-o", "--out_dir", required=True)
    ap.add_argument("--format", choices=("pdb", "ca", "npz"), default="pdb",
                    help="pdb: the single-target CLI's text; ca: its CA records only; npz: arrays")
    ap.add_argument("-n", "--iterations", type=int, default=default_iterations)
    ap.add_argument("-m", "--minsteps", type=int, default=default_minsteps)
    ap.add_argument("-w", "--model_weights", type=str, default=None)
    ap.add_argument("--streams", type=int, default=4, help="targets in flight per GPU")
    ap.add_argument("--static-shards", action="store_true",
                    help="several ranks: fixed partition by estimated cost instead of the shared work queue")
    args = ap.parse_args(argv)
    if not args.list and not args.input:
        ap.error("give -l targets.txt and / or -i alignments ...")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        job_store = shard.job_store(rank, world)
        dist.init_process_group("nccl", store=job_store, rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if world > 1:
        shard.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    targets = (read_target_list(args.list) if args.list else []) + expand_inputs(args.input)
    status, n_failed, broke = 0, 0, 0
    store = job_store if (world > 1 and not args.static_shards) else None
    try:
        n, elapsed, _ = run_batch(targets, args.out_dir, args.iterations, args.minsteps,
                                  weights_file=args.model_weights, streams=args.streams,
                                  device=f"cuda:{local_rank}", rank=rank, world=world, fmt=args.format, store=store)
    except BatchFailures as bf:                      # keep going: the other ranks wait in job_summary
        for aln_path, exc in bf.failed:
            print(f"dmpfold-batch: {aln_path}: {type(exc).__name__}: {exc}", file=sys.stderr)
        n, elapsed, status, n_failed = bf.n_done, bf.elapsed, 1, len(bf.failed)
    except Exception as exc:                         # noqa: BLE001 - whatever went wrong on THIS rank, the others are
        # waiting in job_summary's reduction: report, take part in it, and fail the job through the exit status.  What
        # this rank had already taken from the shared queue is lost with it: the summary says so
        print(f"dmpfold-batch: rank {rank}: {type(exc).__name__}: {exc}", file=sys.stderr)
        n, elapsed, status, broke = 0, 0.0, 2, 1
    total, tmax, failed_all, broke_all = shard.job_summary(n, elapsed, failures=(n_failed, broke))
    if rank == 0:
        summary = {"targets": total, "seconds": tmax, "structures_per_s": total / tmax if tmax > 0 else 0.0,
                   "n_gpus": world, "failed_targets": failed_all, "failed_ranks": broke_all}
        if broke_all:
            summary["note"] = ("%d rank(s) broke down: the targets they had taken are missing from the output directory - "
                               "compare it with the target list" % broke_all)
        print(json.dumps(summary), flush=True)
    if (failed_all or broke_all) and status == 0:
        status = 1                                   # every rank of a job that lost targets fails, rank 0 included
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
