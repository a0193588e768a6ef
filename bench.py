#!/usr/bin/env python
"""Benchmark of the alignment -> structure hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cpu-baseline full|sample|none]

A "step" is one batch of complete predictions (features, sequence trunk, 11 pair-trunk passes with
recycling, MDS, coordinate GRU, 2 x 100 minimiser steps, backbone) of synthetic targets of
the north-star configuration L=300, N_seq=2000, iterations=10, minsteps=100, with the residue
codes already resident in HBM and the packed weights loaded (model construction / weight load is
excluded).  Every rank (one process per GPU) predicts its own targets -
independent alignments are the only parallel axis of this path, so there is no data-path
collective, only the timing barrier.  `--gpus N` with N > 1 starts the N ranks itself
(torch.distributed.run on 127.0.0.1) unless the process already runs under a launcher
(RANK / WORLD_SIZE in the environment, as the driver's command line does).

Two arithmetic flavours are measured, each for the full --steps with its own warm-up and its own HIP-event
intervals around every convolution launch:
  * `value` / `roofline`: option "precision" 0, the default - convolutions and vertical GRU form float32 products from
    two f16 pieces per operand (22 significand bits, float32 accumulate) on the f16 matrix cores -
    tolerance-qualified float32-GRADE arithmetic;
  * `value_f32` / `roofline_f32`: option "precision" 1 - the reference's own arithmetic type END TO END: the exact-f32 MFMA
    convolution (conv_mode 1, bitwise an fmaf chain) AND the float32 vertical GRU (v_mfma_f32_16x16x4_f32, library
    expf / tanhf gates; round 5) - no f16 / bf16 matrix-core kernel runs in this leg; priced against the 157.3 TFLOP/s
    f32 matrix-core peak.
Rank 0 prints ONE JSON line with both, a verification of the outputs against the reference's golden vectors for this
configuration (`verify`), and (N = 1 only) the CPU oracle timed on this host (`cpu_baseline`).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The scheduler keeps several HIP streams busy per GPU; the runtime multiplexes streams onto 4 hardware
# queues by default, which serialises a fifth stream (4 engines + the default stream): 5.3 structures/s
# with 4 engines on 4 queues, 6.5 on 8.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

L_NS, N_NS, ITERS, MINSTEPS = 300, 2000, 10, 100
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (not the 2:1 sparse figure)
CONV_FLOP_PER_LAUNCH = 2.0 * 128 * 512 * 25 * L_NS * L_NS     # one block's 5x5 conv
STUB = os.environ.get("DMP_BENCH_STUB") == "1"   # CPU test of the launch / reduction logic (gloo, no GPU work)


# ---------------------------------------------------------------------------------------------------
# launching N ranks
# ---------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n, argv, port):
    """The command line the driver itself uses for N > 1 (one process per GPU on this node)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks and hand their output through
    (rank 0 prints the JSON line).  Returns the exit code."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    return subprocess.call(launch_command(n, argv, free_port()), env=env)


# ---------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle on this host
# ---------------------------------------------------------------------------------------------------
def _cpu_threads():
    # PyTorch-CPU collapses when given every hardware thread of the 256-thread GPU host (37x
    # slower than 8 threads); a sweep there (tools/gpu_diag.py --cpu-sweep: 8/16/32/64/128)
    # put the optimum at 16 threads for the GRU and 16-32 for the convolutions.
    # DMP_CPU_THREADS overrides.
    return int(os.environ.get("DMP_CPU_THREADS", min(os.cpu_count() or 1, 16)))


def cpu_baseline_full(budget_s=150.0):
    """ONE complete prediction of the north-star workload by the CPU oracle (bench target 0: all 2000
    rows, 11 trunk passes, 2 x 100 minimiser steps), timed per stage.  Before it starts, a probe
    (vertical GRU on 64 rows, one residual block) estimates the total; if that exceeds `budget_s`
    the bounded-sample estimate is reported instead (returns None)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dmpfold_oracle as O
    from dmpfold2_amd import synth
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    sd = synth.synth_weights(0, coord_scale=5.0)
    W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    alnmat = O.encode_aln(synth.synth_msa(L_NS, N_NS, 0))
    with torch.no_grad():
        t0 = time.time()
        x = W["embed.weight"][torch.from_numpy(alnmat[:64].astype(np.int64))]
        O._gru(W, "vgru", x, 22, 512, 2, False, False)
        t_v = (time.time() - t0) * N_NS / 64.0
        xb = torch.randn(1, 128, L_NS, L_NS)
        t0 = time.time()
        O.block_finish(W, 1, O.block_conv(W, 1, xb), xb)
        t_b = (time.time() - t0) * 16 * (ITERS + 1)
    if t_v + t_b > budget_s:
        return None
    stages = {}

    def timed(name):
        fn = getattr(O, name)

        def wrapper(*a, **kw):
            t = time.time()
            try:
                return fn(*a, **kw)
            finally:
                stages[name] = stages.get(name, 0.0) + time.time() - t
        setattr(O, name, wrapper)
        return fn
    names = ["reweight", "fast_dca", "sequence_trunk", "pair_trunk", "mds_top8", "coords_from_mds",
             "refine_coords", "ca_to_backbone"]
    saved = {n: timed(n) for n in names}
    try:
        t0 = time.time()
        coords, confs = O.predict(alnmat, W, None, ITERS, MINSTEPS, "canonical")
        total = time.time() - t0
    finally:
        for n, fn in saved.items():
            setattr(O, n, fn)
    label = {"reweight": "reweight", "fast_dca": "fast_dca", "sequence_trunk": "vgru+hgru",
             "pair_trunk": "pair trunk x11", "mds_top8": "eigh x11", "coords_from_mds": "coord_gru x11",
             "refine_coords": "refine 2x100", "ca_to_backbone": "backbone"}
    return {"value": 1.0 / total, "unit": "structures/s", "cores": cores, "kind": "port",
            "seconds_per_structure": total,
            "stage_seconds": {label[k]: round(v, 3) for k, v in stages.items()},
            "sample": ("ONE complete prediction of bench target 0 (L=300, N=2000, 11 trunk passes, 2 x 100 "
                       "minimiser steps) by the CPU oracle (PyTorch-CPU port of the reference's operator "
                       "sequence) on %d threads of this host: %.1f s" % (cores, total))}


def cpu_baseline_sample():
    """Bounded sample of the NS workload, extrapolated linearly (used when the full run does not fit):
      vgru     first 250 of the 2000 alignment rows (x8; cost is linear in rows)
      features reweight + fast_dca on 500 of the 2000 rows for the covariance (x4 on that part)
               and the full 6300 x 6300 inverse
      trunk    1 of the 11 pair-trunk passes incl. MDS and coordinate GRU (x11)
    """
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dmpfold_oracle as O
    from dmpfold2_amd import synth
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    sd = synth.synth_weights(0, coord_scale=5.0)
    W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    alnmat = O.encode_aln(synth.synth_msa(L_NS, N_NS, 0))
    L = L_NS
    with torch.no_grad():
        t0 = time.time()
        sub = alnmat[:250]
        x = W["embed.weight"][torch.from_numpy(sub.astype(np.int64))]
        v = O._gru(W, "vgru", x, 22, 512, 2, False, False)
        t_vgru = (time.time() - t0) * (N_NS / 250.0)
        t0 = time.time()
        w500 = O.reweight(alnmat[:500])
        t_rw = (time.time() - t0) * (N_NS / 500.0) ** 2
        t0 = time.time()
        O.fast_dca(alnmat[:500], w500)
        t_dca500 = time.time() - t0
        t0 = time.time()
        torch.inverse(torch.eye(21 * L) + 0.01 * torch.rand(21 * L, 21 * L))
        t_inv = time.time() - t0
        # covariance GEMM scales with rows; the inverse and the relayouts do not
        t_dca = t_inv + (t_dca500 - t_inv) * 4.0 if t_dca500 > t_inv else t_dca500 * 4.0
        mat1d = O._gru(W, "hgru", v[-1].unsqueeze(1), 512, 256, 2, True, False)[:, 0].t().contiguous()
        f2d = torch.randn(L, L, 442) * 0.05
        pair = (mat1d.unsqueeze(1) * mat1d.unsqueeze(2)).unsqueeze(0)
        static = torch.cat((pair, f2d.permute(2, 0, 1).unsqueeze(0)), dim=1)
        dmap = torch.zeros(1, 1, L, L) - 1
        t0 = time.time()
        y = O.pair_trunk(W, torch.cat((static, dmap), dim=1))
        dm, conf, M = O.head_to_gram(y)
        mds = O.mds_top8(M, "canonical")
        ca = O.coords_from_mds(W, mat1d, mds)
        t_pass = time.time() - t0
        t0 = time.time()
        O.refine_coords(ca[0], 20)
        t_ref = (time.time() - t0) * (2 * MINSTEPS / 20.0)
    total = t_vgru + t_rw + t_dca + (ITERS + 1) * t_pass + t_ref
    return {"value": 1.0 / total, "unit": "structures/s", "cores": cores, "kind": "port",
            "seconds_per_structure": total,
            "sample": ("EXTRAPOLATED (a complete oracle run would exceed the time budget on this host): "
                       "vgru on 250/2000 rows x8, reweight+fast_dca on 500/2000 rows (GEMM part x4) + "
                       "full 6300^2 inverse, 1/11 trunk passes x11, 20/200 minimiser steps x10; est. "
                       "%.1f s per structure (vgru %.1f, features %.1f, trunk passes %.1f)"
                       % (total, t_vgru, t_rw + t_dca, (ITERS + 1) * t_pass))}


def cpu_baseline(mode):
    if mode == "full":
        r = cpu_baseline_full()
        if r is not None:
            return r
    return cpu_baseline_sample()


# ---------------------------------------------------------------------------------------------------
def interval_union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0.0, None, None
    for a_, b_ in iv:
        if cur_b is None or a_ > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a_, b_
        else:
            cur_b = max(cur_b, b_)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def stub_main(args, rank, world):
    """DMP_BENCH_STUB=1: the launch, barrier and max-over-ranks logic of this file over gloo with a
    stand-in workload (tests/test_host_cpu.py, world_size 2; no GPU)."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        seen = dist.get_world_size()
    t0 = time.perf_counter()
    time.sleep(0.02 * (rank + 1) * args.steps)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * args.steps / elapsed, "n_gpus": world,
                          "ranks": seen, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=4,
                    help="targets in flight per GPU (one context + HIP stream each)")
    ap.add_argument("--batch", type=int, default=0,
                    help="targets per step and GPU (default 2 x streams)")
    ap.add_argument("--cpu-baseline", choices=("full", "sample", "none"), default="full")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="same as --cpu-baseline none")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the exact-f32 convolution leg")
    ap.add_argument("--no-files-leg", action="store_true", help="skip the alignment files -> PDB files measurement (N = 1)")
    ap.add_argument("--legs", choices=("both", "f16x3", "f32"), default="both",
                    help="profiling: run only one arithmetic leg (f32 alone skips the verification and latency extras, "
                         "so that a kernel-stats table of the run holds that leg's launches only)")
    ap.add_argument("--vgru-per-row", action="store_true",
                    help="A/B: the vertical GRU as one launch per alignment row (round 3) instead of the persistent launch")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args.gpus, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if STUB:
        return stub_main(args, rank, world)

    import numpy as np
    import torch
    # DMP_FORCE_DIST=1 exercises the RCCL barrier / reduction path with a single rank (1-GPU boxes)
    distributed = world > 1 or (os.environ.get("DMP_FORCE_DIST") == "1" and "RANK" in os.environ)
    # DMP_BENCH_SHARE_GPU=1 (test of the N > 1 flow on a one-GPU box): every rank works on cuda:0 and the
    # barrier / reductions go over gloo (RCCL refuses two ranks on one device); the throughput is meaningless
    share_gpu = os.environ.get("DMP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    ranks_seen = 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        ranks_seen = dist.get_world_size()
    device = torch.device("cuda", local_rank)
    red_device = torch.device("cpu") if share_gpu else device          # where the reduced scalars live

    from dmpfold2_amd import synth, _lib, shard
    from dmpfold2_amd.predict import Pipeline, encode_aln
    lib = _lib.load()
    if world > 1 and not share_gpu:
        shard.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    S = max(1, args.streams)
    sd = synth.synth_weights(0, coord_scale=5.0)
    pipe = Pipeline(device, L_NS, N_NS, {k: torch.from_numpy(np.array(v)) for k, v in sd.items()},
                    streams=S)
    # two PROCESSES on one GPU (the share-GPU test mode): the persistent vertical GRU needs every CU of the device for
    # itself, and the per-device launch order that keeps a process's own co-resident kernels apart does not reach
    # across processes - one launch per row there
    if args.vgru_per_row or share_gpu:
        for e in pipe.engines:
            e.set_option("vgru_persistent", 0)

    # B synthetic targets per step and rank, all resident in HBM before the clock starts
    B = args.batch if args.batch > 0 else 2 * S
    total = (args.warmup + args.steps) * B
    targets = []
    for i in range(total):
        rows = synth.synth_msa(L_NS, N_NS, seed=100000 * rank + i)
        targets.append(torch.from_numpy(encode_aln(rows)).to(device))

    def sync_all():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(device)

    host_cpu = {}                                        # conv_mode -> host CPU-seconds per structure of the timed region

    def timed_leg(conv_mode):
        """W warm-up steps, then exactly K timed steps of the scheduler in the given convolution arithmetic, bracketed
        by barrier + synchronize; HIP events around every convolution launch of the timed region (recorded on the
        launching streams).  -> (elapsed s, warm-up outputs, timed outputs, launches, mean launch ms, union ms)"""
        for e in pipe.engines:
            e.set_option("precision", conv_mode)          # 0: split f16; 1: float32 convolutions AND float32 vertical GRU
            assert e.get_option("precision") == conv_mode and e.get_option("vgru_f32") == conv_mode
        warm = pipe.run(targets[:args.warmup * B], ITERS, MINSTEPS)
        sync_all()
        cap = 16 * (ITERS + 1) * (args.steps * B // S + 2)
        for e in pipe.engines:
            _lib.check(lib.dmp_profile_enable(e.ctx, 1, cap))
        cpu0 = time.process_time()                       # CPU time of this process, all threads
        t0 = time.perf_counter()
        # the steps are pipelined: step k+1 is queued as soon as every target of step k has started on an
        # engine; the clock stops when all K batches have completed (sync_all)
        tickets = []
        for k in range(args.steps):
            lo = (args.warmup + k) * B
            tickets += [pipe.submit(m, ITERS, MINSTEPS) for m in targets[lo:lo + B]]
            pipe.pump()
        pipe.drain()
        sync_all()
        el = time.perf_counter() - t0
        host_cpu[conv_mode] = (time.process_time() - cpu0) / max(1, len(tickets))
        timed = [pipe.result(t) for t in tickets]
        # the lane keeps two launches in flight: besides the per-launch duration, measure the time during
        # which at least one launch runs (union of the HIP-event intervals of all engines)
        iv = []
        for e in pipe.engines:
            a, b, n = (C.c_float * cap)(), (C.c_float * cap)(), C.c_int()
            _lib.check(lib.dmp_profile_conv_intervals(e.ctx, pipe.engines[0].ctx, a, b, cap, C.byref(n)))
            iv += [(a[i], b[i]) for i in range(n.value)]
        for e in pipe.engines:                           # (the record of engines[0] is the time origin of all of them)
            _lib.check(lib.dmp_profile_enable(e.ctx, 0, 0))
        union = interval_union(iv)
        tot, cnt = sum(b_ - a_ for a_, b_ in iv), len(iv)
        pipe.sync_check()
        for e in pipe.engines:
            e.set_option("precision", 0)
        return el, warm, timed, cnt, tot, union

    only_f32 = args.legs == "f32"
    if args.legs == "f16x3":
        args.no_exact_f32 = True
    if only_f32:
        # profiling run of the exact-f32 leg alone: its numbers go into the f32 keys, nothing else is measured
        exact, warm1, timed1, cnt1, tot1, union1 = timed_leg(1)
        if rank == 0:
            eff1 = union1 / cnt1 if cnt1 else 0.0
            print(json.dumps({"metric": "structures/s at L=300, N_seq=2000, 10 iters+100 min", "legs": "f32",
                              "value_f32": world * args.steps * B / exact, "ms_per_step_f32": exact / args.steps * 1e3,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "roofline_f32": {"achieved": CONV_FLOP_PER_LAUNCH / (eff1 * 1e-3) / 1e12 if eff1 else 0.0,
                                               "peak": PEAK_F32_MFMA_TFLOPS, "chip_ms_per_launch": eff1,
                                               "avg_launch_ms": tot1 / cnt1 if cnt1 else 0.0, "launches_timed": cnt1}}), flush=True)
        if distributed:
            dist.destroy_process_group()
        return 0
    elapsed, outs, timed_outs, conv_cnt, conv_tot, conv_union = timed_leg(0)
    outs += timed_outs
    conv_ms = conv_tot / conv_cnt if conv_cnt else 0.0
    pipe.sync_check()
    ok = all(bool(torch.isfinite(c).all()) and bool(torch.isfinite(f).all()) for c, f in outs)

    # ---- verification of the outputs (untimed) ---------------------------------------------------
    verify = {"finite_outputs": ok}
    e0 = pipe.engines[0]
    first = args.warmup * B                                   # first target of the timed region
    if timed_outs:
        # (1) the scheduler's result of that target == the same target alone on one engine, bit for bit
        c1, f1 = e0.predict_device(targets[first], None, ITERS, MINSTEPS)
        e0.sync_check()
        verify["timed_target_bitwise_equals_single_engine"] = bool(
            torch.equal(c1, timed_outs[0][0]) and torch.equal(f1, timed_outs[0][1]))
        ok = ok and verify["timed_target_bitwise_equals_single_engine"]
        # (2) digest of bench target 0 (seed 0, the full 10 + 100 prediction, alone on one engine - the same
        #     bits the scheduler delivers, by (1)) against the stored one (profiles/bench_digest.json, written
        #     by DMP_WRITE_DIGEST=1); the results are bit-reproducible, so a change means the arithmetic changed.
        #     The key does not depend on --steps / --warmup.
        if rank == 0:
            c0, f0 = e0.predict_device(targets[0], None, ITERS, MINSTEPS)
            e0.sync_check()
            digest = hashlib.sha256(c0.cpu().numpy().tobytes() + f0.cpu().numpy().tobytes()).hexdigest()
            verify["digest"] = digest
            dpath = os.path.join(ROOT, "profiles", "bench_digest.json")
            key = f"L{L_NS}_N{N_NS}_n{ITERS}_m{MINSTEPS}_seed0"
            if not e0.get_option("vgru_persistent"):         # the launch-per-row chain sums K in another order: its own bits
                key += "_vgru_per_row"
            stored = {}
            if os.path.exists(dpath):
                try:
                    stored = json.load(open(dpath))
                except Exception:
                    stored = {}
            if os.environ.get("DMP_WRITE_DIGEST") == "1":
                stored = {key: digest}
                json.dump(stored, open(dpath, "w"), indent=1, sort_keys=True)
            verify["digest_expected"] = stored.get(key)
            verify["digest_match"] = (stored.get(key) == digest) if key in stored else None
            if verify["digest_match"] is False:          # the arithmetic changed: not a valid headline run
                ok = False
    if rank == 0:
        # (3) bench target 0 (seed 0) at iterations=1, minsteps=0 against the vector captured from the
        #     reference itself at this size (tests/golden/synth_L300_N2000_n1_m0.npz; data only)
        gpath = os.path.join(ROOT, "tests", "golden", "synth_L300_N2000_n1_m0.npz")
        if os.path.exists(gpath):
            g = np.load(gpath)
            t0msa = targets[0]                          # synth_msa(300, 2000, seed 0) on rank 0
            sha = hashlib.sha256(t0msa.cpu().numpy().tobytes()).hexdigest()
            gc, gf = e0.predict_device(t0msa, None, 1, 0)
            e0.sync_check()
            d = gc.cpu().numpy()[:, 1].astype(np.float64) - g["coords"][:, 1].astype(np.float64)
            rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
            dconf = float(np.abs(gf.cpu().numpy() - g["confs"]).max())
            same_input = sha == bytes(g["alnmat_sha256"]).decode()
            verify["reference_golden_L300_N2000_n1_m0"] = {
                "ca_rmsd_A": rmsd, "max_dconf": dconf, "same_input": same_input,
                "ok": bool(same_input and rmsd <= 1e-3 and dconf < 1e-4)}
            ok = ok and verify["reference_golden_L300_N2000_n1_m0"]["ok"]
        # (4) THE METRIC'S CONFIGURATION ITSELF against the reference: bench target 0 at iterations=10,
        #     minsteps=100 on the weight set on which the reference is stable there (coord_fc fitted to a
        #     protein-like trace, MDS feedback attenuated: tests/golden/fitns_L300_N2000_n10_m100.npz holds the
        #     reference's output, its thread-count floor and the fitted matrix; data only).  Own engine: the
        #     pipeline's weights stay untouched.
        gpath = os.path.join(ROOT, "tests", "golden", "fitns_L300_N2000_n10_m100.npz")
        if os.path.exists(gpath):
            from dmpfold2_amd.predict import Engine
            g = np.load(gpath)
            sdh = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]))
            eh = Engine(device, L_NS, N_NS)
            if share_gpu or args.vgru_per_row:
                eh.set_option("vgru_persistent", 0)        # (two processes on this GPU: see above)
            try:
                eh.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sdh.items()})
                gc, gf = eh.predict_device(targets[0], None, ITERS, MINSTEPS)
                eh.sync_check()
                d = gc.cpu().numpy()[:, 1].astype(np.float64) - g["coords"][:, 1].astype(np.float64)
                rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
                dconf = float(np.abs(gf.cpu().numpy() - g["confs"]).max())
                floor = float(g["noise_ca_rmsd"])
                same = (hashlib.sha256(targets[0].cpu().numpy().tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
                        and synth.weights_checksum(sdh) == bytes(g["weights_sha256"]).decode())
                verify["reference_golden_L300_N2000_n10_m100"] = {
                    "ca_rmsd_A": rmsd, "max_dconf": dconf, "reference_thread_noise_A": floor, "same_input": bool(same),
                    "ok": bool(same and rmsd <= max(1e-3, 3.0 * floor) and dconf < 1e-4)}
                ok = ok and verify["reference_golden_L300_N2000_n10_m100"]["ok"]
            finally:
                eh.close()

    # ---- host side of one aln_to_coords call (SURVEY 8d's unit of work; outside the timed region, whose
    #      inputs are resident in HBM): alignment file -> rows -> residue codes -> device, and the PDB text
    #      of a result brought back to the host.  Reported, never part of `value`.
    host = None
    if rank == 0 and timed_outs:
        import tempfile
        from dmpfold2_amd.predict import read_aln, pdb_text
        rows0 = synth.synth_msa(L_NS, N_NS, seed=0)
        with tempfile.NamedTemporaryFile("w", suffix=".aln", delete=False) as fh:
            fh.write("\n".join(rows0) + "\n")
        reps = 5
        t_read = t_enc = t_h2d = t_pdb = 0.0
        for _ in range(reps):
            t = time.perf_counter(); rows = read_aln(fh.name); t_read += time.perf_counter() - t
            t = time.perf_counter(); am = encode_aln(rows); t_enc += time.perf_counter() - t
            t = time.perf_counter(); d = torch.from_numpy(am).to(device); torch.cuda.synchronize(device)
            t_h2d += time.perf_counter() - t
            t = time.perf_counter()
            text = pdb_text(timed_outs[0][0], timed_outs[0][1], am)
            t_pdb += time.perf_counter() - t
        os.unlink(fh.name)
        host = {"read_aln_ms": t_read / reps * 1e3, "encode_ms": t_enc / reps * 1e3,
                "h2d_ms": t_h2d / reps * 1e3, "d2h_and_pdb_text_ms": t_pdb / reps * 1e3,
                "total_ms": (t_read + t_enc + t_h2d + t_pdb) / reps * 1e3,
                "note": "host work of one target (file -> codes in HBM, result -> PDB text), single thread, "
                        "outside the timed region; the batch front end overlaps it with the GPU"}

    # ---- latency of ONE prediction of the same configuration on an engine of its own (the reference's use case: one
    #      CLI call; the scheduler above is the throughput mode).  Reported beside `value`, never part of it.
    single = None
    if rank == 0 and timed_outs:
        from dmpfold2_amd.predict import Engine
        es = Engine(device, L_NS, N_NS)
        if share_gpu or args.vgru_per_row:
            es.set_option("vgru_persistent", 0)        # (two processes on this GPU: see above)
        try:
            es.share_weights(pipe.engines[0])
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(device)
                t = time.perf_counter()
                cs, fs = es.predict_device(targets[first], None, ITERS, MINSTEPS)
                es.sync_check()
                ts.append((time.perf_counter() - t) * 1e3)
            # the bits are compared with the scheduler's kernels selected (its engines tridiagonalise with one launch per
            # Householder step; the cluster launch of a lone engine is the same algorithm with its float64 sums
            # associated differently - the same bits on almost every matrix, not on all: round 4 found a target whose
            # minimised trace tells them apart)
            es.set_option("tridiag_cluster", 0)
            cb, fb = es.predict_device(targets[first], None, ITERS, MINSTEPS)
            es.sync_check()
            single = {"ms": min(ts[1:]), "runs_ms": ts,
                      "bitwise_equals_the_scheduler": bool(torch.equal(cb, timed_outs[0][0]) and torch.equal(fb, timed_outs[0][1])),
                      "cluster_tridiagonalisation_same_bits": bool(torch.equal(cs, cb) and torch.equal(fs, fb)),
                      "note": "one target alone on one engine of its own (single stream; the first run builds the launch "
                              "graphs; timed with the cluster tridiagonalisation, compared bit for bit with the scheduler's "
                              "first timed result with the scheduler's per-step tridiagonalisation)"}
            ok = ok and single["bitwise_equals_the_scheduler"]
        finally:
            es.close()

    # ---- the same workload, same K steps and W warm-up steps, with the exact-f32 MFMA convolution (conv_mode 1): the
    #      reference's own arithmetic type, measured exactly like the headline leg
    exact = None
    f32_leg = None
    if not args.no_exact_f32:
        exact, warm1, timed1, cnt1, tot1, union1 = timed_leg(1)
        ok = ok and all(bool(torch.isfinite(c).all()) and bool(torch.isfinite(f).all()) for c, f in warm1 + timed1)
        f32_leg = (cnt1, tot1, union1)
        if timed1 and timed_outs:
            d = (timed1[0][0][:, 1].double() - timed_outs[0][0][:, 1].double())
            verify["f32_vs_f16x3_first_timed_target"] = {
                "ca_rmsd_A": float((d ** 2).sum(-1).mean().sqrt()),
                "max_dconf": float((timed1[0][1] - timed_outs[0][1]).abs().max()),
                "note": "the two arithmetic flavours on the same target (10 + 100 on random weights: the minimiser on a "
                        "collapsed trace amplifies rounding differences; informational)"}

        # the float32 leg's own check against the reference: bench target 0 at iterations=1, minsteps=0 in precision 1
        gpath = os.path.join(ROOT, "tests", "golden", "synth_L300_N2000_n1_m0.npz")
        if rank == 0 and os.path.exists(gpath):
            g = np.load(gpath)
            e0.set_option("precision", 1)
            try:
                gc, gf = e0.predict_device(targets[0], None, 1, 0)
                e0.sync_check()
            finally:
                e0.set_option("precision", 0)
            d = gc.cpu().numpy()[:, 1].astype(np.float64) - g["coords"][:, 1].astype(np.float64)
            rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
            dconf = float(np.abs(gf.cpu().numpy() - g["confs"]).max())
            verify["reference_golden_L300_N2000_n1_m0_precision1"] = {
                "ca_rmsd_A": rmsd, "max_dconf": dconf, "ok": bool(rmsd <= 1e-3 and dconf < 1e-4)}
            ok = ok and verify["reference_golden_L300_N2000_n1_m0_precision1"]["ok"]

    if distributed:
        vals = [elapsed, exact if exact is not None else 0.0]
        t = torch.tensor(vals, dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, el1 = float(t[0].item()), float(t[1].item())
        exact = el1 if exact is not None else None
        flag = torch.tensor([1.0 if ok else 0.0], device=red_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0.5)

    if rank == 0:
        # chip-level rate of the kernel: the launches' algorithmic FLOPs over the time at least one of them
        # runs.  With one launch at a time this is FLOP per launch / average launch duration; with the
        # lane's two launches in flight each launch lasts about twice its share of the chip
        # (avg_launch_ms is the raw per-launch duration the rocprofv3 kernel trace shows).
        eff_ms = conv_union / conv_cnt if conv_cnt else 0.0
        in_flight = conv_tot / conv_union if conv_union > 0 else 0.0
        achieved = CONV_FLOP_PER_LAUNCH / (eff_ms * 1e-3) / 1e12 if eff_ms > 0 else 0.0
        traffic, traffic_current = None, None
        pmc = os.path.join(ROOT, "profiles", "conv5x5_pmc.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                # the PMC passes are a separate profiler run (counters cannot be collected inside the timed region):
                # the file carries the hash of the kernel source it was taken from
                if pj.get("kernel_source_sha256"):
                    src = os.path.join(ROOT, pj.get("kernel_source", "dmpfold2_amd/csrc/conv_f16.h"))
                    traffic_current = hashlib.sha256(open(src, "rb").read()).hexdigest() == pj["kernel_source_sha256"]
            except Exception:
                traffic = None
        verify["ok"] = ok
        # what the matrix cores sustain under the power cap on random operands (measured, not a spec figure):
        # reported beside the spec-peak fraction, never instead of it
        capped = None
        cpath = os.path.join(ROOT, "profiles", "mfma_power_ceiling.json")
        if os.path.exists(cpath):
            try:
                cj = json.load(open(cpath))
                capped = {"mfma_f16_random_operands_tflops": cj["random_f16_operands_tflops"],
                          "executed_f16_frac_of_it": 3.0 * achieved / cj["random_f16_operands_tflops"],
                          "source": "profiles/mfma_power_ceiling.json (tools/ubench_mfma_power.hip: register-"
                                    "resident operands, no memory traffic; zeros reach the 2.5 PFLOP/s spec peak, "
                                    "random f16 data 1.6 at the 1.3 kW power cap)"}
            except Exception:
                capped = None
        line = {
            "metric": "structures/s at L=300, N_seq=2000, 10 iters+100 min",
            "value": world * args.steps * B / elapsed,
            "unit": "structures/s",
            "n_gpus": world,
            "ranks": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # `value` / `roofline`: float32-GRADE products from two f16 pieces per operand (22 significand bits) on the f16
            # matrix cores, float32 accumulate - NOT the f32 instruction; `value_f32` / `roofline_f32` (below) are the
            # same workload in exact f32 MFMA arithmetic, the reference's own type
            "dtype": "f32-grade: 2xf16 split products (22-bit operands) in the convolutions and the vertical GRU, f32 "
                     "accumulate; the reference's f32 arithmetic end to end = value_f32",
            "data": "synthetic",
            "finite_outputs": ok,
            "verify": verify,
            "config": {"workload": "synthetic targets L=300 N_seq=2000, iterations=10, minsteps=100 "
                                   "(BASELINE.json metric config); one step = one batch of "
                                   f"{B} independent targets per GPU",
                       "L": L_NS, "n_seq": N_NS, "iterations": ITERS, "minsteps": MINSTEPS,
                       "targets_per_step_per_gpu": B, "streams_per_gpu": S,
                       "weights": "synthetic seed 0 (reference state_dict shapes)",
                       "parallelism": f"replicas x{world}, no collective on the data path; "
                                      f"{S} HIP streams per GPU"},
            # The convolution forms each float32 product from 2-way f16 splits of its operands: 3 f16
            # MFMA products per float32 product, so the matrix-core ceiling for the ALGORITHMIC
            # (float32) FLOPs is the dense f16 peak / 3.  The exact-f32 MFMA path (conv_mode 1) is bounded by
            # PEAK_F32_MFMA_TFLOPS: `value_f32` / `roofline_f32`.
            "roofline": {"kernel": "conv5x5_f16x3_kernel (5x5 conv 128->512 + bias + 4-way maxout, "
                                   "float32 products from 3 f16 MFMA products, float32 accumulate)",
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
                         "unit": "TFLOP/s", "frac": achieved / (PEAK_F16_MFMA_TFLOPS / 3.0),
                         "traffic": traffic,
                         "traffic_source": "profiles/conv5x5_pmc.json (rocprofv3 --pmc passes of single "
                                           "launches, tools/profile_r05.sh; a separate profiler run)",
                         "traffic_taken_from_this_kernel_source": traffic_current,
                         "launches_timed": conv_cnt,
                         "avg_launch_ms": conv_ms, "launches_in_flight": in_flight,
                         "chip_ms_per_launch": eff_ms,
                         "algorithmic_flop_per_launch": CONV_FLOP_PER_LAUNCH,
                         "executed_f16_tflops": 3.0 * achieved,
                         "power_capped_ceiling": capped,
                         "peak_f16_mfma_tflops": PEAK_F16_MFMA_TFLOPS,
                         "peak_f32_mfma_tflops": PEAK_F32_MFMA_TFLOPS},
        }
        # what the scheduler costs the host: CPU time of the whole process (scheduler thread, runtime helper threads)
        # over the timed region per structure; the scheduler sleeps on blocking-sync events when nothing can be issued
        line["host_cpu_s_per_structure"] = host_cpu.get(0)
        if 1 in host_cpu:
            line["host_cpu_s_per_structure_f32"] = host_cpu.get(1)
        if host is not None:
            line["host_ms_per_target"] = host
        if single is not None:
            line["single_target"] = single
        if exact is not None:
            cnt1, tot1, union1 = f32_leg
            eff1 = union1 / cnt1 if cnt1 else 0.0
            ach1 = CONV_FLOP_PER_LAUNCH / (eff1 * 1e-3) / 1e12 if eff1 > 0 else 0.0
            traffic1, traffic1_current = None, None
            pmc1 = os.path.join(ROOT, "profiles", "conv5x5_f32_pmc.json")
            if os.path.exists(pmc1):
                try:
                    pj = json.load(open(pmc1))
                    traffic1 = pj.get("hbm_bytes_per_launch")
                    if pj.get("kernel_source_sha256"):
                        src = os.path.join(ROOT, pj.get("kernel_source", "dmpfold2_amd/csrc/trunk.hip"))
                        traffic1_current = hashlib.sha256(open(src, "rb").read()).hexdigest() == pj["kernel_source_sha256"]
                except Exception:
                    traffic1 = None
            v = world * args.steps * B / exact
            line["value_f32"] = v
            line["ms_per_step_f32"] = exact / args.steps * 1e3
            line["dtype_f32"] = ("f32 end to end (option precision = 1): convolutions on v_mfma_f32_32x32x2_f32 (bitwise an "
                                 "fmaf chain), vertical GRU on v_mfma_f32_16x16x4_f32 with library expf / tanhf gates; no "
                                 "f16 / bf16 matrix-core kernel runs in this leg")
            line["roofline_f32"] = {
                "kernel": "conv5x5_maxout_kernel (5x5 conv 128->512 + bias + 4-way maxout on the f32 matrix cores)",
                "bound": "mfma", "achieved": ach1, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach1 / PEAK_F32_MFMA_TFLOPS, "traffic": traffic1,
                "traffic_source": "profiles/conv5x5_f32_pmc.json (rocprofv3 --pmc passes of single launches; a separate "
                                  "profiler run)",
                "traffic_taken_from_this_kernel_source": traffic1_current,
                "launches_timed": cnt1, "avg_launch_ms": tot1 / cnt1 if cnt1 else 0.0,
                "launches_in_flight": tot1 / union1 if union1 > 0 else 0.0, "chip_ms_per_launch": eff1,
                "algorithmic_flop_per_launch": CONV_FLOP_PER_LAUNCH,
                "whole_job_conv_tflops": v / world * 16 * (ITERS + 1) * CONV_FLOP_PER_LAUNCH / 1e12,
                "note": "same workload, scheduler, --steps and --warmup as `value`, option precision = 1 (conv_mode 1 + "
                        "float32 vertical GRU); "
                        "achieved = algorithmic FLOP per launch / chip time per launch (union of the HIP-event "
                        "intervals / launches), as for `roofline`"}
        # ---- SURVEY 8d's unit of work is one aln_to_coords call: the whole chain alignment FILES -> PDB FILES through the
        #      batch front end (read + encode + H2D + prediction + D2H + PDB text + write overlapped with the GPU by
        #      dmpfold2_amd.batch), same configuration, on a pipeline of its own that takes over the timed pipeline's
        #      streams.  Reported beside `value` (whose inputs are resident in HBM), never instead of it.
        if world == 1 and not args.no_files_leg and not only_f32:
            import shutil
            import tempfile
            from dmpfold2_amd import batch
            pipe.close()
            tmpd = tempfile.mkdtemp(prefix="dmp_bench_files_")
            try:
                nfiles = 3 * B
                paths = []
                for i in range(nfiles):
                    paths.append(os.path.join(tmpd, "t%03d.aln" % i))
                    synth.write_aln(paths[-1], synth.synth_msa(L_NS, N_NS, seed=700000 + i))
                tw = time.perf_counter()
                nb, secs, outs_f = batch.run_batch([(a, None) for a in paths], os.path.join(tmpd, "out"), ITERS, MINSTEPS,
                                                   state_dict={k: torch.from_numpy(np.array(v)) for k, v in sd.items()},
                                                   streams=S, device=str(device))
                wall = time.perf_counter() - tw
                line["files_to_pdb"] = {
                    "structures_per_s": nb / wall, "targets": nb, "seconds": wall,
                    "pdb_files_written": sum(1 for o in outs_f if os.path.getsize(o) > 0),
                    "note": "alignment files -> PDB files through dmpfold2_amd.batch.run_batch (one call: pipeline set-up - "
                            "contexts, one packed copy of the weights - reading, encoding, H2D, prediction, D2H, PDB text "
                            "and writing all inside the clock; host work overlapped with the GPU); `value` has its inputs "
                            "resident in HBM and stops at tensors on the device"}
            finally:
                shutil.rmtree(tmpd, ignore_errors=True)
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            line["cpu_baseline"] = cpu_baseline(mode)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    # a failed verification (non-finite output, scheduler != single engine, digest or reference-golden mismatch)
    # still prints the line - with verify.ok false - but the process fails
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
