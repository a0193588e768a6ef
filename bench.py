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

Three arithmetic settings of the same workload and scheduler are measured, each with its own warm-up and its own
HIP-event intervals around every convolution launch (recorded on the launching streams):
  * `value` / `roofline` / `dtype` - THE HEADLINE: option "precision" 2, full-width operands at the 16-bit matrix cores'
    rate.  Every float32 operand of the convolutions is split EXACTLY into three bf16 pieces (3 x 8 = the 24 significand
    bits of a float32), the six piece products above 2^-24 are accumulated in float32 (conv5x5_bf16x6_kernel), and the
    vertical GRU's three products are computed the same way (vgru_persist_x3_kernel) with library gate functions.
    Priced against the dense bf16 peak / 6.
    The full --steps / --warmup;
  * `value_f32` / `roofline_f32`: option "precision" 1 - the reference's instruction for instruction: the f32 MFMA
    convolution (bitwise an fmaf chain) and the float32-MFMA vertical GRU; no 16-bit matrix-core kernel runs in this leg;
    priced against the 157.3 TFLOP/s f32 matrix-core peak;
  * `value_split_f16` / `roofline_split_f16`: option "precision" 0, the FAST mode - two f16 pieces per operand (22-23
    significand bits: narrower than float32's 24, so this number is never the metric), 3 f16 MFMA products.
  The two secondary legs run a quarter of the steps (at least 2, one warm-up step): they are reported beside the
  headline, not instead of it.
Rank 0 prints ONE JSON line with all three, a verification of the outputs against the reference's golden vectors for
this configuration in every setting (`verify`), and (N = 1 only) the CPU oracle timed on this host (`cpu_baseline`).
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

# The arithmetic settings (option "precision" of include/dmpfold_hip.h).  `peak` prices the ALGORITHMIC float32 FLOPs of
# the convolution: the dense 16-bit matrix-core peak over the number of piece products per float32 product.
LEGS = {
    "bf16x3": {"precision": 2, "sfx": "", "peak": PEAK_F16_MFMA_TFLOPS / 6.0, "products": 6,
               "kernel": "conv5x5_bf16x6_kernel (5x5 conv 128->512 + bias + 4-way maxout; every float32 operand as three "
                         "bf16 pieces = 24 significand bits, six bf16 MFMA products per float32 product, float32 accumulate)",
               "dtype": "f32 with full-width operands on the bf16 matrix cores (option precision = 2): convolution operands "
                        "split exactly into 3 bf16 pieces (24 significand bits), the 6 piece products above 2^-24 accumulated "
                        "in f32 by v_mfma_f32_32x32x16_bf16; vertical GRU the same way (3 bf16 pieces per operand, 6 products on "
                        "v_mfma_f32_16x16x32_bf16) with library expf / tanhf; "
                        "everything else f32 (f64 statistics and eigensolver)",
               "pmc": "conv5x5_bf16_pmc.json", "src": "dmpfold2_amd/csrc/conv_bf16.h"},
    "f32": {"precision": 1, "sfx": "_f32", "peak": PEAK_F32_MFMA_TFLOPS, "products": 1,
            "kernel": "conv5x5_maxout_kernel (5x5 conv 128->512 + bias + 4-way maxout on the f32 matrix cores)",
            "dtype": "f32 end to end (option precision = 1): convolutions on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain), "
                     "vertical GRU on v_mfma_f32_16x16x4_f32 with library expf / tanhf gates; no f16 / bf16 matrix-core "
                     "kernel runs in this leg",
            "pmc": "conv5x5_f32_pmc.json", "src": "dmpfold2_amd/csrc/trunk.hip"},
    "f16x2": {"precision": 0, "sfx": "_split_f16", "peak": PEAK_F16_MFMA_TFLOPS / 3.0, "products": 3,
              "kernel": "conv5x5_f16x3_kernel (5x5 conv 128->512 + bias + 4-way maxout, float32 products from 3 f16 MFMA "
                        "products of 2 f16 pieces per operand, float32 accumulate)",
              "dtype": "f32-GRADE, NOT the metric's arithmetic (option precision = 0, the fast mode): 2 x f16 split products "
                       "(22-23-bit operands) in the convolutions and the vertical GRU, f32 accumulate",
              "pmc": "conv5x5_pmc.json", "src": "dmpfold2_amd/csrc/conv_f16.h"},
}
HEADLINE = "bf16x3"
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


def thread_cpu_snapshot():
    """CPU seconds of every thread of this process: {tid: (name, utime + stime)} from /proc (who burns the host CPU the
    process reports: the scheduler thread inside the library, or the HIP runtime's own)."""
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open(f"/proc/self/task/{tid}/stat").read()
                name = raw[raw.index("(") + 1:raw.rindex(")")]
                f = raw[raw.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)
            except (OSError, ValueError):
                pass
    except (OSError, ValueError):
        pass
    return out


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


def all_threads_probe(cores16):
    """The host's EVERY hardware thread beside the `cores16` the baseline uses ("the node's own host cores"
    reads as all of them): the same bounded probe with both thread counts - the vertical GRU on 64 of the 2000 rows and
    one residual block - extrapolated to a structure.  PyTorch-CPU collapses on the 256-thread hosts when given every
    thread, which is why the full run uses 16."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dmpfold_oracle as O
    from dmpfold2_amd import synth
    W = {k: torch.from_numpy(np.array(v)) for k, v in synth.synth_weights(0, coord_scale=5.0).items()}
    alnmat = O.encode_aln(synth.synth_msa(L_NS, N_NS, 0))
    out = {}
    every = os.cpu_count() or 1
    for n in sorted({cores16, every}):
        torch.set_num_threads(n)
        with torch.no_grad():
            x = W["embed.weight"][torch.from_numpy(alnmat[:64].astype(np.int64))]
            t0 = time.time()
            O._gru(W, "vgru", x, 22, 512, 2, False, False)
            t_v = (time.time() - t0) * N_NS / 64.0
            xb = torch.randn(1, 128, L_NS, L_NS)
            t0 = time.time()
            O.block_finish(W, 1, O.block_conv(W, 1, xb), xb)
            t_b = (time.time() - t0) * 16 * (ITERS + 1)
        out[str(n)] = {"threads": n, "est_seconds_per_structure": t_v + t_b, "vgru_s": t_v, "pair_trunk_s": t_b}
    torch.set_num_threads(cores16)
    return {"probe": "vertical GRU on 64 of 2000 rows (x 31.25) + one residual block (x 176), per thread count",
            "by_threads": out, "all_threads": every}


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
    ap.add_argument("--no-exact-f32", "--headline-only", dest="headline_only", action="store_true",
                    help="skip the two secondary arithmetic legs (precision 1 and 0)")
    ap.add_argument("--no-files-leg", action="store_true", help="skip the alignment files -> PDB files measurement (N = 1)")
    ap.add_argument("--legs", choices=("all",) + tuple(LEGS), default="all",
                    help="profiling: run ONE arithmetic leg only, with the full --steps (no verification or latency extras, "
                         "so that a kernel-stats table of the run holds that leg's launches only)")
    ap.add_argument("--vgru-per-row", action="store_true",
                    help="A/B: the vertical GRU as one launch per alignment row instead of the persistent launch")
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
    from dmpfold2_amd.predict import Pipeline, Engine, encode_aln
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
    per_row = args.vgru_per_row or share_gpu
    if per_row:
        pipe.set_option("vgru_persistent", 0)

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

    def timed_leg(name, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` timed steps of the scheduler in the leg's arithmetic, bracketed by
        barrier + synchronize; HIP events around every convolution launch of the timed region (recorded on the launching
        streams).  The steps are pipelined: step k+1 is queued as soon as every target of step k has started on an engine;
        the clock stops when all batches have completed."""
        prec = LEGS[name]["precision"]
        pipe.set_option("precision", prec)
        for e in pipe.engines:
            assert e.get_option("precision") == prec and e.get_option("vgru_f32") == prec
        warm = pipe.run(targets[:warmup * B], ITERS, MINSTEPS)
        sync_all()
        cap = 16 * (ITERS + 1) * (steps * B // S + 2)
        for e in pipe.engines:
            _lib.check(lib.dmp_profile_enable(e.ctx, 1, cap))
        cpu0 = time.process_time()                       # CPU time of this process, all threads
        sched0 = pipe.stats()["scheduler_thread_cpu_s"]
        thr0 = thread_cpu_snapshot()
        t0 = time.perf_counter()
        tickets = []
        for k in range(steps):
            lo = (warmup + k) * B
            tickets += pipe.submit_many(targets[lo:lo + B], ITERS, MINSTEPS)
            pipe.pump()
        pipe.drain()
        sync_all()
        el = time.perf_counter() - t0
        host_cpu = (time.process_time() - cpu0) / max(1, len(tickets))
        sched_cpu = (pipe.stats()["scheduler_thread_cpu_s"] - sched0) / max(1, len(tickets))
        thr1 = thread_cpu_snapshot()
        by_thread = sorted(((thr1[t][1] - thr0.get(t, (None, 0.0))[1], thr1[t][0], t) for t in thr1), reverse=True)[:5]
        threads = [{"name": nm, "tid_is_main": tid == os.getpid(), "cpu_s_per_structure": round(c / max(1, len(tickets)), 4)}
                   for c, nm, tid in by_thread if c > 0]
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
        pipe.sync_check()
        finite = all(bool(torch.isfinite(c).all()) and bool(torch.isfinite(f).all()) for c, f in warm + timed)
        return {"elapsed": el, "steps": steps, "warmup": warmup, "timed": timed, "finite": finite, "host_cpu": host_cpu,
                "sched_cpu": sched_cpu, "threads": threads,
                "cnt": len(iv), "tot": sum(b_ - a_ for a_, b_ in iv), "union": interval_union(iv)}

    def roofline_of(name, r):
        """chip-level rate of the leg's convolution kernel: the launches' algorithmic FLOPs over the time at least one of
        them runs.  With one launch at a time this is FLOP per launch / average launch duration; with the lane's two
        launches in flight each launch lasts about twice its share of the chip (avg_launch_ms is the raw per-launch
        duration the rocprofv3 kernel trace shows)."""
        leg = LEGS[name]
        eff = r["union"] / r["cnt"] if r["cnt"] else 0.0
        ach = CONV_FLOP_PER_LAUNCH / (eff * 1e-3) / 1e12 if eff > 0 else 0.0
        traffic, current = None, None
        pmc = os.path.join(ROOT, "profiles", leg["pmc"])
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                # the PMC passes are a separate profiler run (counters cannot be collected inside the timed region):
                # the file carries the hash of the kernel source it was taken from
                if pj.get("kernel_source_sha256"):
                    src = os.path.join(ROOT, pj.get("kernel_source", leg["src"]))
                    current = hashlib.sha256(open(src, "rb").read()).hexdigest() == pj["kernel_source_sha256"]
            except Exception:
                traffic = None
        out = {"kernel": leg["kernel"], "bound": "mfma", "achieved": ach, "peak": leg["peak"], "unit": "TFLOP/s",
               "frac": ach / leg["peak"], "traffic": traffic,
               "traffic_source": "profiles/%s (rocprofv3 --pmc passes of single launches; a separate profiler run)" % leg["pmc"],
               "traffic_taken_from_this_kernel_source": current,
               "launches_timed": r["cnt"], "avg_launch_ms": r["tot"] / r["cnt"] if r["cnt"] else 0.0,
               "launches_in_flight": r["tot"] / r["union"] if r["union"] > 0 else 0.0, "chip_ms_per_launch": eff,
               "algorithmic_flop_per_launch": CONV_FLOP_PER_LAUNCH,
               "peak_is": "dense 16-bit matrix-core peak %.0f / %d piece products per float32 product" % (PEAK_F16_MFMA_TFLOPS, leg["products"])
                          if leg["products"] > 1 else "the f32 matrix-core peak"}
        if leg["products"] > 1:
            out["executed_16bit_mfma_tflops"] = leg["products"] * ach
            # what the matrix cores sustain under the power cap on random operands (measured, not a spec figure):
            # reported beside the spec-peak fraction, never instead of it
            cpath = os.path.join(ROOT, "profiles", "mfma_power_ceiling.json")
            if os.path.exists(cpath):
                try:
                    cj = json.load(open(cpath))
                    key = "random_bf16_operands_tflops" if name == "bf16x3" else "random_f16_operands_tflops"
                    if cj.get(key):
                        out["power_capped_ceiling"] = {
                            "mfma_random_operands_tflops": cj[key], "executed_frac_of_it": leg["products"] * ach / cj[key],
                            "source": "profiles/mfma_power_ceiling.json (tools/ubench_mfma_power.hip: register-resident "
                                      "operands, no memory traffic; zeros reach the 2.5 PFLOP/s spec peak, random data is "
                                      "held lower by the 1.3 kW power cap)"}
                except Exception:
                    pass
        return out

    # ---- a single leg alone (profiling): its numbers under the leg's own keys, nothing else is measured
    if args.legs != "all":
        r = timed_leg(args.legs, args.steps, args.warmup)
        if distributed:
            t = torch.tensor([r["elapsed"]], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r["elapsed"] = float(t[0].item())
        if rank == 0:
            sfx = LEGS[args.legs]["sfx"]
            print(json.dumps({"metric": "structures/s at L=300, N_seq=2000, 10 iters+100 min", "legs": args.legs,
                              "value" + sfx: world * args.steps * B / r["elapsed"],
                              "ms_per_step" + sfx: r["elapsed"] / args.steps * 1e3,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "finite_outputs": r["finite"],
                              "host_cpu_s_per_structure": r["host_cpu"], "scheduler_thread_cpu_s_per_structure": r["sched_cpu"],
                              "host_cpu_by_thread": r["threads"],
                              "roofline" + sfx: roofline_of(args.legs, r)}), flush=True)
        if distributed:
            dist.destroy_process_group()
        return 0

    # ---- the headline leg, then the two secondary ones at a quarter of the steps
    res = {HEADLINE: timed_leg(HEADLINE, args.steps, args.warmup)}
    ok = res[HEADLINE]["finite"]
    k2, w2 = max(2, args.steps // 4), min(1, args.warmup)
    if not args.headline_only:
        for name in LEGS:
            if name != HEADLINE:
                res[name] = timed_leg(name, min(k2, args.steps), w2)
                ok = ok and res[name]["finite"]
    timed_outs = res[HEADLINE]["timed"]

    # ---- verification of the outputs (untimed) ---------------------------------------------------
    verify = {"finite_outputs": ok}
    first = args.warmup * B                                   # first target of the headline leg's timed region
    es = Engine(device, L_NS, N_NS)                           # an engine of its own, with the pipeline's packed weights
    es.share_weights(pipe.engines[0])
    if per_row:
        es.set_option("vgru_persistent", 0)
    single, t0_out = {}, {}
    try:
        for name, r in res.items():
            prec, sfx = LEGS[name]["precision"], LEGS[name]["sfx"]
            es.set_option("precision", prec)
            # (1) the scheduler's result of the leg's first timed target == the same target alone on one engine, bit for
            #     bit.  The bits are compared with the scheduler's kernels selected (its engines tridiagonalise with one
            #     launch per Householder step; the cluster launch of a lone engine is the same algorithm with its float64
            #     sums associated differently - the same bits on almost every matrix, not on all)
            es.set_option("tridiag_cluster", 0)
            lo = (r["warmup"]) * B
            c1, f1 = es.predict_device(targets[lo], None, ITERS, MINSTEPS)
            es.sync_check()
            same = bool(torch.equal(c1, r["timed"][0][0]) and torch.equal(f1, r["timed"][0][1]))
            verify["timed_target_bitwise_equals_single_engine" + sfx] = same
            ok = ok and same
            if rank != 0:
                continue
            # (2) digest of bench target 0 (seed 0, the full 10 + 100 prediction, alone on one engine - the same bits the
            #     scheduler delivers, by (1)) against the stored one (profiles/bench_digest.json, written by
            #     DMP_WRITE_DIGEST=1); the results are bit-reproducible, so a change means the arithmetic changed.
            c0, f0 = es.predict_device(targets[0], None, ITERS, MINSTEPS)
            es.sync_check()
            t0_out[name] = (c0.clone(), f0.clone())
            digest = hashlib.sha256(c0.cpu().numpy().tobytes() + f0.cpu().numpy().tobytes()).hexdigest()
            dpath = os.path.join(ROOT, "profiles", "bench_digest.json")
            key = f"L{L_NS}_N{N_NS}_n{ITERS}_m{MINSTEPS}_seed0" + ("" if prec == 0 else f"_precision{prec}")
            if not es.get_option("vgru_persistent"):         # the launch-per-row chain sums K in another order: its own bits
                key += "_vgru_per_row"
            stored = {}
            if os.path.exists(dpath):
                try:
                    stored = json.load(open(dpath))
                except Exception:
                    stored = {}
            if os.environ.get("DMP_WRITE_DIGEST") == "1":
                stored[key] = digest
                json.dump(stored, open(dpath, "w"), indent=1, sort_keys=True)
            match = (stored.get(key) == digest) if key in stored else None
            verify["digest" + sfx] = {"sha256": digest, "expected": stored.get(key), "match": match}
            if match is False:                               # the arithmetic changed: not a valid run
                ok = False
            # (3) bench target 0 (seed 0) at iterations=1, minsteps=0 against the vector captured from the reference
            #     itself at this size (tests/golden/synth_L300_N2000_n1_m0.npz; data only)
            gpath = os.path.join(ROOT, "tests", "golden", "synth_L300_N2000_n1_m0.npz")
            if os.path.exists(gpath):
                g = np.load(gpath)
                sha = hashlib.sha256(targets[0].cpu().numpy().tobytes()).hexdigest()
                gc, gf = es.predict_device(targets[0], None, 1, 0)
                es.sync_check()
                d = gc.cpu().numpy()[:, 1].astype(np.float64) - g["coords"][:, 1].astype(np.float64)
                rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
                dconf = float(np.abs(gf.cpu().numpy() - g["confs"]).max())
                same_input = sha == bytes(g["alnmat_sha256"]).decode()
                verify["reference_golden_L300_N2000_n1_m0" + sfx] = {
                    "ca_rmsd_A": rmsd, "max_dconf": dconf, "same_input": same_input,
                    "ok": bool(same_input and rmsd <= 1e-3 and dconf < 1e-4)}
                ok = ok and verify["reference_golden_L300_N2000_n1_m0" + sfx]["ok"]
            # ---- latency of ONE prediction of the same configuration (the reference's use case: one CLI call; the
            #      scheduler above is the throughput mode), with the lone engine's cluster tridiagonalisation.
            es.set_option("tridiag_cluster", 1)
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(device)
                t = time.perf_counter()
                es.predict_device(targets[lo], None, ITERS, MINSTEPS)
                es.sync_check()
                ts.append((time.perf_counter() - t) * 1e3)
            single[name] = {"ms": min(ts[1:]), "runs_ms": ts}
    finally:
        es.close()
    if rank == 0:
        # (4) THE METRIC'S CONFIGURATION ITSELF against the reference, in every arithmetic setting: 10 + 100 on the weight
        #     sets on which the reference is stable there (coord_fc fitted to a protein-like trace, MDS feedback
        #     attenuated): tests/golden/fitns_*.npz and (round 6: other alignment, weight and trace seeds) fitns2_*.npz hold
        #     the reference's output, its thread-count floor and the fitted matrix; data only.  Own engine: the
        #     pipeline's weights stay untouched.
        for gname in ("fitns_L300_N2000_n10_m100", "fitns2_L300_N2000_n10_m100"):
            gpath = os.path.join(ROOT, "tests", "golden", gname + ".npz")
            if not os.path.exists(gpath):
                continue
            g = np.load(gpath)
            wseed = int(g["weights_seed"]) if "weights_seed" in g else 0
            mseed = int(g["msa_seed"])
            sdh = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]), seed=wseed)
            msa_h = targets[0] if mseed == 0 else torch.from_numpy(encode_aln(synth.synth_msa(L_NS, N_NS, mseed))).to(device)
            eh = Engine(device, L_NS, N_NS)
            if per_row:
                eh.set_option("vgru_persistent", 0)        # (two processes on this GPU: see above)
            try:
                eh.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sdh.items()})
                same = (hashlib.sha256(msa_h.cpu().numpy().tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
                        and synth.weights_checksum(sdh) == bytes(g["weights_sha256"]).decode())
                floor = float(g["noise_ca_rmsd"])
                for name in res:
                    eh.set_option("precision", LEGS[name]["precision"])
                    gc, gf = eh.predict_device(msa_h, None, ITERS, MINSTEPS)
                    eh.sync_check()
                    d = gc.cpu().numpy()[:, 1].astype(np.float64) - g["coords"][:, 1].astype(np.float64)
                    rmsd = float(np.sqrt((d ** 2).sum(-1).mean()))
                    dconf = float(np.abs(gf.cpu().numpy() - g["confs"]).max())
                    v = {"ca_rmsd_A": rmsd, "max_dconf": dconf, "reference_thread_noise_A": floor, "same_input": bool(same),
                         "ok": bool(same and rmsd <= max(1e-3, 3.0 * floor) and dconf < 1e-4)}
                    verify["reference_golden_" + gname[:gname.index("_")] + "_L300_N2000_n10_m100" + LEGS[name]["sfx"]] = v
                    ok = ok and v["ok"]
            finally:
                eh.close()
        for name in t0_out:
            if name == HEADLINE:
                continue
            a, b = t0_out[HEADLINE], t0_out[name]
            d = (a[0][:, 1].double() - b[0][:, 1].double())
            verify["headline_vs%s_bench_target_0" % LEGS[name]["sfx"]] = {
                "ca_rmsd_A": float((d ** 2).sum(-1).mean().sqrt()),
                "max_dconf": float((a[1] - b[1]).abs().max()),
                "note": "two arithmetic settings on the same target (10 + 100 on random weights: the minimiser on a "
                        "collapsed trace amplifies rounding differences, as the reference's own runs do; informational)"}

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

    if distributed:
        names = list(res)
        t = torch.tensor([res[n]["elapsed"] for n in names], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        for i, n in enumerate(names):
            res[n]["elapsed"] = float(t[i].item())
        flag = torch.tensor([1.0 if ok else 0.0], device=red_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0.5)

    if rank == 0:
        verify["ok"] = ok
        r = res[HEADLINE]
        line = {
            "metric": "structures/s at L=300, N_seq=2000, 10 iters+100 min",
            "value": world * r["steps"] * B / r["elapsed"],
            "unit": "structures/s",
            "n_gpus": world,
            "ranks": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / r["steps"] * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # `value` / `roofline`: the reference's float32 arithmetic with FULL-WIDTH operands (24 significand bits as three
            # bf16 pieces, the six products above 2^-24, float32 accumulate); `value_f32` is the same workload on the f32
            # MFMA instruction; `value_split_f16` the fast 22-bit mode, never the metric
            "dtype": LEGS[HEADLINE]["dtype"],
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
            "roofline": roofline_of(HEADLINE, r),
        }
        line["roofline"]["whole_job_conv_tflops"] = line["value"] / world * 16 * (ITERS + 1) * CONV_FLOP_PER_LAUNCH / 1e12
        # what the scheduler costs the host: CPU time of the whole process (scheduler thread, runtime helper threads)
        # over the timed region per structure
        line["host_cpu_s_per_structure"] = r["host_cpu"]
        # ... of which the scheduler's own thread inside the library (dmp_pipeline_stats; the rest is the HIP runtime's helper
        # threads, which spin for as long as kernels are in flight, and this process's main thread)
        line["scheduler_thread_cpu_s_per_structure"] = r["sched_cpu"]
        line["host_cpu_by_thread"] = r["threads"]          # the five busiest threads of the process over the timed region
        if host is not None:
            line["host_ms_per_target"] = host
        if single:
            line["single_target"] = dict(single[HEADLINE], note="one target alone on one engine of its own (single stream, the "
                                         "headline arithmetic; the first run builds the launch graphs)")
            for name in single:
                if name != HEADLINE:
                    line["single_target" + LEGS[name]["sfx"]] = single[name]
        for name, r2 in res.items():
            if name == HEADLINE:
                continue
            sfx = LEGS[name]["sfx"]
            v = world * r2["steps"] * B / r2["elapsed"]
            line["value" + sfx] = v
            line["ms_per_step" + sfx] = r2["elapsed"] / r2["steps"] * 1e3
            line["steps" + sfx], line["warmup" + sfx] = r2["steps"], r2["warmup"]
            line["dtype" + sfx] = LEGS[name]["dtype"]
            line["roofline" + sfx] = roofline_of(name, r2)
            line["roofline" + sfx]["whole_job_conv_tflops"] = v / world * 16 * (ITERS + 1) * CONV_FLOP_PER_LAUNCH / 1e12
            line["host_cpu_s_per_structure" + sfx] = r2["host_cpu"]
        # ---- SURVEY 8d's unit of work is one aln_to_coords call: the whole chain alignment FILES -> PDB FILES through the
        #      batch front end (read + encode + H2D + prediction + D2H + PDB text + write overlapped with the GPU by
        #      dmpfold2_amd.batch), same configuration and arithmetic as the headline, on a pipeline of its own that takes
        #      over the timed pipeline's streams.  Reported beside `value` (whose inputs are resident in HBM), never
        #      instead of it.
        if world == 1 and not args.no_files_leg:
            import shutil
            import tempfile
            from dmpfold2_amd import batch
            pipe.close()
            tmpd = tempfile.mkdtemp(prefix="dmp_bench_files_")
            keep = os.environ.get("DMPFOLD_PRECISION")
            os.environ["DMPFOLD_PRECISION"] = str(LEGS[HEADLINE]["precision"])
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
                    "note": "alignment files -> PDB files through dmpfold2_amd.batch.run_batch in the headline arithmetic (one "
                            "call: pipeline set-up - contexts, one packed copy of the weights - reading, encoding, H2D, "
                            "prediction, D2H, PDB text and writing all inside the clock; host work overlapped with the GPU); "
                            "`value` has its inputs resident in HBM and stops at tensors on the device"}
            finally:
                shutil.rmtree(tmpd, ignore_errors=True)
                if keep is None:
                    os.environ.pop("DMPFOLD_PRECISION", None)
                else:
                    os.environ["DMPFOLD_PRECISION"] = keep
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if world == 1 and mode != "none":
            line["cpu_baseline"] = cpu_baseline(mode)
            try:
                line["cpu_baseline"]["all_host_threads"] = all_threads_probe(line["cpu_baseline"]["cores"])
            except Exception as exc:                      # informational: never costs the line
                line["cpu_baseline"]["all_host_threads"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    # a failed verification (non-finite output, scheduler != single engine, digest or reference-golden mismatch)
    # still prints the line - with verify.ok false - but the process fails
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
