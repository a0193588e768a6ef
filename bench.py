#!/usr/bin/env python
"""Benchmark of the alignment -> structure hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]

A "step" is one batch of complete predictions (features, sequence trunk, 11 pair-trunk passes with
recycling, MDS, coordinate GRU, 2 x 100 minimiser steps, backbone) of synthetic targets of
the north-star configuration L=300, N_seq=2000, iterations=10, minsteps=100, with the residue
codes already resident in HBM and the packed weights loaded (model construction / weight load is
excluded).  For N > 1 the driver starts one process per GPU
(torch.distributed.run); every rank predicts its own targets - independent alignments are the
only parallel axis of this path, so there is no data-path collective, only the timing barrier.

Rank 0 prints ONE JSON line: structures/s for the whole job, the roofline of the dominant kernel
(conv5x5_f16x3, MFMA bound) measured with HIP events around every launch inside the timed
region, and (N = 1 only) the CPU oracle timed on this host on a bounded sample of the workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The scheduler keeps several HIP streams busy per GPU; the runtime multiplexes streams onto 4 hardware
# queues by default, which serialises a fifth stream (4 engines + the default stream): 5.3 structures/s
# with 4 engines on 4 queues, 6.5 on 8.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np   # noqa: E402
import torch         # noqa: E402

L_NS, N_NS, ITERS, MINSTEPS = 300, 2000, 10, 100
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (not the 2:1 sparse figure)
CONV_FLOP_PER_LAUNCH = 2.0 * 128 * 512 * 25 * L_NS * L_NS     # one block's 5x5 conv


def cpu_baseline():
    """Time the CPU oracle (a port of the reference's operator sequence, validated against the
    reference in tests/) on a bounded sample of the NS workload and extrapolate linearly:
      vgru     first 250 of the 2000 alignment rows (x8; cost is linear in rows)
      features reweight + fast_dca on 500 of the 2000 rows for the covariance (x4 on that part)
               and the full 6300 x 6300 inverse
      trunk    1 of the 11 pair-trunk passes incl. MDS and coordinate GRU (x11)
    """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dmpfold_oracle as O
    from dmpfold2_amd import synth
    # PyTorch-CPU collapses when given every hardware thread of the 256-thread GPU host (37x
    # slower than 8 threads); a sweep there (tools/gpu_diag.py --cpu-sweep: 8/16/32/64/128)
    # put the optimum at 16 threads for the GRU and 16-32 for the convolutions.
    # DMP_CPU_THREADS overrides.
    cores = int(os.environ.get("DMP_CPU_THREADS", min(os.cpu_count() or 1, 16)))
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
            "sample": ("oracle (PyTorch-CPU port) on this host: vgru on 250/2000 rows x8, reweight+"
                       "fast_dca on 500/2000 rows (GEMM part x4) + full 6300^2 inverse, 1/11 trunk "
                       "passes x11, 20/200 minimiser steps x10; est. %.1f s per structure "
                       "(vgru %.1f, features %.1f, trunk passes %.1f)"
                       % (total, t_vgru, t_rw + t_dca, (ITERS + 1) * t_pass))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=4,
                    help="targets in flight per GPU (one context + HIP stream each)")
    ap.add_argument("--batch", type=int, default=0,
                    help="targets per step and GPU (default 2 x streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stagger", action="store_true", help="scheduler: space the engines' phases")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DMP_FORCE_DIST=1 exercises the RCCL barrier / reduction path with a single rank (1-GPU boxes)
    distributed = world > 1 or (os.environ.get("DMP_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)

    from dmpfold2_amd import synth, _lib
    from dmpfold2_amd.predict import Pipeline, encode_aln
    lib = _lib.load()
    S = max(1, args.streams)
    sd = synth.synth_weights(0, coord_scale=5.0)
    pipe = Pipeline(device, L_NS, N_NS, {k: torch.from_numpy(np.array(v)) for k, v in sd.items()},
                    streams=S, stagger=args.stagger)

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

    outs = pipe.run(targets[:args.warmup * B], ITERS, MINSTEPS)
    sync_all()
    for e in pipe.engines:
        _lib.check(lib.dmp_profile_enable(e.ctx, 1, 16 * (ITERS + 1) * (args.steps * B // S + 2)))
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
    elapsed = time.perf_counter() - t0
    outs += [pipe.result(t) for t in tickets]
    # the lane keeps two launches in flight: besides the per-launch duration, measure the time during
    # which at least one launch runs (union of the HIP-event intervals of all engines)
    iv = []
    for e in pipe.engines:
        cap = 16 * (ITERS + 1) * (args.steps * B // S + 2)
        a, b, n = (C.c_float * cap)(), (C.c_float * cap)(), C.c_int()
        _lib.check(lib.dmp_profile_conv_intervals(e.ctx, pipe.engines[0].ctx, a, b, cap, C.byref(n)))
        iv += [(a[i], b[i]) for i in range(n.value)]
    iv.sort()
    conv_union, cur_a, cur_b = 0.0, None, None
    for a_, b_ in iv:
        if cur_b is None or a_ > cur_b:
            if cur_b is not None:
                conv_union += cur_b - cur_a
            cur_a, cur_b = a_, b_
        else:
            cur_b = max(cur_b, b_)
    if cur_b is not None:
        conv_union += cur_b - cur_a
    conv_tot, conv_cnt = 0.0, 0
    for e in pipe.engines:
        ms, n = C.c_float(), C.c_int()
        _lib.check(lib.dmp_profile_conv_ms(e.ctx, C.byref(ms), C.byref(n)))
        _lib.check(lib.dmp_profile_enable(e.ctx, 0, 0))
        conv_tot += ms.value * n.value
        conv_cnt += n.value
    conv_ms = conv_tot / conv_cnt if conv_cnt else 0.0
    pipe.sync_check()
    ok = all(bool(torch.isfinite(c).all()) and bool(torch.isfinite(f).all()) for c, f in outs)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
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
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "conv5x5_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "structures/s at L=300, N_seq=2000, 10 iters+100 min",
            "value": world * args.steps * B / elapsed,
            "unit": "structures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "finite_outputs": ok,
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
            # (float32) FLOPs is the dense f16 peak / 3.  The exact-f32 MFMA path (option
            # conv_f32_exact) is bounded by PEAK_F32_MFMA_TFLOPS and measures 130 TFLOP/s, the 3-way
            # bf16 split (conv_mode 2) 235 TFLOP/s (profiles/).
            "roofline": {"kernel": "conv5x5_f16x3_kernel (5x5 conv 128->512 + bias + 4-way maxout, "
                                   "float32 products from 3 f16 MFMA products, float32 accumulate)",
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
                         "unit": "TFLOP/s", "frac": achieved / (PEAK_F16_MFMA_TFLOPS / 3.0),
                         "traffic": traffic, "launches_timed": conv_cnt,
                         "avg_launch_ms": conv_ms, "launches_in_flight": in_flight,
                         "chip_ms_per_launch": eff_ms,
                         "algorithmic_flop_per_launch": CONV_FLOP_PER_LAUNCH,
                         "executed_f16_tflops": 3.0 * achieved,
                         "peak_f16_mfma_tflops": PEAK_F16_MFMA_TFLOPS,
                         "peak_f32_mfma_tflops": PEAK_F32_MFMA_TFLOPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
