#!/usr/bin/env python
"""Parity error budget by stage substitution (GPU box; developer diagnostic).

For a reference-golden fixture the prediction is composed stage by stage through the C ABI (tests/abi.py) exactly as
dmp_predict composes it, and then repeated with ONE stage at a time computed by the CPU oracle on the HIP path's own
inputs (the oracle is bit-identical to the reference, tests/golden/REPORT.txt).  Each variant's final structure,
confidences and per-pass traces are compared with the vectors captured from the reference itself: the stage whose
substitution moves the result towards the reference is where the deviation comes from; if no substitution does, the
deviation is the reference's own sensitivity (its thread-count noise, stored in the fixture).

    python tools/parity_budget.py [--cases w1x4,l300,fitns,...] [--modes 0,1] [--skip trunk] > profiles/r04_parity_budget.txt

Stages that can be substituted: dca (reweight + covariance + inverse + contacts), vgru, hgru, trunk (stem + 16 blocks +
head + Gram matrix, every pass), mds (float32 LAPACK eigh + sign rule, every pass), mds64 (float64 LAPACK), cgru
(coordinate GRU + coord_fc), refine (minimiser), all (the whole oracle on this host's threads: a sample of the
reference's own thread-count noise).
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dmpfold_oracle as O              # noqa: E402  (checker)
from dmpfold2_amd import synth          # noqa: E402
from dmpfold2_amd.predict import encode_aln   # noqa: E402
from abi import Stages                  # noqa: E402
from conftest import load_golden, ca_rmsd     # noqa: E402


def fixture(name):
    """-> (golden dict, state_dict, alnmat)"""
    if name == "w1x4":
        g = load_golden("w1x4_L128_N500_n3_m0")
        sd = synth.synth_weights(int(g["weights_seed"]), coord_scale=float(g["coord_scale"]), act_scale=float(g["act_scale"]))
        alnmat = g["alnmat"]
    elif name in ("actsmall", "actmixed"):
        g = load_golden(f"{name}_L128_N500_n3_m0")
        sd = synth.synth_weights(int(g["weights_seed"]), coord_scale=float(g["coord_scale"]), act_scale=float(g["act_scale"]))
        if "scaled_blocks" in g:
            sd = synth.scale_block_norms(sd, [int(b) for b in g["scaled_blocks"]], float(g["scaled_blocks_factor"]))
        alnmat = g["alnmat"]
    elif name == "l300":
        g = load_golden("synth_L300_N2000_n10_m0")
        sd = synth.synth_weights(0, coord_scale=5.0)
        alnmat = encode_aln(synth.synth_msa(300, 2000, int(g["msa_seed"])))
    elif name == "fitns":
        g = load_golden("fitns_L300_N2000_n10_m100")
        sd = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]))
        alnmat = encode_aln(synth.synth_msa(300, int(g["msa_rows"]), int(g["msa_seed"])))
    elif name == "fullgain":
        g = load_golden("fullgain_L300_N2000_n10_m5")
        sd = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]))
        alnmat = encode_aln(synth.synth_msa(300, int(g["msa_rows"]), int(g["msa_seed"])))
    elif name == "pf":
        g = load_golden("pf10963_n10_m0")
        sd = synth.synth_weights(0, coord_scale=5.0)
        alnmat = g["alnmat"]
    else:
        raise KeyError(name)
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode(), "weights differ from the fixture's"
    assert hashlib.sha256(np.ascontiguousarray(alnmat).tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    return g, sd, alnmat


def compose(st, W, alnmat, n, m, sub, timing=None):
    """One prediction from stages; `sub` = the set of stages the oracle computes.  Returns (coords, confs, ca_pass,
    conf_means)."""
    L = alnmat.shape[1]
    N = alnmat.shape[0]
    with torch.no_grad():
        if "dca" in sub and N > 1:
            cap = {}
            O.fast_dca(alnmat, O.reweight(alnmat, 0.8), capture=cap)
            inv, contacts = st.to(cap["inv_cov"].numpy()), st.to(cap["contacts"].numpy())
        elif N > 1:
            w = st.msa_weights(alnmat)
            inv = st.spd_inverse(st.cov_build(alnmat, w))
            contacts = st.dca_contacts(inv, L)
        else:
            inv = contacts = None
        if "vgru" in sub:
            x = W["embed.weight"][torch.from_numpy(np.asarray(alnmat).astype(np.int64))]
            v = st.to(O._gru(W, "vgru", x, 22, 512, 2, False, False)[-1].numpy())
        else:
            v = st.gru_vertical(alnmat)
        if "hgru" in sub:
            h = st.to(O._gru(W, "hgru", v.cpu().unsqueeze(1), 512, 256, 2, True, False)[:, 0, :].numpy())
        else:
            h = st.gru_bidir(0, v)
        mat1d = h.t().contiguous()                                 # (512, L)
        if "trunk" in sub:
            m1 = mat1d.cpu()
            pair = (m1.unsqueeze(1) * m1.unsqueeze(2)).unsqueeze(0)
            if N > 1:
                ic = inv.cpu().view(L, 21, L, 21)
                feats = ic.transpose(1, 2).contiguous().reshape(L, L, 441)
                f2d = torch.cat((feats, contacts.cpu()[:, :, None]), dim=2)
            else:
                f2d = torch.zeros((L, L, 442))
            static = torch.cat((pair, f2d.permute(2, 0, 1).unsqueeze(0)), dim=1)
        else:
            z0 = st.stem_static(mat1d, inv, contacts) if N > 1 else st.stem_static(mat1d, 0, 0)
        dmap = st.to(np.full((L, L), -1.0, np.float32))
        ca_pass, means = [], []
        best = None
        for p in range(n + 1):
            t0 = time.perf_counter()
            if "trunk" in sub:
                y = O.pair_trunk(W, torch.cat((static, dmap.cpu().view(1, 1, L, L)), dim=1))
                _, conf, M = O.head_to_gram(y)
                conf, M = st.to(conf[0].numpy()), st.to(M[0].numpy())
            else:
                conf, M = st.trunk_pass(z0, dmap)
            if "mds" in sub:
                mds = st.to(O.mds_top8(M.cpu().unsqueeze(0), "canonical")[0].numpy())
            elif "mds64" in sub:
                lam, vec = torch.linalg.eigh(M.cpu().double(), UPLO="U")
                vec = O.canonical_signs(vec)
                mds = st.to((vec * lam.clamp(min=1e-8).sqrt())[:, -8:].float().numpy())
            else:
                mds = st.eigh_top8(M)
            if "cgru" in sub:
                ca = st.to(O.coords_from_mds(W, mat1d.cpu(), mds.cpu().unsqueeze(0))[0].numpy())
            else:
                ca = st.coords_from_mds(mat1d, mds)
            ca_pass.append(ca.cpu().numpy().copy())
            if p == 0 and m > 0:
                ca = st.to(O.refine_coords(ca.cpu(), m).numpy()) if "refine" in sub else st.refine(ca, m)
            cm = conf.cpu().mean()
            means.append(float(cm))
            if best is None or bool(cm > best[2]):
                best = (conf.clone(), ca.clone(), cm)
            dmap = st.pair_distances(ca)
            if timing is not None:
                timing.append(time.perf_counter() - t0)
        bconf, bca, _ = best
        if m > 0:
            bca = st.to(O.refine_coords(bca.cpu(), m).numpy()) if "refine" in sub else st.refine(bca, m)
        coords, confs = st.backbone(bca, bconf)
        torch.cuda.synchronize()
    return coords.cpu().numpy(), confs.cpu().numpy(), np.stack(ca_pass), np.array(means, dtype=np.float32)


def score(g, coords, confs, ca_pass, means):
    P = g["ca_pass"].shape[0]
    per = [ca_rmsd(ca_pass[p], g["ca_pass"][p]) for p in range(P)]
    return (ca_rmsd(coords[:, 1], g["coords"][:, 1]), float(np.abs(confs - g["confs"]).max()),
            float(np.abs(means - g["conf_mean_pass"]).max()), per)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="w1x4,l300,fitns")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--skip", default="", help="comma-separated substitutions to leave out (e.g. trunk,all)")
    ap.add_argument("--only", default="", help="comma-separated substitutions to run (default: all)")
    a = ap.parse_args()
    skip = set(x for x in a.skip.split(",") if x)
    only = set(x for x in a.only.split(",") if x)
    subs = ["none", "dca", "vgru", "hgru", "trunk", "mds", "mds64", "cgru", "refine", "vgru+hgru+dca", "all"]
    print(f"# parity budget by stage substitution; library {os.environ.get('DMPFOLD_HIP_LIB', 'libdmpfold_hip.so')}; "
          f"host threads {torch.get_num_threads()}", flush=True)
    for case in a.cases.split(","):
        g, sd, alnmat = fixture(case)
        n, m = int(g["iterations"]), int(g["minsteps"])
        N, L = alnmat.shape
        W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
        floor_p = g["noise_ca_pass"] if "noise_ca_pass" in g else None
        print(f"\n## {case}: L={L} N={N} n={n} m={m}; reference's own thread-count floor: final CA-RMSD "
              f"{float(g['noise_ca_rmsd']):.2e} A, |dconf| {float(g['noise_conf']):.2e}"
              + ("" if floor_p is None else "; per pass " + " ".join(f"{x:.1e}" for x in floor_p)), flush=True)
        print(f"{'conv_mode':9s} {'oracle computes':16s} {'final CA-RMSD':>13s} {'max|dconf|':>11s} {'max|dmean|':>11s}  per-pass CA-RMSD vs reference", flush=True)
        st = Stages(sd, L, N)
        try:
            for mode in (int(x) for x in a.modes.split(",")):
                st.eng.set_option("conv_mode", mode)
                for sub in subs:
                    if sub in skip or (only and sub not in only):
                        continue
                    if sub == "refine" and m == 0:
                        continue
                    if sub == "all":
                        if mode != int(a.modes.split(",")[0]):
                            continue
                        t0 = time.perf_counter()
                        cap = {}
                        c, f = O.predict(alnmat, W, None, n, m, "canonical", cap)
                        res = (c.numpy(), f.numpy(), np.stack([cap[f"p{p}.ca"].numpy() for p in range(n + 1)]),
                               np.array([float(cap[f"p{p}.conf"].mean()) for p in range(n + 1)], dtype=np.float32))
                        dt = time.perf_counter() - t0
                    else:
                        t0 = time.perf_counter()
                        res = compose(st, W, alnmat, n, m, set() if sub == "none" else set(sub.split("+")))
                        dt = time.perf_counter() - t0
                        if sub == "none":
                            # the composition is dmp_predict's: same bits
                            c2, f2 = st.eng.predict(alnmat, None, n, m)
                            st.eng.sync_check()
                            same = bool(np.array_equal(c2.cpu().numpy(), res[0]) and np.array_equal(f2.cpu().numpy(), res[1]))
                            d2 = ca_rmsd(c2.cpu().numpy()[:, 1], res[0][:, 1])
                            print(f"#   composed == dmp_predict bit for bit: {same} (CA-RMSD between them {d2:.2e})", flush=True)
                    fin, dc, dm, per = score(g, *res)
                    print(f"{mode:<9d} {sub:16s} {fin:13.2e} {dc:11.2e} {dm:11.2e}  " + " ".join(f"{x:.1e}" for x in per)
                          + f"   [{dt:.1f} s]", flush=True)
        finally:
            st.eng.close()


if __name__ == "__main__":
    main()
