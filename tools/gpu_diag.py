#!/usr/bin/env python
"""Developer diagnostic (GPU box): per-stage error of the HIP path vs the oracle, without
asserting, plus optional stage timings.  Not part of the product or the test-suite.

    python tools/gpu_diag.py [--time L N]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dmpfold_oracle as O                      # noqa: E402
from dmpfold2_amd import synth                  # noqa: E402
from abi import Stages                          # noqa: E402
from conftest import load_golden, ca_rmsd      # noqa: E402


def err(tag, got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(got - ref)
    print(f"{tag:34s} max|d|={d.max():.3e}  scale={np.abs(ref).max():.3e}  rel={d.max() / max(np.abs(ref).max(), 1e-30):.2e}"
          f"  nan={int(np.isnan(got).sum())}", flush=True)


def stage_errors():
    sd = synth.synth_weights(0, coord_scale=5.0)
    W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    st = Stages(sd, max_L=128, max_N=3000)
    g = load_golden("pf10963_n0_m0")
    a = g["alnmat"]
    L = a.shape[1]
    cap = {}
    O.predict(a, W, None, 0, 0, "canonical", cap)
    w = st.msa_weights(a)
    print("msa_weights bit-exact:", np.array_equal(w.cpu().numpy(), g["w"]))
    cov = st.cov_build(a, st.to(cap["w"].numpy()))
    err("cov_reg", cov.cpu(), cap["cov_reg"])
    inv = st.spd_inverse(st.to(cap["cov_reg"].numpy()))
    err("inv_cov", inv.cpu(), cap["inv_cov"])
    con = st.dca_contacts(st.to(cap["inv_cov"].numpy()), L)
    err("contacts", con.cpu(), cap["contacts"])
    v = st.gru_vertical(a)
    err("vgru_last", v.cpu(), g["vgru_last"])
    h = st.gru_bidir(0, st.to(g["vgru_last"]))
    err("hgru (mat1d^T)", h.cpu().numpy().T, g["mat1d"])
    m = cap["mat1d"]
    pair = (m.unsqueeze(1) * m.unsqueeze(2)).unsqueeze(0)
    invf = cap["inv_cov"].view(L, 21, L, 21).transpose(1, 2).reshape(L, L, 441)
    f2d = torch.cat((invf, cap["contacts"][:, :, None]), dim=2).permute(2, 0, 1).unsqueeze(0)
    dmap = torch.zeros(L, L) - 1
    resinp = torch.cat((pair, f2d, dmap.view(1, 1, L, L)), dim=1)
    z0 = st.stem_static(st.to(m.numpy()), st.to(cap["inv_cov"].numpy()), st.to(cap["contacts"].numpy()))
    z0_ref = torch.nn.functional.conv2d(resinp[:, :954], W["resnet.0.lin.weight"][:, :954], W["resnet.0.lin.bias"])
    err("stem_static z0", z0.cpu(), z0_ref[0])
    x0 = st.stem_update(z0, st.to(dmap.numpy()))
    err("stem", x0.cpu(), cap["p0.stem"][0])
    for blk, xin in ((1, cap["p0.stem"]), (16, cap["p0.block1"])):
        u_ref = O.block_conv(W, blk, xin)
        u, stats = st.conv(blk, st.to(xin[0].numpy()))
        err(f"block{blk} conv+maxout", u.cpu(), u_ref[0])
        s_ref = torch.stack((u_ref[0].double().sum(dim=(1, 2)), (u_ref[0].double() ** 2).sum(dim=(1, 2))), 1)
        err(f"block{blk} stats", stats.cpu(), s_ref)
        o_ref = O.block_finish(W, blk, u_ref, xin)
        o = st.norm(blk, st.to(u_ref[0].numpy()), st.to(s_ref.numpy(), torch.float64), st.to(xin[0].numpy()))
        err(f"block{blk} norm+scSE+res", o.cpu(), o_ref[0])
    conf, M = st.head_gram(st.to(cap["p0.block16"][0].numpy()))
    err("head conf", conf.cpu(), cap["p0.conf"])
    err("gram M", M.cpu(), cap["p0.M"])
    tconf, tM = st.trunk_pass(z0, st.to(dmap.numpy()))
    err("trunk_pass conf", tconf.cpu(), cap["p0.conf"])
    err("trunk_pass M", tM.cpu(), cap["p0.M"])
    mds = st.eigh_top8(st.to(cap["p0.M"].numpy()))
    err("mds vs oracle(f32 eigh)", mds.cpu(), cap["p0.mds"])
    lam, vec = torch.linalg.eigh(cap["p0.M"].double(), UPLO="U")
    truth = (O.canonical_signs(vec) * lam.clamp(min=1e-8).sqrt())[:, -8:]
    err("mds vs f64 truth", mds.cpu(), truth)
    err("oracle mds vs f64 truth", cap["p0.mds"], truth)
    ca = st.coords_from_mds(st.to(m.numpy()), st.to(cap["p0.mds"].numpy()))
    err("ca (coord_gru+fc)", ca.cpu(), cap["p0.ca"])
    k = load_golden("kat_refine_backbone")
    for steps in (1, 10, 100, 1000):
        r = st.refine(st.to(k["ca_in"]), steps)
        err(f"refine KAT {steps}", r.cpu(), k[f"refined_{steps}"])
    bb, _ = st.backbone(st.to(k["ca_in"]), st.to(np.zeros(len(k["ca_in"]), np.float32)))
    err("backbone KAT", bb.cpu().numpy().reshape(-1, 3), k["backbone"])
    for name in ["pf10963_n0_m0", "pf10963_n3_m0", "pf10963_n2_m5", "pf10963_default_cli",
                 "synth_L40_N64_n2_m0", "synth_L24_N3050_n1_m0", "synth_L30_N1_n1_m3",
                 "alphabet_L16_N12_n0_m0", "template_L96_N50_n1_m0"]:
        gg = load_golden(name)
        tpl = gg["template_ca"] if "template_ca" in gg else None
        n, mm = int(gg["iterations"]), int(gg["minsteps"])
        c, f = st.eng.predict(gg["alnmat"], tpl, n, mm)
        c, f = c.cpu().numpy(), f.cpu().numpy()
        means = st.eng.fetch("conf_means", n + 1).cpu().numpy()
        cap_ = st.eng.fetch("ca_pass", (n + 1) * gg["alnmat"].shape[1] * 3).cpu().numpy().reshape(n + 1, -1, 3)
        per_pass = [ca_rmsd(cap_[i], gg["ca_pass"][i]) for i in range(min(n + 1, 4))]
        print(f"e2e {name:26s} CA-RMSD={ca_rmsd(c[:, 1], gg['coords'][:, 1]):.3e} max|dcoord|={np.abs(c - gg['coords']).max():.3e} "
              f"dconf={np.abs(f - gg['confs']).max():.3e} dmeans={np.abs(means - gg['conf_mean_pass']).max():.3e} "
              f"noise={float(gg['noise_ca_rmsd']):.1e} per-pass ca rmsd={['%.1e' % v for v in per_pass]}", flush=True)


def timings(L, N, n=10, m=100):
    sd = synth.synth_weights(0, coord_scale=5.0)
    st = Stages(sd, max_L=L, max_N=N)
    rows = synth.synth_msa(L, N, 0)
    a = O.encode_aln(rows)
    torch.cuda.synchronize()

    def timed(tag, fn, reps=1):
        fn()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        print(f"time {tag:28s} {(time.time() - t) / reps * 1e3:10.3f} ms", flush=True)
        return out
    w = timed("msa_weights", lambda: st.msa_weights(a))
    cov = timed("cov_build", lambda: st.cov_build(a, w))
    inv = timed("spd_inverse", lambda: st.spd_inverse(cov))
    con = timed("dca_contacts", lambda: st.dca_contacts(inv, L))
    v = timed("gru_vertical", lambda: st.gru_vertical(a))
    h = timed("gru_bidir(hgru)", lambda: st.gru_bidir(0, v))
    mat1d = h.t().contiguous()
    z0 = timed("stem_static", lambda: st.stem_static(mat1d, inv, con))
    dmap = st.to(np.full((L, L), -1, np.float32))
    x = timed("stem_update", lambda: st.stem_update(z0, dmap))
    timed("conv5x5 (stage API, +pad)", lambda: st.conv(1, x))
    val = st.conv_ms(z0, dmap, 2)
    flops = 2.0 * 128 * 512 * 25 * L * L
    print(f"time conv5x5 kernel only        {val:10.3f} ms  -> {flops / val / 1e9:.1f} TFLOP/s", flush=True)
    conf, M = timed("trunk_pass", lambda: st.trunk_pass(z0, dmap))
    mds = timed("eigh_top8", lambda: st.eigh_top8(M))
    ca = timed("coords_from_mds", lambda: st.coords_from_mds(mat1d, mds))
    timed("refine 100", lambda: st.refine(ca, 100))
    timed(f"predict n={n} m={m}", lambda: st.eng.predict(a, None, n, m))


def cpu_sweep():
    """How many PyTorch-CPU threads does the oracle want on this host?"""
    sd = synth.synth_weights(0, coord_scale=5.0)
    W = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    a = O.encode_aln(synth.synth_msa(300, 100, 0))
    x = W["embed.weight"][torch.from_numpy(a.astype(np.int64))]
    xin = torch.randn(1, 128, 300, 300)
    for nt in (8, 16, 32, 64, 128):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        with torch.no_grad():
            t = time.time(); O._gru(W, "vgru", x, 22, 512, 2, False, False); tv = time.time() - t
            t = time.time(); O.block_conv(W, 1, xin); tc = time.time() - t
            t = time.time(); O.block_conv(W, 2, xin); tc = min(tc, time.time() - t)
        print(f"cpu threads {nt:4d}: vgru 100 rows {tv:7.2f} s   one 5x5 conv block {tc:7.3f} s", flush=True)


if __name__ == "__main__":
    if "--cpu-sweep" in sys.argv:
        cpu_sweep()
    elif "--time" in sys.argv:
        i = sys.argv.index("--time")
        timings(int(sys.argv[i + 1]), int(sys.argv[i + 2]))
    else:
        stage_errors()
