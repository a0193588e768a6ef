"""End-to-end pins of the HEADLINE path against vectors captured from the reference itself
(tests/golden/make_goldens.py): the benchmark's recycling depth (11 trunk passes), the benchmark's
size (L=300, N=2000) and the benchmark's minimiser setting (2 x 100 steps) on protein-like traces -
each in the default split-f16 convolution AND the exact-f32 one - plus the behaviour at the drop-in
boundary when something goes wrong on the device (range fault -> automatic re-run, unknown residue
code -> IndexError, bad weight files -> RuntimeError) and the sharded batch front end.

Tolerances: CA-RMSD <= 1e-3 A and |dconf| < 1e-4 at minsteps=0 (BASELINE.json north_star); with the
minimiser max(1e-3, 3 x the reference's own 8-vs-1-thread deviation stored in the fixture).
"""
import contextlib
import ctypes as C
import hashlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_rows, ca_rmsd

pytestmark = pytest.mark.gpu

import dmpfold_oracle as O          # noqa: E402  (test infrastructure: the checker)

MODES = {"f16x3": 0, "f32": 1, "bf16x3": 2}       # option "precision": split f16 / f32 MFMA / exact 3 x bf16 + f32 GRU


def _engine(sd, max_L, max_N):
    from dmpfold2_amd.predict import Engine
    eng = Engine("cuda:0", max_L, max_N)
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return eng


@pytest.fixture(scope="module")
def small_engine(synth_sd):
    eng = _engine(synth_sd, 128, 512)
    yield eng
    eng.close()


def _check_passes(eng, g, P, L, tol, first=0):
    """Every pass's confidence mean and CA trace - INTERMEDIATE quantities, not outputs of aln_to_coords (the
    final structure is checked by the callers at the plain north-star tolerance wherever the fixture's own
    thread-noise floor allows).  Recycling is expansive over its first passes before it settles: the
    reference's own 8-vs-1-thread runs differ by up to 7e-4 A at pass 5 of the example although the final
    structure (the best pass, usually an early one) agrees to 2e-4, and a 1e-7 change of the vertical-GRU state
    (library expf against the hardware exponential, same convolution arithmetic) moves the HIP path's own pass-5
    trace from 2.3e-4 to 6.6e-4 A from the reference (gpurun_out r03b, tools/perpass_dev.py).  Bound per pass:
    max(tol, 4 x that pass's floor stored in the fixture - the largest deviation among the reference's runs with
    1, 2, 3, 5 (or 4) threads from its 8-thread run: a maximum over a handful of samples)."""
    means = eng.fetch("conf_means", P).cpu().numpy()
    assert np.abs(means - g["conf_mean_pass"]).max() < 1e-3
    ca_pass = eng.fetch("ca_pass", P * L * 3).cpu().numpy().reshape(P, L, 3)
    floor = g["noise_ca_pass"] if "noise_ca_pass" in g else np.full(P, float(g["noise_ca_rmsd"]))
    dev = np.array([ca_rmsd(ca_pass[p], g["ca_pass"][p]) for p in range(P)])
    # first = 1 with minsteps > 0: the fixture's pass-0 trace is the coordinate head's output, the engine records it
    # after the first refinement (network.py:257-258) - that trace is checked through the final structure instead
    assert (dev[first:] <= np.maximum(tol, 4.0 * floor)[first:]).all(), (dev, floor)
    return dev


# ------------------------------------------------------------------ the benchmark's recycling depth
@pytest.mark.parametrize("mode", list(MODES))
def test_pf10963_eleven_passes_vs_reference(small_engine, mode):
    """-n 10 -m 0 on the reference's example alignment: 11 trunk passes (176 convolutions) deep, every
    pass's CA trace and confidence mean against the reference's, then the final structure."""
    g = load_golden("pf10963_n10_m0")
    eng = small_engine
    eng.set_option("precision", MODES[mode])
    try:
        coords, confs = eng.predict(g["alnmat"], None, 10, 0)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        _check_passes(eng, g, 11, 82, 1e-3)
        assert ca_rmsd(coords[:, 1], g["coords"][:, 1]) <= 1e-3
        assert np.abs(confs - g["confs"]).max() < 1e-4
        assert np.abs(coords - g["coords"]).max() < 2e-2
    finally:
        eng.set_option("precision", 0)


# ------------------------------------------------------------------ the benchmark's size
@pytest.fixture(scope="module")
def ns_engine(synth_sd):
    eng = _engine(synth_sd, 300, 2000)
    yield eng
    eng.close()


@pytest.mark.parametrize("mode", list(MODES))
def test_north_star_size_vs_reference(ns_engine, mode):
    """bench.py's target 0 (L=300, N=2000, seed 0), iterations=1, minsteps=0, against the output of the
    reference itself at that size (6 minutes of reference CPU time in the build container)."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import encode_aln
    g = load_golden("synth_L300_N2000_n1_m0")
    alnmat = encode_aln(synth.synth_msa(300, 2000, int(g["msa_seed"])))
    assert hashlib.sha256(alnmat.tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    eng = ns_engine
    eng.set_option("precision", MODES[mode])
    try:
        coords, confs = eng.predict(alnmat, None, 1, 0)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        _check_passes(eng, g, 2, 300, 1e-3)
        assert ca_rmsd(coords[:, 1], g["coords"][:, 1]) <= 1e-3
        assert np.abs(confs - g["confs"]).max() < 1e-4
    finally:
        eng.set_option("precision", 0)


@pytest.mark.parametrize("mode", list(MODES))
def test_north_star_size_and_depth_vs_reference(ns_engine, mode):
    """The benchmark's workload itself minus the minimiser: bench target 0 (L=300, N=2000), iterations=10
    (11 trunk passes = 176 convolutions at L=300), minsteps=0, every pass against the reference's run."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import encode_aln
    g = load_golden("synth_L300_N2000_n10_m0")
    alnmat = encode_aln(synth.synth_msa(300, 2000, int(g["msa_seed"])))
    assert hashlib.sha256(alnmat.tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    eng = ns_engine
    eng.set_option("precision", MODES[mode])
    try:
        coords, confs = eng.predict(alnmat, None, 10, 0)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        _check_passes(eng, g, 11, 300, 1e-3)
        assert ca_rmsd(coords[:, 1], g["coords"][:, 1]) <= 1e-3       # plain north-star tolerance (7.5e-4 in both modes)
        # confidences: the reference's own 8- and 4-thread runs differ by 8.4e-5 at this size and depth; the default
        # convolution lands at 7.8e-5, the exact-f32 one at 1.7e-4
        assert np.abs(confs - g["confs"]).max() < max(1e-4, 3.0 * float(g["noise_conf"]))
    finally:
        eng.set_option("precision", 0)


# ------------------------------------------------------------------ the benchmark's workload itself
@pytest.mark.parametrize("mode", list(MODES))
def test_headline_workload_with_minimiser_vs_reference(mode):
    """BASELINE.json's metric configuration EXACTLY - bench target 0 (L=300, N=2000, alignment seed 0),
    iterations=10, minsteps=100 (11 trunk passes, 2 x 100 minimiser steps) - against the reference's own run.
    On random weights the reference is chaotic there (its 8- and 4-thread runs end 120 A apart on the first
    attempt's fixture), so the fixture's weights were designed for stability (tools/design_coord_fc.py): coord_fc
    fitted to a protein-like 300-residue trace, the coordinate GRU's 8 MDS input columns scaled by 0.02 - the
    traces still move 15-30 A from pass to pass with the trunk's output, the refined structure has 3.77-3.81 A
    bonds, and the reference's own thread-count spread is 1.2e-4 .. 5.8e-4 A per pass and 8.2e-4 A in the final
    structure after both refinements (|dconf| 1.5e-7).  Bounds: every pass max(1e-3, 4 x floor) (all floors are
    below 2.5e-4 x 4 = 1e-3 except passes 1-2), final structure max(1e-3, 3 x 8.2e-4) A, confidences plain 1e-4."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import encode_aln
    g = load_golden("fitns_L300_N2000_n10_m100")
    sd = synth.headline_fixture_weights(g["coord_fc"], float(g["coord_gru_mds_scale"]))
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode()
    alnmat = encode_aln(synth.synth_msa(300, int(g["msa_rows"]), int(g["msa_seed"])))
    assert hashlib.sha256(alnmat.tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    eng = _engine(sd, 300, 2000)
    try:
        eng.set_option("precision", MODES[mode])
        coords, confs = eng.predict(alnmat, None, 10, 100)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        dev = _check_passes(eng, g, 11, 300, 1e-3, first=1)
        final = ca_rmsd(coords[:, 1], g["coords"][:, 1])
        print("headline workload", mode, "per-pass CA-RMSD", dev, "final", final, "max|dconf|", np.abs(confs - g["confs"]).max())
        assert final <= max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
        assert np.abs(confs - g["confs"]).max() < 1e-4
        bonds = np.linalg.norm(coords[1:, 1] - coords[:-1, 1], axis=1)
        assert 3.7 < bonds.min() and bonds.max() < 3.9          # the minimiser ran in its regular regime
    finally:
        eng.close()


@pytest.mark.parametrize("mode", list(MODES))
def test_headline_size_at_full_mds_gain_vs_reference(mode):
    """the headline fixture above scales the coordinate GRU's 8 MDS input columns by 0.02, which
    desensitises exactly the eigensolver -> coordinate GRU -> distance map feedback (network.py:247-255, 272).  Here the
    same target (L=300, N=2000) with those columns UNSCALED: coord_fc fitted to the protein-like trace with ridge 30 -
    the strongest coord_fc at which the reference itself is stable at full gain (tools/explore_fullgain.py: ridge 1e-3,
    the fitns weights, puts its own 8- and 4-thread runs 7.9e-3 A apart in the FIRST pass and 140 A after eleven; ridge 1:
    6.6e-4 -> 2.6e-2 A over four passes; ridge 30: 2e-5 .. 8e-5 A through all eleven passes) -, iterations=10 and
    minsteps=5 (2 x 5 minimiser steps on the resulting compact trace: 3.6e-4 A between the reference's runs).  Every
    pass at the plain 1e-3 A (floors below 1e-4), the final structure at max(1e-3, 3 x floor), confidences plain."""
    import os
    from conftest import GOLDEN
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import encode_aln
    name = "fullgain_L300_N2000_n10_m5"
    if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated: " + name)
    g = load_golden(name)
    assert float(g["coord_gru_mds_scale"]) == 1.0
    sd = synth.headline_fixture_weights(g["coord_fc"], 1.0)
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode()
    alnmat = encode_aln(synth.synth_msa(300, int(g["msa_rows"]), int(g["msa_seed"])))
    assert hashlib.sha256(alnmat.tobytes()).hexdigest() == bytes(g["alnmat_sha256"]).decode()
    eng = _engine(sd, 300, 2000)
    try:
        eng.set_option("precision", MODES[mode])
        n, m = int(g["iterations"]), int(g["minsteps"])
        coords, confs = eng.predict(alnmat, None, n, m)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        dev = _check_passes(eng, g, n + 1, 300, 1e-3, first=1)
        final = ca_rmsd(coords[:, 1], g["coords"][:, 1])
        print("full MDS gain", mode, "per-pass CA-RMSD", dev, "final", final, "max|dconf|", np.abs(confs - g["confs"]).max())
        assert (dev[1:] <= 1e-3).all()
        assert final <= max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
        assert np.abs(confs - g["confs"]).max() < 1e-4
    finally:
        eng.close()


# ------------------------------------------------------------------ a second weight distribution
@pytest.mark.parametrize("mode", ["f16x3", "f32", "bf16x6"])
def test_second_weight_set_vs_reference(mode):
    """Every other golden uses synth_weights(0): this one is seed 1 with InstanceNorm gamma / beta x 4 (the
    residual stream reaches several hundred, block-16 activations 6e9 in sum of squares instead of 1e8), L=128,
    N=500, four trunk passes through the reference itself - the f16 pieces of the default convolution at a
    different activation scale.  Stage tensors by sampled index, every pass, final structure; the bounds carry
    the reference's own thread-count floor (1, 2, 3, 5 against 8 threads) where it exceeds the plain tolerance."""
    from dmpfold2_amd import synth
    g = load_golden("w1x4_L128_N500_n3_m0")
    sd = synth.synth_weights(int(g["weights_seed"]), coord_scale=float(g["coord_scale"]), act_scale=float(g["act_scale"]))
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode()
    alnmat = g["alnmat"]
    L = alnmat.shape[1]
    from abi import Stages
    st = Stages(sd, 128, 512)
    eng = st.eng
    try:
        eng.set_option("conv_mode", {"f16x3": 0, "f32": 1, "bf16x6": 2}[mode])
        eng.predict(alnmat, None, 0, 0)
        eng.sync_check()
        assert np.array_equal(eng.fetch("w", alnmat.shape[0]).cpu().numpy(), g["w"])
        mat1d = eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
        contacts = eng.fetch("contacts", L * L).reshape(L, L).clone()
        inv = eng.fetch("inv_cov", (21 * L) ** 2).reshape(21 * L, 21 * L).clone()
        assert np.abs(mat1d.cpu().numpy() - g["mat1d"]).max() < 1e-5
        assert np.abs(contacts.cpu().numpy() - g["contacts"]).max() <= 1e-5 * max(1.0, np.abs(g["contacts"]).max())
        # the trunk of the first pass stage by stage on the device's own features: sampled entries of the
        # tensors the REFERENCE produced (stem, block 1, block 16), relative 1e-4 of each tensor's scale
        x = st.stem_update(st.stem_static(mat1d, inv, contacts), st.to(np.full((L, L), -1.0, np.float32)))
        def check(name, t):
            ref = g[name + ".val"]
            got = t.cpu().numpy().ravel()[g[name + ".idx"]]
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), name
        check("stem_p0", x)
        for block in range(1, 17):
            u, stats = st.conv(block, x)
            x = st.norm(block, u, stats, x)
            if block == 1:
                check("block1_p0", x)
        check("block16_p0", x)
        coords, confs = eng.predict(alnmat, None, 3, 0)
        eng.sync_check()
        dev = _check_passes(eng, g, 4, L, 1e-3)
        print("second weight set", mode, "per-pass CA-RMSD", dev, "floors", g["noise_ca_pass"])
        # this weight regime is more expansive than the default one: the reference's own runs (1, 2, 3, 5 threads
        # against 8) differ by 7.2e-4 A / 4.4e-5 in the final structure, so the plain 1e-3 / 1e-4 cannot be held
        # reliably by anything; measured here: 6.1e-4 .. 1.04e-3 A, 1.2e-4 (gpurun_out r03b)
        assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) <= max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
        assert np.abs(confs.cpu().numpy() - g["confs"]).max() < max(1e-4, 4.0 * float(g["noise_conf"]))    # 1.2e-4 .. 1.5e-4
    finally:
        eng.close()


# ------------------------------------------------------------------ small activations (f16 low pieces)
def _fixture_weights(g):
    from dmpfold2_amd import synth
    sd = synth.synth_weights(int(g["weights_seed"]), coord_scale=float(g["coord_scale"]), act_scale=float(g["act_scale"]))
    if "scaled_blocks" in g:
        sd = synth.scale_block_norms(sd, [int(b) for b in g["scaled_blocks"]], float(g["scaled_blocks_factor"]))
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode()
    return sd


@pytest.mark.parametrize("name", ["actsmall_L128_N500_n3_m0", "actmixed_L128_N500_n3_m0"])
def test_small_activation_regimes_vs_reference(name):
    """VERDICT r03 item 1 (weak #3): trunks whose activations are SMALL - every InstanceNorm gamma / beta x 1/64
    (`actsmall`: the residual stream stays below 0.5, where the low f16 piece of an unscaled split is a subnormal), and
    odd blocks x 1/256 between O(1) ones (`actmixed`) - through the reference itself, four trunk passes.  The default
    convolution takes its f16 pieces of 2^e x with e chosen per block from the InstanceNorm weights
    (dmp_weights_finalize), so the pieces are as precise here as at O(1).  Plain north-star tolerances in the default
    mode (the fixtures' own thread-count floors are 2.5e-4 A / 4e-5 and 1.2e-4 A / 2e-7); per block the split-f16
    convolution agrees with the exact-f32 one to 1e-5 of the output's scale."""
    g = load_golden(name)
    sd = _fixture_weights(g)
    alnmat = g["alnmat"]
    L = alnmat.shape[1]
    from abi import Stages
    st = Stages(sd, 128, 512)
    eng = st.eng
    try:
        # the scales follow the weights: large where the trunk is small
        e1 = eng.get_option("act_scale_log2_block1")
        assert e1 >= 10 if name.startswith("actsmall") else e1 >= 4
        eng.predict(alnmat, None, 0, 0)
        eng.sync_check()
        assert np.array_equal(eng.fetch("w", alnmat.shape[0]).cpu().numpy(), g["w"])
        mat1d = eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
        contacts = eng.fetch("contacts", L * L).reshape(L, L).clone()
        inv = eng.fetch("inv_cov", (21 * L) ** 2).reshape(21 * L, 21 * L).clone()
        assert np.abs(mat1d.cpu().numpy() - g["mat1d"]).max() < 1e-5
        x = st.stem_update(st.stem_static(mat1d, inv, contacts), st.to(np.full((L, L), -1.0, np.float32)))

        def check(key, t):
            ref = g[key + ".val"]
            got = t.cpu().numpy().ravel()[g[key + ".idx"]]
            assert np.abs(got - ref).max() <= 1e-4 * max(1e-30, float(np.abs(ref).max())), key    # relative to the tensor's scale
        check("stem_p0", x)
        worst = 0.0
        for block in range(1, 17):
            u0, stats = st.conv(block, x)
            eng.set_option("conv_mode", 1)
            u1, _ = st.conv(block, x)
            eng.set_option("conv_mode", 0)
            rel = float((u0 - u1).abs().max() / u1.abs().max())
            worst = max(worst, rel)
            assert rel <= 1e-5, (block, rel)
            x = st.norm(block, u0, stats, x)
            if block == 1:
                check("block1_p0", x)
        check("block16_p0", x)
        print(name, "worst |f16x3 - f32| / scale over the 16 convolutions:", worst)
        for mode in MODES:
            eng.set_option("precision", MODES[mode])
            coords, confs = eng.predict(alnmat, None, 3, 0)
            eng.sync_check()
            dev = _check_passes(eng, g, 4, L, 1e-3)
            final = ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1])
            dconf = float(np.abs(confs.cpu().numpy() - g["confs"]).max())
            print(name, mode, "per-pass CA-RMSD", dev, "final", final, "max|dconf|", dconf)
            assert final <= 1e-3 and dconf < 1e-4
    finally:
        eng.set_option("precision", 0)
        eng.close()


def test_unscaled_pieces_lose_precision_on_small_activations():
    """What the activation scale is for.  On the `actsmall` fixture (trunk at 0.01 .. 1) unscaled pieces are still as
    accurate as scaled ones (1.5e-6 of the output's scale either way, measured: the float32 accumulation dominates),
    so the regime that needs the scale is made here: every InstanceNorm gamma / beta x 2^-16 (the residual stream at
    1e-5, where even the HIGH f16 piece of an unscaled split is a subnormal).  Block 9's convolution against a float64
    convolution of the same input: scaled pieces as accurate as the exact-f32 kernel, unscaled pieces orders of
    magnitude worse."""
    from dmpfold2_amd import synth
    sd = synth.synth_weights(2, coord_scale=5.0, act_scale=2.0 ** -16)
    L = 96
    alnmat = O.encode_aln(synth.synth_msa(L, 40, 31))
    from abi import Stages
    st = Stages(sd, 128, 64)
    eng = st.eng
    try:
        assert eng.get_option("act_scale_log2_block9") >= 20
        eng.predict(alnmat, None, 0, 0)
        eng.sync_check()
        mat1d = eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
        contacts = eng.fetch("contacts", L * L).reshape(L, L).clone()
        inv = eng.fetch("inv_cov", (21 * L) ** 2).reshape(21 * L, 21 * L).clone()
        x = st.stem_update(st.stem_static(mat1d, inv, contacts), st.to(np.full((L, L), -1.0, np.float32)))
        for block in range(1, 9):
            u0, stats = st.conv(block, x)
            x = st.norm(block, u0, stats, x)
        eng.set_option("conv_mode", 1)
        u32, _ = st.conv(9, x)
        eng.set_option("conv_mode", 0)
        us, _ = st.conv(9, x)
        eng.set_option("act_scaling", 0)
        uu, _ = st.conv(9, x)
        eng.sync_check()
        w = torch.from_numpy(np.array(sd["resnet.9.layer1.lin.weight"])).double().cuda()
        b = torch.from_numpy(np.array(sd["resnet.9.layer1.lin.bias"])).double().cuda()
        full = torch.nn.functional.conv2d(x.double().unsqueeze(0), w, b, padding=2).view(128, 4, L, L)
        t = full.max(dim=1)[0]
        scale = float((full - b.view(128, 4, 1, 1)).abs().max())          # the scale of the SUMS (the bias is O(0.02))
        e_scaled = float((us.double() - t).abs().max()) / scale
        e_unscaled = float((uu.double() - t).abs().max()) / scale
        e_f32 = float((u32.double() - t).abs().max()) / scale
        print("block 9 at 2^-16: max error / scale of the sums vs float64: scaled pieces", e_scaled, "unscaled", e_unscaled,
              "f32 MFMA", e_f32, "max|x|", float(x.abs().max()))
        # measured: scaled 2.3e-5 = the exact-f32 kernel's 2.4e-5 (at this scale the float32 inputs themselves limit
        # both), unscaled 4.4e-4
        assert e_scaled <= 1.5 * e_f32 + 1e-7
        assert e_unscaled > 10.0 * e_scaled                  # what the round-3 arithmetic would have done here
    finally:
        eng.set_option("act_scaling", 1)
        eng.close()


# ------------------------------------------------------------------ the benchmark's minimiser setting
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", ["fit3fgx_L96_N50_n0_m100", "fit3fgx_L96_N50_n10_m100"])
def test_minimiser_end_to_end_on_protein_like_traces(synth_sd, name, mode):
    """minsteps=100 (the CLI default, 2 x 100 steps) end to end.  With random weights the first trace
    is a collapsed tangle on which the reference's minimiser is chaotic (0.12 A between its own 8- and
    1-thread runs); these fixtures use synthetic weights whose coord_fc was fitted so that the first
    trace approximates 3FGX chain A (make_goldens.fit_coord_fc), where the reference's spread over 1, 2,
    3, 5 and 8 threads is 2.9e-4 A (n=0) and 4.8e-4 A (n=10): 250 times tighter."""
    g = load_golden(name)
    sd = dict(synth_sd)
    sd["coord_fc.weight"] = g["coord_fc"]
    from dmpfold2_amd import synth
    assert synth.weights_checksum(sd) == bytes(g["weights_sha256"]).decode()
    eng = _engine(sd, 96, 64)
    eng.set_option("precision", MODES[mode])
    try:
        n, m = int(g["iterations"]), int(g["minsteps"])
        coords, confs = eng.predict(g["alnmat"], None, n, m)
        eng.sync_check()
        coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
        tol = max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
        assert tol <= 1.5e-3                                   # the floor of these fixtures is small
        assert ca_rmsd(coords[:, 1], g["coords"][:, 1]) <= tol
        assert np.abs(confs - g["confs"]).max() < max(1e-4, 3.0 * float(g["noise_conf"]))
        means = eng.fetch("conf_means", n + 1).cpu().numpy()
        assert np.abs(means - g["conf_mean_pass"]).max() < 1e-3
        # the trace really is protein-like: bonded CA-CA distances near 3.8 A after the minimiser
        ca = coords[:, 1]
        bond = np.linalg.norm(ca[1:] - ca[:-1], axis=1)
        assert 3.5 < bond.min() and bond.max() < 6.0
    finally:
        eng.close()


# ------------------------------------------------------------------ device faults at the boundary
def _hot_weights(synth_sd):
    """Synthetic weights whose block-6 InstanceNorm scales its output by 4e4: the residual stream
    leaves the f16 range (|x| up to 4e5) from block 7 on.  The head weights are scaled back by the same
    factor, so the distance map stays O(10) and the problem well conditioned (without that the
    REFERENCE differs from itself by 0.4-36 A between 8 and 1 threads; with it by 1e-5..8e-5 A)."""
    sd = dict(synth_sd)
    sd["resnet.6.layer1.norm.weight"] = (sd["resnet.6.layer1.norm.weight"] * 4e4).astype(np.float32)
    sd["resnet.6.layer1.norm.bias"] = (sd["resnet.6.layer1.norm.bias"] * 4e4).astype(np.float32)
    sd["resnet.17.weight"] = (sd["resnet.17.weight"] / 4e4).astype(np.float32)
    return sd


def test_f16_range_fault_is_reported_poisoned_and_cured(synth_sd, tmp_path, capsys, monkeypatch):
    from dmpfold2_amd import aln_to_coords, synth
    from dmpfold2_amd import predict as P
    sd = _hot_weights(synth_sd)
    rows = synth.synth_msa(40, 64, 1)
    alnmat = P.encode_aln(rows)
    ow = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    ref_c, ref_f = O.predict(alnmat, ow, None, 1, 0, "canonical")
    eng = _engine(sd, 64, 64)
    try:
        # round 4: with the per-block activation scale (chosen from the InstanceNorm weights at dmp_weights_finalize)
        # the default convolution scales this trunk DOWN and predicts it without a fault
        assert eng.get_option("act_scaling") == 1
        assert eng.get_option("act_scale_log2_block7") < 0 < eng.get_option("act_scale_log2_block1")
        c, f = eng.predict(alnmat, None, 1, 0)
        assert eng.sync_faults() == 0
        assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3
        assert np.abs(f.cpu().numpy() - ref_f.numpy()).max() < 1e-4
        # the fault machinery itself, with unscaled pieces (act_scaling = 0: the round-3 behaviour).
        # raw call: the fault is recorded, the outputs are NaN, the report clears the word
        eng.set_option("act_scaling", 0)
        coords, confs = eng.predict(alnmat, None, 1, 0)
        bits = eng.sync_faults()
        assert bits == P.FAULT_F16_RANGE
        assert bool(torch.isnan(coords).all()) and bool(torch.isnan(confs).all())
        assert eng.sync_faults() == 0
        with pytest.raises(P.DeviceFault):
            eng.predict(alnmat, None, 1, 0)
            eng.sync_check()
        # the range-free convolutions give the reference's answer
        for mode in (2, 1):
            eng.set_option("conv_mode", mode)
            c, f = eng.predict(alnmat, None, 1, 0)
            eng.sync_check()
            assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3, mode
            assert np.abs(f.cpu().numpy() - ref_f.numpy()).max() < 1e-4, mode
        eng.set_option("conv_mode", 0)
        # the checked entry re-runs by itself and says so
        capsys.readouterr()
        c, f = eng.predict_checked(alnmat, None, 1, 0)
        assert "conv_mode=2" in capsys.readouterr().err
        assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3
        assert eng.get_option("conv_mode") == 0
    finally:
        eng.close()
    # the public function: same cure through aln_to_coords, and the engine is healthy afterwards (the FAST mode, selected
    # through the environment: the drop-in default, precision 2, has float32's range and nothing to cure)
    monkeypatch.setenv("DMPFOLD_PRECISION", "0")
    P._ENGINES.clear()
    wf, aln = str(tmp_path / "hot.pt"), str(tmp_path / "t.aln")
    synth.save_state_dict(wf, sd)
    synth.write_aln(aln, rows)
    cached = P.get_engine("cuda:0", 40, 64, weights_file=wf)          # the engine aln_to_coords will use
    cached.set_option("act_scaling", 0)
    try:
        c, f = aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=0, weights_file=wf)
        assert "conv_mode=2" in capsys.readouterr().err
        assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3
        assert np.abs(f.cpu().numpy() - ref_f.numpy()).max() < 1e-4
    finally:
        cached.set_option("act_scaling", 1)
    c, f = aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=0, weights_file=wf)
    assert "conv_mode=2" not in capsys.readouterr().err              # scaled pieces: no fault, no re-run
    assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3
    assert np.abs(f.cpu().numpy() - ref_f.numpy()).max() < 1e-4
    # ... and in the drop-in default (full-width operands, float32's range) there is nothing to cure even unscaled
    monkeypatch.delenv("DMPFOLD_PRECISION")
    cached = P.get_engine("cuda:0", 40, 64, weights_file=wf)
    assert cached.get_option("precision") == 2
    cached.set_option("act_scaling", 0)
    try:
        c, f = aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=0, weights_file=wf)
        assert "conv_mode=2" not in capsys.readouterr().err
        assert ca_rmsd(c.cpu().numpy()[:, 1], ref_c.numpy()[:, 1]) <= 1e-3
    finally:
        cached.set_option("act_scaling", 1)
        P._ENGINES.clear()


def test_unknown_residue_code_raises_index_error(small_engine, tmp_path, weights_file):
    """network.py:223: a code above 21 fails in the reference's embedding.  File input: raised by the
    host parser; device-resident codes: flagged by the first kernel that reads them."""
    from dmpfold2_amd import aln_to_coords, synth
    rows = synth.synth_msa(24, 8, 3)
    rows[5] = rows[5][:7] + "x" + rows[5][8:]
    aln = tmp_path / "lower.aln"
    synth.write_aln(str(aln), rows)
    with pytest.raises(IndexError):
        aln_to_coords(str(aln), device="cuda:0", iterations=0, minsteps=0, weights_file=weights_file)
    codes = O.encode_aln(synth.synth_msa(24, 8, 3)).copy()
    codes[5, 7] = 22
    eng = small_engine
    d = torch.from_numpy(codes).to("cuda:0")
    coords, confs = eng.predict_device(d, None, 0, 0)
    with pytest.raises(IndexError):
        eng.sync_check()
    assert bool(torch.isnan(coords).all())
    # reported once; the next (valid) prediction is clean
    codes[5, 7] = 21
    coords, confs = eng.predict(codes, None, 0, 0)
    eng.sync_check()
    assert bool(torch.isfinite(coords).all())


def test_scheduler_isolates_a_faulting_target(synth_sd):
    """Pipeline.collect: the one target with a bad residue code comes back as its exception, the
    others as the bits the single engine computes."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Pipeline
    dev = torch.device("cuda:0")
    msas = [O.encode_aln(synth.synth_msa(L, N, 60 + i)) for i, (L, N) in
            enumerate([(40, 30), (33, 12), (48, 64), (24, 5)])]
    bad = msas[1].copy()
    bad[3, 3] = 30
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    pipe = Pipeline(dev, 64, 64, sd, streams=3)
    try:
        order = [msas[0], bad, msas[2], msas[3], msas[1]]
        tickets = [pipe.submit(torch.from_numpy(m).to(dev), 1, 2) for m in order]
        res = pipe.collect(tickets)
        assert isinstance(res[tickets[1]], IndexError)
        single = pipe.engines[0]
        for t, m in zip(tickets, order):
            if t == tickets[1]:
                continue
            c, f = single.predict(m, None, 1, 2)
            single.sync_check()
            assert torch.equal(c, res[t][0]) and torch.equal(f, res[t][1])
    finally:
        pipe.close()


def test_hot_target_in_a_batch_is_re_run_alone(synth_sd):
    """A range fault inside the scheduler: that target is repeated through the checked single-engine
    path (conv_mode 2) and every target of the batch gets a valid result."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Pipeline
    dev = torch.device("cuda:0")
    sd = _hot_weights(synth_sd)
    ow = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    msas = [O.encode_aln(synth.synth_msa(L, N, 70 + i)) for i, (L, N) in enumerate([(40, 30), (32, 16)])]
    pipe = Pipeline(dev, 64, 64, ow, streams=2)
    try:
        for e in pipe.engines:
            e.set_option("act_scaling", 0)         # unscaled pieces, so that these weights do leave the f16 range
        tickets = [pipe.submit(torch.from_numpy(m).to(dev), 1, 0) for m in msas]
        res = pipe.collect(tickets)
        for t, m in zip(tickets, msas):
            rc, rf = O.predict(m, ow, None, 1, 0, "canonical")
            c, f = res[t]
            assert ca_rmsd(c.cpu().numpy()[:, 1], rc.numpy()[:, 1]) <= 1e-3
            assert np.abs(f.cpu().numpy() - rf.numpy()).max() < 1e-4
    finally:
        pipe.close()


# ------------------------------------------------------------------ weight ABI: strict load
def test_weight_abi_rejects_unknown_missing_and_misshaped_tensors(synth_sd):
    """load_state_dict(strict) semantics of predict.py:98: RuntimeError naming the key."""
    from dmpfold2_amd.predict import Engine
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    eng = Engine("cuda:0", 32, 8)
    try:
        extra = dict(sd)
        extra["resnet.18.weight"] = torch.zeros(2, 128, 1, 1)
        with pytest.raises(RuntimeError, match="resnet.18.weight"):
            eng.set_weights(extra)
        missing = {k: v for k, v in sd.items() if k != "coord_fc.weight"}
        with pytest.raises(RuntimeError, match="coord_fc.weight"):
            eng.set_weights(missing)
        shaped = dict(sd)
        shaped["vgru.weight_ih_l0"] = torch.zeros(1536, 21)
        with pytest.raises(RuntimeError, match="vgru.weight_ih_l0"):
            eng.set_weights(shaped)
        flat = dict(sd)
        flat["resnet.17.weight"] = torch.zeros(2, 128)          # right count, wrong rank
        with pytest.raises(RuntimeError, match="resnet.17.weight"):
            eng.set_weights(flat)
        eng.set_weights(sd)                                       # and the good dict still loads
        c, f = eng.predict(O.encode_aln(["ACDEFGHIKLMNPQRS"] * 3), None, 0, 0)
        eng.sync_check()
        assert bool(torch.isfinite(c).all())
    finally:
        eng.close()


def test_two_part_default_weights_route(synth_sd, tmp_path, monkeypatch, weights_file):
    """predict.py:81-92: no -w -> trained_model/FINAL_fullmap_e2e_model_part{1,2}.pt merged.  The two
    halves of the synthetic state_dict in that layout give the bits of the single-file route."""
    from dmpfold2_amd import predict as P
    keys = list(synth_sd)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    d = tmp_path / "trained_model"
    d.mkdir()
    parts = [str(d / f"FINAL_fullmap_e2e_model_part{i}.pt") for i in (1, 2)]
    torch.save({k: sd[k] for k in keys[:100]}, parts[0])
    torch.save({k: sd[k] for k in keys[100:]}, parts[1])
    aln = os.path.join(os.path.dirname(__file__), "golden", "PF10963.aln")
    c1, f1 = P.aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=3, weights_file=weights_file)
    monkeypatch.setattr(P, "default_weight_files", lambda: parts)
    c2, f2 = P.aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=3)
    assert torch.equal(c1, c2) and torch.equal(f1, f2)
    monkeypatch.undo()
    P._ENGINES.clear()


def test_drop_in_default_is_full_width_and_the_environment_selects_the_arithmetic(monkeypatch, weights_file, synth_sd):
    """aln_to_coords / the CLI have no argument for the arithmetic.  Their default - and that of a context made through
    the C ABI or of an Engine - is option "precision" = 2: the reference computes in float32 (predict.py:136,
    network.py:25-31), so float32's 24-bit operands (three exact bf16 pieces, float32 vertical GRU); DMPFOLD_PRECISION
    selects the others (this suite's conftest sets it to 0 for every other test): 1 = the f32 matrix-core
    instructions, 0 = the fast 22-bit mode.  Each gives the bits of an engine set to that precision by hand; the three
    differ from each other by rounding only."""
    from dmpfold2_amd import predict as P
    aln = os.path.join(os.path.dirname(__file__), "golden", "PF10963.aln")
    alnmat = P.encode_aln(P.read_aln(aln))
    monkeypatch.delenv("DMPFOLD_PRECISION", raising=False)
    got = {}
    for env, want in ((None, 2), ("1", 1), ("0", 0), ("2", 2)):
        P._ENGINES.clear()
        if env is None:
            monkeypatch.delenv("DMPFOLD_PRECISION", raising=False)
        else:
            monkeypatch.setenv("DMPFOLD_PRECISION", env)
        c, f = P.aln_to_coords(aln, device="cuda:0", iterations=1, minsteps=0, weights_file=weights_file)
        eng = P._ENGINES[0]
        assert eng.get_option("precision") == want and eng.get_option("vgru_f32") == want
        assert eng.get_option("conv_mode") == want
        if want in got:
            assert torch.equal(got[want][0], c) and torch.equal(got[want][1], f)       # unset == "2"
        got[want] = (c, f)
    monkeypatch.delenv("DMPFOLD_PRECISION")
    P._ENGINES.clear()
    e = P.Engine("cuda:0", alnmat.shape[1], alnmat.shape[0])
    try:
        e.set_weights({k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()})
        assert e.get_option("precision") == 2              # an engine made directly starts in the library's setting ...
        raw = C.c_void_p()                                 # ... which is a context's own: full-width operands
        lib = P._lib.load()
        P._lib.check(lib.dmp_ctx_create(0, 64, 8, C.byref(raw)))
        v = C.c_int(-5)
        P._lib.check(lib.dmp_ctx_get_option(raw, b"precision", C.byref(v)))
        lib.dmp_ctx_destroy(raw)
        assert v.value == 2
        for prec in (0, 1, 2):
            e.set_option("precision", prec)
            c2, f2 = e.predict_checked(alnmat, None, 1, 0)
            assert torch.equal(got[prec][0], c2) and torch.equal(got[prec][1], f2), prec
    finally:
        e.close()
    for a, b in ((0, 1), (0, 2), (1, 2)):
        assert not torch.equal(got[a][0], got[b][0])
        assert float((got[a][0][:, 1] - got[b][0][:, 1]).pow(2).sum(-1).mean().sqrt()) < 1e-3       # ... by rounding only
    monkeypatch.setenv("DMPFOLD_PRECISION", "3")
    with pytest.raises(ValueError, match="DMPFOLD_PRECISION"):
        P.Engine("cuda:0", 64, 8)


def test_aln_to_coords_is_reentrant_two_threads_two_weight_files(tmp_path, synth_sd, weights_file):
    """The reference builds a fresh network per call (predict.py:79), so its aln_to_coords may be called from several
    threads, each with its own weights.  Here the calls of a GPU share cached engines: two threads, two weight files, two
    alignments, interleaved a few times each - every result equals the same call made alone, bit for bit."""
    import threading
    from dmpfold2_amd import predict as P, synth
    sd2 = synth.synth_weights(1, coord_scale=5.0)
    wf2 = str(tmp_path / "seed1.pt")
    synth.save_state_dict(wf2, sd2)
    alns = []
    for i, (L, N) in enumerate(((40, 64), (56, 48))):
        a = str(tmp_path / f"t{i}.aln")
        synth.write_aln(a, synth.synth_msa(L, N, 30 + i))
        alns.append(a)
    jobs = [(alns[0], weights_file), (alns[1], wf2)]
    P._ENGINES.clear()
    alone = [P.aln_to_coords(a, device="cuda:0", iterations=2, minsteps=5, weights_file=w) for a, w in jobs]
    P._ENGINES.clear()
    results, errors = [[], []], []

    def worker(k):
        try:
            for _ in range(4):
                results[k].append(P.aln_to_coords(jobs[k][0], device="cuda:0", iterations=2, minsteps=5, weights_file=jobs[k][1]))
        except Exception as exc:                      # noqa: BLE001
            errors.append(exc)
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert len(results[k]) == 4
        for c, f in results[k]:
            assert torch.equal(c, alone[k][0]) and torch.equal(f, alone[k][1]), k
    assert len(P._ENGINES.lru[0]) == 2               # one cached engine per weights file: no re-packing in between
    P._ENGINES.clear()


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_missing_workgroup_of_the_persistent_chain_times_out_once_and_falls_back(synth_sd, capsys, precision):
    """a workgroup of the persistent vertical GRU that is missing for good - another process holds CUs -
    used to make EVERY row wait the full spin bound again (hours at N = 3000).  With the test option the chain is launched
    one workgroup short: the row barrier of one XCD times out once, DMP_FAULT_VGRU_HANDOFF is raised, the kernel leaves
    its row loop, and predict_checked repeats with one launch per row - within seconds, with the right answer, in all
    three arithmetic settings."""
    import time
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, encode_aln, FAULT_VGRU_HANDOFF
    msa = encode_aln(synth.synth_msa(96, 400, 77))
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    eng = Engine("cuda:0", 96, 400)
    try:
        eng.set_weights(sd)
        eng.set_option("precision", precision)
        eng.set_option("vgru_persistent", 0)
        want_c, want_f = eng.predict_checked(msa, None, 1, 0)          # the launch-per-row answer
        eng.set_option("vgru_persistent", 1)
        eng.set_option("vgru_debug_drop_wg", 1)
        d_msa = torch.from_numpy(msa).to(eng.device)
        t0 = time.perf_counter()
        c, f = eng.predict_device(d_msa, None, 1, 0)
        bits = eng.sync_faults()
        dt = time.perf_counter() - t0
        assert bits & FAULT_VGRU_HANDOFF and bool(torch.isnan(f).all())   # flagged, and never a plausible wrong structure
        assert dt < 5.0, dt                                               # ONE time-out, not one per row (401 rows here)
        t0 = time.perf_counter()
        c, f = eng.predict_device_checked(d_msa, None, 1, 0)            # times out again, then falls back
        dt = time.perf_counter() - t0
        assert "one launch per alignment row" in capsys.readouterr().err
        assert eng.get_option("vgru_persistent") == 0
        assert torch.equal(c, want_c) and torch.equal(f, want_f) and dt < 10.0, dt
    finally:
        eng.close()


def test_pipeline_whose_chains_time_out_repeats_and_switches_every_engine(synth_sd, capsys):
    """in a Pipeline the launch-per-row fallback used to be set on the engine that ran the repeat only - the other
    engines (and a group chain led by one of them) kept faulting, target after target.  Every chain of this pipeline is
    launched one workgroup short: collect() repeats the faulted targets, switches ALL engines, and every target gets the
    single engine's launch-per-row bits."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, Pipeline, encode_aln
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    msas = [encode_aln(synth.synth_msa(L, N, 300 + i)) for i, (L, N) in enumerate([(64, 120), (48, 60), (80, 33), (40, 200), (64, 64)])]
    single = Engine(dev, 80, 200)
    pipe = Pipeline(dev, 80, 200, sd, streams=2)
    try:
        single.set_weights(sd)
        single.set_option("vgru_persistent", 0)
        single.set_option("tridiag_cluster", 0)
        for e in pipe.engines:
            e.set_option("vgru_debug_drop_wg", 1)
        tickets = [pipe.submit(torch.from_numpy(m).to(dev), 1, 2) for m in msas]
        out = pipe.collect(tickets)
        assert "one launch per alignment row" in capsys.readouterr().err
        assert all(e.get_option("vgru_persistent") == 0 for e in pipe.engines)
        for m, t in zip(msas, tickets):
            assert not isinstance(out[t], Exception), out[t]
            c, f = single.predict(m, None, 1, 2)
            single.sync_check()
            assert torch.equal(out[t][0], c) and torch.equal(out[t][1], f), m.shape
        # ... and the next batch runs clean on the switched engines
        more = [pipe.submit(torch.from_numpy(m).to(dev), 1, 2) for m in msas[:3]]
        res = pipe.collect(more)
        for m, t in zip(msas[:3], more):
            c, f = single.predict(m, None, 1, 2)
            single.sync_check()
            assert torch.equal(res[t][0], c) and torch.equal(res[t][1], f)
    finally:
        pipe.close()
        single.close()


# ------------------------------------------------------------------ BASELINE config[3], sharded
def _parse_pdb(text):
    ca, conf = [], None
    for line in text.splitlines():
        if line.startswith("REMARK  CONF:"):
            conf = float(line.split()[-1])
        if line[:4] == "ATOM" and line[12:16] == " CA ":
            ca.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return np.asarray(ca), conf


def test_config3_batch_sharded_two_ways(tmp_path, weights_file):
    """BASELINE configs[3] at reduced count: synthetic targets with L in [100, 300] and N=2000 plus the
    reference's example alignment, 10 iterations, split over world=2 shards (both run here, one after
    the other, on the one GPU).  Every PDB equals the text of the single-target CLI, the shards cover
    every target exactly once, and the example's structure matches the REFERENCE's (-n 10 -m 0 golden)."""
    from dmpfold2_amd import run_dmpfold, synth
    from dmpfold2_amd.batch import run_batch
    rng = np.random.Generator(np.random.Philox(key=256))
    paths = []
    for k, L in enumerate(rng.integers(100, 301, size=5)):
        q = tmp_path / f"t{k}_L{L}.aln"
        synth.write_aln(str(q), synth.synth_msa(int(L), 2000, seed=300 + k))
        paths.append(str(q))
    g = load_golden("pf10963_n10_m0")
    p = tmp_path / "pf10963.aln"
    p.write_text("\n".join(golden_rows(g)) + "\n")
    paths.append(str(p))
    targets = [(a, None) for a in paths]
    outs, counts = [], []
    for rank in (0, 1):
        n, secs, o = run_batch(targets, str(tmp_path / "out"), 10, 0, weights_file=weights_file,
                               streams=4, device="cuda:0", rank=rank, world=2)
        counts.append(n)
        outs += o
    assert sum(counts) == 6 and min(counts) >= 1 and len(set(outs)) == 6
    for a in paths:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            run_dmpfold(["-i", a, "-d", "cuda:0", "-n", "10", "-m", "0", "-w", weights_file])
        name = os.path.splitext(os.path.basename(a))[0] + ".pdb"
        assert (tmp_path / "out" / name).read_text() == buf.getvalue(), a
    ca, conf = _parse_pdb((tmp_path / "out" / "pf10963.pdb").read_text())
    # %8.3f text: 5e-4 A of rounding per coordinate on top of the 1e-3 A tolerance
    assert ca_rmsd(ca, g["coords"][:, 1]) <= 1e-3 + 5e-4
    assert abs(conf - float(g["confs"].mean())) < 1e-4


def test_config3_full_256_targets_world_8(tmp_path, weights_file):
    """BASELINE configs[3] IN FULL: 256 synthetic targets, L uniform in [100, 300], N=2000,
    iterations=10, minsteps=100, split over world=8 (the eight shards run here one after the other on the
    one GPU, each through its own 4-engine scheduler, as eight ranks would).  Every target is written exactly
    once, by the rank plan_shard gives it to; a sample of targets (the two longest, the two shortest and one
    per shard) equals the text of the single-target CLI byte for byte; every structure is finite
    (no poisoned output slips through)."""
    from dmpfold2_amd import run_dmpfold, synth
    from dmpfold2_amd.batch import run_batch, plan_shard
    import subprocess
    import sys
    rng = np.random.Generator(np.random.Philox(key=256))
    lengths = rng.integers(100, 301, size=256)
    paths = [str(tmp_path / f"t{k:03d}_L{L}.aln") for k, L in enumerate(lengths)]
    # the 256 alignments (0.1 GB of text) are written by a pool of fresh processes (no HIP state to fork)
    maker = ("import sys, multiprocessing as mp\n"
             "sys.path.insert(0, sys.argv[1])\n"
             "from dmpfold2_amd import synth\n"
             "def one(a):\n"
             "    synth.write_aln(a[0], synth.synth_msa(a[1], 2000, seed=a[2]))\n"
             "if __name__ == '__main__':\n"
             "    jobs = [l.split() for l in open(sys.argv[2])]\n"
             "    with mp.Pool(min(16, mp.cpu_count())) as pool:\n"
             "        pool.map(one, [(p, int(L), int(sd)) for p, L, sd in jobs], chunksize=4)\n")
    (tmp_path / "make.py").write_text(maker)
    (tmp_path / "jobs.txt").write_text("".join(f"{p} {int(L)} {1000 + k}\n" for k, (p, L) in enumerate(zip(paths, lengths))))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, str(tmp_path / "make.py"), root, str(tmp_path / "jobs.txt")], check=True)
    targets = [(a, None) for a in paths]
    out_dir = tmp_path / "out"
    owner, total_s = {}, 0.0
    for rank in range(8):
        mine = plan_shard(targets, 10, rank, 8)
        n, secs, outs = run_batch(targets, str(out_dir), 10, 100, weights_file=weights_file,
                                  streams=4, device="cuda:0", rank=rank, world=8)
        assert n == len(mine) == len(outs)
        total_s += secs
        for o in outs:
            assert o not in owner, o                     # exactly once over the ranks
            owner[o] = rank
    assert len(owner) == 256
    assert {os.path.basename(o) for o in owner} == {os.path.splitext(os.path.basename(a))[0] + ".pdb" for a in paths}
    print("configs[3]: 256 targets in %.1f s of shard time = %.1f structures/s on one GPU" % (total_s, 256 / total_s))
    for o in owner:
        ca, conf = _parse_pdb(open(o).read())
        assert np.isfinite(ca).all() and 0.0 < conf < 1.0, o
    order = np.argsort(lengths)
    sample = {int(order[0]), int(order[1]), int(order[-1]), int(order[-2])}
    for rank in range(8):
        sample.add(int(plan_shard(targets, 10, rank, 8)[-1]))
    for i in sorted(sample):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            run_dmpfold(["-i", paths[i], "-d", "cuda:0", "-n", "10", "-m", "100", "-w", weights_file])
        name = os.path.splitext(os.path.basename(paths[i]))[0] + ".pdb"
        assert (out_dir / name).read_text() == buf.getvalue(), paths[i]


def test_accuracy_harness_runs_when_trained_weights_exist():
    """TM-score / RMSD of the PF10963 prediction against 3FGX chain A needs the trained
    weights, which are not in the tree (.MISSING_LARGE_BLOBS); the harness is staged and runs as soon
    as they are placed in dmpfold2_amd/trained_model/."""
    from dmpfold2_amd import predict as P
    if not all(os.path.isfile(f) for f in P.default_weight_files()):
        pytest.skip("trained weights absent: " + P.default_weight_files()[0])
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("accuracy_3fgx", os.path.join(root, "tools", "accuracy_3fgx.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.evaluate(os.path.join(root, "tests", "golden", "PF10963.aln"),
                       os.path.join(root, "tests", "golden", "kat_refine_backbone.npz"))
    assert res["tm_score"] > 0.5, res


# ------------------------------------------------------------------ the reference's LAPACK sign flavour
def test_lapack_sign_flavour_given_its_signs(synth_sd):
    """Eigenvector signs of `symeig` are implementation-defined; the HIP solver fixes them by rule
    (largest-magnitude component positive) and is compared with the goldens of that flavour.  The other
    captured flavour - MKL's signs as `torch.linalg.eigh` returns them - differs from it ONLY in those sign
    bits: flipping the MDS columns whose recorded LAPACK sign is negative and running the rest of the path
    (coordinate GRU, coord_fc, backbone) through the stage API reproduces the LAPACK-flavour golden."""
    from abi import Stages
    g = load_golden("pf10963_n0_m0_lapack")
    assert bytes(g["sign_mode"]).decode() == "lapack"
    signs = g["mds_sign_ref"][0]                                   # sign of each eigenvector's largest component
    assert (signs < 0).any() and (signs > 0).any()                 # the two flavours really differ here
    st = Stages(synth_sd, max_L=128, max_N=512)
    try:
        L = 82
        coords, confs = st.eng.predict(g["alnmat"], None, 0, 0)
        st.eng.sync_check()
        assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) > 0.05     # canonical signs: another trace
        mds = st.eng.fetch("mds", L * 8).reshape(L, 8).clone()
        mat1d = st.eng.fetch("mat1d", 512 * L).reshape(512, L).clone()
        flipped = (mds * torch.from_numpy(signs).to(mds.device)).contiguous()
        ca = st.coords_from_mds(mat1d, flipped)
        logit = torch.zeros(L, device=ca.device)
        bb, _ = st.backbone(ca, logit)
        st.eng.sync_check()
        assert ca_rmsd(ca.cpu().numpy(), g["ca_pass"][0]) <= 1e-3
        assert ca_rmsd(bb.cpu().numpy()[:, 1], g["coords"][:, 1]) <= 1e-3
        assert np.abs(bb.cpu().numpy() - g["coords"]).max() < 2e-2
        assert np.abs(confs.cpu().numpy() - g["confs"]).max() < 1e-4               # confidences do not depend on signs
    finally:
        st.eng.close()


# ------------------------------------------------------------------ bench.py with more than one rank
def test_bench_two_ranks_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` outside a launcher: it starts two ranks itself, each runs its own scheduler
    and targets, rank 0 prints one line with n_gpus = ranks = 2 and the whole-job rate.  On this one-GPU box
    both ranks share cuda:0 (DMP_BENCH_SHARE_GPU=1: gloo instead of RCCL for the barrier / reductions), so
    only the flow and the verification are checked, not the rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMP_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--streams", "2", "--cpu-baseline", "none", "--no-exact-f32"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["verify"]["ok"] and d["verify"]["reference_golden_L300_N2000_n1_m0"]["ok"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] + 1e-9
    assert "cpu_baseline" not in d                       # rank 0 times the CPU oracle at N = 1 only


def test_bench_under_the_drivers_launcher_over_rccl():
    """the driver's own command line for N > 1 (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`) with N = 1 and DMP_FORCE_DIST=1,
    so that the RCCL process group (backend "nccl"), its barrier and the MAX / MIN all-reduces of bench.py really
    execute on the MI355X once - the multi-rank test above has to use gloo (two ranks on one device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    env = dict(os.environ, DMP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DMP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    cmd = bench.launch_command(1, ["--gpus", "1", "--steps", "1", "--warmup", "1", "--streams", "2",
                                   "--cpu-baseline", "none"], bench.free_port())
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks"] == 1 and d["verify"]["ok"]
    assert d["value"] > 0 and d["value_f32"] > 0 and 0 < d["roofline_f32"]["frac"] < 1 and 0 < d["roofline"]["frac"] < 1
