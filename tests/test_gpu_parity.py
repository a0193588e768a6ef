"""Parity of the HIP path (through the C ABI) against the CPU oracle and the goldens.

Every test feeds a HIP stage the ORACLE's input for that stage, so one stage's error never
hides in another's.  Tolerances (SURVEY.md section 8c / BASELINE.json north_star):
  integer / index work        bit exact
  covariance, inverse, APC    relative 1e-5 of the tensor's scale
  GRU states                  absolute 1e-5
  trunk activations           absolute 1e-4 * max(1, scale)
  end to end (m = 0)          CA-RMSD <= 1e-3 Angstrom, |dconf| < 1e-4
"""
import io
import contextlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, golden_rows, ca_rmsd

pytestmark = pytest.mark.gpu

import dmpfold_oracle as O          # noqa: E402  (test infrastructure)


@pytest.fixture(scope="module")
def st_engine(synth_sd):
    from abi import Stages
    return Stages(synth_sd, max_L=128, max_N=3000)


CONV_MODES = {"f16x3": 0, "f32": 1, "bf16x6": 2}


@pytest.fixture(params=list(CONV_MODES))
def st(request, st_engine):
    """Every test runs with each convolution path at the SAME tolerances: the default (float32
    products from 2-way f16 splits, 3 MFMA products), the exact-f32 MFMA path, and the 3-way bf16
    split (6 products)."""
    st_engine.eng.set_option("conv_mode", CONV_MODES[request.param])
    yield st_engine
    st_engine.eng.set_option("conv_mode", 0)
    st_engine.eng.sync_check()


@pytest.fixture(scope="module")
def pf():
    return load_golden("pf10963_n0_m0")


@pytest.fixture(scope="module")
def ocap(pf, oracle_weights):
    """Oracle stage tensors for PF10963, n=0 m=0."""
    cap = {}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    O.predict(pf["alnmat"], oracle_weights, None, 0, 0, "canonical", cap)
    return cap


def scale_tol(ref, rel):
    return rel * max(1.0, float(np.abs(ref).max()))


# ---------------------------------------------------------------------------- features
@pytest.mark.parametrize("name", ["pf10963_n0_m0", "synth_L40_N64_n2_m0", "synth_L24_N3050_n1_m0",
                                  "alphabet_L16_N12_n0_m0", "template_L96_N50_n1_m0"])
def test_msa_weights_bit_exact(st, name):
    g = load_golden(name)
    w = st.msa_weights(g["alnmat"]).cpu().numpy()
    assert np.array_equal(w, g["w"])
    assert np.array_equal(w, O.reweight(g["alnmat"]))


def test_msa_weights_ragged_sizes(st):
    rng = np.random.default_rng(1)
    for n, L in [(1, 9), (2, 8), (65, 13), (130, 127), (257, 33)]:
        a = rng.integers(0, 22, size=(n, L)).astype(np.uint8)
        a[n // 2] = a[0]                       # force at least one close pair
        assert np.array_equal(st.msa_weights(a).cpu().numpy(), O.reweight(a))


def test_covariance_inverse_contacts(st, pf, ocap):
    w = st.to(ocap["w"].numpy())
    cov = st.cov_build(pf["alnmat"], w).cpu().numpy()
    ref = ocap["cov_reg"].numpy()
    assert np.abs(cov - ref).max() <= scale_tol(ref, 1e-5)
    inv = st.spd_inverse(st.to(ref)).cpu().numpy()
    iref = ocap["inv_cov"].numpy()
    assert np.abs(inv - iref).max() <= scale_tol(iref, 1e-5)
    con = st.dca_contacts(st.to(iref), pf["alnmat"].shape[1]).cpu().numpy()
    cref = ocap["contacts"].numpy()
    assert np.abs(con - cref).max() <= scale_tol(cref, 1e-5)
    assert np.abs(con - pf["contacts"]).max() <= scale_tol(cref, 1e-5)     # golden from the reference


@pytest.mark.parametrize("name", ["pf10963_n0_m0", "synth_L40_N64_n2_m0"])
def test_dca_features_layout_vs_reference(st, name):
    """dmp_dca_features = fast_dca's return value (L, L, 442) in the reference's own layout
    (predict.py:54-61), against the samples / checksums the goldens hold of the tensor the REFERENCE
    returned: all 441 coupling channels and the contact channel, by flat index."""
    g = load_golden(name)
    f = st.dca_features(g["alnmat"]).cpu().numpy()
    L = g["alnmat"].shape[1]
    assert f.shape == (L, L, 442)
    flat = f.ravel()
    ref = g["f2d.val"]
    assert np.abs(flat[g["f2d.idx"]] - ref).max() <= scale_tol(ref, 1e-5)
    assert abs(flat.astype(np.float64).sum() - float(g["f2d.sum"])) <= 1e-5 * max(1.0, np.abs(flat).sum())
    assert abs((flat.astype(np.float64) ** 2).sum() - float(g["f2d.sumsq"])) <= 2e-5 * float(g["f2d.sumsq"])
    assert np.abs(f[:, :, 441] - g["contacts"]).max() <= scale_tol(g["contacts"], 1e-5)
    # the layout itself: channel 21a+b of pair (i, j) is inv_cov[21i+a, 21j+b]
    st.eng.predict(g["alnmat"], None, 0, 0)
    st.eng.sync_check()
    inv = st.eng.fetch("inv_cov", (21 * L) ** 2).cpu().numpy().reshape(L, 21, L, 21)
    assert np.array_equal(f[:, :, :441].reshape(L, L, 21, 21), inv.transpose(0, 2, 1, 3))


def test_dca_features_single_sequence_is_zero(st):
    """predict.py:139: no covariance features for a one-row alignment."""
    a = np.random.default_rng(0).integers(0, 20, size=(1, 17)).astype(np.uint8)
    assert not st.dca_features(a).cpu().numpy().any()


def test_spd_inverse_pairs_and_lookahead_are_bitwise_the_serial_order(st_engine):
    """Option gj_pairs (round 4, the default): two block steps' trailing updates in one pass over the tiles, each step's
    sum in its own MFMA chain.  Option gj_lookahead (round 4): the next block step's diagonal sweep and panels on a second stream beside this
    step's trailing update, the update issued in two parts.  Every tile sees the same operations in every form: same
    bits, for a single block, odd and even numbers of blocks, a ragged last block and the front end's 21 L sizes."""
    st = st_engine
    rng = np.random.default_rng(5)
    try:
        for D in (100, 128, 300, 512, 21 * 30 + 0, 21 * 61, 1536):
            X = rng.standard_normal((D, 2 * D)).astype(np.float32)
            A = st.to(X @ X.T / (2 * D) + 0.5 * np.eye(D, dtype=np.float32))
            out = {}
            st.eng.set_option("gj_pairs", 0)
            for mode in (0, 2):
                st.eng.set_option("gj_lookahead", mode)
                out[mode] = st.spd_inverse(A).cpu().numpy()
            st.eng.set_option("gj_pairs", 1)                 # block steps in pairs (the default): one pass per two steps
            out["pairs"] = st.spd_inverse(A).cpu().numpy()
            st.eng.sync_check()
            assert np.array_equal(out[0], out[2]), D
            assert np.array_equal(out[0], out["pairs"]), D
            assert np.abs(out[0] @ A.cpu().numpy() - np.eye(D)).max() < 2e-4
            if D > 128:          # (a single block is the sweep's result as it is; from two blocks on the upper
                assert np.array_equal(out[0], out[0].T)      # triangle is the mirrored lower one)
    finally:
        st.eng.set_option("gj_lookahead", 1)
        st.eng.set_option("gj_pairs", 1)


def test_spd_inverse_blocked_diagonal_sweep_vs_the_pivot_chain(st_engine):
    """Round 5 (option gj_diag_blocked = 1): the 128 x 128 diagonal block swept in 8 sub-blocks of 16 pivots
    (pivot block inside one wave, W = T P and the rank-16 update on the f32 matrix cores) against the chain of 128
    barrier-synchronised pivots of rounds 1-4: the same operator in another order - the inverses agree to float32
    rounding - for one block, a ragged last block, the front end's 21 L sizes; both satisfy A inv(A) = I."""
    st = st_engine
    rng = np.random.default_rng(11)
    try:
        for D in (16, 100, 128, 129, 21 * 30, 21 * 61, 1536):
            X = rng.standard_normal((D, 2 * D)).astype(np.float32)
            A = X @ X.T / (2 * D) + 0.5 * np.eye(D, dtype=np.float32)
            out = {}
            for blocked in (1, 0):
                st.eng.set_option("gj_diag_blocked", blocked)
                assert st.eng.get_option("gj_diag_blocked") == blocked
                out[blocked] = st.spd_inverse(st.to(A)).cpu().numpy()
            st.eng.sync_check()
            ref = np.linalg.inv(A.astype(np.float64))
            scale = np.abs(ref).max()
            assert np.abs(out[1] - out[0]).max() <= 1e-5 * scale, D
            print(f"D={D}: max error against the float64 inverse / max|inv|: blocked {np.abs(out[1] - ref).max() / scale:.2e}, "
                  f"chain {np.abs(out[0] - ref).max() / scale:.2e}")
            # ... and the blocked form is no further from the float64 inverse than the chain (x 2)
            assert np.abs(out[1] - ref).max() <= max(2.0 * np.abs(out[0] - ref).max(), 1e-6 * scale), D
            assert np.abs(out[1].astype(np.float64) @ A - np.eye(D)).max() < 5e-5
    finally:
        st.eng.set_option("gj_diag_blocked", 0)


def test_spd_inverse_identity_property(st):
    rng = np.random.default_rng(3)
    for D in (21 * 9, 21 * 13 + 0, 300):
        B = rng.standard_normal((D, D + 40)).astype(np.float32)
        A = (B @ B.T) / (D + 40) + 0.3 * np.eye(D, dtype=np.float32)
        inv = st.spd_inverse(st.to(A)).cpu().numpy().astype(np.float64)
        assert np.abs(inv @ A.astype(np.float64) - np.eye(D)).max() < 5e-5


# ---------------------------------------------------------------------------- sequence trunk
def test_gru_vertical(st, pf, oracle_weights):
    idx = torch.from_numpy(pf["alnmat"].astype(np.int64))
    x = oracle_weights["embed.weight"][idx]
    ref = O._gru(oracle_weights, "vgru", x, 22, 512, 2, False, False)[-1].numpy()
    got = st.gru_vertical(pf["alnmat"]).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5
    assert np.abs(got - pf["vgru_last"]).max() < 1e-5                      # golden


def test_gru_vertical_deep_and_truncated(st, oracle_weights):
    g = load_golden("synth_L24_N3050_n1_m0")
    a = g["alnmat"]                                                        # 3000 x 24 after the cap
    x = oracle_weights["embed.weight"][torch.from_numpy(a.astype(np.int64))]
    ref = O._gru(oracle_weights, "vgru", x, 22, 512, 2, False, False)[-1].numpy()
    assert np.abs(st.gru_vertical(a).cpu().numpy() - ref).max() < 1e-5


def test_gru_bidir(st, pf, ocap, oracle_weights):
    vin = torch.from_numpy(pf["vgru_last"])
    ref = O._gru(oracle_weights, "hgru", vin.unsqueeze(1), 512, 256, 2, True, False)[:, 0].numpy()
    got = st.gru_bidir(0, st.to(pf["vgru_last"])).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5
    assert np.abs(got.T - pf["mat1d"]).max() < 1e-5
    emb = torch.cat((ocap["mat1d"].t(), ocap["p0.mds"]), dim=1)
    ref = O._gru(oracle_weights, "coord_gru", emb.unsqueeze(0), 520, 256, 3, True, True)[0].numpy()
    got = st.gru_bidir(1, st.to(emb.numpy())).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5


# ---------------------------------------------------------------------------- pair trunk
def _resinp(ocap, pf, dmap):
    L = pf["alnmat"].shape[1]
    m = ocap["mat1d"]
    pair = (m.unsqueeze(1) * m.unsqueeze(2)).unsqueeze(0)
    inv = ocap["inv_cov"].view(L, 21, L, 21).transpose(1, 2).reshape(L, L, 441)
    f2d = torch.cat((inv, ocap["contacts"][:, :, None]), dim=2).permute(2, 0, 1).unsqueeze(0)
    return torch.cat((pair, f2d, dmap.view(1, 1, L, L)), dim=1)


def test_stem(st, pf, ocap, oracle_weights):
    L = pf["alnmat"].shape[1]
    dmap = torch.zeros(L, L) - 1
    ref = O.stem(oracle_weights, _resinp(ocap, pf, dmap))[0].numpy()
    z0 = st.stem_static(st.to(ocap["mat1d"].numpy()), st.to(ocap["inv_cov"].numpy()),
                        st.to(ocap["contacts"].numpy()))
    got = st.stem_update(z0, st.to(dmap.numpy())).cpu().numpy()
    assert np.abs(got - ref).max() <= scale_tol(ref, 1e-4)
    # a different distance channel re-uses the same static part
    ca = torch.from_numpy(np.random.default_rng(0).standard_normal((L, 3)).astype(np.float32) * 8)
    dmap2 = O.pair_distances(ca)
    ref2 = O.stem(oracle_weights, _resinp(ocap, pf, dmap2))[0].numpy()
    got2 = st.stem_update(z0, st.pair_distances(st.to(ca.numpy()))).cpu().numpy()
    assert np.abs(got2 - ref2).max() <= scale_tol(ref2, 1e-4)


def test_stem_single_sequence(st, oracle_weights):
    """N = 1: zero DCA features (predict.py:139) -> d_inv = d_contacts = NULL."""
    L = 30
    m = torch.from_numpy(np.random.default_rng(5).standard_normal((512, L)).astype(np.float32) * 0.2)
    pair = (m.unsqueeze(1) * m.unsqueeze(2)).unsqueeze(0)
    dmap = torch.zeros(L, L) - 1
    resinp = torch.cat((pair, torch.zeros(1, 442, L, L), dmap.view(1, 1, L, L)), dim=1)
    ref = O.stem(oracle_weights, resinp)[0].numpy()
    z0 = st.stem_static(st.to(m.numpy()), None, None)
    got = st.stem_update(z0, st.to(dmap.numpy())).cpu().numpy()
    assert np.abs(got - ref).max() <= scale_tol(ref, 1e-4)


@pytest.mark.parametrize("block", [1, 7, 16])
def test_block(st, ocap, oracle_weights, block):
    x = ocap["p0.stem"] if block == 1 else ocap["p0.block1"]
    u_ref = O.block_conv(oracle_weights, block, x)
    u, stats = st.conv(block, st.to(x[0].numpy()))
    assert np.abs(u.cpu().numpy() - u_ref[0].numpy()).max() <= scale_tol(u_ref.numpy(), 1e-5)
    s_ref = torch.stack((u_ref[0].double().sum(dim=(1, 2)), (u_ref[0].double() ** 2).sum(dim=(1, 2))), 1)
    assert np.abs(stats.cpu().numpy() - s_ref.numpy()).max() <= 1e-5 * float(s_ref.abs().max())
    out_ref = O.block_finish(oracle_weights, block, u_ref, x)[0].numpy()
    out = st.norm(block, st.to(u_ref[0].numpy()), st.to(s_ref.numpy(), torch.float64),
                  st.to(x[0].numpy())).cpu().numpy()
    assert np.abs(out - out_ref).max() <= scale_tol(out_ref, 1e-5)


def test_conv_paths_error_vs_float64(st_engine, oracle_weights):
    """The split-product convolutions must be as accurate as a float32 convolution: compare every
    path and PyTorch's float32 conv2d with a float64 convolution of the same float32 data."""
    L, block = 64, 3
    rng = np.random.default_rng(42)
    x = torch.from_numpy((rng.standard_normal((1, 128, L, L)) * 3).astype(np.float32))
    w = oracle_weights[f"resnet.{block}.layer1.lin.weight"]
    b = oracle_weights[f"resnet.{block}.layer1.lin.bias"]
    truth = F.conv2d(x.double(), w.double(), b.double(), padding=2).view(1, 128, 4, L, L).max(dim=2)[0][0]
    f32 = O.block_conv(oracle_weights, block, x)[0].double()
    err_torch = float((f32 - truth).abs().max())
    errs = {}
    for name, mode in CONV_MODES.items():
        st_engine.eng.set_option("conv_mode", mode)
        u, _ = st_engine.conv(block, st_engine.to(x[0].numpy()))
        errs[name] = float((u.cpu().double() - truth).abs().max())
    st_engine.eng.set_option("conv_mode", 0)
    scale = float(truth.abs().max())
    print("max |err| vs float64 (output scale %.2f): torch f32 %.3e" % (scale, err_torch), errs)
    # (measured: torch 2.6e-6, f16x3 7.8e-6, bf16x6 1.1e-5, f32 MFMA 1.2e-5 at scale ~10; PyTorch's
    # blocked summation is the most accurate, the k-ordered fmaf chain of the f32 MFMA the least)
    for name, e in errs.items():
        assert e <= 1e-5 * max(1.0, scale), (name, e)                 # the parity tolerance
        assert e <= 1.5 * errs["f32"] + 1e-7, (name, e, errs["f32"])  # no worse than exact f32 MFMA


def test_f16_range_fault_is_reported(st_engine):
    """Activations beyond the f16 range must raise the context's fault flag, not pass silently."""
    L = 16
    x = torch.zeros(128, L, L)
    x[5, 3, 3] = 7.0e4
    st_engine.eng.set_option("conv_mode", 0)
    st_engine.conv(1, st_engine.to(x.numpy()))
    with pytest.raises(Exception, match="f16 range"):
        st_engine.eng.sync_check()
    # reporting clears the fault words: the remaining tests start clean
    st_engine.eng.sync_check()


@pytest.mark.parametrize("L", [17, 33, 96])
def test_block_odd_sizes(st, oracle_weights, L):
    x = torch.from_numpy(np.random.default_rng(L).standard_normal((1, 128, L, L)).astype(np.float32))
    u_ref = O.block_conv(oracle_weights, 2, x)
    u, stats = st.conv(2, st.to(x[0].numpy()))
    assert np.abs(u.cpu().numpy() - u_ref[0].numpy()).max() <= scale_tol(u_ref.numpy(), 1e-5)
    out_ref = O.block_finish(oracle_weights, 2, u_ref, x)[0].numpy()
    out = st.norm(2, u, stats, st.to(x[0].numpy())).cpu().numpy()
    assert np.abs(out - out_ref).max() <= scale_tol(out_ref, 1e-4)


def test_block_conv_maxout_backward_vs_reference_autograd(st_engine):
    """SURVEY 8f.4, second slice: dmp_block_conv5x5_maxout_bwd - input, weight and bias gradients
    of a residual block's convolution + maxout - against the reference's own autograd through Maxout2d of block 3
    (network.py:25-31; tests/golden/make_goldens.py, bwd_block3_L24).  1e-4 of each tensor's scale."""
    g = load_golden("bwd_block3_L24")
    st = st_engine
    dx, dw, db = st.conv_bwd(int(g["block"]), st.to(g["x"]), st.to(g["du"]))
    st.eng.sync_check()
    dx, dw, db = dx.cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()
    assert np.abs(dx - g["dx"]).max() <= 1e-4 * np.abs(g["dx"]).max()
    assert np.abs(db - g["db"]).max() <= 1e-4 * np.abs(g["db"]).max()
    got = dw.ravel()[g["dw.idx"]]
    scale = float(np.abs(g["dw.val"]).max())
    assert np.abs(got - g["dw.val"]).max() <= 1e-4 * scale
    assert abs(float(dw.astype(np.float64).sum()) - float(g["dw.sum"])) <= 1e-4 * scale * np.sqrt(dw.size)
    assert abs(float((dw.astype(np.float64) ** 2).sum()) - float(g["dw.sumsq"])) <= 1e-4 * float(g["dw.sumsq"])
    # the maxout routes each gradient to ONE channel of its quadruple: the bias gradients of a quadruple sum to the
    # gradient mass of its maxout channel
    assert np.abs(db.reshape(128, 4).sum(axis=1) - g["du"].reshape(128, -1).sum(axis=1)).max() <= 1e-3 * np.abs(g["du"]).sum(axis=(1, 2)).max()


def test_block_backward_vs_reference_autograd_through_the_whole_block(st_engine):
    """SURVEY 8f.4, second slice, both halves: dmp_block_norm_scse_residual_bwd (InstanceNorm +
    scSE + residual, network.py:32, 36-83, 99-101) chained into dmp_block_conv5x5_maxout_bwd against the reference's
    autograd through ResNet_Block 3 in evaluation mode (tests/golden/make_goldens.py, bwd_block3_full_L24): every
    parameter gradient of the block, the gradient at the interface (the maxout output) and the gradient of the
    block's input including the residual branch.  1e-4 of each tensor's scale, in both convolution modes (the
    backward is float32 throughout; the mode must not matter)."""
    g = load_golden("bwd_block3_full_L24")
    st = st_engine
    blk = int(g["block"])

    def close(got, want, tol=1e-4):
        want = np.asarray(want, dtype=np.float32)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-30), (np.abs(got - want).max(), np.abs(want).max())

    first = None
    for mode in (0, 1):
        st.eng.set_option("conv_mode", mode)
        try:
            du, dp = st.norm_bwd(blk, st.to(g["u"]), st.to(g["dout"]))
            dx, dw, db = st.conv_bwd(blk, st.to(g["x"]), du)
            st.eng.sync_check()
        finally:
            st.eng.set_option("conv_mode", 0)
        du, dp, dx, dw, db = (t.cpu().numpy() for t in (du, dp, dx, dw, db))
        close(du, g["du"])
        close(dp[0:128], g["dgamma"])
        close(dp[128:256], g["dbeta"])
        close(dp[256:256 + 1024].reshape(8, 128), g["dfc0"])
        close(dp[1280:1280 + 1024].reshape(128, 8), g["dfc2"])
        close(dp[2304:2432], g["dsse_w"])
        close(dp[2432:2433], g["dsse_b"])
        close(dx + g["dout"], g["dx"])                      # the residual branch is the identity
        close(db, g["db"])
        scale = float(np.abs(g["dw.val"]).max())
        assert np.abs(dw.ravel()[g["dw.idx"]] - g["dw.val"]).max() <= 1e-4 * scale
        assert abs(float((dw.astype(np.float64) ** 2).sum()) - float(g["dw.sumsq"])) <= 1e-4 * float(g["dw.sumsq"])
        if first is None:
            first = (du, dp, dx)
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first, (du, dp, dx)))
    # InstanceNorm's backward removes the mean and the uh-component of the gradient per channel
    uh = g["u"] - g["u"].mean(axis=(1, 2), keepdims=True)
    assert np.abs(du.sum(axis=(1, 2))).max() <= 1e-3 * np.abs(du).sum(axis=(1, 2)).max()
    assert np.abs((du * uh).sum(axis=(1, 2))).max() <= 1e-3 * (np.abs(du) * np.abs(uh)).sum(axis=(1, 2)).max()


def test_head_gram_and_trunk_pass(st, pf, ocap, oracle_weights):
    x = ocap["p0.block16"]
    y = O.head(oracle_weights, x)
    dm, conf, M = O.head_to_gram(y)
    gconf, gM = st.head_gram(st.to(x[0].numpy()))
    assert np.abs(gconf.cpu().numpy() - conf[0].numpy()).max() < 1e-4
    assert np.abs(gM.cpu().numpy() - M[0].numpy()).max() <= scale_tol(M.numpy(), 1e-5)
    assert np.abs(gM.cpu().numpy() - gM.cpu().numpy().T).max() == 0.0        # exactly symmetric
    # whole pass on internal buffers
    L = pf["alnmat"].shape[1]
    z0 = st.stem_static(st.to(ocap["mat1d"].numpy()), st.to(ocap["inv_cov"].numpy()),
                        st.to(ocap["contacts"].numpy()))
    tconf, tM = st.trunk_pass(z0, st.to((torch.zeros(L, L) - 1).numpy()))
    assert np.abs(tconf.cpu().numpy() - ocap["p0.conf"].numpy()).max() < 1e-4
    assert np.abs(tM.cpu().numpy() - ocap["p0.M"].numpy()).max() <= scale_tol(ocap["p0.M"].numpy(), 1e-4)


# ---------------------------------------------------------------------------- coordinates
def test_eigh_top8(st, ocap):
    M = ocap["p0.M"]
    ref = O.mds_top8(M.unsqueeze(0), "canonical")[0].numpy()
    got = st.eigh_top8(st.to(M.numpy())).cpu().numpy()
    # float64 solve on the device vs float32 LAPACK in the oracle: compare with the float64 truth too
    lam, vec = torch.linalg.eigh(M.double(), UPLO="U")
    vec = O.canonical_signs(vec)
    truth = (vec * lam.clamp(min=1e-8).sqrt())[:, -8:].numpy()
    assert np.abs(got - truth).max() <= 2e-5 * max(1.0, np.abs(truth).max())
    assert np.abs(got - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max())


def test_eigh_top8_structured(st):
    """A genuine 3-D distance geometry Gram matrix: 3 large eigenvalues, the rest ~0 / negative."""
    rng = np.random.default_rng(11)
    for L in (8, 31, 96):
        P = rng.standard_normal((L, 3)) * 10
        D = np.linalg.norm(P[:, None] - P[None], axis=2)
        M = (0.5 * (D[0:1, :] ** 2 + D[:, 0:1] ** 2 - D ** 2)).astype(np.float32)
        got = st.eigh_top8(st.to(M)).cpu().numpy().astype(np.float64)
        lam, vec = np.linalg.eigh(M.astype(np.float64))
        lam8 = np.maximum(lam[-8:], 1e-8)
        # the three big columns reproduce the Gram matrix; every column has the right norm
        assert np.abs(np.linalg.norm(got, axis=0) - np.sqrt(lam8)).max() < 1e-3
        G3 = got[:, -3:] @ got[:, -3:].T
        assert np.abs(G3 - M).max() < 1e-2
        assert (got[np.abs(got).argmax(axis=0), np.arange(8)] > 0).all()      # sign rule


def test_eigh_single_and_multi_workgroup_tridiagonalisation_agree(st_engine, ocap):
    """Option tridiag_single: the one-workgroup Householder kernel and the per-step multi-workgroup
    one are the same algorithm; odd sizes exercise the chain padding (64 steps per graph replay)."""
    rng = np.random.default_rng(5)
    mats = [ocap["p0.M"].numpy()]
    for L in (9, 64, 65, 127):
        B = rng.standard_normal((L, L)).astype(np.float32)
        mats.append(((B + B.T) * 3).astype(np.float32))
    for M in mats:
        a = st_engine.eigh_top8(st_engine.to(M)).cpu().numpy()
        st_engine.eng.set_option("tridiag_single", 1)
        try:
            b = st_engine.eigh_top8(st_engine.to(M)).cpu().numpy()
        finally:
            st_engine.eng.set_option("tridiag_single", 0)
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max())


def test_coords_from_mds(st, ocap, oracle_weights):
    ref = O.coords_from_mds(oracle_weights, ocap["mat1d"], ocap["p0.mds"].unsqueeze(0))[0].numpy()
    got = st.coords_from_mds(st.to(ocap["mat1d"].numpy()), st.to(ocap["p0.mds"].numpy())).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4


def test_pair_distances(st):
    ca = torch.from_numpy(np.random.default_rng(2).standard_normal((50, 3)).astype(np.float32) * 7)
    ref = O.pair_distances(ca).numpy()
    got = st.pair_distances(st.to(ca.numpy()), 1).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5
    assert np.allclose(np.diag(got), 1e-4)
    got0 = st.pair_distances(st.to(ca.numpy()), 0).cpu().numpy()
    assert np.all(np.diag(got0) == 0.0)


def test_refine_known_answer(st):
    k = load_golden("kat_refine_backbone")
    ca = st.to(k["ca_in"])
    for steps, tol in ((1, 1e-5), (10, 1e-5), (100, 1e-4), (1000, 1e-3)):
        got = st.refine(ca, steps).cpu().numpy()
        assert np.abs(got - k[f"refined_{steps}"]).max() < tol, steps
    got = st.refine(st.to(k["ca_noisy"]), 100).cpu().numpy()
    assert np.abs(got - k["refined_noisy_100"]).max() < 1e-3


def test_refine_cluster_and_single_workgroup_agree(synth_sd):
    """Option refine_single: the 16-workgroup minimiser (granule hand-off between the workgroups every
    step) and the single-workgroup one are the same iteration with different partial-sum slices;
    well-spread chains, so the iteration is not chaotic.  Also pins the cluster kernel at lengths that
    leave the last workgroups empty or ragged, and its determinism."""
    from abi import Stages
    stg = Stages(synth_sd, max_L=1024, max_N=8)
    rng = np.random.default_rng(3)
    try:
        for L, steps in ((33, 50), (82, 100), (300, 100), (1000, 30)):
            step = rng.standard_normal((L, 3))
            step *= 3.8 / np.linalg.norm(step, axis=1, keepdims=True)
            ca = np.cumsum(step, axis=0).astype(np.float32)
            a = stg.refine(stg.to(ca), steps).cpu().numpy()
            a2 = stg.refine(stg.to(ca), steps).cpu().numpy()
            stg.eng.set_option("refine_single", 1)
            try:
                b = stg.refine(stg.to(ca), steps).cpu().numpy()
            finally:
                stg.eng.set_option("refine_single", 0)
            assert np.array_equal(a, a2), L
            assert np.isfinite(a).all() and np.abs(a - ca).max() > 0
            assert np.abs(a - b).max() < 2e-4, (L, np.abs(a - b).max())
        stg.eng.sync_check()
    finally:
        stg.eng.close()


def test_backbone_known_answer(st):
    k = load_golden("kat_refine_backbone")
    L = k["ca_in"].shape[0]
    logit = np.linspace(-3, 3, L).astype(np.float32)
    coords, conf = st.backbone(st.to(k["ca_in"]), st.to(logit))
    assert np.abs(coords.cpu().numpy().reshape(-1, 3) - k["backbone"]).max() < 1e-4
    assert np.abs(conf.cpu().numpy() - 1 / (1 + np.exp(-logit))).max() < 1e-6


# ---------------------------------------------------------------------------- end to end
E2E = ["pf10963_n0_m0", "pf10963_n3_m0", "synth_L40_N64_n2_m0", "synth_L24_N3050_n1_m0",
       "alphabet_L16_N12_n0_m0", "template_L96_N50_n1_m0"]


@pytest.mark.parametrize("name", E2E)
def test_end_to_end_m0(st, name, tmp_path, weights_file):
    """Full dmp_predict vs the reference's captured outputs: CA-RMSD <= 1e-3 A, |dconf| < 1e-4."""
    g = load_golden(name)
    tpl = g["template_ca"] if "template_ca" in g else None
    coords, confs = st.eng.predict(g["alnmat"], tpl, int(g["iterations"]), int(g["minsteps"]))
    coords, confs = coords.cpu().numpy(), confs.cpu().numpy()
    P = int(g["iterations"]) + 1
    means = st.eng.fetch("conf_means", P).cpu().numpy()
    assert np.abs(means - g["conf_mean_pass"]).max() < 1e-3
    assert ca_rmsd(coords[:, 1], g["coords"][:, 1]) <= 1e-3
    assert np.abs(confs - g["confs"]).max() < 1e-4
    assert np.abs(coords - g["coords"]).max() < 2e-2


def test_end_to_end_single_sequence_with_refinement(st):
    g = load_golden("synth_L30_N1_n1_m3")
    coords, confs = st.eng.predict(g["alnmat"], None, 1, 3)
    tol = max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
    assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) <= tol
    assert np.abs(confs.cpu().numpy() - g["confs"]).max() < 1e-4


def test_end_to_end_with_refinement_within_noise_floor(st):
    """m > 0 on synthetic weights is chaotic in the reference itself (see tests/golden/REPORT.txt):
    the bound is max(1e-3, 3 x the reference's own 8-vs-1-thread deviation)."""
    g = load_golden("pf10963_n2_m5")
    coords, confs = st.eng.predict(g["alnmat"], None, 2, 5)
    tol = max(1e-3, 3.0 * float(g["noise_ca_rmsd"]))
    assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) <= tol
    assert np.abs(confs.cpu().numpy() - g["confs"]).max() < max(1e-4, 3.0 * float(g["noise_conf"]))


def test_throughput_pipeline_matches_single_engine(st_engine, synth_sd):
    """The unit-granular scheduler (3 engines, shared lane, more targets than engines, mixed sizes)
    returns bit-identical results to one engine running the same targets one after the other, and
    the goldens' tolerances hold."""
    from dmpfold2_amd.predict import Pipeline
    names = ["pf10963_n0_m0", "synth_L40_N64_n2_m0", "synth_L30_N1_n1_m3", "pf10963_n2_m5",
             "synth_L40_N64_n2_m0", "pf10963_n0_m0", "synth_L30_N1_n1_m3"]
    iters = {"pf10963_n0_m0": (0, 0), "synth_L40_N64_n2_m0": (2, 0), "synth_L30_N1_n1_m3": (1, 3),
             "pf10963_n2_m5": (2, 5)}
    gs = [load_golden(n) for n in names]
    dev = torch.device("cuda:0")
    pipe = Pipeline(dev, 128, 3000, synth_sd, streams=3)
    tickets = [pipe.submit(torch.from_numpy(np.ascontiguousarray(g["alnmat"])).to(dev), *iters[n])
               for n, g in zip(names, gs)]
    pipe.drain()
    pipe.sync_check()
    # like with like: the scheduler's engines tridiagonalise with one launch per Householder step (LIKE_PIPELINE)
    st_engine.eng.set_option("tridiag_cluster", 0)
    try:
        for n, g, t in zip(names, gs, tickets):
            coords, confs = pipe.result(t)
            ref_c, ref_f = st_engine.eng.predict(g["alnmat"], None, *iters[n])
            st_engine.eng.sync_check()
            assert torch.equal(coords, ref_c) and torch.equal(confs, ref_f), n
            if iters[n][1] == 0:
                assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) <= 1e-3
    finally:
        st_engine.eng.set_option("tridiag_cluster", 1)
        pipe.close()


@pytest.mark.parametrize("persistent", [1, 0])
@pytest.mark.parametrize("group,riders,precision", [(1, 4, 0), (2, 4, 0), (4, 4, 0), (4, 0, 0), (2, 6, 0), (3, 2, 0), (4, 7, 0),
                                                    (4, 4, 1), (2, 6, 1), (4, 4, 2), (2, 6, 2)])
def test_grouped_vertical_gru_is_bitwise_the_ungrouped_one(synth_sd, monkeypatch, group, riders, persistent, precision):
    """dmp_predict_group_vgru: predictions that start together run their vertical GRUs as ONE chain (the group
    leader's units) - ragged in L and N, a one-row alignment among them, more targets than engines, so groups of every
    size up to `group` form.  riders: a group's chain also serves up to that many of the NEXT targets in the queue
    (dmp_predict_group_riders; members + riders <= 8), handed over when those targets start
    (dmp_predict_set_vgru_result).  Every result equals the single engine's, bit for bit - in the persistent
    weight-stationary form of the chain (one launch) and in the launch-per-row form; precision = 1 (round 5): the same
    with the float32 vertical GRU (vgru_f32.hip) and the float32 convolutions."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, Pipeline, encode_aln
    monkeypatch.setenv("DMP_VGRU_GROUP", str(group))
    monkeypatch.setenv("DMP_VGRU_RIDERS", str(riders))
    shapes = [(82, 200), (33, 64), (128, 300), (40, 1), (64, 257), (96, 31), (120, 129), (50, 64), (128, 17)]
    msas = [encode_aln(synth.synth_msa(L, N, 40 + i)) for i, (L, N) in enumerate(shapes)]
    dev = torch.device("cuda:0")
    single = Engine(dev, 128, 512)
    single.set_weights(synth_sd)
    single.set_option("vgru_persistent", persistent)
    single.set_option("precision", precision)
    single.set_option("tridiag_cluster", 0)                 # as the scheduler's engines (LIKE_PIPELINE)
    pipe = Pipeline(dev, 128, 512, synth_sd, streams=4)
    for e in pipe.engines:
        e.set_option("vgru_persistent", persistent)
        e.set_option("precision", precision)
    assert pipe.engines[0].get_option("vgru_f32") == precision
    assert pipe.engines[0].get_option("vgru_persistent") == persistent        # a 256-CU device has the persistent form
    msas = msas + msas[:5]                       # more targets than engines
    tickets = pipe.submit_many([torch.from_numpy(m).to(dev) for m in msas], 1, 3)
    pipe.drain()
    pipe.sync_check()
    stats = pipe.stats()                        # the C scheduler's counters (dmp_pipeline_stats)
    assert stats["max_group"] == (group if group > 1 else 0), stats
    if riders and group > 1:
        assert stats["rider_chains"] > 0 and stats["max_riders"] == min(riders, 8 - group) and stats["riders_left"] == 0, stats
    else:
        assert stats["rider_chains"] == 0, stats
    try:
        for m, t in zip(msas, tickets):
            coords, confs = pipe.result(t)
            ref_c, ref_f = single.predict(m, None, 1, 3)
            single.sync_check()
            assert torch.equal(coords, ref_c) and torch.equal(confs, ref_f), m.shape
    finally:
        pipe.close()
        single.close()


def test_persistent_vertical_gru_vs_oracle_and_launch_chain(st_engine, synth_sd, oracle_weights):
    """Round 4: the chain as ONE persistent weight-stationary launch (vgru_persist_kernel: columns over the XCDs, hidden
    units over the CUs of an XCD, XCD-local row barriers) against the oracle's nn.GRU (1e-5, the tolerance of the
    launch-per-row kernel) on a ragged group - odd lengths, a one-row alignment, more rows than columns - and against
    the launch-per-row form (different K summation order: float32 rounding); a member's bits do not depend on the
    group it ran in; no row barrier timed out."""
    import ctypes as C
    from dmpfold2_amd import _lib, synth
    from dmpfold2_amd.predict import encode_aln
    shapes = [(82, 100), (33, 7), (128, 64), (40, 1), (50, 257), (9, 300)]
    msas = [encode_aln(synth.synth_msa(L, N, 70 + i)) for i, (L, N) in enumerate(shapes)]
    st = st_engine
    eng = st.eng
    assert eng.get_option("vgru_persistent") == 1

    def chain(ms, persistent):
        eng.set_option("vgru_persistent", persistent)
        k = len(ms)
        d = [st.to(m, torch.uint8) for m in ms]
        outs = [st.f32(m.shape[1], 512) for m in ms]
        ctxs = (C.c_void_p * k)(*[eng.ctx] * k)
        mp = (C.c_void_p * k)(*[x.data_ptr() for x in d])
        op = (C.c_void_p * k)(*[x.data_ptr() for x in outs])
        Ns = (C.c_int * k)(*[m.shape[0] for m in ms])
        Ls = (C.c_int * k)(*[m.shape[1] for m in ms])
        _lib.check(st.lib.dmp_gru_vertical_group(ctxs, k, mp, Ns, Ls, op, eng.stream()))
        torch.cuda.synchronize()
        return outs
    try:
        grouped = chain(msas, 1)
        assert eng.sync_faults() == 0
        rows = chain(msas, 0)
        for m, a, b in zip(msas, grouped, rows):
            x = oracle_weights["embed.weight"][torch.from_numpy(m.astype(np.int64))]
            ref = O._gru(oracle_weights, "vgru", x, 22, 512, 2, False, False)[-1]
            assert float((a.cpu() - ref).abs().max()) < 1e-5, m.shape
            assert float((a - b).abs().max()) < 5e-6, m.shape
        for i, m in enumerate(msas):
            alone = chain([m], 1)[0]
            assert torch.equal(alone, grouped[i]), i
            pair = chain([msas[(i + 1) % len(msas)], m], 1)[1]
            assert torch.equal(pair, grouped[i]), i
        assert eng.sync_faults() == 0
    finally:
        eng.set_option("vgru_persistent", 1)


def test_float32_vertical_gru_is_the_references_arithmetic(st_engine, synth_sd, oracle_weights):
    """Round 5: option "precision" = 1 runs the vertical GRU on float32 MFMAs with library gate
    functions (vgru_persist_f32_kernel) - nn.GRU's arithmetic (network.py:189, 223-224) - together with the exact-f32
    convolution; round 6: "precision" = 2 runs it with full-width operands on the bf16 matrix cores (vgru_persist_x3_kernel).  Against the oracle's nn.GRU on a ragged group at a TIGHTER bound than the split-f16 form's 1e-5; the
    launch-per-row fallback is the same kernel without the barrier: the same bits; a member's bits do not depend on its
    group; the option reads back; "vgru_f32" overrides it per context."""
    import ctypes as C
    from dmpfold2_amd import _lib, synth
    from dmpfold2_amd.predict import encode_aln
    shapes = [(82, 100), (33, 7), (128, 64), (40, 1), (50, 257), (9, 300)]
    msas = [encode_aln(synth.synth_msa(L, N, 70 + i)) for i, (L, N) in enumerate(shapes)]
    st = st_engine
    eng = st.eng

    def chain(ms, persistent):
        eng.set_option("vgru_persistent", persistent)
        k = len(ms)
        d = [st.to(m, torch.uint8) for m in ms]
        outs = [st.f32(m.shape[1], 512) for m in ms]
        ctxs = (C.c_void_p * k)(*[eng.ctx] * k)
        mp = (C.c_void_p * k)(*[x.data_ptr() for x in d])
        op = (C.c_void_p * k)(*[x.data_ptr() for x in outs])
        Ns = (C.c_int * k)(*[m.shape[0] for m in ms])
        Ls = (C.c_int * k)(*[m.shape[1] for m in ms])
        _lib.check(st.lib.dmp_gru_vertical_group(ctxs, k, mp, Ns, Ls, op, eng.stream()))
        torch.cuda.synchronize()
        return outs
    g64 = torch.nn.GRU(22, 512, num_layers=2).double().eval()               # the same recurrence in float64
    g64.load_state_dict({k[5:]: v.double() for k, v in oracle_weights.items() if k.startswith("vgru.")})
    try:
        assert eng.get_option("precision") == 0 and eng.get_option("vgru_f32") == 0
        split = chain(msas, 1)
        results = {}
        # precision 1: float32 MFMAs (vgru_f32.hip); precision 2 (round 6): every operand as three exact bf16 pieces, six
        # piece products (vgru_x3.hip) - both with the library gate functions, both held to the same bounds
        for precision in (1, 2):
            eng.set_option("precision", precision)
            assert eng.get_option("precision") == precision and eng.get_option("vgru_f32") == precision
            assert eng.get_option("conv_mode") == precision
            grouped = chain(msas, 1)
            assert eng.sync_faults() == 0
            rows = chain(msas, 0)
            worst = 0.0
            for m, a, b, h in zip(msas, grouped, rows, split):
                x = oracle_weights["embed.weight"][torch.from_numpy(m.astype(np.int64))]
                ref = O._gru(oracle_weights, "vgru", x, 22, 512, 2, False, False)[-1]
                with torch.no_grad():
                    ref64 = g64(x.double())[0][-1]
                worst = max(worst, float((a.cpu() - ref).abs().max()))
                assert float((a.cpu() - ref).abs().max()) < 3e-6, m.shape          # float32 against float32
                # ... and as close to the float64 recurrence as the reference's own float32 run is (x 2)
                assert float((a.cpu().double() - ref64).abs().max()) <= 2.0 * float((ref.double() - ref64).abs().max()) + 1e-7
                assert torch.equal(a, b), m.shape                                  # per-row launches: the same kernel
                assert float((a - h).abs().max()) < 1e-5 and not torch.equal(a, h)  # the split-f16 form is another arithmetic
            print("vertical GRU, precision", precision, ": max |dev| from the oracle's nn.GRU", worst)
            for i, m in enumerate(msas):
                alone = chain([m], 1)[0]
                assert torch.equal(alone, grouped[i]), i
                pair = chain([msas[(i + 1) % len(msas)], m], 1)[1]
                assert torch.equal(pair, grouped[i]), i
            assert eng.sync_faults() == 0
            results[precision] = grouped
        assert not torch.equal(results[1][0], results[2][0])                       # two arithmetics ...
        assert float((results[1][0] - results[2][0]).abs().max()) < 2e-6           # ... of the same width
        # "vgru_f32" overrides: the exact convolutions with the split-f16 GRU ...
        eng.set_option("vgru_f32", 0)
        assert eng.get_option("precision") == -1
        assert torch.equal(chain(msas[:2], 1)[0], split[0])
        # ... or with the float32-MFMA GRU (the round-6 headline before vgru_x3.hip)
        eng.set_option("vgru_f32", 1)
        assert eng.get_option("precision") == -1 and eng.get_option("conv_mode") == 2
        assert torch.equal(chain(msas[:2], 1)[0], results[1][0])
        # ... and back to following the convolution's mode
        eng.set_option("precision", 0)
        assert eng.get_option("precision") == 0 and eng.get_option("conv_mode") == 0
        assert torch.equal(chain(msas[:2], 1)[0], split[0])
        eng.set_option("precision", 2)
        assert eng.get_option("precision") == 2 and eng.get_option("conv_mode") == 2 and eng.get_option("vgru_f32") == 2
        assert torch.equal(chain(msas[:2], 1)[0], results[2][0])
        # conv_mode 2 alone (the range fallback of the fast mode) keeps the split-f16 GRU and reads back as a mixed setting
        eng.set_option("precision", 0)
        eng.set_option("conv_mode", 2)
        assert eng.get_option("precision") == -1 and eng.get_option("vgru_f32") == 0
        eng.set_option("precision", 0)
        assert eng.get_option("precision") == 0 and eng.get_option("conv_mode") == 0 and eng.get_option("vgru_f32") == 0
    finally:
        eng.set_option("precision", 0)
        eng.set_option("vgru_persistent", 1)


def test_gru_vertical_group_stage_call(st, synth_sd):
    """dmp_gru_vertical_group (the stage-level form): three alignments of different shape through one chain,
    each result bit-identical to dmp_gru_vertical alone and within 1e-5 of the oracle's GRU."""
    import ctypes as C
    from dmpfold2_amd import _lib, synth
    from dmpfold2_amd.predict import Engine, encode_aln
    shapes = [(82, 100), (33, 7), (128, 64)]
    msas = [encode_aln(synth.synth_msa(L, N, 70 + i)) for i, (L, N) in enumerate(shapes)]
    alone = [st.gru_vertical(m).clone() for m in msas]
    others = [Engine("cuda:0", 128, 512) for _ in range(2)]
    try:
        for e in others:
            e.set_weights({k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()})
        engs = [st.eng] + others
        d = [st.to(m, torch.uint8) for m in msas]
        outs = [st.f32(L, 512) for L, _ in shapes]
        k = len(engs)
        ctxs = (C.c_void_p * k)(*[e.ctx for e in engs])
        mp = (C.c_void_p * k)(*[x.data_ptr() for x in d])
        op = (C.c_void_p * k)(*[x.data_ptr() for x in outs])
        Ns = (C.c_int * k)(*[N for _, N in shapes])
        Ls = (C.c_int * k)(*[L for L, _ in shapes])
        _lib.check(st.lib.dmp_gru_vertical_group(ctxs, k, mp, Ns, Ls, op, st.eng.stream()))
        torch.cuda.synchronize()
        for a, o in zip(alone, outs):
            assert torch.equal(a, o)
    finally:
        for e in others:
            e.close()


def test_shared_weights_are_the_owners_and_outlive_it(synth_sd):
    """dmp_weights_share: the engines of a scheduler use ONE packed copy of the weights.  A sharing engine predicts
    the owner's bits, keeps working after the owner is destroyed (the buffers are reference counted), and can load
    weights of its own again afterwards."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, encode_aln
    w = {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}
    msa = encode_aln(synth.synth_msa(48, 40, 3))
    a, b = Engine("cuda:0", 64, 64), Engine("cuda:0", 64, 64)
    try:
        a.set_weights(w)
        bytes_owner = a.device_bytes
        b.share_weights(a)
        ca, fa = a.predict(msa, None, 1, 2)
        cb, fb = b.predict(msa, None, 1, 2)
        a.sync_check(); b.sync_check()
        assert torch.equal(ca, cb) and torch.equal(fa, fb)
        assert b.device_bytes < bytes_owner                  # no second copy of the packed weights
        a.close()
        cb2, fb2 = b.predict(msa, None, 1, 2)
        b.sync_check()
        assert torch.equal(cb2, cb) and torch.equal(fb2, fb)
        w2 = dict(w)
        w2["coord_fc.weight"] = w["coord_fc.weight"] * 0.5
        b.set_weights(w2)
        cb3, _ = b.predict(msa, None, 1, 2)
        b.sync_check()
        assert not torch.equal(cb3, cb) and torch.isfinite(cb3).all()
    finally:
        a.close(); b.close()


def test_batch_front_end_matches_cli(weights_file, tmp_path):
    """dmpfold2_amd.batch (targets file -> one PDB per target through the scheduler, template in
    the second column) writes exactly the text the single-target CLI prints."""
    from dmpfold2_amd import run_dmpfold, synth
    from dmpfold2_amd.batch import read_target_list, run_batch
    paths = []
    g = load_golden("pf10963_n0_m0")
    p = tmp_path / "pf10963.aln"
    p.write_text("\n".join(golden_rows(g)) + "\n")
    paths.append(str(p))
    for k, (L, N) in enumerate([(24, 40), (57, 3), (33, 1), (64, 200), (40, 64)]):
        q = tmp_path / f"t{k}.aln"
        synth.write_aln(str(q), synth.synth_msa(L, N, seed=50 + k))
        paths.append(str(q))

    def cli(aln, tpl=None):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            run_dmpfold(["-i", aln, "-d", "cuda:0", "-n", "1", "-m", "2", "-w", weights_file] +
                        (["-t", tpl] if tpl else []))
        return buf.getvalue()

    ref = {a: cli(a) for a in paths}
    tpl = tmp_path / "tpl.pdb"                       # a previous model as the template of the same target
    tpl.write_text(ref[paths[0]])
    lst = tmp_path / "targets.txt"
    lst.write_text("# alignment [template]\n" + "\n".join(paths[1:]) + f"\n{paths[0]} {tpl}\n")
    targets = read_target_list(str(lst))
    assert targets[-1] == (paths[0], str(tpl)) and len(targets) == 6
    n, secs, outs = run_batch(targets, str(tmp_path / "out"), 1, 2, weights_file=weights_file, streams=3,
                              device="cuda:0")
    assert n == 6 and len(outs) == 6
    for a in paths[1:]:
        name = a.rsplit("/", 1)[1].replace(".aln", ".pdb")
        assert (tmp_path / "out" / name).read_text() == ref[a], a
    assert (tmp_path / "out" / "pf10963.pdb").read_text() == cli(paths[0], str(tpl))


def test_backbone_bitwise_stable_beside_convolutions(synth_sd):
    """Regression test for the packed-f32 hazard: dmp_ca_to_backbone launched 3000 times
    while another context runs f16 split-product convolutions; every output bit-identical to the
    output computed alone.  With the vectorised cross products (v_pk_mul_f32 ... op_sel:[0,1]) about
    a quarter of the launches had the C, O, CB atoms of lanes 48..63 wrong (tools/bb_hazard.hip)."""
    import ctypes as C
    import threading
    from dmpfold2_amd import _lib
    from dmpfold2_amd.predict import Engine
    dev = torch.device("cuda:0")
    L = 300
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ea, eb = Engine(dev, L, 8, stream=sa), Engine(dev, L, 8, stream=sb)
    eb.set_weights(synth_sd)
    g = torch.Generator().manual_seed(11)
    ca = torch.cumsum(torch.randn(L, 3, generator=g) * 2.2, 0).to(dev)
    lg = torch.randn(L, generator=g).to(dev)
    slots = 100
    out = torch.empty(slots, L, 15, device=dev)
    cf = torch.empty(slots, L, device=dev)

    def backbone(k):
        _lib.check(ea.lib.dmp_ca_to_backbone(ea.ctx, ca.data_ptr(), lg.data_ptr(), L, out[k].data_ptr(),
                                             cf[k].data_ptr(), ea.stream()))
    backbone(0)
    torch.cuda.synchronize()
    ref_c, ref_f = out[0].clone(), cf[0].clone()

    z0 = torch.randn(384, L, L, device=dev)
    dm = torch.full((L, L), -1.0, device=dev)
    cfb, Mb = torch.empty(L, device=dev), torch.empty(L, L, device=dev)
    sb.wait_stream(torch.cuda.current_stream(dev))

    def convolutions():
        with torch.cuda.device(dev):
            for _ in range(38):                          # 38 x 16 split-product convolutions on the other stream
                _lib.check(eb.lib.dmp_trunk_pass(eb.ctx, z0.data_ptr(), dm.data_ptr(), L, cfb.data_ptr(), Mb.data_ptr(),
                                                 eb.stream()))
            sb.synchronize()
    t = threading.Thread(target=convolutions)
    t.start()
    bad = 0
    for rnd in range(30):
        for k in range(slots):
            backbone(k)
        sa.synchronize()
        bad += int((out != ref_c).any(dim=(1, 2)).sum()) + int((cf != ref_f).any(dim=1).sum())
    t.join()
    ea.close()
    eb.close()
    assert bad == 0


def test_scheduler_results_bitwise_stable_under_corunning_kernels(synth_sd):
    """Canary for cross-kernel interference: 72 short predictions (L=300, 1 iteration, 100 minimiser
    steps) through 3 engines, every result bit-identical to the single-engine one.  1-4 % of them
    used to have the C/O/CB atoms of 16 residues wrong when f16 convolutions of another target shared
    the CUs (tools/corrupt_repro.py; cause and fix in DESIGN section 6)."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, Pipeline, encode_aln
    dev = torch.device("cuda:0")
    L, N = 300, 200
    msas = [encode_aln(synth.synth_msa(L, N, seed=40 + i)) for i in range(4)]
    eng = Engine(dev, L, N)
    eng.set_weights(synth_sd)
    eng.set_option("tridiag_cluster", 0)                    # as the scheduler's engines (LIKE_PIPELINE)
    refs = []
    for m in msas:
        c, f = eng.predict(m, None, 1, 100)
        eng.sync_check()
        refs.append((c.clone(), f.clone()))
    eng.close()
    bad = 0
    for trial in range(6):
        pipe = Pipeline(dev, L, N, synth_sd, streams=3)
        order = [i % 4 for i in range(12)]
        res = pipe.run([torch.from_numpy(msas[i]).to(dev) for i in order], 1, 100)
        pipe.sync_check()
        bad += sum(1 for k, i in enumerate(order)
                   if not (torch.equal(res[k][0], refs[i][0]) and torch.equal(res[k][1], refs[i][1])))
        pipe.close()
    assert bad == 0


def test_unit_api_contract(st_engine):
    """dmp_predict_next_unit / issue_unit: 18 units per pass, blocks are the conv units, pass == units."""
    eng = st_engine.eng
    lib = eng.lib
    g = load_golden("synth_L40_N64_n2_m0")
    ref_c, ref_f = eng.predict(g["alnmat"], None, 2, 0)
    eng.sync_check()
    d_msa = torch.from_numpy(np.ascontiguousarray(g["alnmat"])).cuda()
    n, L = d_msa.shape
    coords = torch.empty((L, 5, 3), device="cuda")
    confs = torch.empty((L,), device="cuda")
    s = eng.stream()
    assert lib.dmp_predict_begin_units(eng.ctx, d_msa.data_ptr(), n, L, None, 0, 2, 0) == 0
    assert lib.dmp_predict_end(eng.ctx, coords.data_ptr(), confs.data_ptr(), s) < 0      # units still outstanding
    kinds = []
    while True:
        k = lib.dmp_predict_next_unit(eng.ctx)
        if k == 0:
            break
        kinds.append(k)
        assert lib.dmp_predict_issue_unit(eng.ctx, s) == 0
    # front end of a 64 x 40 alignment: weights + covariance, 7 inverse block steps in 2 chunks,
    # 65 vertical-GRU launches in 1 chunk, sequence GRU + static stem
    assert kinds == [1] * 5 + ([1] + [2] * 16 + [1]) * 3
    assert lib.dmp_predict_issue_unit(eng.ctx, s) < 0          # nothing left to issue
    assert lib.dmp_predict_end(eng.ctx, coords.data_ptr(), confs.data_ptr(), s) == 0
    eng.sync_check()
    assert lib.dmp_ctx_pending(eng.ctx) == 0
    assert torch.equal(coords, ref_c) and torch.equal(confs, ref_f)


def test_python_api_and_cli(weights_file, tmp_path):
    from dmpfold2_amd import aln_to_coords, run_dmpfold
    g = load_golden("pf10963_n0_m0")
    aln = tmp_path / "pf.aln"
    aln.write_text("\n".join(golden_rows(g)) + "\n")
    coords, confs, alnmat = aln_to_coords(str(aln), device="cuda:0", iterations=0, minsteps=0,
                                          weights_file=weights_file, return_alnmat=True)
    assert coords.shape == (82, 5, 3) and confs.shape == (82,) and coords.is_cuda
    assert alnmat.dtype == np.uint8 and np.array_equal(alnmat, g["alnmat"])
    assert ca_rmsd(coords.cpu().numpy()[:, 1], g["coords"][:, 1]) <= 1e-3
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        run_dmpfold(["-i", str(aln), "-d", "cuda:0", "-n", "0", "-m", "0", "-w", weights_file])
    lines = buf.getvalue().splitlines()
    ref_text = O.pdb_text(torch.from_numpy(g["coords"]), torch.from_numpy(g["confs"]), g["alnmat"])
    ref_lines = ref_text.splitlines()
    assert len(lines) == len(ref_lines) and lines[-1] == "END" and lines[0].startswith("REMARK  CONF:  ")
    # same records; coordinates agree to the printed 3 decimals up to the 1e-3 tolerance
    for a, b in zip(lines[1:-1], ref_lines[1:-1]):
        assert a[:30] == b[:30]
        for lo in (30, 38, 46):
            assert abs(float(a[lo:lo + 8]) - float(b[lo:lo + 8])) <= 0.011


def test_errors(st, weights_file, tmp_path):
    from dmpfold2_amd import aln_to_coords
    ragged = tmp_path / "ragged.aln"
    ragged.write_text("ACDEFGHIKL\nACDEFGHIK\n")
    with pytest.raises(ValueError):
        aln_to_coords(str(ragged), device="cuda:0", weights_file=weights_file)
    short = tmp_path / "short.aln"
    short.write_text("ACDEF\nACDEF\n")
    with pytest.raises(RuntimeError):
        aln_to_coords(str(short), device="cuda:0", weights_file=weights_file)
    with pytest.raises(RuntimeError):
        aln_to_coords(str(short), device="cpu", weights_file=weights_file)
    with pytest.raises(FileNotFoundError):
        aln_to_coords(str(tmp_path / "missing.aln"), device="cuda:0", weights_file=weights_file)


def test_cluster_kernels_agree_with_and_without_the_xcd_local_handoff(synth_sd):
    """The cluster kernels (sequence GRU, minimiser, tridiagonalisation) publish their hand-off granules with plain
    stores when a run-time check finds the cluster on one XCD, and with agent-scope stores otherwise (option
    cluster_local = 0 forces that path, which no longer runs by default on a healthy box).  The two protocols carry
    the same values: a whole prediction with the minimiser is the same bits either way, and no hand-off timed out."""
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import Engine, encode_aln
    eng = Engine("cuda:0", 128, 300)
    eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()})
    try:
        outs = []
        for local in (1, 0, 1):
            eng.set_option("cluster_local", local)
            assert eng.get_option("cluster_local") == local
            res = []
            for L, N, seed in ((82, 120, 3), (128, 300, 4), (33, 17, 5)):
                c, f = eng.predict(encode_aln(synth.synth_msa(L, N, seed)), None, 2, 20)
                eng.sync_check()                      # raises DeviceFault on a hand-off time-out
                res.append((c.clone(), f.clone()))
            outs.append(res)
        for a, b, c in zip(*outs):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    finally:
        eng.close()


def test_c_abi_pipeline_submit_poll_status_release(synth_sd):
    """The throughput scheduler through the C ABI alone (dmp_pipeline_*, what INTEGRATION.md section 3 binds): weights set
    on engine 0 and shared, nine ragged targets submitted with caller-owned buffers, polled to completion without
    touching the scheduler, every ticket DONE with a clean fault word and bit-identical to dmp_predict on a lone context;
    argument errors answer at once; released tickets are forgotten."""
    import ctypes as C
    import time
    from dmpfold2_amd import _lib, synth
    from dmpfold2_amd.predict import Engine, encode_aln
    lib = _lib.load()
    dev = torch.device("cuda:0")
    p = C.c_void_p()
    _lib.check(lib.dmp_pipeline_create(0, 128, 512, 4, C.byref(p)))
    try:
        assert lib.dmp_pipeline_engines(p) == 4
        ctx0 = C.c_void_p(lib.dmp_pipeline_ctx(p, 0))
        # before the weights: submit refuses
        z = torch.zeros((4, 16), dtype=torch.uint8, device=dev)
        o1, o2 = torch.empty(16 * 15, device=dev), torch.empty(16, device=dev)
        assert lib.dmp_pipeline_submit(p, z.data_ptr(), 4, 16, None, 0, 0, o1.data_ptr(), o2.data_ptr(), None) < 0
        for key, val in synth_sd.items():
            arr = np.ascontiguousarray(val, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(lib.dmp_weights_set(ctx0, key.encode(), arr.ctypes.data, shape, arr.ndim))
        _lib.check(lib.dmp_weights_finalize(ctx0))
        _lib.check(lib.dmp_pipeline_weights_ready(p))
        _lib.check(lib.dmp_pipeline_set_option(p, b"precision", 2))
        v = C.c_int(0)
        for i in range(4):
            _lib.check(lib.dmp_ctx_get_option(C.c_void_p(lib.dmp_pipeline_ctx(p, i)), b"precision", C.byref(v)))
            assert v.value == 2
        assert lib.dmp_pipeline_set_option(p, b"no_such_option", 1) < 0
        # capacity and argument errors answer at once
        big = torch.zeros((4, 200), dtype=torch.uint8, device=dev)
        assert lib.dmp_pipeline_submit(p, big.data_ptr(), 4, 200, None, 0, 0, o1.data_ptr(), o2.data_ptr(), None) == -4
        assert lib.dmp_pipeline_submit(p, z.data_ptr(), 4, 7, None, 0, 0, o1.data_ptr(), o2.data_ptr(), None) == -1
        shapes = [(82, 200), (33, 64), (128, 300), (40, 1), (64, 257), (96, 31), (120, 129), (50, 64), (128, 17)]
        msas = [torch.from_numpy(encode_aln(synth.synth_msa(L, N, 90 + i))).to(dev) for i, (L, N) in enumerate(shapes)]
        outs = [(torch.empty((m.shape[1], 5, 3), device=dev), torch.empty((m.shape[1],), device=dev)) for m in msas]
        torch.cuda.synchronize()
        tickets = [lib.dmp_pipeline_submit(p, m.data_ptr(), m.shape[0], m.shape[1], None, 2, 5, c.data_ptr(), f.data_ptr(), None)
                   for m, (c, f) in zip(msas, outs)]
        assert tickets == list(range(9))
        done, buf, n = [], (C.c_int64 * 4)(), C.c_int(0)
        t0 = time.time()
        while len(done) < 9 and time.time() - t0 < 120:
            _lib.check(lib.dmp_pipeline_poll(p, buf, 4, C.byref(n)))
            done += [buf[i] for i in range(n.value)]
            time.sleep(0.001)
        assert sorted(done) == tickets                      # each exactly once
        q, r = C.c_int(1), C.c_int(1)
        _lib.check(lib.dmp_pipeline_backlog(p, C.byref(q), C.byref(r)))
        assert q.value == 0 and r.value == 0
        single = Engine(dev, 128, 512)
        single.set_weights(synth_sd)
        single.set_option("precision", 2)
        single.set_option("tridiag_cluster", 0)
        try:
            st, bits = C.c_int(0), C.c_int(0)
            for t, m, (c, f) in zip(tickets, msas, outs):
                _lib.check(lib.dmp_pipeline_status(p, t, C.byref(st), C.byref(bits)))
                assert st.value == 3 and bits.value == 0
                rc_, rf_ = single.predict_device(m, None, 2, 5)
                single.sync_check()
                assert torch.equal(c, rc_) and torch.equal(f, rf_), tuple(m.shape)
                _lib.check(lib.dmp_pipeline_release(p, t))
                assert lib.dmp_pipeline_status(p, t, C.byref(st), C.byref(bits)) < 0        # forgotten
        finally:
            single.close()
        _lib.check(lib.dmp_pipeline_wait(p, 2))             # nothing outstanding: returns at once
        stats = (C.c_longlong * 6)()
        _lib.check(lib.dmp_pipeline_stats(p, stats, 6))
        assert stats[0] >= 1 and 2 <= stats[1] <= 4 and stats[4] == 0
    finally:
        lib.dmp_pipeline_destroy(p)
    # the same on the CALLER'S streams (dmp_pipeline_create_on, here through the Python wrapper on PyTorch pool streams):
    # the same bits
    from dmpfold2_amd.predict import Pipeline
    pipe = Pipeline(dev, 128, 512, synth_sd, streams=2, precision=2, torch_streams=True)
    try:
        assert [e._stream.cuda_stream for e in pipe.engines] == [st.cuda_stream for st in pipe._torch_streams]
        res = pipe.run(msas[:3], 2, 5)
        torch.cuda.synchronize()
        for (c, f), (c0, f0) in zip(res, outs[:3]):
            assert torch.equal(c, c0) and torch.equal(f, f0)
    finally:
        pipe.close()


@pytest.mark.parametrize("L", [17, 40, 82, 200])
def test_convolution_tile_shapes_give_the_same_bits(synth_sd, L):
    """Round 6: at small L the split-product convolutions can run on 8 x 16 pixel tiles instead of
    16 x 16 (twice the workgroups).  The accumulation order of an output element and the grouping of the InstanceNorm partial sums (per
    8-row half tile in both shapes) do not depend on the shape: convolution output, statistics and a whole prediction are
    the same bits with either, in both split arithmetics; the automatic choice (option "conv_tile_bands" = 0) is the
    half-height tile up to L = 80, where the 16 x 16 shape leaves half the CUs idle (measured: it loses above)."""
    from abi import Stages
    st = Stages(synth_sd, max_L=max(L, 64), max_N=64)
    g = torch.Generator(device="cuda").manual_seed(L)
    x = torch.randn(128, L, L, device="cuda", generator=g) * 3.0
    from dmpfold2_amd import synth
    from dmpfold2_amd.predict import encode_aln
    msa = encode_aln(synth.synth_msa(L, 48, 200 + L))
    try:
        for mode in (0, 2):
            st.eng.set_option("conv_mode", mode)
            got = {}
            for bands in (1, 2, 0):
                st.eng.set_option("conv_tile_bands", bands)
                u, stats = st.conv(3, x)
                c, f = st.eng.predict(msa, None, 2, 5)
                st.eng.sync_check()
                got[bands] = (u.clone(), stats.clone(), c.clone(), f.clone())
            for k in (2, 0):
                for a, b in zip(got[1], got[k]):
                    assert torch.equal(a, b), (mode, k)
            # and the values are a convolution's: against the float32 matrix-core kernel
            st.eng.set_option("conv_mode", 1)
            u32, _ = st.conv(3, x)
            st.eng.sync_check()
            assert float((got[2][0] - u32).abs().max()) <= 2e-5 * float(u32.abs().max())
    finally:
        st.eng.set_option("conv_tile_bands", 0)
        st.eng.set_option("conv_mode", 0)
        st.eng.close()
