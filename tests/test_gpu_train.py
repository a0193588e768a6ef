"""Training-side slice: the backward of the pair trunk's residual blocks and head on the
MI355X against the reference's OWN autograd (train.py:318-344 runs it through network.py:85-103), at the sizes training
runs at - L = 128, the reference's 350 crop (train.py:26-27) - and through the sixteen blocks + head of net.resnet in one
backward pass.

Fixtures: tests/golden/make_goldens.py (bwd_block7_full_L128, bwd_block3_full_L350, bwd_resnet_L96).  The tensors are too
large to store, so the inputs are regenerated here from their Philox keys (NumPy's counter RNG: the same bits on every
machine) and of the large outputs a fixed sample of 4096 entries + sum + sum of squares is compared; the small parameter
gradients are compared in full.  Tolerance: 1e-4 of each tensor's scale (float32 arithmetic on both sides).

Maxout near-ties.  torch.max routes a gradient to ONE channel of a quadruple.  Two float32 implementations of the same
convolution differ by 1e-6 relative (summation order), so a decision whose two largest channels are closer than that can
go either way - about one in a million, i.e. a handful per block at these sizes - and a single flipped decision moves
weight-gradient rows by |du x| >> 1e-4 of scale and, through a chain of blocks, everything upstream of it.  Both
outcomes are valid subgradients; to compare EXACTLY the fixtures list every decision the reference made with a margin
below 1e-4 (a few hundred per block) with its winner, and the tests substitute those into the winners the HIP forward
saved (dmp_block_conv5x5_maxout_winners -> dmp_block_conv5x5_maxout_bwd: the autograd contract - the forward saves the
argmax, the backward consumes it).  How many the HIP forward resolved differently is printed.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def philox_plane(key, shape, scale):
    rng = np.random.Generator(np.random.Philox(key=int(key)))
    return ((2.0 * rng.random(shape)) - 1.0).astype(np.float32) * np.float32(scale)


def close_full(got, want, tol=1e-4, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want, dtype=np.float32).reshape(got.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, (what, err, scale)
    return err / scale


def close_sample(got, g, name, tol=1e-4):
    """got: a GPU tensor; the fixture holds `name`.idx / .val (a fixed sample of its entries) / .sum / .sumsq"""
    flat = got.reshape(-1)
    idx = torch.from_numpy(g[name + ".idx"]).to(flat.device)
    want = g[name + ".val"]
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(flat[idx].cpu().numpy() - want).max())
    assert err <= tol * scale, (name, err, scale)
    d = flat.double()
    # sum: n entries with independent errors of tol x scale at worst add up to tol x scale x sqrt(n) ... x 4 for slack
    assert abs(float(d.sum()) - float(g[name + ".sum"])) <= 4.0 * tol * scale * np.sqrt(flat.numel()), name
    assert abs(float((d * d).sum()) - float(g[name + ".sumsq"])) <= 4.0 * tol * float(g[name + ".sumsq"]), name
    return err / scale


def substitute_near_ties(idx, at, win):
    """the reference's winners at the decisions it made with a margin below the fixture's (in place); -> how many differed"""
    flat = idx.view(-1)
    at = torch.from_numpy(at).to(idx.device)
    win = torch.from_numpy(win).to(idx.device)
    differ = int((flat[at] != win).sum())
    flat[at] = win
    return differ


@pytest.mark.parametrize("name", ["bwd_block7_full_L128", "bwd_block3_full_L350"])
def test_block_backward_at_training_sizes_vs_reference_autograd(synth_sd, name):
    """One whole ResNet_Block (network.py:85-103, evaluation mode) backwards at L = 128 and at the 350 crop: every
    parameter gradient, the gradient at the interface (the maxout output) and the block input's gradient including the
    residual branch; the forward's maxout output on the way."""
    from abi import Stages
    g = load_golden(name)
    L, blk = int(g["L"]), int(g["block"])
    st = Stages(synth_sd, max_L=L, max_N=8)
    try:
        x = st.to(philox_plane(g["x_key"], (128, L, L), float(g["x_scale"])))
        dout = st.to(philox_plane(g["dout_key"], (128, L, L), 1.0))
        u, idx = st.conv_winners(blk, x)
        rel_u = close_sample(u, g, "u", 1e-5)
        own = idx.clone()
        differ = substitute_near_ties(idx, g["tie.at"], g["tie.win"])
        du, dp = st.norm_bwd(blk, u, dout)
        dx, dw, db = st.conv_bwd(blk, x, du, idx)
        st.eng.sync_check()
        rel = {"u": rel_u, "du": close_sample(du, g, "du"), "dx": close_sample(dx + dout, g, "dx"),
               "dw": close_sample(dw, g, "dw"), "db": close_full(db, g["db"], what="db")}
        dp = dp.cpu().numpy()
        close_full(dp[0:128], g["dgamma"], what="dgamma")
        close_full(dp[128:256], g["dbeta"], what="dbeta")
        close_full(dp[256:256 + 1024].reshape(8, 128), g["dfc0"], what="dfc0")
        close_full(dp[1280:1280 + 1024].reshape(128, 8), g["dfc2"], what="dfc2")
        close_full(dp[2304:2432], g["dsse_w"], what="dsse_w")
        close_full(dp[2432:2433], g["dsse_b"], what="dsse_b")
        print(f"{name}: {len(g['tie.at'])} near-ties listed, {differ} resolved differently by the HIP forward; "
              "relative deviations " + ", ".join(f"{k} {v:.1e}" for k, v in rel.items()))
        # without saved winners the backward runs its own forward: the bits of the backward WITH the HIP forward's winners
        a = st.conv_bwd(blk, x, du, own)
        b = st.conv_bwd(blk, x, du, None)
        st.eng.sync_check()
        assert all(torch.equal(p, q) for p, q in zip(a, b))
        # a flipped near-tie is visible: the substitution above was not a no-op whenever the two forwards disagreed
        if differ:
            assert not torch.equal(a[1], dw)
        # the routed gradient keeps its mass: the four bias gradients of a quadruple sum to that of its maxout channel
        assert float((db.view(128, 4).sum(1) - du.double().sum((1, 2)).float()).abs().max()) <= 1e-3 * float(du.abs().sum((1, 2)).max())
    finally:
        st.eng.close()


def test_resnet_backward_through_sixteen_blocks_and_head_vs_reference_autograd(synth_sd):
    """net.resnet[1..17] (the sixteen residual blocks and the 1x1 head; the stem is not part of the slice) forwards in
    float32 through the stage entry points, then backwards block by block - head, then per block the second half
    (InstanceNorm + scSE + residual), the first half (convolution + maxout), the residual add - against ONE backward pass
    of the reference's autograd: the gradient at the head's input, at the trunk's input, every block's bias / norm / sSE /
    cSE gradients and the weight gradients of blocks 1, 8 and 16."""
    from abi import Stages
    g = load_golden("bwd_resnet_L96")
    L = int(g["L"])
    st = Stages(synth_sd, max_L=L, max_N=8)
    try:
        st.eng.set_option("precision", 1)                    # the forward in float32 as well
        x = st.to(philox_plane(g["x_key"], (128, L, L), float(g["x_scale"])))
        g2 = st.to(philox_plane(g["g_key"], (2, L, L), 1.0))
        xs, us, idxs, flips = {}, {}, {}, 0
        for k in range(1, 17):
            u, stats = st.conv(k, x)
            u2, idx = st.conv_winners(k, x)
            assert torch.equal(u, u2)                        # the same kernel with and without the winners
            flips += substitute_near_ties(idx, g[f"b{k}.tie.at"], g[f"b{k}.tie.win"])
            xs[k], us[k], idxs[k] = x, u, idx
            x = st.norm(k, u, stats, x)
        close_sample(x, g, "x16", 1e-4)
        d, hp = st.head_bwd(x, g2)
        hp = hp.cpu().numpy()
        close_full(hp[:256].reshape(2, 128), g["head_dw"], what="head_dw")
        close_full(hp[256:258], g["head_db"], what="head_db")
        worst = {"dx16": close_sample(d, g, "dx16")}
        for k in range(16, 0, -1):
            du, dp = st.norm_bwd(k, us[k], d)
            dx, dw, db = st.conv_bwd(k, xs[k], du, idxs[k])
            d = dx + d
            dp = dp.cpu().numpy()
            r = [close_full(db, g[f"b{k}.db"], what=f"b{k}.db"), close_full(dp[0:128], g[f"b{k}.dgamma"], what=f"b{k}.dgamma"),
                 close_full(dp[128:256], g[f"b{k}.dbeta"], what=f"b{k}.dbeta"),
                 close_full(dp[1280:1280 + 1024].reshape(128, 8), g[f"b{k}.dfc2"], what=f"b{k}.dfc2"),
                 close_full(dp[2304:2432], g[f"b{k}.dsse_w"], what=f"b{k}.dsse_w")]
            if k in (1, 8, 16):
                r.append(close_sample(dw, g, f"b{k}.dw"))
            worst[f"b{k}"] = max(r)
        st.eng.sync_check()
        worst["dx0"] = close_sample(d, g, "dx0")
        print(f"bwd_resnet_L96: {flips} near-ties resolved differently by the HIP forward over 16 blocks; worst relative "
              "deviation per stage: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    finally:
        st.eng.close()


def test_stem_backward_vs_reference_autograd(synth_sd):
    """resnet[0] = Maxout2d(955 -> 128, pool 3, kernel 1) + InstanceNorm (network.py:194, 12-34) on an input built as
    GRUResNet.forward builds it (network.py:226-229: outer product of mat1d, the 442 covariance channels, the distance
    channel), backwards: lin.weight.grad for all 955 channels - outer-product, covariance, contact and distance columns -
    lin.bias.grad, the norm's gradients, and the gradient that flows on into the sequence trunk (d mat1d).  The 955-channel
    input is never materialised on the device: the static stem's Z0 forwards, panels of matrix-core GEMMs backwards."""
    from abi import Stages
    g = load_golden("bwd_stem_L96")
    L = int(g["L"])
    st = Stages(synth_sd, max_L=L, max_N=8)
    try:
        mat1d = st.to(philox_plane(g["mat1d_key"], (512, L), 1.0))
        f2d = philox_plane(g["f2d_key"], (442, L, L), float(g["f2d_scale"]))
        dmap = st.to(np.abs(philox_plane(g["dmap_key"], (L, L), float(g["dmap_scale"]))))
        dy = st.to(philox_plane(g["g_key"], (128, L, L), 1.0))
        # the covariance channels as the inverse they are read from: inv[21 i + a][21 j + b] = f2d[21 a + b][i][j]
        inv = st.to(np.ascontiguousarray(f2d[:441].reshape(21, 21, L, L).transpose(2, 0, 3, 1)).reshape(21 * L, 21 * L))
        contacts = st.to(f2d[441])
        z0 = st.stem_static(mat1d, inv, contacts)
        u, idx = st.stem_winners(z0, dmap)
        rel = {"u": close_sample(u, g, "u", 1e-5)}
        y = st.stem_update(z0, dmap)
        rel["y"] = close_sample(y, g, "y", 1e-4)
        differ = substitute_near_ties(idx, g["tie.at"], g["tie.win"])
        dw, dp, dm = st.stem_bwd(u, idx, dy, mat1d, dmap)
        st.eng.sync_check()
        rel["dw"] = close_sample(dw, g, "dw")
        rel["dw_outer"] = close_sample(dw[:, :512].contiguous(), g, "dw_outer")
        rel["dw_dist"] = close_full(dw[:, 954], g["dw_dist"], what="dw_dist")
        rel["dw_contacts"] = close_full(dw[:, 953], g["dw_contacts"], what="dw_contacts")
        rel["dmat1d"] = close_sample(dm, g, "dmat1d")
        dp = dp.cpu().numpy()
        rel["db"] = close_full(dp[:384], g["db"], what="db")
        rel["dgamma"] = close_full(dp[384:512], g["dgamma"], what="dgamma")
        rel["dbeta"] = close_full(dp[512:640], g["dbeta"], what="dbeta")
        print(f"bwd_stem_L96: {len(g['tie.at'])} near-ties listed, {differ} resolved differently by the HIP forward; relative "
              "deviations " + ", ".join(f"{k} {v:.1e}" for k, v in rel.items()))
        # the block entry points still work after the stem used the shared workspace (the flipped weight pack is rebuilt)
        x = st.to(philox_plane(7, (128, L, L), 1.0))
        du = st.to(philox_plane(8, (128, L, L), 1.0))
        a = st.conv_bwd(3, x, du, None)
        st.stem_bwd(u, idx, dy, mat1d, dmap)
        b = st.conv_bwd(3, x, du, None)
        st.eng.sync_check()
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    finally:
        st.eng.close()


def test_whole_resnet_backward_stem_blocks_head_vs_reference_autograd(synth_sd):
    """ONE backward pass of the reference's autograd through ALL of net.resnet (network.py:194-207: the stem, the sixteen
    residual blocks, the 1x1 head) from the gradient at the two head planes back to mat1d, composed here from the entry
    points of the training slice: forwards in float32 with the winners saved per maxout (the reference's near-ties
    substituted, see the module docstring), backwards head -> blocks 16 .. 1 -> stem."""
    from abi import Stages
    g = load_golden("bwd_resnet_whole_L96")
    L = int(g["L"])
    st = Stages(synth_sd, max_L=L, max_N=8)
    try:
        st.eng.set_option("precision", 1)
        mat1d = st.to(philox_plane(g["mat1d_key"], (512, L), 1.0))
        f2d = philox_plane(g["f2d_key"], (442, L, L), float(g["f2d_scale"]))
        dmap = st.to(np.abs(philox_plane(g["dmap_key"], (L, L), float(g["dmap_scale"]))))
        g2 = st.to(philox_plane(g["g_key"], (2, L, L), 1.0))
        inv = st.to(np.ascontiguousarray(f2d[:441].reshape(21, 21, L, L).transpose(2, 0, 3, 1)).reshape(21 * L, 21 * L))
        z0 = st.stem_static(mat1d, inv, st.to(f2d[441]))
        u0, idx0 = st.stem_winners(z0, dmap)
        flips = substitute_near_ties(idx0, g["b0.tie.at"], g["b0.tie.win"])
        x = st.stem_update(z0, dmap)
        xs, us, idxs = {}, {}, {}
        for k in range(1, 17):
            u, stats = st.conv(k, x)
            _, idx = st.conv_winners(k, x)
            flips += substitute_near_ties(idx, g[f"b{k}.tie.at"], g[f"b{k}.tie.win"])
            xs[k], us[k], idxs[k] = x, u, idx
            x = st.norm(k, u, stats, x)
        # the head's two planes (its forward is part of dmp_head_gram on the prediction path: here by hand, float64)
        W = {k: torch.from_numpy(np.array(v)).to(st.dev) for k, v in synth_sd.items() if k.startswith("resnet.17.")}
        out = (torch.einsum("hc,cij->hij", W["resnet.17.weight"].reshape(2, 128).double(), x.double())
               + W["resnet.17.bias"].double()[:, None, None]).float()
        worst = {"out": close_sample(out, g, "out")}
        d, hp = st.head_bwd(x, g2)
        hp = hp.cpu().numpy()
        worst["head"] = max(close_full(hp[:256].reshape(2, 128), g["head_dw"], what="head_dw"),
                            close_full(hp[256:258], g["head_db"], what="head_db"))
        for k in range(16, 0, -1):
            du, dp = st.norm_bwd(k, us[k], d)
            dx, dw, db = st.conv_bwd(k, xs[k], du, idxs[k])
            d = dx + d
            worst[f"b{k}"] = max(close_full(db, g[f"b{k}.db"], what=f"b{k}.db"),
                                 close_full(dp[0:128], g[f"b{k}.dgamma"], what=f"b{k}.dgamma"))
        sdw, sdp, dm = st.stem_bwd(u0, idx0, d, mat1d, dmap)
        st.eng.sync_check()
        sdp = sdp.cpu().numpy()
        worst["stem"] = max(close_sample(sdw, g, "stem.dw"), close_full(sdp[:384], g["stem.db"], what="stem.db"),
                            close_full(sdp[384:512], g["stem.dgamma"], what="stem.dgamma"),
                            close_full(sdp[512:640], g["stem.dbeta"], what="stem.dbeta"))
        worst["dmat1d"] = close_sample(dm, g, "dmat1d")
        print(f"bwd_resnet_whole_L96: {flips} near-ties resolved differently by the HIP forward; worst relative deviation per "
              "stage: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    finally:
        st.eng.close()


@pytest.mark.parametrize("max_L", [32, 64])
def test_block_backward_in_a_small_context(synth_sd, max_L):
    """the bias gradient's first-stage partial sums (128 x 16 x 4 doubles) live in the workspace's norm
    region, which was sized by the InstanceNorm scratch alone - smaller than those partials for contexts of max_L <= 72, so
    they ran into the padded input planes behind them and dW of input channel 0 came out wrong, silently.  The whole-block
    fixture at L = 24 (the reference's autograd through ResNet_Block 3) in contexts created for max_L = 32 and 64."""
    from abi import Stages
    g = load_golden("bwd_block3_full_L24")
    st = Stages(synth_sd, max_L=max_L, max_N=8)
    blk = int(g["block"])
    du, dp = st.norm_bwd(blk, st.to(g["u"]), st.to(g["dout"]))
    dx, dw, db = st.conv_bwd(blk, st.to(g["x"]), du)
    st.eng.sync_check()
    close_full(du, g["du"], what="du")
    close_full(dx + st.to(g["dout"]), g["dx"], what="dx")
    close_full(db, g["db"], what="db")
    close_full(dp[0:128], g["dgamma"], what="dgamma")
    close_sample(dw, g, "dw")
    # input channel 0's rows of dW are the ones the overflow destroyed: every sampled entry of them, explicitly
    flat = dw.reshape(512, 128, 25)
    ch0 = torch.from_numpy(g["dw.idx"][(g["dw.idx"] // 25) % 128 == 0]).to(dw.device)
    want = g["dw.val"][(g["dw.idx"] // 25) % 128 == 0]
    if len(want):
        assert float(np.abs(flat.reshape(-1)[ch0].cpu().numpy() - want).max()) <= 1e-4 * float(np.abs(g["dw.val"]).max())
