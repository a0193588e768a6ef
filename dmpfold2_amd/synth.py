"""Platform-independent synthetic inputs for the DMPfold2 hot path.

The trained weights of the reference are not in its tree
(/root/reference/.MISSING_LARGE_BLOBS) and there is no network, so parity and
benchmarks run on synthetic weights that have exactly the reference's
``state_dict`` keys and shapes (reference dmpfold/network.py:182-215, the
"weight ABI") and on synthetic alignments.  Both generators use NumPy's
counter-based Philox bit generator with uniform variates only, so the same
seed gives the same bytes in this container and on the GPU box; the golden
fixtures store a checksum of every generated tensor.
"""
from __future__ import annotations

import hashlib
from math import sqrt

import numpy as np

AA20 = "ARNDCQEGHILKMFPSTWYV"

WIDTH = 512        # GRU width (reference network.py:182 GRUResNet(512, 128))
CWIDTH = 128       # pair-trunk width
NBLOCKS = 16       # network.py:200
NUM_DCA = 442      # network.py:10
STEM_IN = NUM_DCA + WIDTH + 1   # 955
STEM_POOL = 3
BLOCK_POOL = 4
KSIZE = 5


def weight_spec():
    """[(key, shape)] in the order of ``GRUResNet(512,128).state_dict()``."""
    spec = [("embed.weight", (22, 22))]

    def gru(prefix, nin, hid, layers, bidir):
        out = []
        for l in range(layers):
            lin = nin if l == 0 else hid * (2 if bidir else 1)
            for sfx in ([""] + (["_reverse"] if bidir else [])):
                out += [(f"{prefix}.weight_ih_l{l}{sfx}", (3 * hid, lin)),
                        (f"{prefix}.weight_hh_l{l}{sfx}", (3 * hid, hid)),
                        (f"{prefix}.bias_ih_l{l}{sfx}", (3 * hid,)),
                        (f"{prefix}.bias_hh_l{l}{sfx}", (3 * hid,))]
        return out

    spec += gru("vgru", 22, WIDTH, 2, False)
    spec += gru("hgru", WIDTH, WIDTH // 2, 2, True)
    spec += [("resnet.0.lin.weight", (CWIDTH * STEM_POOL, STEM_IN, 1, 1)),
             ("resnet.0.lin.bias", (CWIDTH * STEM_POOL,)),
             ("resnet.0.norm.weight", (CWIDTH,)),
             ("resnet.0.norm.bias", (CWIDTH,))]
    for k in range(1, NBLOCKS + 1):
        p = f"resnet.{k}"
        spec += [(f"{p}.layer1.lin.weight", (CWIDTH * BLOCK_POOL, CWIDTH, KSIZE, KSIZE)),
                 (f"{p}.layer1.lin.bias", (CWIDTH * BLOCK_POOL,)),
                 (f"{p}.layer1.norm.weight", (CWIDTH,)),
                 (f"{p}.layer1.norm.bias", (CWIDTH,)),
                 (f"{p}.scSE.cSE.fc.0.weight", (CWIDTH // 16, CWIDTH)),
                 (f"{p}.scSE.cSE.fc.2.weight", (CWIDTH, CWIDTH // 16)),
                 (f"{p}.scSE.sSE.conv.weight", (1, CWIDTH, 1, 1)),
                 (f"{p}.scSE.sSE.conv.bias", (1,))]
    spec += [(f"resnet.{NBLOCKS + 1}.weight", (2, CWIDTH, 1, 1)),
             (f"resnet.{NBLOCKS + 1}.bias", (2,))]
    spec += gru("coord_gru", WIDTH + 8, WIDTH // 2, 3, True)
    spec += [("coord_fc.weight", (3, WIDTH))]
    return spec


def _uniform(rng, shape, bound):
    u = rng.random(size=shape, dtype=np.float64)
    return ((2.0 * u - 1.0) * bound).astype(np.float32)


def synth_weights(seed: int = 0, coord_scale: float = 1.0, act_scale: float = 1.0):
    """Synthetic ``state_dict`` (name -> float32 ndarray).

    Scale rules follow the reference's initialisers so activations stay in a
    sane range: GRU U(+-1/sqrt(H)); convolutions Xavier-uniform with the
    per-block gain 1/sqrt(block) of network.py:20-23; InstanceNorm gamma
    1+-0.2 and beta +-0.5 (non-trivial, so the cSE gate is exercised).
    ``coord_fc`` is scaled so C-alpha traces span tens of Angstroms rather
    than collapsing onto a point.  ``act_scale`` multiplies every InstanceNorm
    gamma and beta (1.0 leaves the bytes of the default set unchanged): the
    residual stream of the pair trunk grows by about that factor per block, a
    second activation regime for the parity fixtures.
    """
    rng = np.random.Generator(np.random.Philox(key=int(seed) + 0x5EED))
    sd = {}
    for name, shape in weight_spec():
        if name == "embed.weight":
            w = np.eye(22, dtype=np.float32)
        elif name.startswith(("vgru.", "hgru.", "coord_gru.")):
            hid = shape[0] // 3
            w = _uniform(rng, shape, 1.0 / sqrt(hid))
        elif name.endswith("norm.weight"):
            w = np.float32(act_scale) * (1.0 + _uniform(rng, shape, 0.2))
        elif name.endswith("norm.bias"):
            w = np.float32(act_scale) * _uniform(rng, shape, 0.5)
        elif name.endswith("lin.weight"):
            cout, cin, kh, kw = shape
            block = int(name.split(".")[1])
            gain = 1.0 / sqrt(block) if block > 0 else 1.0
            bound = gain * sqrt(6.0 / (cin * kh * kw + cout * kh * kw))
            w = _uniform(rng, shape, bound)
        elif name.endswith("lin.bias"):
            block = int(name.split(".")[1])
            fan_in = STEM_IN if block == 0 else CWIDTH * KSIZE * KSIZE
            w = _uniform(rng, shape, 1.0 / sqrt(fan_in))
        elif ".cSE.fc." in name:
            w = _uniform(rng, shape, 1.0 / sqrt(shape[1]))
        elif ".sSE.conv." in name or name.startswith(f"resnet.{NBLOCKS + 1}."):
            w = _uniform(rng, shape, 1.0 / sqrt(CWIDTH))
        elif name == "coord_fc.weight":
            w = _uniform(rng, shape, coord_scale)
        else:  # pragma: no cover
            raise KeyError(name)
        sd[name] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


def weights_checksum(sd) -> str:
    h = hashlib.sha256()
    for name, _ in weight_spec():
        h.update(name.encode())
        h.update(np.ascontiguousarray(sd[name], dtype=np.float32).tobytes())
    return h.hexdigest()


def headline_fixture_weights(coord_fc, mds_scale, seed: int = 0, coord_scale: float = 5.0):
    """The weight set of the fixture that pins the benchmark's full setting (tests/golden/make_goldens.py,
    fitns_L300_N2000_n10_m100): synth_weights(seed) with the 8 MDS columns of the coordinate GRU's first-layer
    input weights scaled by `mds_scale` and `coord_fc` replaced by the fitted matrix stored in the fixture."""
    sd = dict(synth_weights(seed, coord_scale=coord_scale))
    for k in ("coord_gru.weight_ih_l0", "coord_gru.weight_ih_l0_reverse"):
        w = np.array(sd[k]).copy()
        w[:, WIDTH:WIDTH + 8] *= np.float32(mds_scale)
        sd[k] = w
    sd["coord_fc.weight"] = np.ascontiguousarray(coord_fc, dtype=np.float32)
    return sd


def scale_block_norms(sd, blocks, factor):
    """A copy of `sd` with the InstanceNorm gamma and beta of the trunk blocks in `blocks` (0 = stem,
    1..16 = residual blocks) multiplied by `factor`: a mixed activation regime (some blocks add
    O(1) terms to the residual stream, others O(factor) ones) for the parity fixtures."""
    out = dict(sd)
    for k in blocks:
        p = "resnet.0.norm" if k == 0 else f"resnet.{k}.layer1.norm"
        for sfx in (".weight", ".bias"):
            out[p + sfx] = np.ascontiguousarray(np.array(sd[p + sfx]) * np.float32(factor), dtype=np.float32)
    return out


def synth_msa(L: int, N: int, seed: int = 0):
    """Synthetic alignment as a list of N strings of length L.

    Row 0 is uniform over the 20 standard residues (never gap/unknown, so the
    PDB writer's residue table applies); each other row copies row 0, mutating
    each position to a uniform residue w.p. 0.3 and to a gap w.p. 0.05.
    """
    rng = np.random.Generator(np.random.Philox(key=int(seed) + 0xA11))
    q = rng.integers(0, 20, size=L)
    rows = np.broadcast_to(q, (N, L)).copy()
    u = rng.random(size=(N, L))
    sub = rng.integers(0, 20, size=(N, L))
    mut = u < 0.3
    gap = (u >= 0.3) & (u < 0.35)
    rows[mut] = sub[mut]
    rows[0] = q
    letters = np.frombuffer((AA20 + "-").encode(), dtype=np.uint8)
    rows[gap] = 20
    rows[0] = q
    txt = letters[rows]
    return [bytes(r).decode("ascii") for r in txt]


def write_aln(path, rows):
    with open(path, "w") as f:
        for r in rows:
            f.write(r + "\n")


def save_state_dict(path, sd):
    """Save in the reference's weight-file format (a pickled tensor dict)."""
    import torch
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, path)


def protein_like_trace(L, seed, bond=3.8, min_sep=4.5, radius=None):
    """A self-avoiding CA trace of L points (builder's own generator, Philox counter RNG): 3.8 A bonds,
    i / i+2 distance in [5, 7] A, every other pair at least `min_sep` apart, confined to a sphere of the
    radius a compact chain of that length fills.  A regression target for `fit_coord_fc` at lengths for
    which the reference tree holds no structure (3FGX chain A has 96 residues)."""
    rng = np.random.Generator(np.random.Philox(key=int(seed) + 0xCA))
    radius = radius or 3.3 * L ** (1.0 / 3.0) + 6.0
    pts = [np.zeros(3), np.array([bond, 0.0, 0.0])]
    fails = 0
    while len(pts) < L:
        u = rng.standard_normal(3)
        cand = pts[-1] + bond * u / np.linalg.norm(u)
        P = np.asarray(pts)
        d2 = np.linalg.norm(cand - P[-2])
        ok = 5.0 <= d2 <= 7.0 and np.linalg.norm(cand - P.mean(0)) <= radius
        if ok and len(pts) > 2:
            ok = np.linalg.norm(P[:-2] - cand, axis=1).min() >= min_sep
        if ok:
            pts.append(cand)
            fails = 0
        else:
            fails += 1
            if fails > 400:                      # dead end: back out of it
                del pts[max(2, len(pts) - 6):]
                fails = 0
    return np.asarray(pts, dtype=np.float32)
