"""CPU oracle for the DMPfold2 ``aln_to_coords`` hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (NumPy + PyTorch-CPU operators) of the
algorithm in the reference tree, written from the behaviour documented in
SURVEY.md; every function cites the reference lines it follows
(paths relative to /root/reference/).  It exists so that the HIP product path
can be checked against something that runs without the reference being
present (the reference's Python cannot travel to the GPU box).

Rules (see DESIGN.md "Oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import this module; the product package never does;
  * parity pin: tests/golden/*.npz were produced by importing the REAL
    reference in the build container (tests/golden/make_goldens.py) and this
    oracle is checked against them in tests/test_oracle_vs_golden.py.  The
    reference's own test-suite holds no numerical fixtures for this path
    (its CI only checks exit codes), so those captured outputs are the pin.

Eigenvector signs.  ``torch.symeig`` (network.py:247,292) no longer exists in
PyTorch; the reference runs only with a ``symeig`` provider patched in.  Signs
of eigenvectors are implementation-defined, so two providers are supported and
both were used to capture goldens:
  ``eig_sign="lapack"``    - ``torch.linalg.eigh(UPLO='U')`` as returned by MKL;
  ``eig_sign="canonical"`` - the same, then each eigenvector is flipped so that
                             its largest-magnitude component is positive
                             (first such index on ties).  This is the rule the
                             HIP solver implements.
"""
from __future__ import annotations

from math import asin, cos, pi, sin, sqrt

import numpy as np
import torch
import torch.nn.functional as F

MAX_SEQS = 3000          # predict.py:130-132
_ALPHABET = "ARNDCQEGHILKMFPSTWYV"
_RESNAMES = ("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER "
             "THR TRP TYR VAL").split()


# ----------------------------------------------------------------------------
# alignment text -> codes            (predict.py:100-104, 124-132)
# ----------------------------------------------------------------------------
def read_aln(path):
    """Keep every line that does not start with '>' and strip trailing
    whitespace (predict.py:100-104)."""
    rows = []
    with open(path, "r") as fh:
        for line in fh.readlines():
            if line[:1] != ">":
                rows.append(line.rstrip())
    return rows


def _code_table():
    # predict.py:124: 20 residues -> 0..19, BJOUXZ -> 20, '-' and '.' -> 21;
    # every other byte b maps to (b - 65) mod 256 (uint8 wrap of "- ord('A')").
    tab = (np.arange(256, dtype=np.int64) - 65).astype(np.uint8)
    for i, ch in enumerate(_ALPHABET):
        tab[ord(ch)] = i
    for ch in "BJOUXZ":
        tab[ord(ch)] = 20
    for ch in "-.":
        tab[ord(ch)] = 21
    return tab


def encode_aln(rows):
    """rows -> uint8 (N, L) matrix, capped at 3000 rows (predict.py:126-132).
    Raises ValueError on ragged input exactly like the reference's reshape."""
    nseqs = len(rows)
    length = len(rows[0])
    flat = np.frombuffer("".join(rows).encode("latin-1"), dtype=np.uint8)
    mat = _code_table()[flat].reshape(nseqs, length)
    if nseqs > MAX_SEQS:
        mat = mat[:MAX_SEQS]
    return mat


# ----------------------------------------------------------------------------
# sequence weights                    (predict.py:32-37)
# ----------------------------------------------------------------------------
def reweight(alnmat, cutoff=0.8):
    """w_n = 1 / #{m : matches(n,m) > L*cutoff}, matches counted on
    min(code,20) (gap and unknown share one-hot class 20, predict.py:136).
    The reference does the comparison in float32 on exact integer counts."""
    a = np.minimum(np.asarray(alnmat), 20).astype(np.uint8)
    n, L = a.shape
    id_min = np.float32(L * cutoff)
    counts = np.empty(n, dtype=np.int64)
    blk = max(1, min(n, (1 << 26) // max(1, n * L)))   # bound the temporary
    for s in range(0, n, blk):
        eq = (a[s:s + blk, None, :] == a[None, :, :]).sum(axis=2)
        counts[s:s + blk] = (eq.astype(np.float32) > id_min).sum(axis=1)
    return (np.float32(1.0) / counts.astype(np.float32)).astype(np.float32)


# ----------------------------------------------------------------------------
# DCA features                        (predict.py:41-61)
# ----------------------------------------------------------------------------
def fast_dca(alnmat, w, penalty=4.5, capture=None):
    """Shrunk-covariance inverse and APC contacts.  Returns (L, L, 442) f32."""
    a = torch.from_numpy(np.minimum(np.asarray(alnmat), 20).astype(np.int64))
    w = torch.as_tensor(w, dtype=torch.float32)
    n, L = a.shape
    ns = 21
    x = F.one_hot(a, ns).float().reshape(n, L * ns)
    neff = w.sum()
    num_points = neff - torch.sqrt(w.mean())
    mean = (x * w[:, None]).sum(dim=0, keepdim=True) / num_points
    xc = (x - mean) * torch.sqrt(w[:, None])
    cov = (xc.t() @ xc) / num_points
    cov_reg = cov + torch.eye(L * ns) * penalty / torch.sqrt(neff)
    inv_cov = torch.inverse(cov_reg)
    blocks = inv_cov.view(L, ns, L, ns)
    feats = blocks.transpose(1, 2).contiguous().reshape(L, L, ns * ns)
    off = 1.0 - torch.eye(L)
    norms = torch.sqrt((blocks[:, :-1, :, :-1] ** 2).sum(dim=(1, 3))) * off
    apc = norms.sum(dim=0, keepdim=True) * norms.sum(dim=1, keepdim=True) / norms.sum()
    contacts = (norms - apc) * off
    if capture is not None:
        capture["cov_reg"] = cov_reg
        capture["inv_cov"] = inv_cov
        capture["contacts"] = contacts
    return torch.cat((feats, contacts[:, :, None]), dim=2)


# ----------------------------------------------------------------------------
# network pieces                      (network.py)
# ----------------------------------------------------------------------------
def _gru(weights, prefix, x, nin, hid, layers, bidir, batch_first):
    """torch.nn.GRU semantics (gate order r,z,n; h0 = 0), weights by key."""
    g = torch.nn.GRU(nin, hid, num_layers=layers, bidirectional=bidir,
                     batch_first=batch_first)
    g.load_state_dict({k[len(prefix) + 1:]: v for k, v in weights.items()
                       if k.startswith(prefix + ".")})
    g.eval()
    with torch.no_grad():
        return g(x)[0]


def gru_manual(weights, prefix, x, layers, bidir):
    """Explicit GRU recurrence on a (T, B, nin) tensor - the definition the
    HIP kernels implement; checked against ``_gru`` in the tests.
        r = s(W_ir x + b_ir + W_hr h + b_hr);  z likewise
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1-z) n + z h
    """
    inp = x
    for l in range(layers):
        outs = []
        for sfx in ([""] + (["_reverse"] if bidir else [])):
            wih = weights[f"{prefix}.weight_ih_l{l}{sfx}"]
            whh = weights[f"{prefix}.weight_hh_l{l}{sfx}"]
            bih = weights[f"{prefix}.bias_ih_l{l}{sfx}"]
            bhh = weights[f"{prefix}.bias_hh_l{l}{sfx}"]
            H = whh.shape[1]
            h = torch.zeros(inp.shape[1], H)
            seq = range(inp.shape[0]) if sfx == "" else range(inp.shape[0] - 1, -1, -1)
            out = torch.zeros(inp.shape[0], inp.shape[1], H)
            for t in seq:
                gi = inp[t] @ wih.t() + bih
                gh = h @ whh.t() + bhh
                r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                nn_ = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                h = (1.0 - z) * nn_ + z * h
                out[t] = h
            outs.append(out)
        inp = torch.cat(outs, dim=2)
    return inp


def sequence_trunk(weights, alnmat):
    """embed + vgru + hgru -> mat1d (512, L)      (network.py:223-226)."""
    idx = torch.from_numpy(np.asarray(alnmat).astype(np.int64))
    x = weights["embed.weight"][idx]                       # (N, L, 22)
    v = _gru(weights, "vgru", x, 22, 512, 2, False, False)   # time = N, batch = L
    h = _gru(weights, "hgru", v[-1].unsqueeze(1), 512, 256, 2, True, False)
    return h[:, 0, :].t().contiguous()                     # (512, L)


def cse_gate(weights, k):
    """Channel gate of block k.  avgpool(InstanceNorm(x)) equals the norm's
    bias exactly in real arithmetic, so the gate is input independent
    (network.py:37-52 applied to the output of network.py:32)."""
    beta = weights[f"resnet.{k}.layer1.norm.bias"]
    w1 = weights[f"resnet.{k}.scSE.cSE.fc.0.weight"]
    w2 = weights[f"resnet.{k}.scSE.cSE.fc.2.weight"]
    return torch.sigmoid(w2 @ torch.relu(w1 @ beta))


def pair_trunk(weights, resinp, capture=None, tag=""):
    """Stem + 16 maxout/scSE residual blocks + head   (network.py:12-103,
    194-207).  resinp: (1, 955, L, L) -> (1, 2, L, L)."""
    x = stem(weights, resinp)
    if capture is not None:
        capture[tag + "stem"] = x
    for k in range(1, 17):
        x = block_finish(weights, k, block_conv(weights, k, x), x)
        if capture is not None and k in (1, 16):
            capture[tag + f"block{k}"] = x
    return head(weights, x)


def stem(weights, resinp):
    """network.py:25-32 with pool 3, kernel 1: conv, max over channel triples, InstanceNorm."""
    L = resinp.shape[-1]
    x = F.conv2d(resinp, weights["resnet.0.lin.weight"], weights["resnet.0.lin.bias"])
    x = x.view(1, 128, 3, L, L).max(dim=2)[0]
    return F.instance_norm(x, weight=weights["resnet.0.norm.weight"],
                           bias=weights["resnet.0.norm.bias"], eps=1e-5)


def block_conv(weights, k, x):
    """network.py:25-31 for block k: 5x5 conv (pad 2) + max over channel quadruples."""
    L = x.shape[-1]
    p = f"resnet.{k}"
    u = F.conv2d(x, weights[p + ".layer1.lin.weight"], weights[p + ".layer1.lin.bias"], padding=2)
    return u.view(1, 128, 4, L, L).max(dim=2)[0]


def block_finish(weights, k, u, x):
    """network.py:32 + 37-82 + 99-101: InstanceNorm, scSE gates, residual add."""
    p = f"resnet.{k}"
    y = F.instance_norm(u, weight=weights[p + ".layer1.norm.weight"],
                        bias=weights[p + ".layer1.norm.bias"], eps=1e-5)
    # cSE: per-channel gate from the spatial mean of y
    m = y.mean(dim=(2, 3))
    gate = torch.sigmoid(torch.relu(m @ weights[p + ".scSE.cSE.fc.0.weight"].t())
                         @ weights[p + ".scSE.cSE.fc.2.weight"].t())
    # sSE: per-pixel gate from a 1x1 conv over channels
    s = torch.sigmoid(F.conv2d(y, weights[p + ".scSE.sSE.conv.weight"],
                               weights[p + ".scSE.sSE.conv.bias"]))
    return y * gate.view(1, 128, 1, 1) + y * s + x


def head(weights, x):
    """network.py:207: 1x1 conv 128 -> 2."""
    return F.conv2d(x, weights["resnet.17.weight"], weights["resnet.17.bias"])


def canonical_signs(v):
    """Flip each column so its largest-|.| entry (first index on ties) is > 0."""
    idx = v.abs().argmax(dim=-2, keepdim=True)
    s = torch.sign(torch.gather(v, -2, idx))
    s = torch.where(s == 0, torch.ones_like(s), s)
    return v * s


def mds_top8(M, eig_sign="canonical"):
    """network.py:247-250: ascending eigen-decomposition, eigenvalues clamped to
    >= 1e-8, V*sqrt(lambda), last 8 columns."""
    lam, vec = torch.linalg.eigh(M.float(), UPLO="U")
    if eig_sign == "canonical":
        vec = canonical_signs(vec)
    elif eig_sign != "lapack":
        raise ValueError(eig_sign)
    lam = torch.clamp(torch.relu(lam), min=1e-8)
    return (vec * lam.sqrt().unsqueeze(-2))[..., -8:]


def head_to_gram(y):
    """network.py:237-246: dm, conf, Gram matrix from the 2-channel head."""
    dm = y[:, 0]
    conf = y[:, 1].mean(dim=2)
    dm = torch.abs((dm + dm.transpose(1, 2)) / 2)
    L = dm.shape[-1]
    M = 0.5 * (dm[:, 0:1, :].expand(-1, L, -1) ** 2 + dm[:, :, 0:1].expand(-1, -1, L) ** 2
               - dm ** 2)
    return dm, conf, M


def coords_from_mds(weights, mat1d, mds):
    """network.py:251-255: bi-GRU over the sequence + linear -> CA (1, L, 3)."""
    emb = torch.cat((mat1d.t().unsqueeze(0), mds), dim=2)          # (1, L, 520)
    g = _gru(weights, "coord_gru", emb, 520, 256, 3, True, True)
    return g @ weights["coord_fc.weight"].t()


def pair_distances(ca):
    """network.py:272: clamp(sum sq, 1e-8).sqrt() -> diagonal 1e-4."""
    d = ca.unsqueeze(1) - ca.unsqueeze(0)
    return torch.clamp((d * d).sum(dim=2), min=1e-8).sqrt()


def refine_coords(coords, n_steps):
    """Steric/bond minimiser, network.py:106-137.  coords (L, 3)."""
    c = coords
    for _ in range(n_steps):
        d = c.unsqueeze(0) - c.unsqueeze(1)            # d[i, j] = c[j] - c[i]
        dist = d.norm(dim=2).clamp(min=0.01, max=10.0)
        unit = d / dist.unsqueeze(2)
        push = 100.0 * ((dist < 3.0).to(torch.float) * (3.0 - dist))
        acc = c * 0 + (push.unsqueeze(2) * unit).sum(dim=0)
        b = c[1:] - c[:-1]
        bl = b.norm(dim=1).clamp(min=0.1)
        pull = (100.0 * (bl - 3.78).clamp(max=3.0)).unsqueeze(1) * (b / bl.unsqueeze(1))
        acc[:-1] += pull
        acc[1:] -= pull
        c = c + acc.clamp(min=-100.0, max=100.0) * 0.001
    return c


def ca_to_backbone(ca):
    """N, CA, C, O, CB from a CA trace, network.py:141-177.  ca (1, L, 3) ->
    (1, 5L, 3) in atom order N, CA, C, O, CB per residue."""
    def unit(v):
        return F.normalize(v, dim=2)
    a0, a1, a2 = ca[:, 0:1], ca[:, 1:2], ca[:, 2:3]
    z0, z1, z2 = ca[:, -1:], ca[:, -2:-1], ca[:, -3:-2]
    nterm = a0 + 3.82 * unit(torch.cross(a0 - a1, a2 - a1, dim=2))
    cterm = z0 + 3.82 * unit(torch.cross(z0 - z1, z2 - z1, dim=2))
    ext = torch.cat((nterm, ca, cterm), dim=1)
    prev = ext[:, :-2] - ext[:, 1:-1]
    nxt = ext[:, 2:] - ext[:, 1:-1]
    mid = (ext[:, 1:] + ext[:, :-1]) / 2
    nrm = unit(torch.cross(prev, nxt, dim=2))
    n_at = mid[:, :-1] - prev / 8 + nrm / 4
    c_sh = mid[:, :-1] + prev / 8 - nrm / 2
    o_sh = mid[:, :-1] - nrm * 1.8
    c_last = mid[:, -1:] - nxt[:, -1:] / 8 + nrm[:, -1:] / 2
    o_last = mid[:, -1:] + nrm[:, -1:] * 2.0
    c_at = torch.cat((c_sh[:, 1:], c_last), dim=1)
    o_at = torch.cat((o_sh[:, 1:], o_last), dim=1)
    vn = ca - n_at
    vc = ca - c_at
    cr = torch.cross(vn, vc, dim=2)
    vb = vn + vc
    ang = pi / 2 - asin(1 / sqrt(3))
    sx = (1.5 * cos(ang) / vb.norm(dim=2)).unsqueeze(2)
    sy = (1.5 * sin(ang) / cr.norm(dim=2)).unsqueeze(2)
    cb = ca + sx * vb + sy * cr
    out = torch.stack((n_at, ca, c_at, o_at, cb), dim=2)
    return out.reshape(ca.size(0), 5 * ca.size(1), 3)


def forward(weights, alnmat, f2d, seed_dmap, nloops, refine_steps,
            eig_sign="canonical", capture=None):
    """GRUResNet.forward, network.py:218-314.  Returns (coords (1,5L,3),
    conf (1,L))."""
    L = alnmat.shape[1]
    mat1d = sequence_trunk(weights, alnmat)                       # (512, L)
    pair = (mat1d.unsqueeze(1) * mat1d.unsqueeze(2)).unsqueeze(0)  # [c, i, j] = m[c,j] m[c,i]
    static = torch.cat((pair, f2d.permute(2, 0, 1).unsqueeze(0)), dim=1)   # 954 channels
    if capture is not None:
        capture["mat1d"] = mat1d

    def one_pass(dmap, tag):
        y = pair_trunk(weights, torch.cat((static, dmap.view(1, 1, L, L)), dim=1),
                       capture, tag)
        dm, conf, M = head_to_gram(y)
        mds = mds_top8(M, eig_sign)
        ca = coords_from_mds(weights, mat1d, mds)
        if capture is not None:
            capture[tag + "dm"] = dm[0]
            capture[tag + "conf"] = conf[0]
            capture[tag + "M"] = M[0]
            capture[tag + "mds"] = mds[0]
            capture[tag + "ca"] = ca[0]
        return conf, ca

    conf, ca = one_pass(seed_dmap, "p0.")
    if refine_steps > 0:
        ca = refine_coords(ca[0], refine_steps).unsqueeze(0)
    best_conf, best_ca = conf, ca
    for it in range(nloops):
        conf, ca = one_pass(pair_distances(ca[0]), f"p{it + 1}.")
        if conf.mean() > best_conf.mean():            # strict, network.py:302
            best_conf, best_ca = conf, ca
    if refine_steps > 0:
        best_ca = refine_coords(best_ca[0], refine_steps).unsqueeze(0)
    if capture is not None:
        capture["best_ca"] = best_ca[0]
    return ca_to_backbone(best_ca), torch.sigmoid(best_conf)


# ----------------------------------------------------------------------------
# top level                           (predict.py:74-158, 160-208)
# ----------------------------------------------------------------------------
def read_template_ca(path):
    """predict.py:106-117: CA atoms of ATOM records, fixed columns."""
    xyz = []
    with open(path, "r") as fh:
        for line in fh:
            if line[:4] == "ATOM" and line[12:16] == " CA ":
                xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return torch.tensor(np.asarray(xyz, dtype=np.float32)).reshape(-1, 3)


def load_weights(weights_file):
    sd = torch.load(weights_file, map_location="cpu")
    return {k: v.float() for k, v in sd.items()}


def predict(alnmat, weights, template_ca=None, iterations=10, minsteps=100,
            eig_sign="canonical", capture=None):
    """Everything after parsing: codes + weights -> (coords (L,5,3), confs (L,))."""
    with torch.no_grad():
        n, L = alnmat.shape
        w = reweight(alnmat, 0.8)
        if capture is not None:
            capture["w"] = torch.from_numpy(w)
        if n > 1:
            f2d = fast_dca(alnmat, w, capture=capture).float()
        else:
            f2d = torch.zeros((L, L, 442))
        if template_ca is not None:
            t = template_ca
            seed = (t.unsqueeze(0) - t.unsqueeze(1)).pow(2).sum(dim=2).sqrt()   # predict.py:143
        else:
            seed = torch.zeros((L, L)) - 1                                       # predict.py:145
        coords, conf = forward(weights, alnmat, f2d, seed, max(iterations, 0),
                               max(minsteps, 0), eig_sign, capture)
        return coords.view(-1, L, 5, 3)[0], conf[0]


def aln_to_coords(input_file, template=None, iterations=10, minsteps=100,
                  weights_file=None, return_alnmat=False, eig_sign="canonical",
                  capture=None):
    if weights_file is None:
        raise FileNotFoundError("the oracle has no bundled weights; pass weights_file")
    weights = load_weights(weights_file)
    alnmat = encode_aln(read_aln(input_file))
    tca = read_template_ca(template) if template is not None else None
    coords, confs = predict(alnmat, weights, tca, iterations, minsteps, eig_sign, capture)
    if return_alnmat:
        return coords, confs, alnmat
    return coords, confs


def pdb_text(coords, confs, alnmat):
    """predict.py:189-208: the exact text the CLI prints."""
    out = ["REMARK  CONF:  " + repr(confs.mean().item())]
    names = (" N  ", " CA ", " C  ", " O  ", " CB ")
    serial = 1
    for ri in range(coords.size(0)):
        code = int(alnmat[0, ri])
        for ai, an in enumerate(names):
            if code == 7 and ai == 4:
                continue                                  # no CB on glycine
            if code > 19:
                raise KeyError(code)
            out.append("ATOM   %4d %s %s  %4d    %8.3f%8.3f%8.3f  1.00%6.2f" % (
                serial, an, _RESNAMES[code], ri + 1, coords[ri, ai, 0].item(),
                coords[ri, ai, 1].item(), coords[ri, ai, 2].item(), confs[ri]))
            serial += 1
    out.append("END")
    return "\n".join(out) + "\n"
