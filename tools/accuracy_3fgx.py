#!/usr/bin/env python
"""Accuracy check against the only ground-truth structure in the reference tree:
predict the reference's example alignment PF10963.aln with the TRAINED weights and compare the CA
trace with chain A of example/3FGX.pdb (TM-score, RMSD after superposition).

    python tools/accuracy_3fgx.py [-w weights.pt] [-n 10] [-m 100]

Needs the two FINAL_fullmap_e2e_model_part*.pt files (absent from the reference tree, see
.MISSING_LARGE_BLOBS) in dmpfold2_amd/trained_model/ or a -w file.  The native CA trace and sequence
come from tests/golden/kat_refine_backbone.npz (data captured from the example PDB file).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AA3 = "ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split()
AA1 = "ARNDCQEGHILKMFPSTWYV"


def needleman_wunsch(a, b, match=2, mismatch=-1, gap=-2):
    """Global alignment of two strings; returns [(i, j)] of aligned (non-gap) positions."""
    n, m = len(a), len(b)
    S = np.zeros((n + 1, m + 1), dtype=np.int32)
    S[:, 0] = gap * np.arange(n + 1)
    S[0, :] = gap * np.arange(m + 1)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            S[i, j] = max(S[i - 1, j - 1] + (match if a[i - 1] == b[j - 1] else mismatch),
                          S[i - 1, j] + gap, S[i, j - 1] + gap)
    pairs, i, j = [], n, m
    while i > 0 and j > 0:
        if S[i, j] == S[i - 1, j - 1] + (match if a[i - 1] == b[j - 1] else mismatch):
            pairs.append((i - 1, j - 1))
            i, j = i - 1, j - 1
        elif S[i, j] == S[i - 1, j] + gap:
            i -= 1
        else:
            j -= 1
    return pairs[::-1]


def kabsch(P, Q):
    """Rotation R and translation t minimising |R P + t - Q| (rows are points)."""
    pc, qc = P.mean(0), Q.mean(0)
    H = (P - pc).T @ (Q - qc)
    U, _, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1.0, 1.0, d]) @ U.T
    return R, qc - R @ pc


def tm_score(P, Q, l_norm):
    """TM-score of the paired traces P -> Q normalised by l_norm residues: maximum over superpositions
    started from fragments of the pairing and refined on the pairs closer than d0 (the search of the
    TM-score program, reduced to a handful of seeds)."""
    d0 = max(1.24 * (l_norm - 15) ** (1.0 / 3.0) - 1.8, 0.5) if l_norm > 15 else 0.5
    n = len(P)
    best = 0.0
    for frag in sorted({n, max(n // 2, 4), max(n // 4, 4)}, reverse=True):
        for start in range(0, n - frag + 1, max(frag // 2, 1)):
            idx = np.arange(start, start + frag)
            for _ in range(20):
                R, t = kabsch(P[idx], Q[idx])
                d = np.linalg.norm(P @ R.T + t - Q, axis=1)
                best = max(best, float((1.0 / (1.0 + (d / d0) ** 2)).sum() / l_norm))
                new = np.nonzero(d < max(d0, 4.5))[0]
                if len(new) < 3 or (len(new) == len(idx) and (new == idx).all()):
                    break
                idx = new
    return best


def evaluate(aln_path, native_npz, weights_file=None, iterations=10, minsteps=100, device="cuda:0"):
    from dmpfold2_amd import aln_to_coords
    from dmpfold2_amd.predict import read_aln
    coords, confs = aln_to_coords(aln_path, device=device, iterations=iterations, minsteps=minsteps,
                                  weights_file=weights_file)
    ca = coords[:, 1].cpu().numpy().astype(np.float64)
    nat = np.load(native_npz)
    native = nat["ca_in"].astype(np.float64)
    native_seq = bytes(nat["seq1"]).decode()
    query = read_aln(aln_path)[0]
    pairs = needleman_wunsch(query, native_seq)
    qi = np.array([i for i, _ in pairs])
    nj = np.array([j for _, j in pairs])
    P, Q = ca[qi], native[nj]
    R, t = kabsch(P, Q)
    rmsd = float(np.sqrt(((P @ R.T + t - Q) ** 2).sum(1).mean()))
    ident = float(np.mean([query[i] == native_seq[j] for i, j in pairs]))
    return {"aligned_pairs": len(pairs), "sequence_identity": ident, "ca_rmsd_A": rmsd,
            "tm_score": tm_score(P, Q, len(native)), "tm_score_by_query_length": tm_score(P, Q, len(query)),
            "mean_conf": float(confs.mean()), "query_length": len(query), "native_length": len(native)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-w", "--model_weights", default=None)
    ap.add_argument("-n", "--iterations", type=int, default=10)
    ap.add_argument("-m", "--minsteps", type=int, default=100)
    ap.add_argument("-d", "--device", default="cuda:0")
    a = ap.parse_args()
    res = evaluate(os.path.join(ROOT, "tests", "golden", "PF10963.aln"),
                   os.path.join(ROOT, "tests", "golden", "kat_refine_backbone.npz"),
                   a.model_weights, a.iterations, a.minsteps, a.device)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
