import os
import sys

import numpy as np
import pytest

# Engines made by the GPU suite start in the FAST mode (option precision 0: the arithmetic most stage-level tests were
# written against - activation scales, the f16 range fault, its bf16 fallback); every golden test selects all three
# settings explicitly, and the tests of the defaults (precision 2 for a context, an Engine and the drop-in entry points)
# remove this variable.
os.environ.setdefault("DMPFOLD_PRECISION", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # the oracle is test infrastructure only
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: g[k] for k in g.files}


def golden_rows(g):
    return bytes(g["aln_text"]).decode("latin-1").split("\n")


@pytest.fixture(scope="session")
def synth_sd():
    from dmpfold2_amd import synth
    return synth.synth_weights(0, coord_scale=5.0)


@pytest.fixture(scope="session")
def weights_file(tmp_path_factory, synth_sd):
    from dmpfold2_amd import synth
    p = tmp_path_factory.mktemp("w") / "synth_seed0.pt"
    synth.save_state_dict(str(p), synth_sd)
    return str(p)


@pytest.fixture(scope="session")
def oracle_weights(synth_sd):
    import torch
    return {k: torch.from_numpy(np.array(v)) for k, v in synth_sd.items()}


def ca_rmsd(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum(-1).mean()))
