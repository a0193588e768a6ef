"""dmpfold2_amd: the DMPfold2 alignment -> backbone path on AMD MI355X.

Public surface = the reference's: ``aln_to_coords`` and ``run_dmpfold``
(reference dmpfold/__init__.py:1).  Everything numeric runs in libdmpfold_hip.so.
"""
from .predict import aln_to_coords, run_dmpfold, Engine, get_engine  # noqa: F401

__all__ = ["aln_to_coords", "run_dmpfold", "Engine", "get_engine"]
