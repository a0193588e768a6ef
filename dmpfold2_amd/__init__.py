"""dmpfold2_amd: the DMPfold2 alignment -> backbone path on AMD MI355X.

Public surface = the reference's: ``aln_to_coords`` and ``run_dmpfold``
(reference dmpfold/__init__.py:1).  Everything numeric runs in libdmpfold_hip.so.
"""
import os as _os

# several HIP streams per GPU (Pipeline): the runtime's default of 4 hardware queues serialises the
# fifth stream; effective only if the HIP runtime has not been initialised yet
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .predict import aln_to_coords, run_dmpfold, Engine, get_engine  # noqa: F401

__all__ = ["aln_to_coords", "run_dmpfold", "Engine", "get_engine"]
