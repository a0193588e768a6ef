"""Drop-in name: ``from dmpfold import aln_to_coords`` resolves to the MI355X engine."""
from dmpfold2_amd import aln_to_coords, run_dmpfold  # noqa: F401
