"""manta_amd -- MI355X-native engine for MANTA's NMF multiplicative-update hot path.

Only the solver path is here: ``run_nmf`` with the reference signature, the
``Engine`` handle over libmnmf.so (hand-written HIP for gfx950), a document-sharded driver and a
synthetic TF-IDF generator.  Importing the package does not load the HIP library; using the engine
without it (or without a GPU) raises.
"""
from .nmf import find_max_rank, install, run_nmf, run_nmf_sweep, sweeping_run_nmf  # noqa: F401

__all__ = ["run_nmf", "run_nmf_sweep", "sweeping_run_nmf", "install", "find_max_rank"]
__version__ = "0.1.0"
