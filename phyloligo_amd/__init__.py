"""phyloligo_amd -- MI355X-native all-by-all contig distance path of PhylOligo.

Host side in Python (mirroring /root/reference/phylopackage/bin/phyloligo.py's dispatcher
interface), compute in hand-written HIP kernels behind the C ABI of include/phyloligo_amd.h.
"""
from ._lib import PhyloligoError, LIB_PATH, METRICS, STRANDS  # noqa: F401
from .api import Context, device_count, fasta_index, normalise_pattern, pattern_info, write_mat_text  # noqa: F401

__version__ = "0.1"
