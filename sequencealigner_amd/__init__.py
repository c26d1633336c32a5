"""sequencealigner_amd -- MI355X (gfx950) DP alignment engine behind the reference aligner's
hot path (LCS / Needleman-Wunsch / Smith-Waterman fill + traceback).

The product is the C-ABI library `libseqalign_hip.so` (include/seqalign_hip.h) and the
C++ host CLI `aligner`; this package is a thin ctypes view of the same C-ABI for tests,
bench.py and Python callers.  It fails loudly when the HIP library is missing.
"""
from .engine import (  # noqa: F401
    AlignResult,
    Engine,
    SeqAlignError,
    globalalign,
    lcs,
    lib_path,
    load_library,
    localalign,
)
