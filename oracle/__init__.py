"""CPU oracle for the DP hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package.  The product (sequencealigner_amd/) never does.
"""
from .pyoracle import (  # noqa: F401
    Oracle,
    OracleResult,
    build,
    ops_to_alignment,
    ops_to_path,
    ref_binary,
    run_reference,
)
