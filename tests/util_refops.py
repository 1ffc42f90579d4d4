"""Oracle operator namespace for the re-hosted model (moved to oracle/ref_ops.py so that bench.py's
cpu_baseline leg can time the whole CPU model with it); re-exported here for the tests."""
from oracle.ref_ops import RefOps  # noqa: F401
