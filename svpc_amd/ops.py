"""placeholder — replaced below by the HIP-backed implementation."""
from .ops_common import *  # noqa
from .ops_common import SeqInfo, Idx, FIdx  # noqa
