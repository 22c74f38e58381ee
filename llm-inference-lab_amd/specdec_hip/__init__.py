"""specdec_hip — Python binding of the gfx950 speculative-decoding C-ABI library.

PyTorch-ROCm is used for device memory, streams and torch.distributed only; the
compute is in csrc/*.hip behind include/specdec_hip.h.
"""

from . import _abi  # noqa: F401
from ._abi import HipLibraryError, load  # noqa: F401
