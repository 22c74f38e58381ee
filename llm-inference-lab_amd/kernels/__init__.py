"""Top-level alias: `import kernels` is `src.kernels` (one module object, one registry).

The reference's internal callers import the top-level name (policies.py:17,
hf_wrappers.py:927, pipeline.py:454) while its tests import `src.kernels`; both
spellings must reach the same registry table.
"""

import importlib
import sys

_real = importlib.import_module("src.kernels")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("src.kernels."):
        sys.modules["kernels." + _name[len("src.kernels."):]] = _mod
sys.modules["kernels"] = _real
