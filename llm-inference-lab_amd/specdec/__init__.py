"""Top-level alias: `import specdec` is `src.specdec` (the reference's internal callers and
its tests use both spellings)."""

import importlib
import sys

_real = importlib.import_module("src.specdec")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("src.specdec."):
        sys.modules["specdec." + _name[len("src.specdec."):]] = _mod
sys.modules["specdec"] = _real
