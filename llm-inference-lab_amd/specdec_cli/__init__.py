"""Top-level alias: `import specdec_cli` is `src.specdec_cli` (the reference installs it as the `specdec` console script)."""

import importlib
import sys

_real = importlib.import_module("src.specdec_cli")
sys.modules["specdec_cli"] = _real
