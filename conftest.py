"""Repo-level pytest bootstrap: put the package directory on sys.path.

`llm-inference-lab_amd/` is not an importable name (hyphen), so its contents are
reached as top-level packages: `specdec_hip`, `src.*`, and the aliases `kernels`
/ `specdec` that the reference's callers use.
"""

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "llm-inference-lab_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
