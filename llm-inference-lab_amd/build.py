"""In-tree build of the gfx950 C-ABI library (libspecdec_hip.so).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the
CPU-only container as the "does it build" check and the resulting .so travels to
the GPU box with the repo snapshot. Objects are cached by source hash under
csrc/.obj/ so an unchanged file is not recompiled.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libspecdec_hip.so"
INCLUDE = PKG_DIR.parent / "include"

ARCH = "gfx950"
COMMON_FLAGS = [
    "-O3",
    "-std=c++17",
    "-fPIC",
    f"--offload-arch={ARCH}",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
    f"-I{INCLUDE}",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _sources() -> list[Path]:
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(_flags_for(src)).encode())
    h.update(src.read_bytes())
    for hdr in sorted(list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))):
        h.update(hdr.read_bytes())
    return h.hexdigest()[:16]


# Per-file optimisation level. csrc/persist.hip at -Os: the persistent kernel is one wave per SIMD through ~80 KB of code, every
# instruction is issued at full latency and the hot paths of a layer barely fit the instruction cache two CUs share; same-box A/B of
# eight flag sets (profiles/round4_persist_ab.md): -O3 595.6 us per 1B forward, -O2 583.8, -Os 582.3, -align-all-nofallthru-blocks=6
# 591.4, -align-all-blocks=4 605, the max-ilp / max-memory-clause scheduling strategies 595-596.
FILE_FLAGS = {"persist.hip": ["-Os"]}


def _flags_for(src: Path) -> list[str]:
    extra = FILE_FLAGS.get(src.name, [])
    return [f for f in COMMON_FLAGS if not (extra and f == "-O3")] + extra


def _compile_one(hipcc: str, src: Path, obj_dir: Path, verbose: bool) -> Path:
    obj = obj_dir / f"{src.stem}.{_digest(src)}.o"
    if obj.exists():
        return obj
    for stale in obj_dir.glob(f"{src.stem}.*.o"):
        stale.unlink()
    cmd = [hipcc, *_flags_for(src), "-x", "hip", "-c", str(src), "-o", str(obj)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every csrc/*.hip|*.cpp for gfx950 and link libspecdec_hip.so."""
    hipcc = _hipcc()
    obj_dir = CSRC / ".obj"
    obj_dir.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    if force:
        for o in obj_dir.glob("*.o"):
            o.unlink()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(hipcc, s, obj_dir, verbose), srcs))
    stamp = LIB_DIR / ".link_stamp"
    want = ",".join(o.name for o in objs)
    if LIB_PATH.exists() and stamp.exists() and stamp.read_text() == want and not force:
        return LIB_PATH
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
           *map(str, objs), "-ldl", "-o", str(LIB_PATH)]   # -ldl: csrc/prefill_gemm.hip opens rocBLAS at first use
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    stamp.write_text(want)
    return LIB_PATH


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(f"built {path}")
