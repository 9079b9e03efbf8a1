"""Build recipe for libconformer_hip.so (gfx950 only, in-tree so the .so travels with the repo snapshot).

    python -m conformer_amd.build            # incremental
    python -m conformer_amd.build --force

hipcc cross-compiles without a GPU.  The library links libamdhip64 by SONAME only (no rpath to
/opt/rocm): inside a PyTorch-ROCm process it binds to the HIP runtime torch already loaded, so there is
exactly one runtime per process.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "lib", "libconformer_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "conformer_hip.h"))
    return max(os.path.getmtime(d) for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_dep()):
        return obj
    cmd = [HIPCC, *CFLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "--no-hip-rt" if False else "",
               "-o", LIB, *objs]
        cmd = [c for c in cmd if c]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"[conformer_amd.build] {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, {len(objs)} objects)")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
