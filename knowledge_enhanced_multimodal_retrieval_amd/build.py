"""Build libkemr.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m knowledge_enhanced_multimodal_retrieval_amd.build [--force]

No torch extension machinery: the library has a plain C ABI (include/kemr.h) and is loaded with ctypes,
so a direct ``hipcc -shared -fPIC`` is all there is.  Objects are rebuilt only when a source or header is
newer than the object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libkemr.so")
SOURCES = ["api.hip", "gemm.hip", "gemm256.hip", "gemm256u.hip", "gemm_skinny.hip", "layernorm.hip", "attention.hip", "embed.hip", "sim.hip", "rank.hip", "preprocess.hip"]
# Experiment kernels kept for A/B timing from tools/ only -- earlier persistent-GEMM generations (gemm_variant 4, 5, 6, 9), the
# attention variants of round 3 (attn_v 1..4), and, inside the product sources behind -DKEMR_AB_VARIANTS, the staggered 256x256
# GEMM, the long-interval K loop, the stamped instantiations: built only with `python -m ...build --ab-variants` (or
# KEMR_BUILD_AB=1), never into the product library, whose kemr_debug_set refuses the values that would select them.
AB_SOURCES = ["gemm256p.hip", "gemm256q.hip", "gemm256w.hip", "gemm256r.hip", "attention_ab.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "kemr.h")]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, extra) -> str:
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    if force or _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
               "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", obj] + list(extra)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
        if res.stderr.strip():
            sys.stderr.write(res.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags=(), ab_variants: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libkemr.so; returns its path."""
    ab_variants = ab_variants or os.environ.get("KEMR_BUILD_AB", "") == "1"
    sources = SOURCES + (AB_SOURCES if ab_variants else [])
    stamp = os.path.join(CSRC, ".ab_variants")
    if ab_variants != os.path.exists(stamp):         # the flag changes gemm.hip's dispatch: rebuild when it flips
        force = True
        (open(stamp, "w").close() if ab_variants else os.remove(stamp))
    if ab_variants:
        extra_flags = list(extra_flags) + ["-DKEMR_AB_VARIANTS"]
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra_flags), sources))
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB_PATH] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, ab_variants="--ab-variants" in sys.argv)
