"""Experiment switches and diagnostics of libkemr.so (include/kemr_debug.h) for tools/ and tests/.

Process-wide and not thread-safe: nothing in the product path (engine, evaluators, retriever, dist, bench.py's timed legs)
calls into this module.  A switch changes which kernel or route runs, never what comes out (tests/ hold every route to the same
results); what does change numerics -- the residual fusion -- is a per-model option (ClipEngine.set_residual_fusion).
"""
from __future__ import annotations

import contextlib
import ctypes as C

from . import _lib

KEYS = ("gemm_variant", "gemm_flags", "gemm_order", "gemm_grid", "gemm_conc", "gemm_kl", "attn_v", "attn_xcd", "attn_waves", "sim_lists", "ln_nt")


def ab_variants() -> bool:
    """True when libkemr.so was built with the experiment kernels (``build.py --ab-variants``); the product library refuses the
    switch values that select them."""
    return get("ab_variants") == 1


def set(key: str, value: int) -> None:      # noqa: A001 (module-level verb of a tiny module)
    _lib.check(_lib.lib().kemr_debug_set(key.encode(), int(value)), f"debug_set({key})")


def get(key: str) -> int:
    v = C.c_int(0)
    _lib.check(_lib.lib().kemr_debug_get(key.encode(), C.byref(v)), f"debug_get({key})")
    return v.value


@contextlib.contextmanager
def override(**kv):
    """with debug.override(gemm_variant=2): ...  -- sets the switches, restores what they held before."""
    old = {k: get(k) for k in kv}
    try:
        for k, v in kv.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)


def set_gemm_variant(packed: int) -> None:
    """The packed form tools/ grew up with: bits 0-7 variant, 8-15 flags (both always written), 16-19 tile order + 1,
    20-21 concurrent epilogues + 1, 24-27 attention waves + 1 (written only when non-zero)."""
    set("gemm_variant", packed & 0xff)
    set("gemm_flags", (packed >> 8) & 0xff)
    if (packed >> 16) & 0xf:
        set("gemm_order", ((packed >> 16) & 0xf) - 1)
    if (packed >> 20) & 0x3:
        set("gemm_conc", ((packed >> 20) & 0x3) - 1)
    if (packed >> 24) & 0xf:
        set("attn_waves", ((packed >> 24) & 0xf) - 1)


def sim_lists(workspace, nq: int, ng: int, kdim: int, k: int):
    """flag / longest list / capacity / chunks / sampled rows / records per query of the last candidate-list search that used
    `workspace` (a torch uint8 tensor, e.g. the one engine.sim_topk(..., return_workspace=True) hands back)."""
    out = (C.c_int32 * 6)()
    _lib.check(_lib.lib().kemr_debug_sim_lists(C.c_void_p(workspace.data_ptr()), nq, ng, kdim, k, C.cast(out, C.c_void_p)),
               "debug_sim_lists")
    return list(out)


def gemm_stamps(n_words: int):
    buf = (C.c_uint * n_words)()
    _lib.check(_lib.lib().kemr_debug_gemm_stamps(buf, n_words), "debug_gemm_stamps")
    return buf
