"""CPU: the C-ABI library loads, exports exactly what include/kemr.h declares, and its host-side argument
checking works without a GPU (no compute call is made here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from knowledge_enhanced_multimodal_retrieval_amd import _lib, debug

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="kemr.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kemr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    """include/kemr.h (the product ABI) and include/kemr_debug.h (switches of tools/ and tests/) against the ctypes tables and
    against what the library really exports: nothing undeclared, nothing missing, no debug entry point in the product header."""
    names, dbg = header_functions(), header_functions("kemr_debug.h")
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names, "python binding table and include/kemr.h disagree"
    assert sorted(_lib.DEBUG_SIGNATURES) == dbg, "python debug table and include/kemr_debug.h disagree"
    assert not set(names) & set(dbg)
    assert not [n for n in names if "debug" in n or n.startswith("kemr_set_")], "debug state belongs in kemr_debug.h"
    lib = _lib.lib()
    for n in names + dbg:
        assert hasattr(lib, n), n
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (kemr_[a-z0-9_]+)", out))
    assert exported == set(names) | set(dbg), "library exports differ from the headers"
    assert lib.kemr_abi_version() == _lib.ABI_VERSION == 4
    text = open(os.path.join(ROOT, "include", "kemr.h")).read()
    assert re.search(r"#define\s+KEMR_ABI_VERSION\s+4\b", text)


def test_model_options_and_debug_switches_are_separate():
    """The residual fusion is a per-model option (one model's setting does not leak into another); the process-wide switches take
    one key each and refuse unknown keys / out-of-range values.  Host-side only."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    lib = _lib.lib()
    h1, h2 = C.c_void_p(), C.c_void_p()
    assert lib.kemr_model_create(C.byref(_cfg()), C.byref(h1)) == 0 and lib.kemr_model_create(C.byref(_cfg()), C.byref(h2)) == 0
    v = C.c_int(-1)
    assert lib.kemr_model_get_option(h1, b"residual_fusion", C.byref(v)) == 0 and v.value == 1
    assert lib.kemr_model_set_option(h1, b"residual_fusion", 0) == 0
    assert lib.kemr_model_get_option(h1, b"residual_fusion", C.byref(v)) == 0 and v.value == 0
    assert lib.kemr_model_get_option(h2, b"residual_fusion", C.byref(v)) == 0 and v.value == 1
    assert lib.kemr_model_set_option(h1, b"residual_fusion", 3) == -1 and lib.kemr_model_set_option(h1, b"nope", 1) == -1
    assert lib.kemr_model_get_option(h1, b"last_block_pooled_row", C.byref(v)) == 0 and v.value == 1
    assert lib.kemr_model_set_option(h1, b"last_block_pooled_row", 0) == 0 and lib.kemr_model_set_option(h1, b"last_block_pooled_row", 2) == -1
    assert lib.kemr_model_get_option(h2, b"last_block_pooled_row", C.byref(v)) == 0 and v.value == 1
    assert lib.kemr_model_get_option(h1, b"residual_stream_24bit", C.byref(v)) == 0 and v.value == 1          # the default since round 4
    assert lib.kemr_model_set_option(h1, b"residual_stream_24bit", 0) == 0 and lib.kemr_model_set_option(h1, b"residual_stream_24bit", 2) == -1
    assert lib.kemr_model_get_option(h1, b"residual_stream_24bit", C.byref(v)) == 0 and v.value == 0
    assert lib.kemr_model_get_option(h2, b"residual_stream_24bit", C.byref(v)) == 0 and v.value == 1
    lib.kemr_model_destroy(h1)
    lib.kemr_model_destroy(h2)
    before = {k: debug.get(k) for k in debug.KEYS}
    assert before["gemm_variant"] == 0 and before["gemm_kl"] == 0 and before["sim_lists"] == 1 and before["ln_nt"] == 3
    with debug.override(gemm_order=0, gemm_variant=2):
        assert debug.get("gemm_order") == 0 and debug.get("gemm_variant") == 2 and debug.get("gemm_conc") == before["gemm_conc"]
        debug.set_gemm_variant(7 | (1 << 16))                     # the packed form touches variant and flags (and what else is non-zero) only ...
        assert debug.get("gemm_variant") == 7 and debug.get("gemm_flags") == 0 and debug.get("gemm_order") == 0
        debug.set_gemm_variant(0)
    assert {k: debug.get(k) for k in debug.KEYS} == before
    with pytest.raises(RuntimeError, match="unknown key"):
        debug.set("gemm_nope", 1)
    with pytest.raises(RuntimeError, match="not in"):
        debug.set("gemm_variant", 12)
    with pytest.raises(RuntimeError, match="read-only"):
        debug.set("ab_variants", 1)


def test_product_library_refuses_the_experiment_kernels():
    """the default libkemr.so holds one attention kernel per shape and one K loop; the switch values that select
    the experiments of rounds 1-3 exist only in a `build.py --ab-variants` library and are refused here."""
    if debug.ab_variants():
        pytest.skip("this libkemr.so was built with --ab-variants")
    for key, value in (("attn_v", 1), ("attn_v", 4), ("attn_waves", 6), ("gemm_kl", 1), ("gemm_variant", 3), ("gemm_variant", 4),
                       ("gemm_variant", 9), ("gemm_flags", 64), ("ln_nt", 0)):
        with pytest.raises(RuntimeError, match="A/B experiment kernel"):
            debug.set(key, value)
        assert debug.get(key) == (3 if key == "ln_nt" else 0)
    for key, value in (("gemm_variant", 1), ("gemm_variant", 2), ("gemm_variant", 7), ("gemm_variant", 8), ("attn_xcd", 0), ("sim_lists", 3)):
        with debug.override(**{key: value}):
            assert debug.get(key) == value
    syms = open(_lib.LIB_PATH, "rb").read()                     # host stubs AND the gfx950 code objects carry the kernels' names
    for name in ("attention32_kernel", "attention_w8_kernel", "attention_s2_kernel", "attention_pd_kernel", "gemm256s_bf16_nt_kernel",
                 "gemm256p", "gemm256q", "gemm256w", "gemm256r"):
        assert name.encode() not in syms, name


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "kemr.h")).read()
    assert "torch" not in text.replace("PyTorch", "").replace("torch /", "").lower() or "at::" not in text
    assert "#include <hip" not in text          # plain C: streams cross as void*


def _cfg(**kw):
    base = dict(embed_dim=128, image_size=32, patch=8, v_width=256, v_layers=2, t_width=256, t_layers=2, vocab=512, ctx=16)
    base.update(kw)
    return _lib.KemrCfg(**base)


def test_model_create_validates_config():
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.kemr_model_create(C.byref(_cfg()), C.byref(h)) == 0
    n = lib.kemr_model_num_tensors(h)
    names = [lib.kemr_model_tensor_name(h, i).decode() for i in range(n)]
    assert n == 13 + 2 * 2 * 12 and "visual.transformer.resblocks.1.mlp.c_proj.weight" in names
    # 51 tokens -> 256 rows x 18 B x width (x f32, h, 2 deltas, 4W big) + the last block's pooled-row area: 3 items -> 256 compact rows
    # x 22 B x width (x f32, h, q, attention out, 2 deltas, 4W hidden) + 256 B of row indices
    assert lib.kemr_workspace_bytes(h, _lib.TOWER_VISION, 3) == 256 * 256 * 18 + 256 * 256 * 22 + 256
    assert lib.kemr_text_packed_workspace_bytes(h, 40, 3) == 256 * 256 * 18 + 256 * 256 * 22 + 256 + 256
    assert lib.kemr_text_packed_workspace_bytes(h, 2, 3) == 0          # fewer rows than texts
    lib.kemr_model_destroy(h)
    for bad in (dict(v_width=200), dict(image_size=30), dict(embed_dim=0), dict(ctx=400), dict(patch=0)):
        h2 = C.c_void_p()
        assert lib.kemr_model_create(C.byref(_cfg(**bad)), C.byref(h2)) == -1, bad
        assert b"cfg" in lib.kemr_last_error()


def test_load_tensor_is_strict_before_any_gpu_work():
    import numpy as np
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.kemr_model_create(C.byref(_cfg()), C.byref(h)) == 0
    x = np.zeros((256, 128), np.float32)
    shape = (C.c_int64 * 2)(256, 128)
    assert lib.kemr_model_load_tensor(h, b"visual.proj", x.ctypes.data_as(C.c_void_p), _lib.KEMR_F32, shape, 2) == 0
    assert lib.kemr_model_load_tensor(h, b"visual.nope", x.ctypes.data_as(C.c_void_p), _lib.KEMR_F32, shape, 2) == -1
    assert b"unexpected key" in lib.kemr_last_error()
    bad = (C.c_int64 * 2)(128, 256)
    assert lib.kemr_model_load_tensor(h, b"visual.proj", x.ctypes.data_as(C.c_void_p), _lib.KEMR_F32, bad, 2) == -1
    assert b"size mismatch" in lib.kemr_last_error()
    assert lib.kemr_model_load_tensor(h, b"visual.proj", x.ctypes.data_as(C.c_void_p), _lib.KEMR_BF16, shape, 2) == -1
    assert lib.kemr_model_load_tensor(h, b"logit_scale", x.ctypes.data_as(C.c_void_p), _lib.KEMR_F32, shape, 0) == 0
    assert lib.kemr_model_finalize(h, _lib.PREC_BF16) == -2          # missing keys: refused before touching the GPU
    assert b"missing key" in lib.kemr_last_error()
    assert lib.kemr_encode_image(h, None, 1, None, 0, None, 0, None) == -1
    assert lib.kemr_encode_text_packed(h, None, None, 1, 1, None, 0, None, 0, None) == -1
    lib.kemr_model_destroy(h)


def test_host_side_shape_checks():
    lib = _lib.lib()
    assert lib.kemr_panel_kdim(768, 1, 1) == 768 and lib.kemr_panel_kdim(768, 2, 3) == 4608
    assert lib.kemr_panel_kdim(100, 1, 3) == 384 and lib.kemr_panel_kdim(768, 1, 2) == -1
    assert lib.kemr_sim_workspace_bytes(1024, 43000, 768, 10) > 0 and lib.kemr_sim_workspace_bytes(0, 5, 768, 10) == 0
    assert lib.kemr_sim_topk(None, 1, None, 1, 64, 0, 10, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.kemr_op_gemm(None, None, None, None, 1, 128, 64, 0, None) == -1
    assert lib.kemr_profile_end(None, None, 0) == -2
    with pytest.raises(RuntimeError, match="libkemr"):
        _lib.check(-1, "x")


def test_host_e4m3_conversion_matches_torch():
    """Weights are quantised to OCP e4m3 on the host at finalize (KEMR_PREC_FP8): round to nearest even, subnormals,
    saturation at +-448 -- against torch's float8_e4m3fn cast on every value it represents and on ties between them."""
    import ctypes as C
    import numpy as np
    import torch
    lib = _lib.lib()
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(torch.float8_e4m3fn).float()
    finite = vals[~torch.isnan(vals)]
    grid = torch.sort(finite).values
    mids = (grid[1:] + grid[:-1]) / 2                       # exact ties
    g = torch.Generator().manual_seed(0)
    rnd = (torch.rand(20000, generator=g) * 2 - 1) * 440 * torch.rand(20000, generator=g) ** 6
    x = torch.cat([finite, mids, rnd, torch.tensor([0.0, -0.0, 1e-9, 0.0009765, 0.00098, 447.9])]).contiguous()
    out = np.empty(x.numel(), dtype=np.uint8)
    assert lib.kemr_op_e4m3_host(C.c_void_p(x.data_ptr()), C.c_void_p(out.ctypes.data), x.numel()) == 0
    ref = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    same_value = torch.from_numpy(out).view(torch.float8_e4m3fn).float() == torch.from_numpy(ref).view(torch.float8_e4m3fn).float()
    assert bool(same_value.all()), x[~same_value][:10]
    big = torch.tensor([448.0, 449.0, 1e6, -1e6], dtype=torch.float32)      # saturating, unlike torch's cast
    out2 = np.empty(4, dtype=np.uint8)
    assert lib.kemr_op_e4m3_host(C.c_void_p(big.data_ptr()), C.c_void_p(out2.ctypes.data), 4) == 0
    assert out2.tolist() == [0x7e, 0x7e, 0x7e, 0xfe]


def test_graft_entry_build_runs():
    """The driver's build check (`__graft_entry__.build()`): compiles (or finds up to date) the library, loads it and checks the ABI
    version against the binding table -- it must not lag behind a version bump."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()
