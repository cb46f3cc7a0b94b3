"""GPU: each hand-written kernel through the C ABI against a plain fp32 torch statement of the same op
(the floating-point kernels keep a torch fp32 reference; the end-to-end path is checked against oracle/)."""
import numpy as np
import pytest
import torch

from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine

pytestmark = pytest.mark.gpu


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _skip_unless_built(variant):
    """gemm_variant 3 (the staggered 256x256 kernel) exists only in a `build.py --ab-variants` library."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    if variant == 3 and not debug.ab_variants():
        pytest.skip("gemm_variant 3 is an A/B kernel (build.py --ab-variants)")


@pytest.mark.parametrize("m,n,k", [(257, 256, 256), (300, 768, 256), (1000, 1024, 1024), (514, 256, 1024), (63 * 257, 1024, 1024)])
@pytest.mark.parametrize("epi", [_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16, _lib.EPI_BIAS_RESID_F32])
@pytest.mark.parametrize("variant", [1, 2, 3, 7])
def test_gemm_epilogues(device, m, n, k, epi, variant):
    _skip_unless_built(variant)
    engine.set_gemm_variant(variant)      # 1: 128x128 tiles, 2: 256x256 tiles (every N here is a multiple of 256)
    try:
        # the persistent kernel (variant 7, more than 512 rows) stores whole 256-row tiles: C's pad rows are scratch
        _gemm_epilogue_case(device, m, n, k, epi, pad_rows_kept=not (variant == 7 and m > 512))
    finally:
        engine.set_gemm_variant(0)


def _gemm_epilogue_case(device, m, n, k, epi, pad_rows_kept=True):
    g = torch.Generator().manual_seed(m * 7 + n + k + epi)
    m_alloc = (m + 255) // 256 * 256
    a = torch.randn(m_alloc, k, generator=g)
    w = torch.randn(n, k, generator=g) * (k ** -0.5)
    bias = torch.randn(n, generator=g)
    a_bf, w_bf = a.to(torch.bfloat16), w.to(torch.bfloat16)
    ref = a_bf.float()[:m] @ w_bf.float().T + bias
    c0 = None
    if epi == _lib.EPI_BIAS_QGELU_BF16:
        ref = ref * torch.sigmoid(1.702 * ref)
    if epi == _lib.EPI_BIAS_RESID_F32:
        c0 = torch.randn(m_alloc, n, generator=g)
        ref = ref + c0[:m]
    c_dev = None if c0 is None else c0.clone().to(device)
    out = engine.op_gemm(a_bf.to(device), w_bf.to(device), bias.to(device), m, epi, c=c_dev)
    torch.cuda.synchronize()
    got = out.float().cpu()
    tol = 2e-2 if epi != _lib.EPI_BIAS_RESID_F32 else 2e-4    # bf16 output rounding vs fp32 output
    err = (got[:m] - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < tol
    if epi == _lib.EPI_BIAS_RESID_F32 and pad_rows_kept:      # rows >= m are never written (tile kernels with row masks)
        assert torch.equal(got[m:], c0[m:])
    # bf16 epilogues: rows in [m, m_alloc) are scratch (the persistent kernel stores whole tiles)


@pytest.mark.parametrize("variant", [1, 2, 3, 7])
def test_gemm_identity_asymmetric(device, variant):
    _skip_unless_built(variant)
    """A = I against an asymmetric W catches transposed / permuted fragment maps (cdna guide section 3)."""
    k = n = 512
    m_alloc = 512
    a = torch.eye(m_alloc, k)
    w = (torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 251) - 125.0        # exact in bf16
    engine.set_gemm_variant(variant)
    try:
        out = engine.op_gemm(a.to(torch.bfloat16).to(device), w.to(torch.bfloat16).to(device), None, 512, _lib.EPI_BIAS_BF16)
    finally:
        engine.set_gemm_variant(0)
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu(), w.T.contiguous().to(torch.bfloat16).float())


@pytest.mark.parametrize("variant", [7])
def test_gemm_persistent_many_tiles_per_cu(device, variant):
    """Persistent kernel: > 256 tiles so that every workgroup walks several tiles (hand-over path), ragged M, with and
    without bias, both bf16 epilogues."""
    g = torch.Generator().manual_seed(11)
    m, n, k = 5 * 256 * 13 + 77, 1024, 256           # 66 row tiles x 4 = 264 tiles
    m_alloc = (m + 255) // 256 * 256
    a = torch.randn(m_alloc, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g)
    ref = a.float()[:m] @ w.float().T
    engine.set_gemm_variant(variant)
    try:
        for epi, b in ((_lib.EPI_BIAS_BF16, None), (_lib.EPI_BIAS_BF16, bias), (_lib.EPI_BIAS_QGELU_BF16, bias)):
            r = ref + (b if b is not None else 0)
            if epi == _lib.EPI_BIAS_QGELU_BF16:
                r = r * torch.sigmoid(1.702 * r)
            for _ in range(2):                          # twice: catches state leaking between launches
                out = engine.op_gemm(a.to(device), w.to(device), None if b is None else b.to(device), m, epi)
                torch.cuda.synchronize()
                got = out.float().cpu()[:m]
                assert float(((got - r).abs() / (r.abs() + 1.0)).max()) < 2e-2
    finally:
        engine.set_gemm_variant(0)


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("k", [64, 128, 192, 4096])
def test_gemm256_short_and_long_k(device, k, variant):
    _skip_unless_built(variant)
    """Pipeline prologue / tail of the 256x256 kernel: 1, 2, 3 and 64 K-tiles (K = 64 falls back to 128x128)."""
    g = torch.Generator().manual_seed(k)
    m, n = 700, 512
    a = torch.randn(768, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16)
    ref = a.float()[:m] @ w.float().T
    engine.set_gemm_variant(variant)
    try:
        out = engine.op_gemm(a.to(device), w.to(device), None, m, _lib.EPI_BIAS_RESID_F32,
                             c=torch.zeros(768, n, device=device))
    finally:
        engine.set_gemm_variant(0)
    torch.cuda.synchronize()
    assert float((out.cpu()[:m] - ref).abs().max()) < 2e-4 * (1 + float(ref.abs().max()))


@pytest.mark.parametrize("variant", [7])
@pytest.mark.parametrize("k", [128, 192, 320, 4096])
def test_gemm_cross_tile_pipeline_k_tiles(device, k, variant):
    """Persistent kernels whose K-tile pipeline runs across tile switches: 2, 3, 5 (odd: the LDS buffer parity flips
    between tiles) and 64 K-tiles, > 256 tiles, ragged M."""
    g = torch.Generator().manual_seed(k)
    m, n = 256 * 70 + 19, 1024                       # 71 x 4 = 284 tiles
    m_alloc = (m + 255) // 256 * 256
    a = torch.randn(m_alloc, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g)
    ref = a.float()[:m] @ w.float().T + bias
    engine.set_gemm_variant(variant)
    try:
        for _ in range(2):
            out = engine.op_gemm(a.to(device), w.to(device), bias.to(device), m, _lib.EPI_BIAS_BF16)
            torch.cuda.synchronize()
            got = out.float().cpu()[:m]
            assert float(((got - ref).abs() / (ref.abs() + 1.0)).max()) < 2e-2
    finally:
        engine.set_gemm_variant(0)


@pytest.mark.parametrize("m,n,k", [(77, 2304, 768), (77, 768, 3072), (1, 768, 768), (257, 1024, 1024), (300, 3072, 768), (512, 256, 64), (129, 512, 192)])
@pytest.mark.parametrize("epi", [_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16])
def test_gemm_skinny(device, m, n, k, epi):
    """Online-query shapes (M = 77 for one text, 257 for one image): split-K kernel, forced and as the automatic choice."""
    for variant in (8, 0):
        engine.set_gemm_variant(variant)
        try:
            _gemm_epilogue_case(device, m, n, k, epi)
        finally:
            engine.set_gemm_variant(0)


@pytest.mark.parametrize("m,n,k", [(64 * 257, 1024, 1024), (64 * 257, 3072, 1024), (128 * 257, 1024, 512), (256 * 70 + 100, 1024, 256), (200 * 257, 1024, 256)])
def test_gemm_ragged_tail_split(device, m, n, k):
    """Automatic choice only: a ragged last row tile that would add a round to the persistent kernel is computed by
    the skinny kernel (two launches); same results, pad rows aside."""
    for epi in (_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16):
        _gemm_epilogue_case(device, m, n, k, epi)


def _fp8(x):
    return x.to(torch.float8_e4m3fn)


@pytest.mark.parametrize("m,n,k", [(300, 768, 256), (1000, 1024, 1024), (256 * 70 + 19, 1024, 384), (63 * 257, 3072, 1024)])
@pytest.mark.parametrize("epi", [_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16])
def test_gemm_fp8(device, m, n, k, epi):
    """fp8 e4m3 operands through the block-scaled MFMA (K = 128 per instruction) against fp32 math on the same bytes."""
    g = torch.Generator().manual_seed(m + n + k + epi)
    m_alloc = (m + 255) // 256 * 256
    a = _fp8(torch.randn(m_alloc, k, generator=g) * 1.5)
    w = _fp8(torch.randn(n, k, generator=g) * 20)
    wscale = (torch.rand(n, generator=g) + 0.5) * (k ** -0.5) / 20
    bias = torch.randn(n, generator=g)
    ref = (a.float()[:m] @ w.float().T) * wscale + bias
    if epi == _lib.EPI_BIAS_QGELU_BF16:
        ref = ref * torch.sigmoid(1.702 * ref)
    for _ in range(2):
        out = engine.op_gemm_fp8(a.to(device), w.to(device), wscale.to(device), bias.to(device), m, epi)
        torch.cuda.synchronize()
        got = out.float().cpu()[:m]
        assert float(((got - ref).abs() / (ref.abs() + 1.0)).max()) < 2e-2


def test_gemm_fp8_identity_asymmetric(device):
    """A = I against an asymmetric integer W: catches a wrong k order inside the 32-byte fragments."""
    k = n = 512
    a = _fp8(torch.eye(512, k))
    w = ((torch.arange(n * k, dtype=torch.float32).reshape(n, k) * 7) % 17) - 8.0            # exact in e4m3
    out = engine.op_gemm_fp8(a.to(device), _fp8(w).to(device), torch.ones(n, device=device), None, 512, _lib.EPI_BIAS_BF16)
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu(), w.T.contiguous())


def test_layernorm_fp8_output(device):
    width, rows = 1024, 301
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, width, generator=g) * 2
    d1 = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    gamma[5] = 400.0                                                    # some outputs beyond +-448: must saturate, not NaN
    gd, bd = gamma.to(device), beta.to(device)
    for delta, wb in ((None, True), (d1, False), (d1, True)):
        xd = x.clone().to(device)
        y = engine.op_layernorm_rows(xd, None if delta is None else delta.to(device), gd, bd, writeback=wb, out_fp8=True)
        torch.cuda.synchronize()
        xs = x if delta is None else x + delta.float()
        ref = torch.nn.functional.layer_norm(xs, (width,), gamma, beta, 1e-5).clamp(-448, 448)
        got = y.float().cpu()
        assert not torch.isnan(got).any()
        # e4m3 has 3 mantissa bits: half an ulp is 1/16 relative for normals, 2^-10 absolute below 2^-6
        assert float(((got - ref).abs() - ref.abs() / 16).max()) < 2e-3


@pytest.mark.parametrize("m,n,k", [(65535, 1024, 1024), (19635, 768, 3072), (1000, 256, 128), (65535, 1024, 4096)])
def test_gemm_residual_add_epilogue(device, m, n, k):
    """EPI_BIAS_RESADD_BF16: the persistent kernel reads the bf16 residual tile its stores overwrite (a rolling window of loads
    with counted vmcnt) and writes bf16(bf16(A.W^T + bias) + x): bit-identical to the store-only epilogue followed by a bf16
    add, on shapes with one to many tiles per CU and a ragged last row tile."""
    g = torch.Generator(device=device).manual_seed(m + n + k)
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=device).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=device) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=device)
    x = (torch.randn(ma, n, generator=g, device=device) * 3).to(torch.bfloat16)
    engine.set_gemm_variant(7)                               # the reference through the same kernel (small shapes go elsewhere)
    try:
        delta = engine.op_gemm(a, w, bias, m, _lib.EPI_BIAS_BF16)
    finally:
        engine.set_gemm_variant(0)
    want = (delta[:m].float() + x[:m].float()).to(torch.bfloat16)
    got = x.clone()
    for _ in range(2):                                       # second round on top of the first: in-place accumulation
        engine.op_gemm(a, w, bias, m, _lib.EPI_BIAS_RESADD_BF16, c=got)
        assert torch.equal(got[:m], want)
        want = (delta[:m].float() + want.float()).to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        engine.op_gemm(a[:256], w, bias, 100, _lib.EPI_BIAS_RESADD_BF16, c=x[:256].clone())      # too few rows for the persistent kernel


@pytest.mark.parametrize("kl", [1, 0])
@pytest.mark.parametrize("m,n,k", [(65535, 1024, 4096), (65535, 1024, 1024), (65527, 768, 3072), (1000, 256, 128)])
def test_gemm_residual_epilogues_against_fp32_torch(device, m, n, k, kl):
    """Both in-place residual epilogues of the persistent kernel at the towers' own shapes (fc2 of ViT-L/14 at B = 255: 65 535 x
    1 024 x 4 096, out-proj, the 851-text fc2) against an INDEPENDENT fp32 statement of the op (torch fp32 matmul of the same bf16
    operands): EPI_BIAS_RESID_F32 x += A.W^T + b exactly in fp32 (no rounding but the accumulation order), EPI_BIAS_RESADD_BF16
    within the two bf16 roundings it documents.  Both K loops (debug switch gemm_kl: four / eight barrier intervals per K-tile)."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    if kl and not debug.ab_variants():
        pytest.skip("the long-interval K loop is an A/B kernel (build.py --ab-variants)")
    g = torch.Generator(device=device).manual_seed(m + n + k)
    ma = (m + 255) // 256 * 256
    a = torch.randn(ma, k, generator=g, device=device).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=device) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, generator=g, device=device)
    x0 = torch.randn(ma, n, generator=g, device=device) * 3
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        upd = a[:m].float() @ w.float().t() + bias                   # fp32 reference of the update
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    with debug.override(gemm_variant=7, gemm_kl=kl):
        x = x0.clone()
        engine.op_gemm(a, w, bias, m, _lib.EPI_BIAS_RESID_F32, c=x)
        err = (x[:m] - (x0[:m] + upd)).abs().max().item()
        assert err < 2e-5 * k ** 0.5 + 1e-5, (err, "resid f32")      # fp32 accumulation over k terms of size ~k^-1/2 each
        engine.op_gemm(a, w, bias, m, _lib.EPI_BIAS_RESID_F32, c=x)  # in place, second round
        err = (x[:m] - (x0[:m] + 2 * upd)).abs().max().item()
        assert err < 4e-5 * k ** 0.5 + 2e-5, (err, "resid f32 x2")
        xb = x0.to(torch.bfloat16)
        want = (upd.to(torch.bfloat16).float() + xb[:m].float()).to(torch.bfloat16)
        got = xb.clone()
        engine.op_gemm(a, w, bias, m, _lib.EPI_BIAS_RESADD_BF16, c=got)
        # one bf16 ulp where the fp32 accumulation order moved a value across a rounding boundary, nothing beyond
        d = (got[:m].float() - want.float()).abs()
        assert float((d / (want.float().abs() + 1.0)).max()) < 1.6e-2
        assert float((d > 0).float().mean()) < 0.02


def test_gemm_rejects_bad_shapes(device):
    a = torch.zeros(256, 96, dtype=torch.bfloat16, device=device)
    w = torch.zeros(128, 96, dtype=torch.bfloat16, device=device)
    with pytest.raises(RuntimeError, match="K % 64"):
        engine.op_gemm(a, w, None, 10, _lib.EPI_BIAS_BF16)
    # the staggered 256x256 kernel and the earlier persistent generations (variants 3, 4, 5, 6, 9) are A/B kernels for tools/: the
    # product library refuses to select them (tests/test_abi.py::test_product_library_refuses_the_experiment_kernels)
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    if not debug.ab_variants():
        for v in (3, 4, 5, 6, 9):
            with pytest.raises(RuntimeError, match="A/B experiment kernel"):
                engine.set_gemm_variant(v)
            assert debug.get("gemm_variant") == 0


@pytest.mark.parametrize("width", [256, 512, 768, 1024])
@pytest.mark.parametrize("out_bf16", [True, False])
def test_layernorm(device, width, out_bf16):
    g = torch.Generator().manual_seed(width)
    rows = 517
    x = torch.randn(rows, width, generator=g) * 3 + 0.7
    gamma = 1 + 0.1 * torch.randn(width, generator=g)
    beta = 0.1 * torch.randn(width, generator=g)
    ref = torch.nn.functional.layer_norm(x, (width,), gamma, beta, 1e-5)
    got = engine.op_layernorm(x.to(device), gamma.to(device), beta.to(device), out_bf16).float().cpu()
    tol = 2e-2 if out_bf16 else 2e-5
    assert float((got - ref).abs().max()) < tol
    if not out_bf16:
        xd = x.to(device)     # in place (ln_pre): y aliases x
        L = _lib.lib()
        import ctypes as C
        gd, bd = gamma.to(device), beta.to(device)      # keep the device copies alive across the call
        _lib.check(L.kemr_op_layernorm(C.c_void_p(xd.data_ptr()), C.c_void_p(gd.data_ptr()),
                                       C.c_void_p(bd.data_ptr()), C.c_void_p(xd.data_ptr()), rows, width,
                                       _lib.KEMR_F32, None))
        torch.cuda.synchronize()
        assert float((xd.cpu() - ref).abs().max()) < tol


@pytest.mark.parametrize("width", [256, 768, 1024])
def test_layernorm_with_fused_residual(device, width):
    g = torch.Generator().manual_seed(width + 1)
    rows = 301
    x = torch.randn(rows, width, generator=g) * 2
    delta = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    xs = x + delta.float()
    ref = torch.nn.functional.layer_norm(xs, (width,), gamma, beta, 1e-5)
    xd = x.clone().to(device)
    y = engine.op_layernorm_resid(xd, delta.to(device), gamma.to(device), beta.to(device))
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), xs)                                  # residual stream updated in place, exactly
    assert float((y.float().cpu() - ref).abs().max()) < 2e-2


@pytest.mark.parametrize("width", [256, 768, 1024])
def test_layernorm_bf16_rows(device, width):
    """KEMR_PREC_BF16_RES16: the residual stream is stored as bf16; the add and the statistics stay fp32."""
    g = torch.Generator().manual_seed(width + 2)
    rows = 301
    x = (torch.randn(rows, width, generator=g) * 2).to(torch.bfloat16)
    delta = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    gd, bd = gamma.to(device), beta.to(device)
    xs = x.float() + delta.float()
    xd = x.clone().to(device)
    y = engine.op_layernorm_rows(xd, delta.to(device), gd, bd)
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), xs.to(torch.bfloat16))              # written back with one RNE rounding
    assert float((y.float().cpu() - torch.nn.functional.layer_norm(xs, (width,), gamma, beta, 1e-5)).abs().max()) < 2e-2
    # without a delta x is only read; fp32 output of bf16 rows; bf16 rows normalised in place (ln_pre form)
    y32 = engine.op_layernorm_rows(xd, None, gd, bd, out_bf16=False)
    ref = torch.nn.functional.layer_norm(xd.float().cpu(), (width,), gamma, beta, 1e-5)
    assert float((y32.cpu() - ref).abs().max()) < 2e-5
    L = _lib.lib()
    import ctypes as C
    _lib.check(L.kemr_op_layernorm_rows(C.c_void_p(xd.data_ptr()), _lib.KEMR_BF16, None, None, 0, C.c_void_p(gd.data_ptr()),
                                        C.c_void_p(bd.data_ptr()), C.c_void_p(xd.data_ptr()), rows, width, _lib.KEMR_BF16, None))
    torch.cuda.synchronize()
    assert float((xd.float().cpu() - ref).abs().max()) < 4e-2


@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_layernorm_two_pending_deltas(device, xdt):
    """ln_2 form: LN(x + d1) without touching x; next ln_1 form: x += d1 + d2 written back once -- equal to two updates."""
    width, rows = 1024, 301
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(rows, width, generator=g) * 2).to(xdt)
    d1 = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    d2 = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    gd, bd = gamma.to(device), beta.to(device)
    xd = x.clone().to(device)
    y2 = engine.op_layernorm_rows(xd, d1.to(device), gd, bd, writeback=False)
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x)                                                   # untouched
    ref2 = torch.nn.functional.layer_norm(x.float() + d1.float(), (width,), gamma, beta, 1e-5)
    assert float((y2.float().cpu() - ref2).abs().max()) < 2e-2
    y1 = engine.op_layernorm_rows(xd, d1.to(device), gd, bd, delta2=d2.to(device))
    torch.cuda.synchronize()
    xs = (x.float() + d1.float()) + d2.float()                                       # same association as two updates
    assert torch.equal(xd.cpu(), xs.to(xdt))
    assert float((y1.float().cpu() - torch.nn.functional.layer_norm(xs, (width,), gamma, beta, 1e-5)).abs().max()) < 2e-2
    with pytest.raises(RuntimeError, match="second delta"):
        engine.op_layernorm_rows(xd, d1.to(device), gd, bd, delta2=d2.to(device), writeback=False)


def _attention_ref(qkv, batch, t, width, causal):
    heads = width // 64
    q, k, v = qkv.float().view(batch, t, 3, heads, 64).permute(2, 0, 3, 1, 4)      # [B,H,T,64]; q already scaled
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((t, t), float("-inf")).triu_(1)
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(batch * t, width)


@pytest.mark.parametrize("t,causal", [(17, False), (16, True), (50, False), (77, True), (100, True), (197, False), (257, False)])
def test_attention(device, t, causal):
    g = torch.Generator().manual_seed(t)
    batch, width = 3, 256
    qkv = torch.randn(batch * t, 3 * width, generator=g)
    qkv[:, :width] *= 0.125 * 2.0           # pre-scaled queries, logits of a few units
    qkv_bf = qkv.to(torch.bfloat16)
    ref = _attention_ref(qkv_bf, batch, t, width, causal)
    got = engine.op_attention(qkv_bf.to(device), batch, t, width, causal).float().cpu()
    assert float((got - ref).abs().max()) < 3e-2          # P and the output are rounded to bf16
    assert float((got - ref).abs().mean()) < 3e-3


def test_attention_softmax_spike(device):
    """One dominant key per query (large logits): exercises the max-subtraction path."""
    t, batch, width = 257, 1, 256
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(batch * t, 3 * width, generator=g) * 0.1
    qkv[:, :width] = 0.0
    qkv[:, 0] = 4.0                          # q[:, d0] = 4 for head 0
    qkv[:, width] = torch.linspace(-8, 8, t)  # k[:, d0] spread: logits up to 32
    qkv_bf = qkv.to(torch.bfloat16)
    ref = _attention_ref(qkv_bf, batch, t, width, False)
    got = engine.op_attention(qkv_bf.to(device), batch, t, width, False).float().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("batch,width", [(3, 256), (9, 1024)])
def test_attention_257_both_kernels_and_exact_structure(device, batch, width):
    """T = 257 (the vision towers' shape): the product kernel (16-query tiles) and -- in a `build.py --ab-variants` library only --
    round 3's experiments (debug switch attn_v = 1..4).  All against the fp32 torch statement; then EXACT structure with
    integer-valued data that bf16 and fp32 hold exactly -- a one-hot softmax (one key 40 logits ahead per query) must return that
    key's V row, for every query of every tile including the lone 257th, which catches a wrong key <-> k-slot permutation or V
    transposition outright (the other 256 keys weigh e^-40: they move a zero entry by 1e-17 and nothing else)."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    t = 257
    g = torch.Generator().manual_seed(batch + width)
    qkv = torch.randn(batch * t, 3 * width, generator=g)
    qkv[:, :width] *= 0.25
    qkv_bf = qkv.to(torch.bfloat16)
    ref = _attention_ref(qkv_bf, batch, t, width, False)
    outs = {}
    variants = (0, 1, 2, 3, 4, 5) if debug.ab_variants() else (0,)
    for v in variants:
        with debug.override(attn_v=v):
            outs[v] = engine.op_attention(qkv_bf.to(device), batch, t, width, False).float().cpu()
        assert float((outs[v] - ref).abs().max()) < 3e-2 and float((outs[v] - ref).abs().mean()) < 3e-3, v
    if 1 in outs:
        assert float((outs[0] - outs[1]).abs().max()) < 2e-2
    for v in variants[:3:2]:                                    # grid order instead of the images dealt to the XCDs (the default): another
        with debug.override(attn_v=v, attn_xcd=0):        # workgroup numbering, the same results
            assert torch.equal(engine.op_attention(qkv_bf.to(device), batch, t, width, False).float().cpu(), outs[v]), v
    # one-hot: query i of head hd looks for key perm[i]: q = 64 * e_(c(i)), k_j = e_(c'(j)) built so that q_i . k_j = 64 iff j == perm[i]
    heads = width // 64
    perm = torch.randperm(t, generator=g)
    x = torch.zeros(batch * t, 3 * width)
    vals = torch.randint(-64, 65, (batch * t, width), generator=g).float()            # exact in bf16
    x[:, 2 * width:] = vals
    code = torch.arange(t)                                                             # 257 codes as 2 base-17 digits -> two one-hot groups of 17 dims
    d0, d1 = code % 17, code // 17
    for hd in range(heads):
        kk = torch.zeros(t, 64)
        kk[torch.arange(t), d0] = 1.0
        kk[torch.arange(t), 17 + d1] = 1.0
        qq = torch.zeros(t, 64)
        qq[torch.arange(t), d0[perm]] = 40.0
        qq[torch.arange(t), 17 + d1[perm]] = 40.0          # q_i . k_j = 80 when j == perm[i], 40 or 0 otherwise: exp(-40) of the rest vanishes under fp32 sums
        for b in range(batch):
            x[b * t:(b + 1) * t, hd * 64:(hd + 1) * 64] = qq
            x[b * t:(b + 1) * t, width + hd * 64:width + (hd + 1) * 64] = kk
    xb = x.to(torch.bfloat16)
    want = torch.empty(batch * t, width)
    for b in range(batch):
        want[b * t:(b + 1) * t] = vals[b * t:(b + 1) * t][perm]
    for v in variants:
        with debug.override(attn_v=v):
            got = engine.op_attention(xb.to(device), batch, t, width, False).float().cpu()
        assert float((got - want).abs().max()) < 1e-6, (v, float((got - want).abs().max()))


@pytest.mark.parametrize("spike_key", [3, 150, 200, 256])
def test_attention_257_online_softmax_rescale_is_exercised(device, spike_key):
    """attn_v = 2 takes a tile's keys in two halves (tiles 0-9 and 10-17) with a running maximum: one key far ahead of the rest,
    in the FIRST half (the second half must not disturb it), in the SECOND half (o and l of the first half are rescaled by
    exp(m_old - m_new) ~ e^-30), at the seam, and as the lone 257th key -- against the fp32 torch statement over the FULL tensor, and
    against the single-pass kernel (cdna guide rule 26: a rare rescale branch needs an input that forces it)."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    if not debug.ab_variants():
        pytest.skip("attn_v = 2..4 are A/B kernels (build.py --ab-variants); the product kernel has no rescale branch")
    t, batch, width = 257, 2, 256
    g = torch.Generator().manual_seed(spike_key)
    qkv = torch.randn(batch * t, 3 * width, generator=g) * 0.3
    qkv[:, 0] = 6.0                                       # q[:, d0] = 6 for head 0, every query
    for b in range(batch):
        qkv[b * t + spike_key, width] = 5.0               # k[spike, d0] = 5: logit 30 ahead of the rest
        qkv[b * t + (spike_key + 100) % t, width] = 2.5   # a runner-up in the other half: logit 15
    qkv_bf = qkv.to(torch.bfloat16)
    ref = _attention_ref(qkv_bf, batch, t, width, False)
    outs = {}
    for v in (0, 2, 3, 4):
        with debug.override(attn_v=v):
            outs[v] = engine.op_attention(qkv_bf.to(device), batch, t, width, False).float().cpu()
        assert torch.isfinite(outs[v]).all()
        assert float((outs[v] - ref).abs().max()) < 3e-2, (v, float((outs[v] - ref).abs().max()))
    assert float((outs[0] - outs[2]).abs().max()) < 2e-2
    assert float((outs[2] - outs[3]).abs().max()) < 4e-3   # the persistent form: the same arithmetic per head up to hipcc's contraction



@pytest.mark.gpu
def test_attention_257_persistent_kernel_walks_several_items(device):
    """attn_v = 3: one workgroup per CU walks a list of (image, head) items, the K / V rows of the next one arriving by LDS-DMA in
    the other buffer while the tiles of the current one run.  100 images x 8 heads = 800 items over 256 workgroups: up to four
    items per workgroup (both buffers reused), XCDs with 13 and with 12 images (100 = 12 * 8 + 4) -- against the one-item-per-
    workgroup kernel with the same arithmetic (bit-identical) and against the fp32 torch statement on a sample of the images."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    if not debug.ab_variants():
        pytest.skip("attn_v = 3 is an A/B kernel (build.py --ab-variants)")
    t, batch, width = 257, 100, 512
    g = torch.Generator().manual_seed(11)
    qkv_bf = (torch.randn(batch * t, 3 * width, generator=g) * 0.7).to(torch.bfloat16)
    x = qkv_bf.to(device)
    with debug.override(attn_v=2):
        one_item = engine.op_attention(x, batch, t, width, False)
    outs = {}
    for waves in (0, 8):                                  # 16 waves, one tile each (the default of attn_v = 3) / 8 waves, two tiles each
        with debug.override(attn_v=3, attn_waves=waves):
            outs[waves] = engine.op_attention(x, batch, t, width, False)
            again = engine.op_attention(x, batch, t, width, False)
        torch.cuda.synchronize()
        assert torch.equal(again, outs[waves])            # nothing left over in LDS, no dependence on the previous launch
        # the same source arithmetic per head as the one-item kernel, compiled in another context (hipcc contracts a*b+c differently:
        # 1 element of 13 M off by one bf16 ulp on the round-3 device): equal up to rounding, not bit for bit
        d = (outs[waves].float() - one_item.float()).abs()
        assert float(d.max()) < 4e-3 and int((d > 0).sum()) < 1e-5 * d.numel(), (waves, float(d.max()), int((d > 0).sum()))
    pick = [0, 7, 8, 63, 95, 96, 99]                      # first / last image of an XCD's list, the ragged tail
    rows = torch.cat([torch.arange(b * t, (b + 1) * t) for b in pick])
    ref = _attention_ref(qkv_bf[rows], len(pick), t, width, False)
    for waves in (0, 8):
        got = outs[waves].float().cpu()[rows]
        assert float((got - ref).abs().max()) < 3e-2 and float((got - ref).abs().mean()) < 3e-3, waves


@pytest.mark.gpu
@pytest.mark.parametrize("t,causal,batch,width", [(77, True, 19, 192), (257, False, 19, 192), (50, False, 9, 128), (257, False, 1, 1024)])
def test_attention_images_dealt_to_the_xcds(device, t, causal, batch, width):
    """The default numbering of the attention workgroups (XCD x takes the images = x mod 8; a grid of heads x batch rounded up to 8
    whose surplus workgroups leave at once) against grid order: bit-identical, for batches that are not multiples of 8, head counts
    that are not powers of two and a single image; and against the fp32 torch statement."""
    from knowledge_enhanced_multimodal_retrieval_amd import debug
    g = torch.Generator().manual_seed(batch * 1000 + t)
    qkv = torch.randn(batch * t, 3 * width, generator=g)
    qkv[:, :width] *= 0.25
    qkv_bf = qkv.to(torch.bfloat16)
    x = qkv_bf.to(device)
    assert debug.get("attn_xcd") == 1
    dealt = engine.op_attention(x, batch, t, width, causal)
    with debug.override(attn_xcd=0):
        plain = engine.op_attention(x, batch, t, width, causal)
    assert torch.equal(dealt, plain)
    ref = _attention_ref(qkv_bf, batch, t, width, causal)
    assert float((dealt.float().cpu() - ref).abs().max()) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize("width", [768, 1024])
def test_layernorm_24bit_rows(device, width):
    """The LayerNorm forms on 24-bit-float rows (csrc/common.h f24_t: a row = W bf16 upper halves + W third bytes; model option
    residual_stream_24bit): the packing helpers round-trip (to nearest on 24 bits, relative error <= 2^-16), ln_2 form LN(x + d1)
    leaves x untouched, ln_1 form writes x += d1 + d2 back as the 24-bit rounding of the fp32 sum, outputs against torch's fp32
    LayerNorm of the exactly unpacked rows."""
    rows = 301
    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, width, generator=g) * 3
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -123456.789])
    x24 = engine.pack_f24_rows(x)
    xq = engine.unpack_f24_rows(x24, width)
    assert x24.shape == (rows, 3 * width) and x24.dtype == torch.uint8
    assert float(((xq - x).abs() / x.abs().clamp_min(1e-20)).max()) <= 2.0 ** -16 + 1e-9 and torch.equal(engine.pack_f24_rows(xq), x24)
    d1 = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    d2 = (torch.randn(rows, width, generator=g) * 0.5).to(torch.bfloat16)
    gamma, beta = 1 + 0.1 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    gd, bd = gamma.to(device), beta.to(device)
    xd = x24.clone().to(device)

    def close(y, ref):                                    # bf16 output: 2^-8 relative (the outlier row normalises to |y| ~ 30) + 2e-2 absolute
        return bool(((y.float().cpu() - ref).abs() <= 2e-2 + 2.0 ** -8 * ref.abs()).all())

    y0 = engine.op_layernorm_rows_f24(xd, width, None, gd, bd, writeback=False)
    assert close(y0, torch.nn.functional.layer_norm(xq, (width,), gamma, beta, 1e-5))
    y2 = engine.op_layernorm_rows_f24(xd, width, d1.to(device), gd, bd, writeback=False)
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x24)                                                 # untouched
    assert close(y2, torch.nn.functional.layer_norm(xq + d1.float(), (width,), gamma, beta, 1e-5))
    y1 = engine.op_layernorm_rows_f24(xd, width, d1.to(device), gd, bd, delta2=d2.to(device))
    torch.cuda.synchronize()
    xs = (xq + d1.float()) + d2.float()
    assert torch.equal(xd.cpu(), engine.pack_f24_rows(xs))                            # the fp32 sum, rounded once to 24 bits
    assert close(y1, torch.nn.functional.layer_norm(xs, (width,), gamma, beta, 1e-5))
