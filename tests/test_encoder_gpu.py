"""GPU: the HIP encoder towers through the C ABI against the fp32 CPU oracle (oracle/clip_ref.py).
Parity bar (BASELINE.json north_star): cosine(build, fp32 oracle) >= 1 - 1e-3 per embedding."""
import os

import numpy as np
import pytest
import torch

from knowledge_enhanced_multimodal_retrieval_amd import engine
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS
from oracle import clip_ref

pytestmark = pytest.mark.gpu
COS_TOL = 1e-3


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)


PRECISIONS = ["bf16", "bf16-res16", "bf16-x24"]     # fp32 / bf16 / 24-bit-float residual stream (include/kemr.h kemr_precision, options)


_ORACLE_CACHE = {}


def _oracle(name, nimg, ntxt, outliers=False):
    """(state dict, pixels, ids, oracle image / text embeddings) of the small full-size cases, computed once per session: the fp32
    CPU oracle takes seconds per ViT-L/14 item and the precision-parametrised tests all compare against the same numbers."""
    key = (name, nimg, ntxt, outliers)
    if key not in _ORACLE_CACHE:
        oa = clip_ref.ARCHS[name]
        sd = clip_ref.random_state_dict(oa, seed=0, outliers=outliers)
        g = torch.Generator().manual_seed(1234)
        px = torch.randn(nimg, 3, oa["image_size"], oa["image_size"], generator=g)
        ids = clip_ref.synthetic_ids(oa, ntxt)
        _ORACLE_CACHE[key] = (sd, px, ids, clip_ref.encode_image(sd, oa, px), clip_ref.encode_text(sd, oa, ids))
    return _ORACLE_CACHE[key]


def _engine(name, device, seed=0, precision="bf16"):
    arch = ARCHS[name]
    sd = clip_ref.random_state_dict(clip_ref.ARCHS[name], seed=seed)
    eng = engine.ClipEngine(arch, device, precision=precision)
    eng.load_state_dict(sd)
    return arch, sd, eng


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name,nimg,ntxt", [("tiny", 9, 11), ("tiny-long", 5, 7)])
def test_tiny_archs_match_oracle(device, name, nimg, ntxt, precision):
    arch, sd, eng = _engine(name, device, precision=precision)
    oa = clip_ref.ARCHS[name]
    g = torch.Generator().manual_seed(1234)
    px = torch.randn(nimg, 3, arch.image_size, arch.image_size, generator=g)
    ids = clip_ref.synthetic_ids(oa, ntxt)
    ref_i, ref_t = clip_ref.encode_image(sd, oa, px), clip_ref.encode_text(sd, oa, ids)
    got_i = eng.encode_image(px.to(device)).cpu()
    got_t = eng.encode_text(ids.to(device)).cpu()
    ci, ct = _cos(got_i, ref_i), _cos(got_t, ref_t)
    print(f"{name} {precision}: image cos min {ci.min():.6f}, text cos min {ct.min():.6f}")
    assert float((1 - ci).max()) < COS_TOL and float((1 - ct).max()) < COS_TOL
    # un-normalised outputs keep their scale; normalised ones are unit length and equal the oracle's rule
    assert float((got_i.norm(dim=-1) / ref_i.norm(dim=-1) - 1).abs().max()) < 2e-2
    got_n = eng.encode_image(px.to(device), normalize=True).cpu()
    assert float((got_n.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((got_n - got_i / got_i.norm(dim=-1, keepdim=True)).abs().max()) < 1e-5


@pytest.mark.parametrize("precision", PRECISIONS)
def test_golden_hf_fixture(device, golden_dir, precision):
    """Same inputs as the committed HF-from-config vectors (tests/golden/clip_hf_tiny-long.npz)."""
    z = np.load(os.path.join(golden_dir, "clip_hf_tiny-long.npz"))
    arch, sd, eng = _engine("tiny-long", device, precision=precision)
    got_i = eng.encode_image(torch.from_numpy(z["pixels"]).to(device)).cpu()
    got_t = eng.encode_text(torch.from_numpy(z["ids"]).to(device)).cpu()
    assert float((1 - _cos(got_i, torch.from_numpy(z["image_features"]))).max()) < COS_TOL
    assert float((1 - _cos(got_t, torch.from_numpy(z["text_features"]))).max()) < COS_TOL


def test_batch_slicing_and_determinism(device):
    """Batches above the per-launch slice are processed in slices; results do not depend on the slicing."""
    arch, sd, eng = _engine("tiny", device)
    g = torch.Generator().manual_seed(7)
    px = torch.randn(engine.MAX_IMAGE_BATCH + 37, 3, arch.image_size, arch.image_size, generator=g).to(device)
    full = eng.encode_image(px)
    part = eng.encode_image(px[-37:])
    again = eng.encode_image(px)
    assert torch.equal(full, again)
    assert torch.equal(full[-37:], part)
    ids = clip_ref.synthetic_ids(clip_ref.ARCHS["tiny"], engine.MAX_TEXT_BATCH + 5).to(device)
    eng.pack_text = False                  # the full-context path slices by MAX_TEXT_BATCH texts (the packed one by token rows:
    t_full = eng.encode_text(ids)          # test_text_rows_behind_the_eot_are_not_needed)
    assert torch.equal(t_full[-5:], eng.encode_text(ids[-5:]))
    eng.pack_text = True
    assert eng.encode_image(px[:0]).shape == (0, arch.embed_dim)


def test_eot_pooling_uses_first_argmax(device):
    arch, sd, eng = _engine("tiny", device)
    oa = clip_ref.ARCHS["tiny"]
    ids = clip_ref.synthetic_ids(oa, 4)
    ids[0, 5] = oa["vocab"] - 1
    ids[0, 9] = oa["vocab"] - 1           # two EOTs: torch.argmax takes the first
    ref = clip_ref.encode_text(sd, oa, ids)
    got = eng.encode_text(ids.to(device)).cpu()
    assert float((1 - _cos(got, ref)).max()) < COS_TOL


def test_strict_load_errors(device):
    arch = ARCHS["tiny"]
    sd = clip_ref.random_state_dict(clip_ref.ARCHS["tiny"], seed=0)
    eng = engine.ClipEngine(arch, device)
    bad = dict(sd)
    del bad["visual.proj"]
    with pytest.raises(RuntimeError, match="missing key 'visual.proj'"):
        eng.load_state_dict(bad)
    eng2 = engine.ClipEngine(arch, device)
    with pytest.raises(RuntimeError, match="unexpected key"):
        eng2.load_state_dict({**sd, "visual.bogus": torch.zeros(3)})
    eng3 = engine.ClipEngine(arch, device)
    with pytest.raises(RuntimeError, match="size mismatch"):
        eng3.load_state_dict({**sd, "visual.proj": torch.zeros(3, 3)})
    with pytest.raises(RuntimeError, match="load_state_dict"):
        engine.ClipEngine(arch, device).encode_image(torch.zeros(1, 3, 32, 32, device=device))
    with pytest.raises(RuntimeError, match="GPU"):
        eng.encode_image(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name,nimg,ntxt", [("ViT-B/32", 3, 4), ("ViT-B/16", 2, 3), ("ViT-L/14", 2, 3)])
def test_full_size_models_match_oracle(device, name, nimg, ntxt, precision):
    """BASELINE configs[0]/[1] architectures at full width/depth on a few items (the CPU oracle takes seconds), and ViT-B/16 -- the
    third `--model_name` choice of the reference's CLIs (evaluator.py:264-266): 197 tokens, the run-time-length attention kernel."""
    arch = ARCHS[name]
    sd, px, ids, ref_i, ref_t = _oracle(name, nimg, ntxt)
    eng = engine.ClipEngine(arch, device, precision=precision)
    eng.load_state_dict(sd)
    got_i = eng.encode_image(px.to(device)).cpu()
    got_t = eng.encode_text(ids.to(device)).cpu()
    ci, ct = _cos(got_i, ref_i), _cos(got_t, ref_t)
    print(f"{name} {precision}: image 1-cos max {float((1 - ci).max()):.2e}, text 1-cos max {float((1 - ct).max()):.2e}")
    assert float((1 - ci).max()) < COS_TOL and float((1 - ct).max()) < COS_TOL


@pytest.mark.parametrize("precision", ["bf16", "bf16-x24", "bf16-res16", "fp8", "fp8-x24"])
@pytest.mark.parametrize("name,nimg,ntxt", [("ViT-B/32", 4, 6), ("ViT-L/14", 3, 4)])
def test_heavy_tailed_weights_match_oracle(device, name, nimg, ntxt, precision):
    """every other parity test runs on N(0, sigma) weights.  oracle.clip_ref.add_outliers gives the towers what
    trained CLIP checkpoints are known for -- two massive residual channels (x 60), LayerNorm gains of 3 .. 10 (uncompensated) and
    of 30 .. 100 (migrated out of the consuming weight columns: the LayerNorm OUTPUT reaches ~400 there, next to e4m3's +-448), a
    class-token-like row, a sharp head per block -- and the path's 1e-3 cosine bar must hold for the default, for the 24-bit
    residual stream and for the fp8 modes (whose A operand carries per-channel scales since round 4; with round 3's unit-scale
    saturating store the same weights gave 1 - cos 9e-2 .. 3e-1).  Measured (tools/outlier_stress.py): bf16 3e-6 / 4e-5 (image /
    text), fp8 1e-4 / as bf16.  What these weights do NOT contain: uncompensated gains of 30 .. 100, which put the attention
    logits at ~1e4 and make the towers chaotic -- there a CPU emulation of bf16 operands (clip_ref ... bf16_operands=True, no
    kernel involved) leaves the fp32 oracle by 3e-2 .. 1e-1 just the same: a statement about the precision class, not testable as
    parity."""
    arch = ARCHS[name]
    sd, px, ids, ref_i, ref_t = _oracle(name, nimg, ntxt, outliers=True)
    assert max(float(v.abs().max()) for k, v in sd.items() if k.endswith("ln_1.weight")) > 25.0
    eng = engine.ClipEngine(arch, device, precision=precision)
    eng.load_state_dict(sd)
    got_i, got_t = eng.encode_image(px.to(device)).cpu(), eng.encode_text(ids.to(device)).cpu()
    assert bool(torch.isfinite(got_i).all()) and bool(torch.isfinite(got_t).all())
    ci, ct = _cos(got_i, ref_i), _cos(got_t, ref_t)
    print(f"{name} {precision} heavy-tailed: image 1-cos max {float((1 - ci).max()):.2e}, text 1-cos max {float((1 - ct).max()):.2e}")
    assert float((1 - ci).max()) < COS_TOL and float((1 - ct).max()) < COS_TOL


def test_bench_shape_batch_matches_oracle(device):
    """The shape bench.py runs: ViT-L/14, 255 items per encoder call = 65 535 token rows = 256 row tiles of the
    persistent GEMM with the ragged last row, the long-K fc2 path and 12-16 tiles per workgroup; texts 255 x 77 rows.  Nine
    images and nine texts spread over the batch (first rows, middle, the last row included) against the fp32 oracle, for
    every residual / operand precision of the product; the fp8 bars are those of test_fp8_encoders_match_oracle."""
    name, B = "ViT-L/14", 255
    oa = clip_ref.ARCHS[name]
    sd = clip_ref.random_state_dict(oa, seed=0)
    g = torch.Generator().manual_seed(4321)
    px = torch.randn(B, 3, 224, 224, generator=g)
    ids = clip_ref.synthetic_ids(oa, B)
    pick = torch.tensor([0, 1, 2, 126, 127, 128, 252, 253, 254])
    ref_i = clip_ref.encode_image(sd, oa, px[pick])
    ref_t = clip_ref.encode_text(sd, oa, ids[pick])
    pxd, idd = px.to(device), ids.to(device)
    # (round 4: fp8 operands are confined to the vision tower, whose embedding averages 257 rows -- the fp8 modes are held to the
    #  path's own 1e-3 bar like every other precision; round 3 allowed them 5e-3 because of the text tower's 1.6e-3)
    for precision, tol in (("bf16-x24", COS_TOL), ("bf16", COS_TOL), ("bf16-res16", COS_TOL), ("fp8", COS_TOL), ("fp8-res16", COS_TOL)):
        eng = engine.ClipEngine(ARCHS[name], device, precision=precision)
        eng.load_state_dict(sd)
        got_i = eng.encode_image(pxd).cpu()
        got_t = eng.encode_text(idd).cpu()
        assert bool(torch.isfinite(got_i).all()) and bool(torch.isfinite(got_t).all())
        ci, ct = _cos(got_i[pick], ref_i), _cos(got_t[pick], ref_t)
        print(f"{name} B={B} {precision}: image 1-cos max {float((1 - ci).max()):.2e}, text 1-cos max {float((1 - ct).max()):.2e}")
        assert float((1 - ci).max()) < tol and float((1 - ct).max()) < tol, precision
        # the batch position must not matter: the same items encoded alone (3 row tiles, skinny text GEMM) agree closely
        solo_i = eng.encode_image(pxd[pick[:2]]).cpu()
        assert float((1 - _cos(solo_i, got_i[pick[:2]])).max()) < (1e-3 if precision.startswith("fp8") else 1e-4)      # (the skinny-M GEMM rounds in another order; res16 measured 2.9e-5)
        del eng


@pytest.mark.parametrize("precision,tol", [("fp8", COS_TOL), ("fp8-res16", COS_TOL), ("fp8-mlp", COS_TOL)])
@pytest.mark.parametrize("name,nimg,ntxt", [("tiny", 9, 11), ("ViT-B/32", 3, 4), ("ViT-L/14", 2, 3)])
def test_fp8_encoders_match_oracle(device, name, nimg, ntxt, precision, tol):
    """BASELINE config 5: the vision tower's QKV (and, "fp8-mlp", fc1) on fp8 e4m3 operands; the text tower stays on bf16 operands
    (round 4: e4m3 q / k / v cost its one pooled row 1 - cos 1.4e-3 .. 5e-3, over north_star's 1e-3, for 1 % of throughput).  At
    the full-size models every fp8 mode is inside the path's 1e-3 cosine bar (round 3 allowed 5e-3); the 2-layer, 17-token "tiny"
    image tower has nothing to average the e4m3 error over and keeps a 5e-3 bound.  What config 5 itself asks for is Recall@10
    within 0.2 points of bf16 (next test)."""
    if name == "tiny":
        tol = 5e-3 if precision != "fp8-mlp" else 2e-2
    arch = ARCHS[name]
    sd, px, ids, ref_i, ref_t = _oracle(name, nimg, ntxt)
    eng = engine.ClipEngine(arch, device, precision=precision)
    eng.load_state_dict(sd)
    # fp8 operands are confined to the vision tower: the text embeddings are, bit for bit, those of the bf16 precision with the same residual stream
    twin = engine.ClipEngine(arch, device, precision="bf16-res16" if precision == "fp8-res16" else "bf16")
    twin.load_state_dict(sd)
    assert torch.equal(eng.encode_text(ids.to(device)), twin.encode_text(ids.to(device)))
    assert not torch.equal(eng.encode_image(px.to(device)), twin.encode_image(px.to(device)))
    del twin
    ci, ct = _cos(eng.encode_image(px.to(device)).cpu(), ref_i), _cos(eng.encode_text(ids.to(device)).cpu(), ref_t)
    print(f"{name} {precision}: image 1-cos max {float((1 - ci).max()):.2e}, text 1-cos max {float((1 - ct).max()):.2e}")
    assert float((1 - ci).max()) < tol and float((1 - ct).max()) < tol


@pytest.mark.parametrize("precision", ["bf16", "bf16-res16"])          # (a 24-bit stream has no in-place epilogue: the option is idle there)
def test_residual_fusion_switch(device, precision):
    """Option "residual_fusion" (per model): residual add inside the out-proj / fc2 epilogues (default) against the store-only
    epilogues + LayerNorm updates: both inside the path's bar against the oracle, close to each other, the switch really
    switches, and it is THIS model's switch -- a second engine keeps its own setting."""
    arch, sd, eng = _engine("ViT-B/32", device, precision=precision)
    other = engine.ClipEngine(arch, device, precision=precision)
    other.load_state_dict(sd)
    oa = clip_ref.ARCHS["ViT-B/32"]
    g = torch.Generator().manual_seed(77)
    px = torch.randn(40, 3, arch.image_size, arch.image_size, generator=g)            # 2 000 token rows: the persistent GEMM's side of the switch
    ids = clip_ref.synthetic_ids(oa, 24)
    ref_i, ref_t = clip_ref.encode_image(sd, oa, px[:4]), clip_ref.encode_text(sd, oa, ids[:4])
    assert eng.residual_fusion() == 1 and other.residual_fusion() == 1               # the default level: bf16 streams only
    assert eng.residual_fusion_active() == (precision == "bf16-res16")
    level = 1 if precision == "bf16-res16" else 2                                     # fp32 streams fuse at level 2 only (slower: opt-in)
    outs = {}
    for on in (True, False):
        eng.set_residual_fusion(level if on else 0)
        assert eng.residual_fusion() == (level if on else 0) and eng.residual_fusion_active() == on and other.residual_fusion() == 1
        outs[on] = (eng.encode_image(px.to(device)).cpu(), eng.encode_text(ids.to(device)).cpu())
        assert float((1 - _cos(outs[on][0][:4], ref_i)).max()) < COS_TOL and float((1 - _cos(outs[on][1][:4], ref_t)).max()) < COS_TOL
    assert torch.equal(other.encode_image(px.to(device)).cpu(), outs[precision == "bf16-res16"][0])          # untouched by eng's switch
    assert not torch.equal(outs[True][0], outs[False][0])
    close = 3e-4 if precision == "bf16-res16" else 2e-5
    assert float((1 - _cos(outs[True][0], outs[False][0])).max()) < close and float((1 - _cos(outs[True][1], outs[False][1])).max()) < close


RECALL_BAR = 0.2        # percentage points of Recall@10 (BASELINE.json configs[4]); FIXED -- a mode that misses it is opt-in, not a wider bar


@pytest.mark.parametrize("name", ["ViT-B/32", "ViT-L/14"])
def test_recall_at_10_default_against_oracle_and_fp8_within_0p2(device, golden_dir, name):
    """(Round 4, VERDICT r3 1(iv): also on ViT-L/14, the model BASELINE configs[4] names -- there the oracle's embeddings of the 256
    anchor items come from tests/golden/recall_anchor_ViT-L-14.npz, computed by the fp32 CPU oracle in the build container
    (tests/golden/make_golden.py recall-anchor: minutes of CPU time), and those items' pixels from a CPU generator so that they are
    the same bytes here and there.)
    Retrieval with a noisy copy of every gallery image as the query, at three noise levels (Recall@10 about 99 / 77 / 42 %).
    (1) The DEFAULT precision is anchored to the ORACLE: on the first 256 gallery items and their queries the fp32 CPU oracle
        encodes the same pixels; the default engine's ground-truth ranks may differ from the oracle's only where the oracle itself
        scores a competitor within 2e-4 of the ground truth, and its Recall@10 on that subset is the oracle's within those queries.
    (2) BASELINE config 5's bar, fixed at 0.2 points: the "fp8" engine's Recall@10 over all N = 16 384 items stays within 0.2 points
        of the default engine's at every level (0.2 points are 33 queries; at 4 096 items a change of the GEMM's rounding order
        alone moved Recall@10 by 0.14 points between rounds 1 and 2).
    (3) The other modes are RECORDS, printed with inside_bar true / false and not asserted: "bf16-res16" / "fp8-res16" (bf16
        residual stream: 0.3-0.7 points where recall is noise-limited) and "fp8-mlp" (about one point) are opt-in because they
        miss the bar; round 2 widened it four times to keep them green as defaults, which is what this test no longer does."""
    from knowledge_enhanced_multimodal_retrieval_amd import _lib, metrics
    from oracle import metrics_ref
    n, chunk, n_sub = 16384, 1024, 256
    fixture = None
    if name != "ViT-B/32":
        import json
        fixture = np.load(os.path.join(golden_dir, "recall_anchor_%s.npz" % name.replace("/", "-")))
        fmeta = json.loads(bytes(fixture["meta_json"]).decode())
        assert fmeta["arch"] == name and fmeta["n_sub"] == n_sub
        gc = torch.Generator().manual_seed(fmeta["seed"])
        anchor_base = torch.randn(n_sub, 3, 224, 224, generator=gc)
        anchor_noise = torch.randn(n_sub, 3, 224, 224, generator=gc)
        assert abs(float(anchor_base.double().abs().sum()) - fmeta["input_abs_sums"]["base"]) < 1e-6 * fmeta["input_abs_sums"]["base"]
        assert abs(float(anchor_noise.double().abs().sum()) - fmeta["input_abs_sums"]["noise"]) < 1e-6 * fmeta["input_abs_sums"]["noise"]
    arch = ARCHS[name]
    oa = clip_ref.ARCHS[name]
    sd = clip_ref.random_state_dict(oa, seed=0)
    levels = (1.5, 2.0, 2.5) if fixture is None else tuple(fmeta["levels"])
    default = _lib.DEFAULT_PRECISION
    assert default == "bf16-x24", "the default precision must be one that meets the bar (fp32 residual arithmetic; stored as 24-bit floats since round 4)"
    res, emb_default = {}, {}
    precs = (default, "fp8", "bf16-res16", "fp8-res16", "fp8-mlp", "bf16", "fp8-x24")
    for prec in precs:
        eng = engine.ClipEngine(arch, device, precision=prec)
        eng.load_state_dict(sd)
        gal, qry = [], {lvl: [] for lvl in levels}
        for c in range(n // chunk):
            g = torch.Generator(device=device).manual_seed(1000 + c)
            base = torch.randn(chunk, 3, 224, 224, generator=g, device=device)
            noise = torch.randn(chunk, 3, 224, 224, generator=g, device=device)
            if fixture is not None and c == 0:              # the anchor items: the fixture's pixels
                base[:n_sub], noise[:n_sub] = anchor_base.to(device), anchor_noise.to(device)
            gal.append(eng.encode_image(base, normalize=True))
            for lvl in levels:
                qry[lvl].append(eng.encode_image(base + lvl * noise, normalize=True))
        gal = torch.cat(gal)
        for lvl in levels:
            q = torch.cat(qry[lvl])
            res[(prec, lvl)] = metrics.compute_retrieval_metrics(q, gal, "T2I")
            if prec == default:
                emb_default[lvl] = (q[:n_sub].cpu().numpy(), gal[:n_sub].cpu().numpy())
        del eng
    for lvl in levels:
        print(lvl, {k: tuple(round(res[(p, lvl)][k], 2) for p in precs) for k in ("T2I_R@1", "T2I_R@10", "T2I_MRR")})
    # ---- (1) the default engine against the oracle on the first n_sub items (the same pixels, regenerated on the host)
    if fixture is None:
        g = torch.Generator(device=device).manual_seed(1000)
        base = torch.randn(chunk, 3, 224, 224, generator=g, device=device)
        noise = torch.randn(chunk, 3, 224, 224, generator=g, device=device)
        base, noise = base[:n_sub].cpu(), noise[:n_sub].cpu()
        with torch.no_grad():
            o_gal = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, base)).numpy()
    else:
        o_gal = fixture["gallery"]
        with torch.no_grad():                              # the fixture is this oracle's output: two rows re-run here pin it
            two = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, anchor_base[:2])).numpy()
        assert float(np.abs(two - o_gal[:2]).max()) < 5e-6
    anchor, record = [], {}
    for lvl in levels:
        if fixture is None:
            with torch.no_grad():
                o_q = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, base + lvl * noise)).numpy()
        else:
            o_q = fixture[f"query_{lvl}"]
        S = metrics_ref.similarity(o_q, o_gal)
        o_ranks = metrics_ref.ranks_by_count(S.astype(np.float64))
        hq, hg = emb_default[lvl]
        assert float((1 - (hq * o_q).sum(1)).max()) < COS_TOL and float((1 - (hg * o_gal).sum(1)).max()) < COS_TOL
        h_ranks = metrics_ref.ranks_by_count(metrics_ref.similarity(hq, hg).astype(np.float64))
        sgt = S[np.arange(n_sub), np.arange(n_sub)]
        near = (np.abs(S - sgt[:, None]) <= 2e-4).sum(axis=1) - 1
        moved = np.abs(h_ranks - o_ranks)
        r10_o, r10_h = 100.0 * np.mean(o_ranks <= 10), 100.0 * np.mean(h_ranks <= 10)
        may_cross = ((o_ranks - near <= 10) & (o_ranks > 10)) | ((o_ranks + near > 10) & (o_ranks <= 10))
        print(f"level {lvl}, first {n_sub} items: Recall@10 oracle {r10_o:.2f} / default engine {r10_h:.2f}; ranks identical "
              f"{int((moved == 0).sum())}/{n_sub}, beyond the oracle's own 2e-4 neighbourhood: {int((moved > near).sum())}")
        anchor.append({"level": lvl, "recall10_oracle": round(r10_o, 2), "recall10_default_engine": round(r10_h, 2),
                       "ranks_identical": int((moved == 0).sum()), "of": n_sub, "beyond_oracle_neighbourhood": int((moved > near).sum())})
        assert int((moved > near).sum()) <= n_sub // 100, (lvl, int((moved > near).sum()))
        assert abs(r10_h - r10_o) <= 100.0 * may_cross.sum() / n_sub + 1e-9 and abs(r10_h - r10_o) <= 2 * RECALL_BAR + 1e-9, (lvl, r10_h, r10_o)
    # ---- (2) config 5's bar for the fp8 encoders, (3) records for the opt-in modes
    for lvl in levels:
        ref = res[(default, lvl)]["T2I_R@10"]
        rec = {p: {"R@10": round(res[(p, lvl)]["T2I_R@10"], 3), "delta": round(res[(p, lvl)]["T2I_R@10"] - ref, 3),
                   "inside_bar": bool(abs(res[(p, lvl)]["T2I_R@10"] - ref) <= RECALL_BAR + 1e-9)} for p in precs[1:]}
        print(f"level {lvl}: default R@10 {ref:.3f};", rec)
        record[str(lvl)] = {"default_recall10": round(ref, 3), **rec}
        assert rec["fp8"]["inside_bar"], (lvl, rec["fp8"])
    if os.environ.get("KEMR_RECALL_JSON"):                # profiles/r03_recall_bar.json is this record (bench.py reads inside_recall_bar from it)
        import json
        inside = {p: all(record[str(lvl)][p]["inside_bar"] for lvl in levels) for p in precs[1:]}
        inside[default] = True
        path = os.environ["KEMR_RECALL_JSON"] if name == "ViT-B/32" else os.environ["KEMR_RECALL_JSON"].replace(".json", "_" + name.replace("/", "-") + ".json")
        with open(path, "w") as f:
            json.dump({"source": "tests/test_encoder_gpu.py::test_recall_at_10_default_against_oracle_and_fp8_within_0p2 on MI355X: 16 384 synthetic "
                                 "images retrieved by noisy copies, " + name + ", three noise levels; bar = 0.2 points of Recall@10 against the default "
                                 "precision (bf16 operands, fp32 residual stream), FIXED",
                       "bar_points": RECALL_BAR, "default": default, f"oracle_anchor_first_{n_sub}_items": anchor, "levels": record,
                       "inside_bar_at_every_level": inside}, f, indent=1)


@pytest.mark.parametrize("precision", ["bf16", "bf16-res16", "fp8"])
@pytest.mark.parametrize("name,ntxt", [("tiny", 37), ("ViT-B/32", 300), ("ViT-L/14", 130)])
def test_text_rows_behind_the_eot_are_not_needed(device, name, ntxt, precision):
    """kemr_encode_text_packed (the default of ClipEngine.encode_text): every text computed on its first argmax + 1 positions only,
    all texts packed one behind the other, against kemr_encode_text on all ctx positions (causal mask + pooling at the end-of-text
    token: reference `x[arange, text.argmax(-1)]`), for lengths from 1 to ctx, host ids (no device sync), device ids + host
    lengths, device ids alone, and several calls' worth of rows.  The two are the same arithmetic per row; what differs is the number
    of rows M of every GEMM, and with it the kernel a launch is routed to (gemm.hip launch_gemm: persistent / skinny / ragged last
    tile) and its fp32 summation order -- exactly what another batch size does to the full-context path (checked below), 1 - cos of
    a few 1e-5 against the parity bar's 1e-3.  Calls that route alike are bit-identical."""
    arch, sd, eng = _engine(name, device, precision=precision)
    oa = clip_ref.ARCHS[name]
    ids = clip_ref.synthetic_ids(oa, ntxt)
    eot = int(ids.max())
    ids[0] = 0; ids[0, 0] = eot                              # length 1: the pooled row is the first
    ids[1] = 1; ids[1, arch.ctx - 1] = eot                   # length ctx
    ids[2, :] = 0                                            # all zero: argmax 0
    lens = engine.text_lengths(ids)
    assert int(lens[0]) == 1 and int(lens[1]) == arch.ctx and int(lens[2]) == 1 and int(lens.min()) >= 1
    close = lambda x, y: float((1 - _cos(x, y)).max())
    eng.pack_text = False
    full = eng.encode_text(ids.to(device), normalize=True)
    other_batch = close(eng.encode_text(ids[:ntxt // 2].to(device), normalize=True), full[:ntxt // 2])
    eng.pack_text = True
    a = eng.encode_text(ids, normalize=True)                                  # host ids
    b = eng.encode_text(ids.to(device), normalize=True, lens=lens)           # device ids + host lengths
    c = eng.encode_text(ids.to(device), normalize=True)                      # device ids alone (one small D2H copy)
    assert torch.equal(a, b) and torch.equal(a, c)
    tol = 2e-4 if precision == "fp8" else 1e-4
    print(f"{name} {precision}: packed vs full-context 1 - cos max {close(a, full):.2e}; full-context, half the batch: {other_batch:.2e}")
    assert close(a, full) < tol and other_batch < tol
    ref = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, ids[:24]))
    assert close(a[:24].cpu(), ref) < (COS_TOL if precision != "fp8" else 5e-3)
    old = engine.TEXT_ROW_BUDGET
    try:                                                                     # many small calls: the chunking by rows
        engine.TEXT_ROW_BUDGET = 5 * arch.ctx
        d = eng.encode_text(ids, normalize=True)
    finally:
        engine.TEXT_ROW_BUDGET = old
    assert close(d, full) < 2 * tol
    # a length that stops short of the end-of-text token pools the last computed row: not the reference's embedding, but finite and
    # in bounds, and the other texts of the call are untouched; longer than needed changes nothing beyond the route
    short = lens.clone(); short[3] = max(1, int(lens[3]) - 2)
    e = eng.encode_text(ids.to(device), normalize=True, lens=short)
    assert torch.isfinite(e).all() and close(e[4:], a[4:]) < tol
    longer = (lens + 3).clamp(max=arch.ctx)
    assert close(eng.encode_text(ids.to(device), normalize=True, lens=longer), a) < tol
    with pytest.raises(RuntimeError, match="lengths"):
        eng.encode_text(ids.to(device), lens=lens[:-1])


@pytest.mark.parametrize("precision", ["bf16", "bf16-res16", "fp8", "fp8-mlp", "bf16-x24"])
@pytest.mark.parametrize("name,nimg,ntxt", [("tiny", 9, 37), ("ViT-B/32", 70, 200), ("ViT-L/14", 20, 90)])
def test_last_block_on_the_pooled_row_only(device, name, nimg, ntxt, precision):
    """Option last_block_pooled_row (default on): only the class / end-of-text row leaves a tower, so the LAST block computes K and V
    for every row and everything behind the scores -- query, attention output, out-proj, ln_2, MLP -- for that row alone (compact
    [items, W] buffers, attention_pooled_kernel).  Against the same engine with the option off (every row through the last block, as
    the reference computes it): the same arithmetic on the pooled rows up to the smaller GEMMs' summation order and the fp32 dot
    products of the one-query attention -- 1 - cos of a few 1e-5 -- and against the fp32 oracle within the parity bar; for images,
    full-context texts and packed texts, including a text of length 1 and one of length ctx."""
    arch, sd, eng = _engine(name, device, precision=precision)
    oa = clip_ref.ARCHS[name]
    g = torch.Generator().manual_seed(77)
    px = torch.randn(nimg, 3, arch.image_size, arch.image_size, generator=g).to(device)
    ids = clip_ref.synthetic_ids(oa, ntxt)
    eot = int(ids.max())
    ids[0] = 0; ids[0, 0] = eot
    ids[1] = 1; ids[1, arch.ctx - 1] = eot
    assert eng.last_block_pooled_row()
    close = lambda x, y: float((1 - _cos(x, y)).max())
    got = {}
    for on in (True, False):
        eng.set_last_block_pooled_row(on)
        eng.pack_text = True
        got[on] = [eng.encode_image(px, normalize=True), eng.encode_text(ids, normalize=True)]
        eng.pack_text = False
        got[on].append(eng.encode_text(ids.to(device), normalize=True))
    eng.set_last_block_pooled_row(True)
    eng.pack_text = True
    d = [close(a, b) for a, b in zip(got[True], got[False])]
    print(f"{name} {precision}: pooled-row last block vs full, 1 - cos max: images {d[0]:.2e}, packed texts {d[1]:.2e}, full-context texts {d[2]:.2e}")
    assert max(d) < (1e-4 if not precision.startswith("fp8") else 1e-3)    # (fp8: the pooled rows' query goes through another e4m3 launch shape)
    if (precision == "bf16-res16" and name != "tiny") or precision == "fp8-mlp":
        return                                   # (the option is idle: residual add in the GEMM epilogues / fc1 on fp8)
    ref_i = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, px[:6].cpu()))
    ref_t = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, ids[:12]))
    tol = COS_TOL if precision != "fp8" else 5e-3
    assert close(got[True][0][:6].cpu(), ref_i) < tol and close(got[True][1][:12].cpu(), ref_t) < tol
