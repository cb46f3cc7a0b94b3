"""Image transform: the oracle restatement against Pillow itself (CPU, bit-exact) and the HIP kernels
against the host transform the drop-in modules use (GPU, bit-exact)."""
import numpy as np
import pytest
import torch

from knowledge_enhanced_multimodal_retrieval_amd.preprocess import ClipPreprocess, ClipPreprocessGPU
from oracle import preprocess_ref

SIZES = [(300, 400), (224, 500), (999, 224), (1080, 1920), (231, 229), (100, 80), (640, 480), (225, 224), (224, 224),
         (37, 1000), (2000, 1500)]


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:                                   # smooth content as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    return base


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resize_equals_pillow(h, w):
    from PIL import Image
    arr = _image(h, w, h + w)
    nw, nh = preprocess_ref.resized_size(h, w, 224)
    pil = np.asarray(Image.fromarray(arr).resize((nw, nh), Image.BICUBIC))
    assert np.array_equal(preprocess_ref.resize_bicubic_u8(arr, nw, nh), pil)


@pytest.mark.parametrize("h,w", SIZES[:6])
def test_oracle_transform_equals_host_transform(h, w):
    from PIL import Image
    arr = _image(h, w, h * 3 + w)
    want = ClipPreprocess(224)(Image.fromarray(arr)).numpy()
    assert np.array_equal(preprocess_ref.clip_preprocess(arr, 224), want)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_gpu_transform_is_bit_exact(device, h, w):
    from PIL import Image
    arr = _image(h, w, h + 2 * w)
    want = ClipPreprocess(224)(Image.fromarray(arr))
    pre = ClipPreprocessGPU(224, device)
    got = pre(torch.from_numpy(arr))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)
    assert torch.equal(pre(Image.fromarray(arr)).cpu(), want)          # PIL input, workspace reused


@pytest.mark.gpu
def test_gpu_batch_transform_is_bit_exact(device):
    """One launch pair for a whole batch of mixed sizes (repeats share one filter table) == the host transform per image;
    a second, larger batch reuses and grows the workspace; pinned and pageable sources give the same bytes."""
    from PIL import Image
    from knowledge_enhanced_multimodal_retrieval_amd.preprocess import pack_raw
    pre = ClipPreprocessGPU(224, device)
    host = ClipPreprocess(224)
    for sizes in (SIZES[:5] + SIZES[:2], SIZES + [(224, 224)] * 3 + SIZES[::-1]):
        arrs = [_image(h, w, 5 * h + w + i) for i, (h, w) in enumerate(sizes)]
        want = torch.stack([host(Image.fromarray(a)) for a in arrs])
        packed = pack_raw([torch.from_numpy(a) for a in arrs])
        assert len(packed) == len(sizes) and packed.offsets[0] == 0 and packed.data.numel() == sum(h * w * 3 for h, w in sizes)
        got = pre.batch(packed)
        got_pinned = pre.batch(packed.pin_memory())
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want) and torch.equal(got_pinned.cpu(), want)
    assert pre.batch(pack_raw([])).shape == (0, 3, 224, 224)
    with pytest.raises(RuntimeError, match="uint8"):
        pack_raw([torch.zeros(4, 4, 3)])


def test_collate_packs_raw_images_and_tokenizes_in_the_loader():
    from knowledge_enhanced_multimodal_retrieval_amd import datasets
    from knowledge_enhanced_multimodal_retrieval_amd.evaluators import default_tokenize
    from knowledge_enhanced_multimodal_retrieval_amd.preprocess import PackedRaw
    ds = datasets.SyntheticRawImageDataset(6, seed=3)
    a, b = ds[4], ds[4]
    assert a[0].dtype == torch.uint8 and a[0].shape[2] == 3 and torch.equal(a[0], b[0]) and a[1:] == b[1:]
    images, q, t, ids = datasets.CollateAndTokenize(default_tokenize)([ds[i] for i in range(6)])
    assert isinstance(images, PackedRaw) and len(images) == 6 and ids[5] == "synthetic-000005"
    assert q.shape == (6, 77) and t.shape == (6, 77) and torch.equal(q, default_tokenize([ds[i][1] for i in range(6)]))
    hs, ws, offs = images.heights, images.widths, images.offsets
    for i in range(6):
        im = ds[i][0]
        assert (hs[i], ws[i]) == tuple(im.shape[:2])
        assert torch.equal(images.data[offs[i]:offs[i] + im.numel()], im.reshape(-1))
    # through a DataLoader with worker processes: same batches
    from torch.utils.data import DataLoader
    got = list(DataLoader(ds, batch_size=4, num_workers=2, collate_fn=datasets.CollateAndTokenize(default_tokenize)))
    assert [len(g[0]) for g in got] == [4, 2] and torch.equal(got[0][1], q[:4]) and torch.equal(got[1][0].data[-10:], images.data[-10:])


@pytest.mark.gpu
def test_encode_dataset_through_loader_workers_matches_per_image_path(device):
    """The whole host pipeline (worker processes generate / pack / tokenise, pin thread, one H2D copy and one preprocess
    launch pair per loader batch, 255-item encoder calls) against the pieces called one by one on the main thread."""
    import warnings
    import clip
    from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
    ds = datasets.SyntheticRawImageDataset(70, seed=11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = clip.load("ViT-B/32", device="cuda")
        got_i, got_q, got_t, ids = evaluators.encode_dataset(model, ds, batch_size=16, seed=1, num_workers=2)
        pre = ClipPreprocessGPU(224, device)
        items = [ds[i] for i in range(len(ds))]
        px = torch.stack([pre(im) for im, *_ in items])
        want_i = model.encode_image(px, normalize=True)
        want_q = model.encode_text(evaluators.default_tokenize([it[1] for it in items]).to(device), normalize=True)
    assert ids == [it[3] for it in items]
    # rows are independent of the grouping up to the GEMM path a batch size selects (skinny kernel below 512 rows)
    assert float((1 - torch.nn.functional.cosine_similarity(got_i, want_i)).max()) < 1e-4
    assert float((1 - torch.nn.functional.cosine_similarity(got_q, want_q)).max()) < 1e-4
    assert got_t.shape == got_q.shape and bool(torch.isfinite(got_t).all())


@pytest.mark.gpu
def test_gpu_transform_other_resolution_and_errors(device):
    from PIL import Image
    arr = _image(500, 333, 7)
    want = ClipPreprocess(336)(Image.fromarray(arr))
    assert torch.equal(ClipPreprocessGPU(336, device)(torch.from_numpy(arr)).cpu(), want)
    with pytest.raises(RuntimeError, match="uint8"):
        ClipPreprocessGPU(224, device)(torch.zeros(10, 10, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        ClipPreprocessGPU(224, "cpu")


@pytest.mark.gpu
def test_encode_dataset_with_gpu_preprocessing_is_identical(device):
    """A dataset that hands out raw uint8 images (preprocess.RawRGB) is preprocessed on the device by encode_dataset: the
    embeddings equal those of the host transform bit for bit (same pixels in, deterministic kernels)."""
    import warnings
    from PIL import Image
    import clip
    from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
    from knowledge_enhanced_multimodal_retrieval_amd.preprocess import RawRGB
    sizes = [(300, 400), (224, 224), (500, 333), (231, 229), (640, 480), (100, 80), (225, 224)]
    rows = [{"image": Image.fromarray(_image(h, w, i)), "query_text": f"vase {i} bronze", "target_text": f"bronze vase number {i}",
             "uuid": f"u{i}"} for i, (h, w) in enumerate(sizes)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, preprocess = clip.load("ViT-B/32", device="cuda")
        host = evaluators.encode_dataset(model, datasets.CLIPEvalDatasetHF(rows, preprocess), 3, 1)
        raw = evaluators.encode_dataset(model, datasets.CLIPEvalDatasetHF(rows, RawRGB()), 3, 1)
    assert torch.equal(host[0], raw[0]) and torch.equal(host[1], raw[1]) and host[3] == raw[3]


@pytest.mark.gpu
def test_undecodable_items_reach_the_encoder_as_the_reference_zero_tensor(device):
    """VERDICT r3 1(i).  The reference feeds torch.zeros(3, 224, 224) -- zeros AFTER normalisation -- for an image that fails to
    decode (/root/reference/src/clip/datasets/clip_dataset.py:120-125).  On the default (GPU-transform) route such an item is an
    empty descriptor of kemr_preprocess_u8_batch: a split with two undecodable items gives embeddings torch.equal to the
    host-transform run, and the embedding of those items is the fp32 oracle's embedding of a zero tensor within 1e-3 cosine.
    Then the reference's own loop shape (DataLoader(collate_fn_eval) -> images.to(device) -> model.encode_image(images),
    evaluator.py:96-121) on the packed batches."""
    import warnings
    from PIL import Image
    import clip
    from torch.utils.data import DataLoader
    from knowledge_enhanced_multimodal_retrieval_amd import datasets, evaluators
    from knowledge_enhanced_multimodal_retrieval_amd.preprocess import PackedRaw, pack_raw
    from oracle import clip_ref
    sizes = [(300, 400), None, (224, 224), (500, 333), None, (231, 229), (640, 480)]
    rows = [{"image": None if hw is None else Image.fromarray(_image(hw[0], hw[1], i)), "query_text": f"vase {i} bronze",
             "target_text": f"bronze vase number {i}", "uuid": f"u{i}"} for i, hw in enumerate(sizes)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, preprocess = clip.load("ViT-B/32", device="cuda")
    assert preprocess.defer_to_gpu                                    # the default route of the drop-in CLIs
    gpu_ds = datasets.CLIPEvalDatasetHF(rows, preprocess)
    host_ds = datasets.CLIPEvalDatasetHF(rows, ClipPreprocess(224))
    assert tuple(gpu_ds[1][0].shape) == (0, 0, 3) and torch.equal(host_ds[1][0], torch.zeros(3, 224, 224))
    gpu = evaluators.encode_dataset(model, gpu_ds, 3, 1)             # batches of 3: [ok, bad, ok] [ok, bad, ok] [ok]
    host = evaluators.encode_dataset(model, host_ds, 3, 1)
    assert torch.equal(gpu[0], host[0]) and torch.equal(gpu[1], host[1]) and gpu[3] == host[3]
    # the kernel's own output for such an item: exact zeros, also when the whole batch is undecodable
    pre = ClipPreprocessGPU(224, device)
    px = pre.batch(pack_raw([gpu_ds[0][0], gpu_ds[1][0], gpu_ds[2][0]]))
    assert torch.equal(px[1], torch.zeros_like(px[1])) and torch.equal(px[0].cpu(), host_ds[0][0]) and torch.equal(px[2].cpu(), host_ds[2][0])
    allbad = pre.batch(pack_raw([gpu_ds[1][0], gpu_ds[4][0]]))
    assert tuple(allbad.shape) == (2, 3, 224, 224) and float(allbad.abs().max()) == 0.0
    assert float(pre(gpu_ds[1][0]).abs().max()) == 0.0
    # against the oracle: the embedding of a zero tensor
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    oa = clip_ref.ARCHS["ViT-B/32"]
    ref = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, torch.zeros(1, 3, 224, 224)))
    for i in (1, 4):
        cos = torch.nn.functional.cosine_similarity(gpu[0][i].double().cpu(), ref[0].double(), dim=0).item()
        assert cos > 1 - 1e-3, (i, cos)
    black = model.encode_image(pre.batch(pack_raw([torch.zeros(224, 224, 3, dtype=torch.uint8)])), normalize=True)
    assert float(1 - torch.nn.functional.cosine_similarity(black[0], gpu[0][1], dim=0)) > 1e-3      # what round 3 fed instead: another embedding
    # the reference's loop on this dataset: the packed batch moves with .to() and encode_image accepts it
    loader = DataLoader(gpu_ds, batch_size=3, shuffle=False, num_workers=0, collate_fn=datasets.collate_fn_eval)
    feats = []
    for images, queries, targets, uuids in loader:
        assert isinstance(images, PackedRaw)
        images = images.to("cuda")
        f = model.encode_image(images)
        feats.append(f / f.norm(dim=-1, keepdim=True))
    feats = torch.cat(feats)
    assert float((1 - torch.nn.functional.cosine_similarity(feats, host[0])).max()) < 1e-5    # another batch size routes the GEMMs differently
