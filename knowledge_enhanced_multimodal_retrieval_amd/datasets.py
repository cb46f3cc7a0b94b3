"""Host-side input contract of the evaluators (reference: /root/reference/src/clip/datasets/clip_dataset.py:81-132,
176-180): a sample is ``(image f32[3,S,S], query_text, target_text, uuid)``, a batch is
``(f32[B,3,S,S], list[str], list[str], list[str])``.  ``CLIPEvalDatasetHF`` wraps a HuggingFace split the same way
(word-truncation to 150 words, zero image when decoding fails); ``SyntheticRetrievalDataset`` provides the same
contract from a seed so that every entry point runs offline."""
from __future__ import annotations

import logging
from typing import List

import torch
from torch.utils.data import Dataset

logger = logging.getLogger(__name__)


def truncate_words(text: str, max_words: int) -> str:
    words = text.split()
    return " ".join(words[:max_words]) if len(words) > max_words else text


class CLIPEvalDatasetHF(Dataset):
    """Evaluation dataset over a HuggingFace split with columns image / query_text / target_text / uuid."""

    def __init__(self, hf_dataset, preprocessor=None, max_text_length: int = 150, image_size: int = 224):
        self.dataset, self.preprocessor = hf_dataset, preprocessor
        self.max_text_length, self.image_size = max_text_length, image_size
        # The preprocess object of clip.load / load_clip_model may say "defer to the GPU" (preprocess.ClipPreprocess): the item is
        # then the decoded image as uint8 [H, W, 3]; collate_fn_eval packs a batch of them and evaluators.encode_dataset runs the
        # transform on the device, bit-identical to the host one.  The reference's call site stays as it is.
        if getattr(preprocessor, "defer_to_gpu", False):
            from .preprocess import RawRGB
            self.host_preprocessor, self.preprocessor = preprocessor, RawRGB()
        logger.info(f"Evaluation dataset initialized: {len(self.dataset)} samples"
                    + (" (image transform on the GPU)" if getattr(preprocessor, "defer_to_gpu", False) else ""))

    def __len__(self):
        return len(self.dataset)

    def _truncate_text(self, text: str) -> str:
        return truncate_words(text, self.max_text_length)

    def __getitem__(self, idx):
        sample = self.dataset[idx]
        try:
            image = sample["image"]
            if image.mode != "RGB":
                image = image.convert("RGB")
            if self.preprocessor:
                image = self.preprocessor(image)
        except Exception as e:  # undecodable image -> zero image, like the reference (clip_dataset.py:120-125)
            logger.error(f"Error loading image {sample.get('uuid', idx)}: {e}")
            if self.preprocessor:
                # The reference substitutes torch.zeros(3, 224, 224): zeros AFTER normalisation.  On the raw route that is an
                # EMPTY uint8 [0, 0, 3] item, which pack_raw / kemr_preprocess_u8_batch turn into that same all-zero tensor
                # (a black uint8 picture would normalise to (-1.79, -1.75, -1.48) and give a different embedding).
                from .preprocess import RawRGB
                image = (torch.zeros(0, 0, 3, dtype=torch.uint8) if isinstance(self.preprocessor, RawRGB)
                         else torch.zeros(3, self.image_size, self.image_size))
            else:
                from PIL import Image
                image = Image.new("RGB", (self.image_size, self.image_size))
        return (image, self._truncate_text(sample["query_text"]), self._truncate_text(sample["target_text"]),
                sample["uuid"])


CLIPEvaluationDataset = CLIPEvalDatasetHF

_WORDS = ("amphora vase bronze marble portrait landscape oil canvas roman greek medieval baroque gilded wooden panel "
          "statue relief fresco mosaic coin sword helmet textile manuscript folio saint king queen river harbour "
          "ceramic glazed terracotta figure seated standing holding crown dated century workshop attributed school "
          "museum collection fragment inscription").split()


class SyntheticRetrievalDataset(Dataset):
    """Seeded stand-in for the (unavailable offline) ``xuemduan/reevaluate-image-text-pairs`` split: normalised-pixel
    noise images and word-salad query / target texts that share a few item-specific words."""

    def __init__(self, n: int, image_size: int = 224, seed: int = 42):
        self.n, self.image_size, self.seed = n, image_size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        image = torch.randn(3, self.image_size, self.image_size, generator=g)
        pick = lambda k: [_WORDS[int(i)] for i in torch.randint(0, len(_WORDS), (k,), generator=g)]
        shared = pick(4)
        query = " ".join(shared[:2] + pick(int(torch.randint(3, 12, (1,), generator=g))))
        target = " ".join(shared + pick(int(torch.randint(10, 40, (1,), generator=g))))
        return image, query, target, f"synthetic-{idx:06d}"


class SyntheticHFSplit:
    """A stand-in for one split of the HuggingFace dataset the reference loads (columns image / query_text / target_text /
    uuid, ``image`` a PIL image of whatever size the file has): the camera-like pictures of :class:`SyntheticRawImageDataset`
    as PIL images, so that ``CLIPEvalDatasetHF(split, preprocess)`` -- the reference's own call -- runs offline end to end."""

    def __init__(self, n: int, seed: int = 42):
        self._raw = SyntheticRawImageDataset(n, seed)

    def __len__(self):
        return len(self._raw)

    def __getitem__(self, idx):
        from PIL import Image
        image, query, target, uid = self._raw[idx]
        return {"image": Image.fromarray(image.numpy()), "query_text": query, "target_text": target, "uuid": uid}


class SyntheticJPEGSplit:
    """As :class:`SyntheticHFSplit`, with the image stored the way the HuggingFace ``datasets`` Image feature stores it -- ENCODED
    bytes (JPEG here, as photographs are) that are decoded with Pillow when the row is read, i.e. inside the loader worker
    (reference: ``CLIPEvalDatasetHF.__getitem__`` receives the decoded PIL image from ``datasets``, clip_dataset.py:110-113).  The
    bytes are encoded in memory at construction (no files, no network): ``distinct`` different pictures over the eight camera-like
    sizes, row idx reads picture idx % distinct -- what a row costs is its decode, which does not care that pictures repeat."""

    def __init__(self, n: int, seed: int = 42, distinct: int = 256, quality: int = 90):
        import io
        from PIL import Image
        self.n = n
        self._raw = SyntheticRawImageDataset(max(n, 1), seed)
        self._jpeg = []
        for i in range(min(distinct, max(n, 1))):
            buf = io.BytesIO()
            Image.fromarray(self._raw[i][0].numpy()).save(buf, format="JPEG", quality=quality)
            self._jpeg.append(buf.getvalue())

    def __len__(self):
        return self.n

    def mean_jpeg_bytes(self) -> float:
        return sum(len(b) for b in self._jpeg) / max(len(self._jpeg), 1)

    def __getitem__(self, idx):
        import io
        from PIL import Image
        _, query, target, uid = self._raw._texts[idx]
        image = Image.open(io.BytesIO(self._jpeg[idx % len(self._jpeg)]))
        image.load()                                   # decode now (datasets' Image feature does the same on access)
        return {"image": image, "query_text": query, "target_text": target, "uuid": uid}


class SyntheticRawImageDataset(Dataset):
    """The same texts as :class:`SyntheticRetrievalDataset`, with the image as a camera-like uint8 ``[H, W, 3]`` array of a
    seeded size (the reference's dataset returns PIL images of whatever size the file has, clip_dataset.py:110-125): the
    input of the device-side preprocessing."""

    SIZES = ((375, 500), (500, 375), (480, 640), (333, 500), (256, 256), (600, 800), (224, 224), (427, 640))

    def __init__(self, n: int, seed: int = 42):
        self.n, self.seed = n, seed
        self._texts = SyntheticRetrievalDataset(n, 8, seed)

    def __len__(self):
        return self.n

    _bases = {}

    def _base(self, size):
        """One seeded camera-like picture per size (coarse random field upsampled by repetition, plus fine noise), made once
        per process: generating 600 KB of random pixels per item would make the benchmark measure numpy's generator."""
        import numpy as np
        key = (self.seed, size)
        if key not in self._bases:
            h, w = size
            rng = np.random.default_rng(self.seed * 7919 + h * 4099 + w)
            coarse = rng.integers(0, 256, size=((h + 15) // 16, (w + 15) // 16, 3), dtype=np.uint8)
            image = np.repeat(np.repeat(coarse, 16, axis=0), 16, axis=1)[:h, :w]
            self._bases[key] = np.ascontiguousarray(image ^ rng.integers(0, 32, size=(h, w, 1), dtype=np.uint8))
        return self._bases[key]

    def __getitem__(self, idx):
        import numpy as np
        rng = np.random.default_rng(self.seed * 1_000_003 + idx)
        size = self.SIZES[int(rng.integers(0, len(self.SIZES)))]
        # every item its own picture: the base rolled by an item-specific offset and XOR-ed with an item-specific byte
        image = np.roll(self._base(size), int(rng.integers(0, size[1])), axis=1) ^ np.uint8(rng.integers(0, 64))
        _, query, target, uid = self._texts[idx]
        return torch.from_numpy(np.ascontiguousarray(image)), query, target, uid


def collate_fn_eval(batch):
    images, queries, targets, uuids = zip(*batch)
    if torch.is_tensor(images[0]) and images[0].dtype == torch.uint8:      # raw [H, W, 3] images of any size (preprocess.RawRGB):
        from .preprocess import pack_raw                                   # one flat pinned-able buffer per batch,
        return pack_raw(images), list(queries), list(targets), list(uuids)    # preprocessed on the GPU by encode_dataset
    return torch.stack(images, dim=0), list(queries), list(targets), list(uuids)


collate_fn_train = collate_fn_eval


class CollateAndTokenize:
    """``collate_fn_eval`` + tokenisation of both text columns INSIDE the loader worker (evaluator.py:126,132 tokenises in
    the main process between two encoder calls): the main process then only queues copies and kernels."""

    def __init__(self, tokenize_fn):
        self.tokenize_fn = tokenize_fn

    # Workers started from a fork server (evaluators.loader_context) import this package afresh: the opt-ins the parent made with
    # allow_hash_tokenizer() / allow_random_weights() travel with the collate object instead of with forked module globals.
    def __getstate__(self):
        from . import clip_api, tokenizer
        return {"tokenize_fn": self.tokenize_fn, "allow_hash": tokenizer.hash_tokenizer_allowed(), "allow_random": clip_api.random_weights_allowed()}

    def __setstate__(self, state):
        self.tokenize_fn = state["tokenize_fn"]
        if state.get("allow_hash") or state.get("allow_random"):
            from . import clip_api, tokenizer
            if state.get("allow_hash"):
                tokenizer.allow_hash_tokenizer(True)
            if state.get("allow_random"):
                clip_api.allow_random_weights(True)

    def __call__(self, batch):
        images, queries, targets, uuids = collate_fn_eval(batch)
        return images, self.tokenize_fn(queries), self.tokenize_fn(targets), uuids


def collate_fn_eval_texts(batch):
    queries, targets = zip(*batch)
    return list(queries), list(targets)
