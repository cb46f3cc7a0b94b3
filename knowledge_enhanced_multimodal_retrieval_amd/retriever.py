"""Online retrieval: resident gallery embeddings + single-query search, and the reference's ``RetrievalEngine`` API.

Reference call stack: ``RetrievalEngine.retrieve_text`` (/root/reference/src/retrieval.py:79-95)
-> ``CLIPRetrieval.retrieval`` (src/clip/clip_retrieval.py:39-40) -> ``retriever.search(query, alpha)`` of a
``CLIPRetriever`` whose source the reference downloads from the HF Hub and ``exec``s
(clip_retrieval.py:15-37) -- that file is not in the reference checkout, so ``search``'s exact semantics are
UNPINNED.  This build's local ``CLIPRetriever`` takes the documented argument names at face value:
``score = alpha * <q, image_i> + (1 - alpha) * <q, target_text_i>`` over gallery embeddings precomputed under
``data/embeddings`` (the reference's ``local_embeddings_dir``), best ``top_k`` first.

Store format (this build's definition): ``<dir>/image_embeddings.npy`` and ``<dir>/text_embeddings.npy``
(float32 [N, D], L2-normalised rows) + ``<dir>/uuids.json`` (list of N strings).  ``EmbeddingStore.load`` puts both
sets into HBM as one bf16 split panel ([image ; text] along k), built once; a query is one fused kernel pass.

The SPARQL side (Mistral + GraphDB over HTTP, src/text2sparql/*) is out of scope; ``RetrievalEngine`` takes any object
with ``retrieval(query) -> List[uuid]`` and defaults to one that returns no hits.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, engine, ranking

MAX_TOP_K = 32


class EmbeddingStore:
    """Gallery embeddings resident in HBM, ready for the fused similarity kernel."""

    def __init__(self, image_embeddings, text_embeddings, uuids: Sequence[str], device=None, precision: str = "fp32x3"):
        self.uuids = list(uuids)
        self.image = ranking.to_device_f32(image_embeddings, device)
        self.text = ranking.to_device_f32(text_embeddings, self.image.device)
        if self.image.shape != self.text.shape or self.image.shape[0] != len(self.uuids):
            raise ValueError("image / text embeddings and uuids must describe the same N items")
        self.precision = precision
        self.panel = engine.build_panel([self.image, self.text], _lib.SIDE_GALLERY, ranking.PRECISION_TERMS[precision])

    def __len__(self):
        return len(self.uuids)

    @property
    def dim(self) -> int:
        return self.image.shape[1]

    def save(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        np.save(os.path.join(directory, "image_embeddings.npy"), self.image.cpu().numpy())
        np.save(os.path.join(directory, "text_embeddings.npy"), self.text.cpu().numpy())
        with open(os.path.join(directory, "uuids.json"), "w") as f:
            json.dump(self.uuids, f)

    @classmethod
    def load(cls, directory: str, device=None, precision: str = "fp32x3") -> "EmbeddingStore":
        img = np.load(os.path.join(directory, "image_embeddings.npy"), allow_pickle=False)
        txt = np.load(os.path.join(directory, "text_embeddings.npy"), allow_pickle=False)
        with open(os.path.join(directory, "uuids.json")) as f:
            uuids = json.load(f)
        return cls(img, txt, uuids, device, precision)

    @classmethod
    def build(cls, model, dataset, batch_size: int = 64, tokenize_fn=None, precision: str = "fp32x3") -> "EmbeddingStore":
        """Encode a dataset of (image, query, target, uuid) once; the gallery keeps image and TARGET-text embeddings."""
        from .evaluators import encode_dataset
        image, _, target, uuids = encode_dataset(model, dataset, batch_size, tokenize_fn=tokenize_fn)
        return cls(image, target, uuids, image.device, precision)


class CLIPRetriever:
    """Local counterpart of the reference's remote ``CLIPRetriever`` (see module docstring: semantics unpinned)."""

    def __init__(self, model, store: EmbeddingStore, tokenize_fn=None):
        self.model, self.store = model, store
        if tokenize_fn is None:
            from .evaluators import default_tokenize
            tokenize_fn = default_tokenize
        self.tokenize_fn = tokenize_fn

    @classmethod
    def from_pretrained(cls, repo_id: str = "xuemduan/reevaluate-clip-retriever", local_embeddings_dir: str = "data/embeddings",
                        model_name: Optional[str] = None, token: Optional[str] = None, device: str = "cuda", **_):
        """Nothing is fetched: ``repo_id`` / ``token`` are accepted for signature compatibility; weights come from
        ``clip.load`` (see clip_api.py) and the gallery from ``local_embeddings_dir``."""
        from . import clip_api
        model, _ = clip_api.load(model_name or "ViT-L/14", device=device)
        return cls(model, EmbeddingStore.load(local_embeddings_dir, device))

    @torch.no_grad()
    def search_batch(self, queries: Sequence[str], alpha: float = 0.5, top_k: int = 10):
        if not 1 <= top_k <= MAX_TOP_K:
            raise ValueError(f"top_k must be in 1..{MAX_TOP_K}")
        ids = self.tokenize_fn(list(queries))          # host ids: the engine takes the text lengths from them before the upload (no device sync)
        q = self.model.encode_text(ids, normalize=True)
        qp = engine.build_panel([q, q], _lib.SIDE_QUERY, ranking.PRECISION_TERMS[self.store.precision],
                                part_scale=[alpha, 1.0 - alpha])
        return engine.sim_topk(qp, self.store.panel, min(top_k, len(self.store)))

    def search(self, query: str, alpha: float = 0.5, top_k: int = 10) -> List[Dict]:
        scores, idx = self.search_batch([query], alpha, top_k)
        scores, idx = scores[0].cpu().tolist(), idx[0].cpu().tolist()
        return [{"uuid": self.store.uuids[i], "score": float(s)} for s, i in zip(scores, idx) if i >= 0]


class CLIPRetrieval:
    """``CLIPRetrieval(model_name=None).retrieval(query, alpha=0.5)`` (reference src/clip/clip_retrieval.py:10-40),
    without the hub download / ``exec`` / ``login``."""

    def __init__(self, model_name=None, retriever: Optional[CLIPRetriever] = None, embeddings_dir: str = "data/embeddings"):
        self.retriever = retriever or CLIPRetriever.from_pretrained(
            "xuemduan/reevaluate-clip-retriever", local_embeddings_dir=embeddings_dir, model_name=model_name)

    def retrieval(self, query: str, alpha: float = 0.5):
        return self.retriever.search(query, alpha=alpha)


class NoText2SPARQL:
    """Placeholder for the out-of-scope LLM + SPARQL retriever: never any hit."""

    def retrieval(self, query: str) -> List[str]:
        return []


class RetrievalEngine:
    """Same public methods as the reference (src/retrieval.py:11-107); both retrievers are injectable."""

    def __init__(self, clip_retriever=None, t2s_retriever=None):
        self.clip_retriever = clip_retriever or CLIPRetrieval()
        self.t2s_retriever = t2s_retriever or NoText2SPARQL()
        self.cir_endpoint = os.getenv("CIR_ENDPOINT")
        self.cir_headers = {"accept": "application/json", "X-API-Key": os.getenv("CIR_ENDPOINT_KEY")}

    def _fuse_clip_sparql_linear(self, clip_results: List[Dict], sparql_results: List[str], alpha: float = 0.8,
                                 beta: float = 0.2) -> List[Dict]:
        """score = round(alpha * clip + beta * [uuid in sparql], 4), best first (stable), no normalisation."""
        if not clip_results:
            return []
        hits = set(sparql_results)
        fused = [{"uuid": it["uuid"], "score": round(alpha * it["score"] + beta * (1.0 if it["uuid"] in hits else 0.0), 4)}
                 for it in clip_results]
        fused.sort(key=lambda x: x["score"], reverse=True)
        return fused

    def retrieve_text(self, query: str, alpha: float = 0.8, beta: float = 0.2, alpha_clip: float = 0.5, threshold: float = 0):
        clip_results = self.clip_retriever.retrieval(query, alpha=alpha_clip)
        t2s_results = self.t2s_retriever.retrieval(query)
        fused = self._fuse_clip_sparql_linear(clip_results=clip_results, sparql_results=t2s_results, alpha=alpha, beta=beta)
        return [{"uuid": it["uuid"], "score": it["score"]} for it in fused if it.get("score", 0) >= threshold]

    def retrieve_text_noknowledge(self, query: str, alpha: float = 0.8, beta: float = 0.2, alpha_clip: float = 0.5,
                                  threshold: float = 0):
        results = self.clip_retriever.retrieval(query, alpha=alpha_clip)
        return [{"uuid": it["uuid"], "score": it["score"]} for it in results if it.get("score", 0) >= threshold]
