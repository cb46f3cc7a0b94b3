"""Evaluation loops and CLIs of the reference, device-resident.

Counterparts of /root/reference/src/clip/eval/evaluator.py (``evaluate_clip_model`` :53-221,
``evaluate_clip_model_for_training`` :224-257, ``main`` :260-390) and evaluator_baseline.py (``evaluate_clip_model``
:38-146, ``main`` :150-274), with the defects listed in SURVEY.md section 3.5 routed around (4-tuple batches,
``--splits_file`` accepted, ``val`` -> ``validation``, import-time side effects made lazy).

What changes on the hot path: the reference copies every batch of embeddings back to the host
(``.cpu().numpy()``, evaluator.py:123,129,135) and ranks with numpy; here the three embedding sets stay in HBM and go
straight into the fused similarity / rank kernels, L2 normalisation is fused into the encoder tail.
"""
from __future__ import annotations

import argparse
import logging
import os
import random
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import _lib
from . import engine
from . import metrics as M
from . import sparql_fusion as SF
from .datasets import CLIPEvalDatasetHF, CollateAndTokenize, SyntheticHFSplit, SyntheticRawImageDataset, SyntheticRetrievalDataset, collate_fn_eval
from .preprocess import ClipPreprocessGPU, PackedRaw
from .logging_utils import save_metrics_to_json, setup_logger

logger = logging.getLogger(__name__)

TEXT2SPARQL_DIR = "experiments/text2sparql/results"
ALPHAS = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1]


def seed_worker(worker_id):
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


def load_text2sparql_results(directory: str = TEXT2SPARQL_DIR) -> Dict[str, List[str]]:
    """uuid -> list of artefact URIs, one file per query (reference evaluator.py:43-50, but lazy and tolerant)."""
    out: Dict[str, List[str]] = {}
    if not os.path.isdir(directory):
        return out
    for fn in os.listdir(directory):
        with open(os.path.join(directory, fn), "r") as f:
            out[fn.split(".")[0]] = [line.strip() for line in f.readlines()]
    return out


def default_tokenize(texts: Sequence[str]) -> torch.Tensor:
    from .tokenizer import tokenize
    return tokenize(list(texts), truncate=True)


ENCODE_ITEMS = 255          # items per encoder call in encode_dataset (see there)


def default_loader_workers() -> int:
    """Loader processes of the drop-in CLIs (the reference's scripts leave the DataLoader at 4, evaluator_baseline.py:83-92, or at
    0, evaluator.py:97-106): KEMR_LOADER_WORKERS, else min(12, half the cores this process may use).  With the image transform on
    the GPU a worker only decodes and tokenises; 12 of them deliver 17 k items/s, 0 leaves 1.5 k on the consumer."""
    v = os.environ.get("KEMR_LOADER_WORKERS", "")
    if v.strip():
        return max(0, int(v))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(0, min(12, cores // 2))


_FORKSERVER_PRELOAD = ["torch", "numpy", "PIL.Image", "knowledge_enhanced_multimodal_retrieval_amd.datasets",
                       "knowledge_enhanced_multimodal_retrieval_amd.preprocess", "knowledge_enhanced_multimodal_retrieval_amd.tokenizer",
                       "knowledge_enhanced_multimodal_retrieval_amd.evaluators"]


def loader_context():
    """Start method of the loader processes: "forkserver" (KEMR_LOADER_CONTEXT overrides: fork | spawn | forkserver).
    The evaluators' process has an initialised HIP context, GPU-mapped pinned buffers and gigabytes of weights by the time the
    loader starts; workers forked from IT share all of that copy-on-write.  Measured (round 3, tools/probe_pipeline_state.py, same
    box, same 4 080 items): 4.2 k items/s when the loader forks right after clip.load, 1.2-1.6 k with one more piece of state in
    front of the fork, 0.5 k after an encoder engine has run -- the consumer then blocks ~300 ms per loader batch inside its
    launches although the device work of the call takes 0.9 ms.  Workers forked from a clean fork server (torch and this package
    pre-imported, no HIP state -- the library makes no HIP call at load time) do not touch the GPU process's address space."""
    import multiprocessing as mp
    method = os.environ.get("KEMR_LOADER_CONTEXT", "forkserver")
    ctx = mp.get_context(method)
    if method == "forkserver":
        ctx.set_forkserver_preload(_FORKSERVER_PRELOAD)          # takes effect when the server starts (first use in this process)
    return ctx


def _local_world_size() -> int:
    for key in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(key, "")
        if v.strip().isdigit() and int(v) > 0:
            return int(v)
    return 1


def usable_loader_workers(dataset, collate, num_workers: int) -> int:
    """Worker processes the loader can actually be given.  They come from a fork server (loader_context), which PICKLES the
    dataset and the collate object; the reference's loaders fork (evaluator_baseline.py:83-92) and therefore accept lambdas,
    closures and in-memory objects.  A caller who passes such a dataset / tokenize_fn gets num_workers = 0 and a warning
    instead of a crash inside DataLoader; under torchrun the count is divided by the local world size."""
    if num_workers <= 0:
        return 0
    num_workers = max(1, num_workers // _local_world_size())
    if os.environ.get("KEMR_LOADER_CONTEXT", "forkserver") == "fork":
        return num_workers
    import pickle

    class _Discard:                                            # a big in-memory split is serialised once here (as it will be once per
        def write(self, b):                                    # worker by the loader), but never held: the bytes are dropped
            return len(b)

    try:
        pickle.Pickler(_Discard(), protocol=pickle.HIGHEST_PROTOCOL).dump((collate, dataset))
    except Exception as e:                                     # noqa: BLE001 - anything that cannot cross to a fresh process
        logger.warning(f"loader workers need a picklable dataset / tokenize_fn ({type(e).__name__}: {e}); using num_workers=0")
        return 0
    return num_workers


def eval_loader(dataset, batch_size: int, seed: int, num_workers: int, tokenize_fn: Callable, pin: bool) -> DataLoader:
    """The evaluation DataLoader (evaluator.py:96-105: no shuffle, seeded workers), with tokenisation and the packing of raw
    images into one buffer moved INTO the loader (its worker processes when there are any); the pin thread pins both.
    What is left on the consumer's thread per loader batch: three asynchronous copies and the kernel launches."""
    g = torch.Generator()
    g.manual_seed(seed)
    num_workers = usable_loader_workers(dataset, CollateAndTokenize(tokenize_fn), num_workers)
    return DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=pin,
                      collate_fn=CollateAndTokenize(tokenize_fn), worker_init_fn=seed_worker if num_workers else None,
                      generator=g, prefetch_factor=4 if num_workers else None,
                      multiprocessing_context=loader_context() if num_workers else None)


@torch.no_grad()
def encode_dataset(model, dataset, batch_size: int = 64, seed: int = 42, num_workers: Optional[int] = 0,
                   tokenize_fn: Optional[Callable] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[str]]:
    """Hot loop A: (image, query, target) -> three L2-normalised embedding sets that stay on the GPU.
    num_workers None = default_loader_workers()."""
    if num_workers is None:
        num_workers = default_loader_workers()
    model.eval()
    device = next(model.parameters()).device
    tokenize_fn = tokenize_fn or default_tokenize
    loader = eval_loader(dataset, batch_size, seed, num_workers, tokenize_fn, pin=device.type == "cuda")
    img, qry, tgt, uuids = [], [], [], []
    logger.info(f"Computing embeddings for {len(dataset)} samples...")
    # The loader's batch size is the host pipeline's business (the reference scripts pass 64); the encoders are fed the number
    # of items per call that fills the persistent GEMM's rounds (engine.tile_friendly_batch), whatever it is: 255 images of
    # ViT-L/14 are 65 535 token rows = 256 row tiles, while e.g. 64 images are 65 row tiles -> 260 tiles on 256 CUs, a second
    # round for 4 tiles; texts go 850 to a call (queries and targets of 425 items together: 256 row tiles, whole rounds in every
    # GEMM; 255 texts leave 10 % of the out-proj round empty).  Rows are independent, so the embeddings do
    # not depend on the grouping.
    arch = getattr(model, "arch", None)
    n_img = ENCODE_ITEMS if arch is None else engine.tile_friendly_batch(arch.v_tokens, arch.v_width, ENCODE_ITEMS // 2, ENCODE_ITEMS)
    n_txt = ENCODE_ITEMS if arch is None else max(1, engine.tile_friendly_batch(arch.ctx, arch.t_width, ENCODE_ITEMS, engine.MAX_TEXT_BATCH) // 2)
    pend_i, pend_q, pend_t, pend_ql, pend_tl = [], [], [], [], []
    count = {"i": 0, "t": 0, "rows": 0}
    # (Round 4 measured the text calls on a SIDE stream beside the image calls -- tools/bench_two_streams.py, profiles/r04_two_streams.txt:
    # +3.2 % items/s when every image call has a text call beside it, +0.7 % in the bench's mix of one text call per three image calls,
    # consecutive image calls on two streams +1.4 % -- and left it out: the persistent GEMM takes every CU's LDS, so only the HBM-bound
    # kernels of the other tower fit beside it, and a gain of that size does not pay for two-stream bookkeeping in every caller.)
    # Packed text calls (engine.ClipEngine.encode_text: a text is computed up to its end-of-text token only): the tokenizer-side
    # lengths travel with the ids, and a call takes as many (query, target) pairs as fill engine.TEXT_ROW_BUDGET token rows.
    by_rows = bool(getattr(model, "accepts_text_lengths", False)) and arch is not None and arch.ctx <= 128 and \
        os.environ.get("KEMR_TEXT_PACKED", "1") != "0"

    def flush_images(final: bool):
        take = count["i"] if final else (count["i"] // n_img) * n_img              # whole encoder calls; the rest waits
        if take <= 0:
            return
        ci = torch.cat(pend_i)
        img.append(model.encode_image(ci[:take], normalize=True))
        pend_i[:] = [ci[take:]]
        count["i"] -= take

    def flush_texts(final: bool):
        if by_rows:
            if count["t"] <= 0:
                return
            cq, ct, lq, lt = torch.cat(pend_q), torch.cat(pend_t), torch.cat(pend_ql), torch.cat(pend_tl)
            pair_rows = torch.cumsum((lq + lt).to(torch.int64), 0)
            s0, base = 0, 0
            while s0 < count["t"]:
                s1 = s0 + max(1, int((pair_rows[s0:] - base <= engine.TEXT_ROW_BUDGET).sum()))
                if s1 >= count["t"] and not final:
                    break                                                          # not a whole call yet: wait for more
                both = model.encode_text(torch.cat([cq[s0:s1], ct[s0:s1]]), normalize=True, lens=torch.cat([lq[s0:s1], lt[s0:s1]]))
                qry.append(both[: s1 - s0])
                tgt.append(both[s1 - s0:])
                base, s0 = int(pair_rows[s1 - 1]), s1
            pend_q[:], pend_t[:], pend_ql[:], pend_tl[:] = [cq[s0:]], [ct[s0:]], [lq[s0:]], [lt[s0:]]
            count["t"] -= s0
            count["rows"] = int(pair_rows[-1]) - base if s0 < pair_rows.numel() else 0
            return
        take = count["t"] if final else (count["t"] // n_txt) * n_txt
        if take <= 0:
            return
        cq, ct = torch.cat(pend_q), torch.cat(pend_t)
        for s0 in range(0, take, n_txt):                                           # queries and targets of n_txt items in ONE call
            s1 = min(take, s0 + n_txt)
            both = model.encode_text(torch.cat([cq[s0:s1], ct[s0:s1]]), normalize=True)
            qry.append(both[: s1 - s0])
            tgt.append(both[s1 - s0:])
        pend_q[:], pend_t[:] = [cq[take:]], [ct[take:]]
        count["t"] -= take

    gpu_pre = None
    for images, q_ids, t_ids, ids in loader:
        if isinstance(images, PackedRaw):      # raw uint8 images (preprocess.RawRGB): resize / crop / normalise on the device,
            if gpu_pre is None:                # one launch pair per loader batch
                gpu_pre = ClipPreprocessGPU(int(getattr(getattr(model, "visual", None), "input_resolution", 224)), device)
            images = gpu_pre.batch(images)
        pend_i.append(images.to(device, non_blocking=True))
        if by_rows:                            # lengths from the host copy, before the upload
            lq_, lt_ = engine.text_lengths(q_ids.cpu()), engine.text_lengths(t_ids.cpu())
            pend_ql.append(lq_)
            pend_tl.append(lt_)
            count["rows"] += int(lq_.sum()) + int(lt_.sum())
        pend_q.append(q_ids.to(device, non_blocking=True))
        pend_t.append(t_ids.to(device, non_blocking=True))
        count["i"] += int(images.shape[0])
        count["t"] += int(images.shape[0])
        uuids.extend(ids)
        if count["i"] >= n_img:
            flush_images(False)
        if (count["rows"] > engine.TEXT_ROW_BUDGET) if by_rows else (count["t"] >= n_txt):
            flush_texts(False)
    flush_images(True)
    flush_texts(True)
    return torch.cat(img), torch.cat(qry), torch.cat(tgt), uuids


def sparql_sweep(query, target, image, uuids, text2sparql_results, t2i_weight, t2t_weight) -> Dict[str, Dict[str, float]]:
    """T2I / T2T / fused metrics + the 9-alpha weighted SPARQL fusion of evaluator.py:164-218, every variant one fused
    kernel pass (no N x N matrix, no dense 0/1 hit matrix)."""
    out = {"T2I": M.compute_retrieval_metrics(query, image), "T2T": M.compute_retrieval_metrics(query, target),
           "Fused": M.compute_retrieval_metrics_final(query, target, image, t2i_weight=t2i_weight, t2t_weight=t2t_weight)}
    for alpha in ALPHAS:
        out[f"weighted_alpha={alpha}"] = SF.fused_metrics(
            [query, query], [image, target], [t2i_weight, t2t_weight], text2sparql_results, uuids, uuids, "weighted",
            {"alpha": alpha, "sparql_weight": 1 - alpha})
    return out


@torch.no_grad()
def evaluate_clip_model(model, dataset, batch_size: int = 64, device: str = "cuda", seed: int = 42,
                        tasks: List[str] = ("T2I", "I2T", "T2T"), compute_recall: bool = True, compute_mrr: bool = True,
                        tokenize_fn: Optional[Callable] = None, num_workers: Optional[int] = 0,
                        text2sparql_results: Optional[Dict[str, List[str]]] = None, analysis: bool = True
                        ) -> Dict[str, float]:
    """evaluator.py semantics: per-task metrics are returned; the fused / SPARQL-sweep analysis is logged and kept in
    ``evaluate_clip_model.last_analysis``.  ``num_workers`` defaults to the reference's 0 (evaluator.py:101); the CLIs opt into
    ``default_loader_workers()``, a library caller passes a number (None = that default)."""
    image, query, target, uuids = encode_dataset(model, dataset, batch_size, seed, num_workers, tokenize_fn)
    logger.info(f"Image embeddings: {tuple(image.shape)}")
    logger.info(f"Query embeddings: {tuple(query.shape)}")
    logger.info(f"Target embeddings: {tuple(target.shape)}")
    result = M.compute_all_retrieval_metrics(query, target, image, tasks=tasks, compute_recall=compute_recall,
                                             compute_mrr=compute_mrr)
    evaluate_clip_model.last_analysis = {}
    if analysis and compute_recall:
        results = load_text2sparql_results() if text2sparql_results is None else text2sparql_results
        for wi, wt in ((0.5, 0.5), (0.1, 0.9)):
            sweep = sparql_sweep(query, target, image, uuids, results, wi, wt)
            evaluate_clip_model.last_analysis[f"{wi}_{wt}"] = sweep
            for name, m in sweep.items():
                logger.info(f"[t2i={wi} t2t={wt}] {name}: " + ", ".join(f"{k}={v:.2f}" for k, v in m.items()))
    return result


evaluate_clip_model.last_analysis = {}


@torch.no_grad()
def evaluate_clip_model_for_training(model, dataset, batch_size: int = 64, device: str = "cuda", seed: int = 42,
                                     tasks: List[str] = ("T2I", "I2T", "T2T"), **kw) -> Dict[str, float]:
    return evaluate_clip_model(model, dataset, batch_size, device, seed, tasks, compute_recall=False, compute_mrr=True, **kw)


@torch.no_grad()
def evaluate_clip_model_baseline(model, dataset, batch_size: int = 64, device: str = "cuda", seed: int = 42,
                                 tasks: List[str] = ("T2I", "I2T", "T2T"), compute_recall: bool = True,
                                 compute_mrr: bool = True, t2i_weight: float = 0.5, t2t_weight: float = 0.5,
                                 tokenize_fn: Optional[Callable] = None, num_workers: Optional[int] = 4) -> Dict[str, float]:
    """evaluator_baseline.py semantics: metrics of the fused score w_i * T2I + w_t * T2T (un-prefixed keys); 4 loader workers
    as there (evaluator_baseline.py:87)."""
    image, query, target, _ = encode_dataset(model, dataset, batch_size, seed, num_workers, tokenize_fn)
    return M.compute_retrieval_metrics_final(query, target, image, compute_recall=compute_recall, compute_mrr=compute_mrr,
                                             t2i_weight=t2i_weight, t2t_weight=t2t_weight)


# ------------------------------------------------------------------------------------------------ CLIs
def _common_args(parser, baseline: bool):
    parser.add_argument("--model_name", type=str, default="ViT-L/14", choices=["ViT-B/32", "ViT-B/16", "ViT-L/14"])
    parser.add_argument("--checkpoint", type=str, help="Path to checkpoint, if None uses pretrained model")
    parser.add_argument("--images_dir", type=str, default=None)
    parser.add_argument("--texts_dir", type=str, default=None, help="Directory containing query-target JSON files")
    parser.add_argument("--split", type=str, default="test", choices=["train", "val", "test"])
    parser.add_argument("--splits_file", type=str, default=None, help="accepted for the shipped scripts; unused")
    parser.add_argument("--tasks", type=str, nargs="+", default=["T2I", "I2T", "T2T"], choices=["T2I", "I2T", "T2T"])
    parser.add_argument("--mrr_only", action="store_true", help="Only compute MRR (faster, for training validation)")
    parser.add_argument("--batch_size", type=int, default=32)
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--output_file", type=str, required=True)
    parser.add_argument("--seed", type=int, default=42)
    parser.add_argument("--synthetic", type=int, default=0, metavar="N",
                        help="evaluate on N seeded synthetic items instead of the HuggingFace dataset (offline)")
    parser.add_argument("--synthetic_images", type=str, default="float", choices=["float", "uint8"],
                        help="synthetic items: 'float' = already normalised pixel tensors; 'uint8' = camera-sized PIL images through "
                             "CLIPEvalDatasetHF(split, preprocess), the reference's own dataset call (image transform on the GPU "
                             "unless KEMR_GPU_PREPROCESS=0)")
    parser.add_argument("--num_workers", type=int, default=-1, help="DataLoader workers (-1: KEMR_LOADER_WORKERS or min(12, cores / 2))")
    parser.add_argument("--dataset", type=str, default="xuemduan/reevaluate-image-text-pairs")
    if baseline:
        parser.add_argument("--t2i_weight", type=float, default=0.5)
        parser.add_argument("--t2t_weight", type=float, default=0.5)


def _run(args, baseline: bool, log_name: str):
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    out_dir = Path(args.output_file).parent
    out_dir.mkdir(parents=True, exist_ok=True)
    setup_logger(log_name, str(out_dir / "evaluation.log"))
    log = logging.getLogger(log_name)
    log.info("=" * 80)
    log.info("CLIP Model Evaluation (MI355X HIP engine)")
    log.info(f"Model: {args.model_name}  Checkpoint: {args.checkpoint}  Split: {args.split}  Tasks: {args.tasks}  "
             f"MRR only: {args.mrr_only}  Seed: {args.seed}")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the encoders and the ranking run only in the HIP kernels (no CPU fallback)")
    device = args.device if args.device.startswith("cuda") else "cuda"
    from . import clip_api, tokenizer
    from .clip_model import load_clip_model
    if args.synthetic > 0:           # synthetic data: seeded random weights / hash token ids are acceptable, and recorded below
        clip_api.allow_random_weights(True)
        tokenizer.allow_hash_tokenizer(True)
    model, preprocess = load_clip_model(model_name=args.model_name, checkpoint_path=args.checkpoint, device=device)
    if args.synthetic > 0 and args.synthetic_images == "float":
        dataset = SyntheticRetrievalDataset(args.synthetic, model.arch.image_size, args.seed)
    elif args.synthetic > 0:
        dataset = CLIPEvalDatasetHF(hf_dataset=SyntheticHFSplit(args.synthetic, args.seed), preprocessor=preprocess)
    else:
        from datasets import load_dataset
        ds = load_dataset(args.dataset)
        split = {"val": "validation"}.get(args.split, args.split)
        dataset = CLIPEvalDatasetHF(hf_dataset=ds[split], preprocessor=preprocess)        # the reference's call (evaluator.py:330-333)
    workers = default_loader_workers() if args.num_workers < 0 else args.num_workers
    log.info(f"Image transform: {'GPU (batched, bit-identical)' if getattr(preprocess, 'defer_to_gpu', False) else 'host (PIL, per sample)'}; loader workers: {workers}")
    if baseline:
        metrics = evaluate_clip_model_baseline(model, dataset, args.batch_size, device, args.seed, args.tasks,
                                               compute_recall=not args.mrr_only, compute_mrr=True,
                                               t2i_weight=args.t2i_weight, t2t_weight=args.t2t_weight, num_workers=workers)
    else:
        metrics = evaluate_clip_model(model, dataset, args.batch_size, device, args.seed, args.tasks,
                                      compute_recall=not args.mrr_only, compute_mrr=True, num_workers=workers)
    log.info("EVALUATION RESULTS")
    for name, value in sorted(metrics.items()):
        log.info(f"{name}: {value:.2f}" + ("" if "Mean_Rank" in name else "%"))
    results = {"model_name": args.model_name, "checkpoint": args.checkpoint, "split": args.split,
               "num_samples": len(dataset), "seed": args.seed, "metrics": metrics,
               # provenance (not in the reference's file): what the numbers were computed with
               "weights_source": getattr(model, "weights_source", "unknown"), "tokenizer": tokenizer.tokenizer_name(),
               "precision": os.environ.get("KEMR_PRECISION", _lib.DEFAULT_PRECISION), "data": "synthetic" if args.synthetic > 0 else args.dataset,
               "image_transform": {"RawRGB": "gpu (batched kernels, bit-identical to the host transform)", "ClipPreprocess": "host (PIL, per sample)"}.get(
                   type(getattr(dataset, "preprocessor", None)).__name__, "none (pre-normalised tensors)"),
               "loader_workers": workers}
    if not baseline:
        results["tasks"] = list(args.tasks)
    save_metrics_to_json(results, args.output_file)
    log.info(f"Results saved to {args.output_file}")
    return results


def main_evaluator(argv=None):
    parser = argparse.ArgumentParser(description="Evaluate CLIP model")
    _common_args(parser, baseline=False)
    return _run(parser.parse_args(argv), baseline=False, log_name="src.clip.eval.evaluator")


def main_baseline(argv=None):
    parser = argparse.ArgumentParser(description="Evaluate CLIP model (fused T2I + T2T score)")
    _common_args(parser, baseline=True)
    return _run(parser.parse_args(argv), baseline=True, log_name="src.clip.eval.evaluator_baseline")


# ------------------------------------------------------------------------------------------------ learned fusion heads
@torch.no_grad()
def evaluate_fusion_model(fusion_model, dataset, batch_size: int = 64, device: str = "cuda", seed: int = 42,
                          tokenize_fn: Optional[Callable] = None, num_workers: Optional[int] = 4) -> Dict[str, float]:
    """Counterpart of /root/reference/src/clip/eval/evaluator_fusion.py:28-144 (4 loader workers as at :42).  The reference fills an N x N numpy
    matrix in 50 x 500 blocks with an H2D/D2H round trip and ``empty_cache()`` per block (:76-121); here the head's
    score is one fused kernel pass over resident embeddings (``FusionModel.rank``)."""
    fusion_model.eval()
    image, query, target, _ = encode_dataset(fusion_model.clip_model, dataset, batch_size, seed, num_workers, tokenize_fn)
    ranks, _, _ = fusion_model.rank(query, image, target, k=0)
    from . import ranking
    result = ranking.metrics_from_ranks(ranks, [1, 5, 10, 20])
    logger.info("Fusion Model Evaluation Results")
    for k, v in result.items():
        logger.info(f"{k}: {v:.2f}" + ("%" if ("R@" in k or "MRR" in k) else ""))
    return result


def main_fusion(argv=None):
    import json
    parser = argparse.ArgumentParser(description="Evaluate Fusion Model")
    parser.add_argument("--model_name", type=str, default="ViT-L/14")
    parser.add_argument("--clip_checkpoint", type=str, default=None)
    parser.add_argument("--fusion_checkpoint", type=str, default=None)
    parser.add_argument("--fusion_type", type=str, required=True,
                        choices=["linear", "cross_attention", "gated", "simple_gated", "simple_gated_with_bias", "bilinear"])
    parser.add_argument("--images_dir", type=str, default=None)
    parser.add_argument("--texts_dir", type=str, default=None)
    parser.add_argument("--splits_file", type=str, default=None)
    parser.add_argument("--split", type=str, default="test", choices=["train", "val", "test"])
    parser.add_argument("--max_text_length", type=int, default=150)
    parser.add_argument("--batch_size", type=int, default=64)
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--output_file", type=str, default=None)
    parser.add_argument("--synthetic", type=int, default=0, metavar="N")
    parser.add_argument("--dataset", type=str, default="xuemduan/reevaluate-image-text-pairs")
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    from . import clip_api, tokenizer
    from .clip_model import load_clip_model
    from .fusion_model import FusionModel
    if args.synthetic > 0:
        clip_api.allow_random_weights(True)
        tokenizer.allow_hash_tokenizer(True)
    clip_model, preprocess = load_clip_model(model_name=args.model_name, checkpoint_path=args.clip_checkpoint, device=args.device)
    embed_dim = 768 if "L/14" in args.model_name else 512
    fusion_model = FusionModel(clip_model=clip_model, fusion_type=args.fusion_type, embed_dim=embed_dim).to(args.device)
    if args.fusion_checkpoint:
        ckpt = torch.load(args.fusion_checkpoint, map_location="cpu", weights_only=True)
        fusion_model.fusion_head.load_state_dict(ckpt["fusion_head_state_dict"])
    if args.synthetic > 0:
        dataset = SyntheticRetrievalDataset(args.synthetic, clip_model.arch.image_size)
    else:
        from datasets import load_dataset
        ds = load_dataset(args.dataset)
        dataset = CLIPEvalDatasetHF(ds[{"val": "validation"}.get(args.split, args.split)], preprocess, args.max_text_length)
    result = evaluate_fusion_model(fusion_model, dataset, args.batch_size, args.device)
    results = {"model_name": args.model_name, "clip_checkpoint": args.clip_checkpoint,
               "fusion_checkpoint": args.fusion_checkpoint, "fusion_type": args.fusion_type, "split": args.split,
               "num_samples": len(dataset), "metrics": result,
               "weights_source": getattr(clip_model, "weights_source", "unknown"), "tokenizer": tokenizer.tokenizer_name(),
               "precision": os.environ.get("KEMR_PRECISION", _lib.DEFAULT_PRECISION), "data": "synthetic" if args.synthetic > 0 else args.dataset}
    if args.output_file:
        Path(args.output_file).parent.mkdir(parents=True, exist_ok=True)
        with open(args.output_file, "w") as f:
            json.dump(results, f, indent=2)
    else:
        print(json.dumps(results, indent=2))
    return results
