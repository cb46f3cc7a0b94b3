#!/usr/bin/env python3
"""Headline benchmark: CLIP ViT-L/14 gallery encode (images + texts / s) and Q x 43k similarity + top-10 (ms).

    python bench.py                      # N = 1, the whole 43k gallery: ceil(43000 / 255) = 169 steps, ~9 s timed
    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] ("CLIP ViT-L/14 zero-shot, 43k gallery encode + T2I top-10 on 1xMI355X bf16"):
one step = one gallery batch through the hot path exactly as the reference's eval loop encodes it
(evaluator_baseline.py:100-121): `batch` images [B,3,224,224] + B query texts + B target texts -> three sets of
L2-normalised embeddings, resident in HBM.  Synthetic inputs and seeded random-init weights of the ViT-L/14
architecture (no network).  `value` = (images + texts) encoded per second over all ranks (weak scaling: every rank
encodes its own `batch` gallery items per step).  After the timed region the same process measures
  * the fused similarity + top-10 kernel on the 43k gallery (sharded over the ranks) for Q = 1024 and Q = 43000,
  * per-kernel-class time with hipEvents on the launch stream (roofline of the dominant kernel, the bf16 GEMM),
  * on rank 0 at N = 1: the CPU oracle on a bounded sample of the same workload (cpu_baseline).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from knowledge_enhanced_multimodal_retrieval_amd import _lib, engine  # noqa: E402
from knowledge_enhanced_multimodal_retrieval_amd.config import ARCHS  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
GALLERY = 43000


def random_weights(arch, seed=0):
    """Seeded N(0, std) weights in OpenAI-CLIP naming (same shapes/stds as oracle.clip_ref.random_state_dict,
    restated here so that the product path does not import the oracle)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std: torch.randn(*s, generator=g) * std
    sd = {}
    vw, tw, D, p = arch.v_width, arch.t_width, arch.embed_dim, arch.patch
    sd["visual.conv1.weight"] = rn(vw, 3, p, p, std=(3 * p * p) ** -0.5)
    sd["visual.class_embedding"] = rn(vw, std=vw ** -0.5)
    sd["visual.positional_embedding"] = rn(arch.v_tokens, vw, std=vw ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = 1.0 + rn(vw, std=0.1)
        sd[f"visual.{nm}.bias"] = rn(vw, std=0.1)
    sd["visual.proj"] = rn(vw, D, std=vw ** -0.5)

    def blocks(prefix, w, layers):
        for i in range(layers):
            b = f"{prefix}.resblocks.{i}"
            sd[f"{b}.ln_1.weight"] = 1.0 + rn(w, std=0.1)
            sd[f"{b}.ln_1.bias"] = rn(w, std=0.1)
            sd[f"{b}.attn.in_proj_weight"] = rn(3 * w, w, std=w ** -0.5)
            sd[f"{b}.attn.in_proj_bias"] = rn(3 * w, std=0.02)
            sd[f"{b}.attn.out_proj.weight"] = rn(w, w, std=(w ** -0.5) * ((2 * layers) ** -0.5))
            sd[f"{b}.attn.out_proj.bias"] = rn(w, std=0.02)
            sd[f"{b}.ln_2.weight"] = 1.0 + rn(w, std=0.1)
            sd[f"{b}.ln_2.bias"] = rn(w, std=0.1)
            sd[f"{b}.mlp.c_fc.weight"] = rn(4 * w, w, std=(2 * w) ** -0.5)
            sd[f"{b}.mlp.c_fc.bias"] = rn(4 * w, std=0.02)
            sd[f"{b}.mlp.c_proj.weight"] = rn(w, 4 * w, std=(w ** -0.5) * ((2 * layers) ** -0.5))
            sd[f"{b}.mlp.c_proj.bias"] = rn(w, std=0.02)

    blocks("visual.transformer", vw, arch.v_layers)
    sd["token_embedding.weight"] = rn(arch.vocab, tw, std=0.02)
    sd["positional_embedding"] = rn(arch.ctx, tw, std=0.01)
    blocks("transformer", tw, arch.t_layers)
    sd["ln_final.weight"] = 1.0 + rn(tw, std=0.1)
    sd["ln_final.bias"] = rn(tw, std=0.1)
    sd["text_projection"] = rn(tw, D, std=tw ** -0.5)
    return sd


def synthetic_ids(arch, n, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(n, arch.ctx, dtype=torch.int32)
    lens = torch.randint(8, arch.ctx, (n,), generator=g)
    body = torch.randint(1, arch.sot, (n, arch.ctx), generator=g, dtype=torch.int32)
    pos = torch.arange(arch.ctx)[None, :]
    ids = torch.where(pos < lens[:, None], body, ids)
    ids[:, 0] = arch.sot
    ids[torch.arange(n), lens] = arch.eot
    return ids


def ids_with_lengths(arch, lens, seed):
    """Token ids whose end-of-text token sits at position lens[i] - 1 (a text of lens[i] positions, SOT and EOT included)."""
    g = torch.Generator().manual_seed(seed)
    n = lens.numel()
    body = torch.randint(1, arch.sot, (n, arch.ctx), generator=g, dtype=torch.int32)
    pos = torch.arange(arch.ctx)[None, :]
    ids = torch.where(pos < (lens[:, None] - 1), body, torch.zeros(n, arch.ctx, dtype=torch.int32))
    ids[:, 0] = arch.sot
    ids[torch.arange(n), (lens - 1).long()] = arch.eot
    return ids


def gemm_flops_per_step(arch, batch):
    """Algorithmic FLOPs of the bf16 GEMM launches of one step (SURVEY.md 8(d) terms that run in gemm_bf16_nt_kernel)
    and the number of launches."""
    def tower(tokens, w, layers, items):
        m = items * tokens
        return 2.0 * m * layers * (w * 3 * w + w * w + 2 * w * 4 * w), 4 * layers
    fi, li = tower(arch.v_tokens, arch.v_width, arch.v_layers, batch)
    patch = 2.0 * batch * (arch.v_tokens - 1) * 3 * arch.patch * arch.patch * arch.v_width
    ft, lt = tower(arch.ctx, arch.t_width, arch.t_layers, 2 * batch)   # flops of both text calls together
    return fi + patch + ft, li + 1 + 2 * lt      # two encode_text calls (query, target) per step


def gemm256u_census(arch, calls):
    """(launches, algorithmic bytes) of the persistent-GEMM launches behind a list of encoder calls [(kind, items)]: per layer
    QKV, out-proj, fc1, fc2 with A + W read once and C written once, all bf16 (the per-launch minimum the
    PMC traffic is compared with); with the residual add in the epilogue the out-proj and fc2 launches read their C tile too."""
    launches, nbytes = 0, 0.0
    for kind, n, resadd, rows, pooled in calls:
        tokens, w, layers = (arch.v_tokens, arch.v_width, arch.v_layers) if kind == "image" else (arch.ctx, arch.t_width, arch.t_layers)
        m = rows                                         # token rows of the call (texts: only the positions up to the end-of-text token)
        if m <= 512:
            continue                                     # the skinny kernel takes these
        full = layers - 1 if pooled else layers          # pooled: the last block is ONE persistent launch (K, V: N = 2 W), the rest small kernels
        for nn, kk in ((3 * w, w), (w, w), (4 * w, w), (w, 4 * w)):
            # resadd: 0 = store-only bf16 C; 2 / 4 = the out-proj / fc2 launches read and write their C tile in place as bf16 / fp32
            c_bytes = 2.0 * m * nn if not (resadd and nn == w) else 2.0 * resadd * m * nn
            nbytes += full * (2.0 * (m * kk + nn * kk) + c_bytes)
        launches += 4 * full
        if pooled:
            nbytes += 2.0 * (m * w + 2 * w * w) + 2.0 * m * 2 * w
            launches += 1
    return launches, nbytes


def gemm_flops_per_step_text(arch, batch):
    """The text towers' share of gemm_flops_per_step (2 * batch texts, every one of the ctx positions)."""
    return 2.0 * (2 * batch * arch.ctx) * arch.t_layers * (arch.t_width * 3 * arch.t_width + arch.t_width * arch.t_width + 2 * arch.t_width * 4 * arch.t_width)


def gemm_flops_per_text_row(arch):
    """GEMM FLOPs one token row of the text tower costs (QKV, out-proj, fc1, fc2 of every layer)."""
    return 2.0 * arch.t_layers * (arch.t_width * 3 * arch.t_width + arch.t_width * arch.t_width + 2 * arch.t_width * 4 * arch.t_width)


def tower_gemm_flops(width, layers, rows, items, pooled):
    """GEMM FLOPs a tower call EXECUTES: 12 W^2 multiply-adds per token row and block (QKV 3, out-proj 1, fc1 4, fc2 4) -- and, when
    the last block runs its query path on the pooled row only (option last_block_pooled_row), 2 W^2 per row (K, V) + 10 W^2 per
    item in that block."""
    unit = 2.0 * width * width
    if pooled and layers > 0:
        return unit * ((layers - 1) * 12.0 * rows + 2.0 * rows + 10.0 * items)
    return unit * layers * 12.0 * rows


def main():
    if os.environ.get("KEMR_DEBUG_SET"):               # A/B runs only (tools/ab_env.sh): "key=value,key=value" for kemr_debug_set
        from knowledge_enhanced_multimodal_retrieval_amd import debug
        for kv in os.environ["KEMR_DEBUG_SET"].split(","):
            k, v = kv.split("=")
            debug.set(k.strip(), int(v))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps; 0 = the whole gallery shard, ceil(43000 / gpus / batch)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=255, help="gallery items per rank per step")
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--precision", default=_lib.DEFAULT_PRECISION, choices=["bf16", "bf16-res16", "fp8", "fp8-res16", "fp8-mlp", "bf16-x24", "fp8-x24"],
                    help="bf16-x24 (the product's default: bf16 operands, fp32 residual arithmetic, the stream stored as 24-bit floats), bf16 (4-byte stream), bf16-res16 (bf16 stream); fp8 variants")
    ap.add_argument("--image-slice", type=int, default=0, help="experiment: images per encoder launch (default: the engine's 255)")
    ap.add_argument("--gemm-variant", type=int, default=0, help="A/B only: force a GEMM tile variant (0 = the library's choice)")
    ap.add_argument("--text-group", type=int, default=0, help="texts per encoder call (0 = engine.tile_friendly_batch: 851 for ViT-L/14; 255 = one call per text column and step, round 1 / early round 2)")
    ap.add_argument("--full-context", action="store_true", help="compute all 77 positions of every text (the reference's arithmetic; default: only the positions up to the end-of-text token, which are the ones that can reach the embedding)")
    ap.add_argument("--resadd", type=int, default=-1, help="A/B: 1 / 0 = residual add inside the out-proj / fc2 epilogues on / off (default: the library's setting)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sim", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the host-pipeline sub-result (N = 1: uint8 sources through encode_dataset)")
    ap.add_argument("--pipeline-items", type=int, default=24480, help="gallery items of the host-pipeline sub-result")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp8 / bf16-res16 sub-results (two more engines, ~20 steps each)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev = torch.device("cuda", int(os.environ.get("KEMR_LOCAL_DEVICE", local_rank)))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl" on ROCm).  KEMR_DIST_BACKEND=gloo and KEMR_LOCAL_DEVICE=0 exist only to rehearse the N > 1 code path
        # with several processes on ONE GPU (RCCL refuses two ranks on one device).
        backend = os.environ.get("KEMR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    arch = ARCHS[args.model]
    B = args.batch
    if args.steps <= 0:
        args.steps = -(-GALLERY // (world * B))          # every rank encodes its whole contiguous shard of the gallery
    eng = engine.ClipEngine(arch, dev, precision=args.precision)
    if args.image_slice:
        engine.MAX_IMAGE_BATCH = args.image_slice
    if args.gemm_variant:
        engine.set_gemm_variant(args.gemm_variant)
    if args.resadd >= 0:
        eng.set_residual_fusion(bool(args.resadd))
    # residual add inside the out-proj / fc2 epilogues: the library's default (KEMR_RESADD=0 turns it off)
    resadd_on = eng.residual_fusion_active()
    eng.load_state_dict(random_weights(arch, seed=0))

    g = torch.Generator().manual_seed(1234 + rank)
    pixels = torch.randn(B, 3, arch.image_size, arch.image_size, generator=g).to(dev)
    q_host, t_host = synthetic_ids(arch, B, 1235 + rank), synthetic_ids(arch, B, 4321 + rank)     # what a tokenizer hands over: host ids
    q_ids, t_ids = q_host.to(dev), t_host.to(dev)
    q_lens, t_lens = engine.text_lengths(q_host), engine.text_lengths(t_host)                    # positions up to the end-of-text token
    pack = eng.pack_text and not args.full_context
    eng.pack_text = pack

    # Encoder calls sized for the persistent GEMM's rounds (engine.tile_friendly_batch): B = 255 images are 256 row tiles; the 2 B
    # texts a step brings are pooled and go text_group = 851 to a call (256 row tiles: whole rounds in every text GEMM; 255 texts
    # leave 10 % of the out-proj round empty; same box: 16 340 items/s at 255, 16 590 at 565, 16 710 at 848), the pool's rest is encoded when the timed region ends.  Every step
    # still encodes exactly B images + 2 B texts on average, and the region as a whole exactly steps x (B + 2 B) items.
    # Packed texts (the default: only the positions up to a text's end-of-text token are computed, engine.encode_text): the pool is
    # sized by TOKEN ROWS instead -- as many of the step's texts as fill engine.TEXT_ROW_BUDGET = 65 536 rows (256 row tiles).
    pair_lens = torch.cat([q_lens, t_lens])

    def make_pool(packed, pair_ids=None, lens=None):
        """The step's 2 B texts (query column, target column) pooled into encoder calls; pair_ids / lens: another set of texts than
        the headline's (the length-distribution sub-results)."""
        import types
        lens = pair_lens if lens is None else lens
        pair_ids = torch.cat([q_ids, t_ids]) if pair_ids is None else pair_ids
        if args.text_group:
            group = args.text_group
        elif packed:
            reps_ = -(-engine.TEXT_ROW_BUDGET // int(lens.sum())) + 1
            group = max(2 * B, int((torch.cumsum(lens.repeat(reps_).to(torch.int64), 0) <= engine.TEXT_ROW_BUDGET).sum()))
        else:
            group = engine.tile_friendly_batch(arch.ctx, arch.t_width, B, engine.MAX_TEXT_BATCH) if B == 255 else 2 * B
        reps_ = -(-group // (2 * B))
        lens_ = lens.repeat(reps_)[:group].contiguous()                        # host tensor: the launches are sized without asking the device
        csum = [0] + torch.cumsum(lens_.to(torch.int64), 0).tolist()           # rows of the first k texts of the pool
        return types.SimpleNamespace(packed=packed, group=group, ids=pair_ids.repeat(reps_, 1)[:group].contiguous(), lens=lens_,
                                     rows=(lambda k: csum[k]) if packed else (lambda k: k * arch.ctx))

    main_pool = make_pool(pack)
    text_group = main_pool.group

    all_calls = []                 # every encoder call of this process, in order: (kind, items, resadd bytes, token rows)

    class Stepper:
        def __init__(self, e, resadd, pool=None):
            self.pool = pool or main_pool
            self.resadd = (2 if e.precision.endswith("res16") else 4) if resadd else 0      # bytes per in-place C element, census
            self.e, self.pending, self.images, self.texts, self.calls = e, 0, 0, 0, all_calls      # pooled texts; items encoded so far; the process-wide call list
            self.text_rows = 0                         # token rows the text calls computed
            # whether this engine's last blocks run on the pooled row only (store-only epilogues, fc1 not on fp8: api.hip run_blocks)
            self.pooled = bool(e.last_block_pooled_row()) and not resadd and e.precision != "fp8-mlp"

        def step(self):
            a = self.e.encode_image(pixels, normalize=True)
            self.calls.append(("image", B, self.resadd, B * arch.v_tokens, self.pooled))
            self.images += B
            self.pending += 2 * B
            p = self.pool
            self.e.pack_text = p.packed
            while self.pending >= p.group:
                self.e.encode_text(p.ids, normalize=True, lens=p.lens)
                self.calls.append(("text", p.group, self.resadd, p.rows(p.group), self.pooled))
                self.pending -= p.group
                self.texts += p.group
                self.text_rows += p.rows(p.group)
            return a

        def drain(self):
            if self.pending:
                p = self.pool
                self.e.pack_text = p.packed
                self.e.encode_text(p.ids[:self.pending], normalize=True, lens=p.lens[:self.pending])
                self.calls.append(("text", self.pending, self.resadd, p.rows(self.pending), self.pooled))
                self.texts += self.pending
                self.text_rows += p.rows(self.pending)
                self.pending = 0

        def check_outputs(self):            # the step's own items, for the oracle / cross-precision comparisons (untimed)
            pk = self.pool.packed
            self.e.pack_text = pk
            self.calls += [("image", B, self.resadd, B * arch.v_tokens, self.pooled), ("text", B, self.resadd, int(q_lens.sum()) if pk else B * arch.ctx, self.pooled),
                           ("text", B, self.resadd, int(t_lens.sum()) if pk else B * arch.ctx, self.pooled)]
            return (self.e.encode_image(pixels, normalize=True), self.e.encode_text(q_ids, normalize=True, lens=q_lens),
                    self.e.encode_text(t_ids, normalize=True, lens=t_lens))

    main_steps = Stepper(eng, resadd_on)
    step, drain = main_steps.step, main_steps.drain

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    out = main_steps.check_outputs()
    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if os.environ.get("KEMR_DIST_BACKEND") == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = max_over_ranks(elapsed)
    assert all(torch.isfinite(o).all().item() for o in out)

    rccl_ranks = dist.get_world_size() if dist is not None else 1
    per_shard = (GALLERY + world - 1) // world
    shard_bounds = [[r * per_shard, min(GALLERY, (r + 1) * per_shard)] for r in range(world)]
    items = 3 * B * world * args.steps                       # images + query texts + target texts
    value = items / elapsed
    row_frac = float(pair_lens.sum()) / (2 * B * arch.ctx) if pack else 1.0      # share of the text token rows that is computed
    pooled_main = bool(eng.last_block_pooled_row()) and not resadd_on and args.precision != "fp8-mlp"
    skip_v = (10.0 / 12.0) / arch.v_layers if pooled_main else 0.0                # share of a tower's work the pooled-row last block leaves out
    skip_t = (10.0 / 12.0) / arch.t_layers if pooled_main else 0.0
    flops_item_step = B * (arch.image_flops() * (1 - skip_v) + 2 * arch.text_flops() * row_frac * (1 - skip_t))  # executed, not the reference's count
    result = {
        "metric": "gallery images+texts encoded/sec (ViT-L/14; synthetic texts per SURVEY 8(d): end-of-text uniform in positions 8..76, mean "
                  f"{float(pair_lens.float().mean()):.1f} of 77 positions computed -- value_every_text_77_positions is the rate when every text fills the context) and 43k x Q sim+top-10 ms",
        "value": value, "unit": "items/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp8": "fp8 (QKV) + bf16", "fp8-res16": "fp8 (QKV) + bf16", "fp8-mlp": "fp8 (QKV, fc1) + bf16"}.get(args.precision, "bf16"), "data": "synthetic",
        "config": {"workload": "CLIP ViT-L/14 zero-shot: 43k-gallery encode (1 image + query + target text per item, "
                               "224x224 / 77 tokens) + T2I top-10, BASELINE configs[1]",
                   "architecture": args.model, "residual_stream": "bf16" if args.precision.endswith("res16") else ("fp24" if args.precision.endswith("x24") else "fp32"), "batch_per_gpu": B, "text_group": text_group, "gallery": GALLERY, "parallelism": f"dp{world} (gallery sharded)",
                   "items_timed": items, "rccl_ranks": rccl_ranks, "backend": (os.environ.get("KEMR_DIST_BACKEND", "nccl") if world > 1 else None),
                   "shard_bounds": shard_bounds},
        "images_per_s": B * world * args.steps / elapsed,
        "texts_per_s": 2 * B * world * args.steps / elapsed,
        "encode_tflops_per_gpu": flops_item_step * args.steps / elapsed / 1e12,
        "encode_frac_of_bf16_peak": flops_item_step * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS,
    }

    def pipeline_leg():
        # ------------------------------------------------------------------ the same encode through the host input pipeline
        # camera-sized uint8 sources -> CLIPEvalDatasetHF(split, preprocess), the reference's own dataset call
        # (evaluator.py:330-333), with the preprocess object clip.load returns -> DataLoader workers (decode stand-in, tokenise, pack) ->
        # one pinned H2D copy + one preprocess launch pair per loader batch -> the three encoders (evaluators.encode_dataset, what the
        # drop-in CLIs run).  PCIe-inclusive by construction; worker start-up is inside the timed region.  N = 1 only.
        if world == 1 and not args.no_pipeline and args.model == "ViT-L/14":
            import warnings
            from knowledge_enhanced_multimodal_retrieval_amd import clip_api, datasets as kds, evaluators, tokenizer
            clip_api.allow_random_weights(True)
            tokenizer.allow_hash_tokenizer(True)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                pm, ppre = clip_api.load(args.model, device=str(dev))
            n_pipe, workers = args.pipeline_items, evaluators.default_loader_workers()
            split = kds.SyntheticHFSplit(n_pipe, 11)
            evaluators.encode_dataset(pm, kds.CLIPEvalDatasetHF(kds.SyntheticHFSplit(510, 12), ppre), 64, 1, 0)      # warm: kernels, workspaces
            barrier()
            t1 = time.perf_counter()                                       # the host side alone (loader processes, pin thread): its ceiling,
            n_l, t_first, n_first = 0, None, 0                             # and how long the worker processes take to deliver their first batch
            for b_ in evaluators.eval_loader(kds.CLIPEvalDatasetHF(kds.SyntheticHFSplit(n_pipe // 4, 13), ppre), 64, 1, workers,
                                             evaluators.default_tokenize, True):
                n_l += len(b_[3])
                if t_first is None:
                    t_first, n_first = time.perf_counter() - t1, n_l
            t_all = time.perf_counter() - t1
            loader_only = 3 * n_l / t_all
            loader_steady = 3 * (n_l - n_first) / max(t_all - t_first, 1e-9)
            t1 = time.perf_counter()
            pi, pq, pt, pids = evaluators.encode_dataset(pm, kds.CLIPEvalDatasetHF(split, ppre), 64, 1, workers)
            barrier()
            dtp = time.perf_counter() - t1
            assert pi.shape[0] == n_pipe and len(pids) == n_pipe and bool(torch.isfinite(pi).all())
            result["pipeline"] = {"items_per_s": 3 * n_pipe / dtp, "images_per_s": n_pipe / dtp, "seconds": dtp, "items": n_pipe,
                                  "loader_workers": workers, "loader_batch": 64, "host_cores": os.cpu_count(), "loader_only_items_per_s": loader_only, "loader_first_batch_s": t_first,
                                  "loader_only_items_per_s_after_first_batch": loader_steady, "loader_context": os.environ.get("KEMR_LOADER_CONTEXT", "forkserver"),
                                  "image_transform": "gpu" if ppre.defer_to_gpu else "host",
                                  "source": "uint8 PIL images of 8 camera-like sizes (224x224 .. 600x800) -> CLIPEvalDatasetHF(split, preprocess) -> "
                                            "evaluators.encode_dataset; worker start-up and PCIe included"}
            del pi, pq, pt
            # ---- the same with JPEG decode inside the loader workers: rows hold ENCODED bytes, as the HuggingFace
            # Image feature does, and Pillow decodes them when the row is read (clip_dataset.py:110-113 receives the decoded image).
            # Loader alone at several worker counts (start-up reported apart), then the whole pipeline at the default count.
            n_jpeg = max(1020, n_pipe // 2)
            jsplit = kds.SyntheticJPEGSplit(n_jpeg, 21)
            sweep = {}
            for w_ in sorted({4, 8, workers, 16}):
                if w_ <= 0:
                    continue
                t1 = time.perf_counter()
                n_l, t_first, n_first = 0, None, 0
                for b_ in evaluators.eval_loader(kds.CLIPEvalDatasetHF(kds.SyntheticJPEGSplit(4080, 22), ppre), 64, 1, w_, evaluators.default_tokenize, True):
                    n_l += len(b_[3])
                    if t_first is None:
                        t_first, n_first = time.perf_counter() - t1, n_l
                t_all = time.perf_counter() - t1
                sweep[str(w_)] = {"images_per_s_after_first_batch": (n_l - n_first) / max(t_all - t_first, 1e-9), "first_batch_s": t_first}
            t1 = time.perf_counter()
            ji, jq, jt, jids = evaluators.encode_dataset(pm, kds.CLIPEvalDatasetHF(jsplit, ppre), 64, 1, workers)
            barrier()
            dtj = time.perf_counter() - t1
            assert ji.shape[0] == n_jpeg and bool(torch.isfinite(ji).all())
            start_up = sweep[str(workers)]["first_batch_s"] if str(workers) in sweep else None
            encoder_images_per_s = result["images_per_s"]
            best = max(sweep.values(), key=lambda v: v["images_per_s_after_first_batch"])["images_per_s_after_first_batch"]
            result["pipeline_jpeg"] = {
                "items_per_s": 3 * n_jpeg / dtj, "images_per_s": n_jpeg / dtj, "seconds": dtj, "items": n_jpeg, "loader_workers": workers,
                "items_per_s_without_start_up": (3 * n_jpeg / (dtj - start_up)) if start_up and dtj > start_up else None, "loader_start_up_s": start_up,
                "loader_only_by_workers": sweep, "mean_jpeg_bytes": jsplit.mean_jpeg_bytes(),
                "decode_bound": bool(best < encoder_images_per_s),
                "note": f"encoders alone take {encoder_images_per_s:.0f} images/s (headline); the loader with Pillow's JPEG decode delivers at most {best:.0f} images/s "
                        "at the worker counts tried -- whichever is lower bounds the drop-in CLIs on JPEG sources",
                "source": "JPEG bytes (quality 90, 8 camera-like sizes 224x224 .. 600x800, encoded in memory) decoded by Pillow inside the loader workers -> "
                          "CLIPEvalDatasetHF(split, preprocess) -> evaluators.encode_dataset; worker start-up and PCIe included"}
            del pm, ji, jq, jt

    # ------------------------------------------------------------------ roofline: per-class hipEvent timing
    L = _lib.lib()
    prof_steps = 10                                  # whole text groups only inside the profiled region; the rest drains after it
    patch_flops = 2.0 * B * (arch.v_tokens - 1) * 3 * arch.patch * arch.patch * arch.v_width

    def profile_region(st):
        """10 steps of `st` under the library's per-class hipEvent timing -> (ms per class and step, GEMM launches per step, GEMM flops per step)."""
        i0, t0_, r0_ = st.images, st.texts, st.text_rows
        _lib.check(L.kemr_profile_begin(4096 * prof_steps))
        for _ in range(prof_steps):
            st.step()
        ms_ = (C.c_double * 5)()
        cnt_ = (C.c_int64 * 5)()
        _lib.check(L.kemr_profile_end(ms_, cnt_, 5))
        n_img, n_txt = st.images - i0, st.texts - t0_
        st.drain()
        # GEMM FLOPs the region actually executed: whole image calls, the text calls by their token rows and items
        fl = ((tower_gemm_flops(arch.v_width, arch.v_layers, B * arch.v_tokens, B, st.pooled) + patch_flops) * n_img / B +
              tower_gemm_flops(arch.t_width, arch.t_layers, st.text_rows - r0_, n_txt, st.pooled)) / prof_steps
        return [m_ / prof_steps for m_ in ms_], cnt_[0] / prof_steps, fl

    ms, gemm_n, gemm_flops = profile_region(main_steps)
    gemm_ms = ms[0]
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
    # L2-miss (fabric) bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of this command
    # (tools/profile_round.sh <tag> pmc: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), committed under
    # profiles/ with the commit they were measured on; bench.py cannot collect PMC counters itself, and the figure is
    # attached only to the configuration it was measured for.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
    tj_ = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if tj_ and args.model == "ViT-L/14" and B == 255 and args.precision == tj_.get("precision", "bf16-res16") and resadd_on == bool(tj_.get("residual_fusion_active", True)) and \
            bool(pack) == bool(tj_.get("text_packed", False)) and pooled_main == bool(tj_.get("last_block_pooled_row", False)) and args.gemm_variant == 0 and not args.text_group:
        with open(tpath) as f:
            tj = json.load(f)
        traffic, traffic_source = tj.get("bytes_per_launch"), {k: tj.get(k) for k in ("profile", "round", "commit", "algorithmic_bytes_per_launch")}
    result["roofline"] = {
        "kernel": "gemm256u_bf16_nt_kernel (all launches of the GEMM class timed by hipEvents on the launch stream over 10 steps: the persistent GEMMs of the image calls and of the pooled text calls + the patch-embedding GEMM)",
        "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
        "launches_per_step": round(gemm_n, 1), "avg_launch_us": 1e3 * gemm_ms / max(gemm_n, 1),
        "flops_per_launch": gemm_flops / max(gemm_n, 1),
    }
    result["kernel_ms_per_step"] = {"gemm": ms[0], "layernorm": ms[1], "attention": ms[2], "embed_tail": ms[3]}
    result["config"]["residual_add_in_gemm_epilogue"] = resadd_on

    pipeline_leg()

    # ------------------------------------------------------------------ similarity + top-10 on the 43k gallery
    if not args.no_sim:
        per = (GALLERY + world - 1) // world
        lo, hi = rank * per, min(GALLERY, (rank + 1) * per)
        gg = torch.Generator(device=dev).manual_seed(7)
        gal_all = torch.nn.functional.normalize(torch.randn(GALLERY, arch.embed_dim, generator=gg, device=dev), dim=-1)
        qry_all = torch.nn.functional.normalize(gal_all + 0.04 * torch.randn(GALLERY, arch.embed_dim, generator=gg, device=dev), dim=-1)
        tgt_all = torch.nn.functional.normalize(gal_all + 0.5 * torch.randn(GALLERY, arch.embed_dim, generator=gg, device=dev), dim=-1)
        sim = {}
        # (label, queries, bf16 terms, parts): parts == 2 is BASELINE configs[2], fused T2I + T2T scoring (0.5 / 0.5) as ONE
        # contraction over the concatenated weighted panels (reference metrics.py:145-148); k == 0 = ranks only (what the
        # Recall@K / MRR metrics need)
        cases = (("q1024_bf16", 1024, 1, 1, 10), ("q43000_bf16", GALLERY, 1, 1, 10), ("q43000_bf16_rank_only", GALLERY, 1, 1, 0),
                 ("q43000_fp32x3", GALLERY, 3, 1, 10), ("q43000_bf16_c3_fused_t2i_t2t", GALLERY, 1, 2, 10))
        from knowledge_enhanced_multimodal_retrieval_amd import dist as kdist
        for label, nq, terms, parts, k in cases:
            # The product class at every N: dist.ShardedGallery holds this rank's shard of the gallery; a search is
            # all-gather of the ranks' query slices -> local fused top-k with global ids -> all-gather + merge of the candidates,
            # ranks add the all-reduced ground-truth score and `ahead` counts (no collective at all when world == 1).
            gparts = [gal_all[lo:hi]] + ([tgt_all[lo:hi]] if parts == 2 else [])
            sg = kdist.ShardedGallery(gparts, GALLERY, precision="bf16" if terms == 1 else "fp32x3")
            sg.check_rows = False                                      # fixed query slices: the call has no host synchronisation
            nql = nq // world                                          # every rank brings an equal slice of the query batch
            nq = nql * world
            qs = qry_all[rank * nql:(rank + 1) * nql]
            gt = torch.arange(rank * nql, (rank + 1) * nql, dtype=torch.int32, device=dev)

            def run():
                if k == 0:                                             # ranks only (what Recall@K / MRR need)
                    ranks, _, _ = sg.ranks([qs] * parts, gt, weights=[1.0 / parts] * parts, k=0)
                    return ranks, None
                return sg.search([qs] * parts, weights=[1.0 / parts] * parts, k=k)      # queries arrive as fp32 embeddings

            run()
            barrier()
            reps = 5 if nq <= 2048 else 3
            t1 = time.perf_counter()
            for _ in range(reps):
                s_, i_ = run()
            barrier()
            dt = (time.perf_counter() - t1) / reps
            dt = max_over_ranks(dt)
            kdim = arch.embed_dim * terms * parts
            flops = 2.0 * nq * GALLERY * kdim
            sim[label] = {"ms": 1e3 * dt, "mfma_tflops_per_gpu": flops / dt / 1e12 / world, "kdim": kdim, "queries": nq,
                          "via": "dist.ShardedGallery.%s" % ("ranks" if k == 0 else "search")}
            if k:
                sim[label]["top1_hit"] = float((i_[:, 0].long() == torch.arange(nq, device=dev)).float().mean())
            else:
                sim[label]["rank1"] = float((s_ == 1).float().mean())
            del sg
        result["sim_top10"] = sim
        # roofline of the similarity kernel at the headline size (MFMA-bound: SURVEY 8(d)); min_bytes = both panels read once +
        # the lists written
        hs = sim["q43000_bf16"]
        result["roofline_sim"] = {"kernel": "gemm256u_bf16_nt_kernel<SIM 3 sample pass + SIM 2 list pass> + threshold / select kernels "
                                            "(whole kemr_sim_topk call at Q = 43000, panel build of the queries included)", "bound": "mfma",
                                  "flops": 2.0 * GALLERY * GALLERY * arch.embed_dim, "ms": hs["ms"],
                                  "achieved": hs["mfma_tflops_per_gpu"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                  "frac": hs["mfma_tflops_per_gpu"] / PEAK_BF16_TFLOPS,
                                  "min_bytes": 2 * GALLERY * arch.embed_dim * 2 + GALLERY * 10 * 8, "traffic": None}
        spath = os.path.join(ROOT, "profiles", "sim_traffic.json")
        if os.path.exists(spath) and world == 1 and args.model == "ViT-L/14":      # measured for this exact call (separate --pmc passes)
            with open(spath) as f:
                sj = json.load(f)
            result["roofline_sim"]["traffic"] = sj.get("bytes_per_launch")
            result["roofline_sim"]["traffic_source"] = {k: sj.get(k) for k in ("kernel", "profile", "round", "commit", "method")}

    # ------------------------------------------------------------------ sub-result: the same steps with every row computed
    # The headline leaves out rows that cannot reach an output: a text is computed only up to its end-of-text token (causal mask,
    # reference pooling x[arange, text.argmax(-1)]), and the last block of each tower runs its query path on the one row per item that
    # leaves the tower.  This leg runs the reference's arithmetic in full -- 77 positions per text, 851 texts to a call, every row
    # through every block -- on the same engine, and the embeddings of the two are compared.
    result["text_packing"] = {"enabled": bool(pack), "text_rows_computed_fraction": row_frac,
                              "mean_positions_per_text": float(pair_lens.float().mean()) if pack else float(arch.ctx), "context": arch.ctx,
                              "texts_per_call": text_group,
                              "lengths": "synthetic_ids: end-of-text token uniform in positions 8 .. 76 (unchanged since round 1)"}
    result["last_block_pooled_row"] = {"enabled": pooled_main, "tower_work_left_out_vision": skip_v, "tower_work_left_out_text": skip_t,
                                       "what": "the last block of a tower computes K and V for every row and the query path (attention output, out-proj, "
                                               "ln_2, MLP) for the class / end-of-text row only: the one row per item that leaves the tower"}
    if (pack or pooled_main) and not args.no_extras:
        pooled_option = eng.last_block_pooled_row()
        eng.set_last_block_pooled_row(False)           # this leg: every position of every text, every row through every block
        s4 = Stepper(eng, resadd_on, make_pool(False))
        for _ in range(3):
            s4.step()
        s4.drain()
        barrier()
        n4 = 20
        t1 = time.perf_counter()
        for _ in range(n4):
            s4.step()
        s4.drain()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        o4 = s4.check_outputs()
        eng.pack_text = pack
        eng.set_last_block_pooled_row(pooled_option)
        cos4 = [float(torch.nn.functional.cosine_similarity(a.double(), b.double()).min()) for a, b in zip(o4, out)]
        result["reference_arithmetic"] = {"what": "the same steps with every text position and every row of every block computed (KEMR_TEXT_PACKED=0, "
                                                  "last_block_pooled_row off): what the reference's model does; same engine, same run",
                                          "items_per_s": 3 * B * world * n4 / dt, "ms_per_step": 1e3 * dt / n4, "steps": n4,
                                          "headline_speedup_over_it": value / (3 * B * world * n4 / dt),
                                          "min_cosine_headline_vs_it_image_query_target": cos4}

    # ------------------------------------------------------------------ sub-results: other text-length distributions
    # The headline's texts follow SURVEY 8(d) (end-of-text uniform in 8 .. 76: 43.8 of 77 positions on average).  The reference cuts
    # texts at 150 WORDS (clip_dataset.py:103-108): its target descriptions routinely fill all 77 positions, where computing a text
    # only up to its end-of-text token saves nothing, while its user-like queries are short.  Same engine, same steps, texts of
    # other lengths: every text 77 positions (the conservative figure), and queries of 16 / targets of 77 positions.
    if pack and not args.no_extras:
        by_len = {}
        for label, lq_, lt_ in (("every_text_77_positions", arch.ctx, arch.ctx), ("queries_16_targets_77_positions", 16, arch.ctx)):
            lens_ = torch.cat([torch.full((B,), lq_, dtype=torch.int32), torch.full((B,), lt_, dtype=torch.int32)])
            ids_ = ids_with_lengths(arch, lens_, 99).to(dev)
            s5 = Stepper(eng, resadd_on, make_pool(True, ids_, lens_))
            for _ in range(3):
                s5.step()
            s5.drain()
            barrier()
            n5 = 20
            t1 = time.perf_counter()
            for _ in range(n5):
                s5.step()
            s5.drain()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            by_len[label] = {"items_per_s": 3 * B * world * n5 / dt, "ms_per_step": 1e3 * dt / n5, "steps": n5,
                             "mean_positions_per_text": float(lens_.float().mean()), "texts_per_call": s5.pool.group}
        eng.pack_text = pack
        result["text_packing"]["by_length_distribution"] = by_len
        result["text_packing"]["headline"] = {"items_per_s": value, "mean_positions_per_text": float(pair_lens.float().mean())}
        result["value_every_text_77_positions"] = by_len["every_text_77_positions"]["items_per_s"]

    # ------------------------------------------------------------------ sub-results: the same step at other precisions
    recall_bar = {}
    rpath = os.path.join(ROOT, "profiles", "r04_recall_bar_ViT-L-14.json")     # the fixed-bar record on the bench's model (ViT-B/32: r04_recall_bar.json)
    if os.path.exists(rpath):
        with open(rpath) as f:
            recall_bar = json.load(f)
    if not args.no_extras and args.precision == _lib.DEFAULT_PRECISION:
        extras = {}
        for prec in ("bf16", "bf16-res16", "fp8", "fp8-res16", "fp8-x24"):
            e2 = engine.ClipEngine(arch, dev, precision=prec)
            e2.load_state_dict(random_weights(arch, seed=0))
            e2.pack_text = pack

            s2 = Stepper(e2, e2.residual_fusion_active())
            for _ in range(3):
                s2.step()
            s2.drain()
            barrier()
            n2 = 20
            t1 = time.perf_counter()
            for _ in range(n2):
                s2.step()
            s2.drain()
            barrier()
            dt = time.perf_counter() - t1
            o2 = s2.check_outputs()
            dt = max_over_ranks(dt)
            cos = [float(torch.nn.functional.cosine_similarity(a.double(), b.double()).min()) for a, b in zip(o2, out)]
            extras[prec] = {"items_per_s": 3 * B * world * n2 / dt, "ms_per_step": 1e3 * dt / n2, "steps": n2,
                            "min_cosine_vs_default_image_query_target": cos, "speedup_vs_default": (3 * B * world * n2 / dt) / value,
                            "inside_recall_bar": recall_bar.get("inside_bar_at_every_level", {}).get(prec)}
            del e2
        result["other_precisions"] = extras
        # BASELINE configs[4] asks for fp8 encoders with Recall@10 within 0.2 % of bf16: `inside_recall_bar` is the record of the fixed-bar
        # stress test (tests/test_encoder_gpu.py; profiles/r03_recall_bar.json) -- a speed-up of a mode outside the bar is not a
        # config-5 result, and the mode inside it ("fp8": QKV only) buys what its line says
        result["recall_bar"] = {"points": recall_bar.get("bar_points"), "default_precision_inside": recall_bar.get("inside_bar_at_every_level", {}).get(args.precision),
                                "source": "profiles/r04_recall_bar_ViT-L-14.json (tests/test_encoder_gpu.py::test_recall_at_10_...[ViT-L/14]; ViT-B/32: profiles/r04_recall_bar.json)"}
        if resadd_on:
            # the same engine with the store-only epilogues (updates applied by the LayerNorms): the GEMM class then holds GEMM
            # work only, which is the configuration the GEMM's own roofline fraction is best read on
            eng.set_residual_fusion(False)
            s3 = Stepper(eng, False)
            for _ in range(3):
                s3.step()
            s3.drain()
            barrier()
            t1 = time.perf_counter()
            for _ in range(20):
                s3.step()
            s3.drain()
            barrier()
            dt = time.perf_counter() - t1
            ms3, n3, fl3 = profile_region(s3)
            eng.set_residual_fusion(True)
            a3 = fl3 / (ms3[0] * 1e-3) / 1e12
            result["roofline_store_only_epilogues"] = {
                "config": "KEMR_RESADD=0: out-proj / fc2 store deltas, the LayerNorms apply them (round 2's earlier default)",
                "items_per_s": 3 * B * world * 20 / dt, "bound": "mfma", "achieved": a3, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": a3 / PEAK_BF16_TFLOPS, "launches_per_step": round(n3, 1),
                "kernel_ms_per_step": {"gemm": ms3[0], "layernorm": ms3[1], "attention": ms3[2], "embed_tail": ms3[3]}}

    # ------------------------------------------------------------------ CPU baseline (oracle, bounded sample)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import clip_ref, metrics_ref
        oa = clip_ref.ARCHS[args.model]
        threads = torch.get_num_threads()
        sd = random_weights(arch, seed=0)
        # 64+ items of the same synthetic batch in the step's 1 : 2 image : text mix, spread over the batch
        # (first / middle / last rows, the last one included: token row 65 534 of the 256-row GEMM tiles), and the
        # reference-style ranking (sgemm + two full argsorts) at N = 8 192
        n_img, n_txt = 22, 44
        pick = lambda n, k: torch.unique(torch.cat([torch.arange(0, k // 3), torch.arange(n // 2 - k // 6, n // 2 - k // 6 + k // 3),
                                                    torch.arange(n - (k - 2 * (k // 3)), n)]))
        ii, it = pick(B, n_img), pick(B, n_txt // 2)
        px = pixels[ii].cpu()
        ids = torch.cat([q_ids[it], t_ids[it]]).cpu()
        clip_ref.encode_text(sd, oa, ids[:2])                       # warm the allocator / thread pool
        t1 = time.perf_counter()
        ci = clip_ref.l2_normalize(clip_ref.encode_image(sd, oa, px))
        ct = clip_ref.l2_normalize(clip_ref.encode_text(sd, oa, ids))
        t_enc = time.perf_counter() - t1
        cpu_items_per_s = (len(ii) + len(ids)) / t_enc
        n_rank = 8192
        im, qq, tt_ = metrics_ref.planted_embeddings(n_rank, arch.embed_dim, seed=0)
        t1 = time.perf_counter()
        metrics_ref.retrieval_metrics(qq, im, "T2I")
        t_rank = time.perf_counter() - t1
        cosf = lambda a, b: float(torch.nn.functional.cosine_similarity(a.cpu().double(), b.double()).min())
        cos_img = cosf(out[0][ii], ci)
        cos_txt = min(cosf(out[1][it], ct[:len(it)]), cosf(out[2][it], ct[len(it):]))
        assert cos_img > 1 - 1e-3 and cos_txt > 1 - 1e-3, (cos_img, cos_txt)          # north_star parity bar
        result["cpu_baseline"] = {
            "value": cpu_items_per_s, "unit": "items/s", "cores": threads, "kind": "port",
            "sample": f"oracle/clip_ref fp32 torch on {threads} threads: {len(ii)} images + {len(ids)} texts of the same "
                      f"synthetic batch, rows from its start, middle and end ({t_enc:.1f} s); oracle/metrics_ref sgemm + full argsort "
                      f"R@K/MRR at N={n_rank}: {t_rank:.2f} s",
            "rank_metrics_n8192_s": t_rank, "gpu_vs_oracle_min_cosine_images": cos_img, "gpu_vs_oracle_min_cosine_texts": cos_txt,
        }

    # what the PMC passes of tools/profile_round.sh average over: every persistent-GEMM launch of this process
    c_launches, c_bytes = gemm256u_census(arch, all_calls)
    result["roofline"]["census"] = {"gemm256u_launches": c_launches, "algorithmic_bytes_per_launch": c_bytes / max(c_launches, 1)}
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
