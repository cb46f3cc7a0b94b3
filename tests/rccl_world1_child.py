"""Child process of tests/test_dist_rccl_world1.py (not collected by pytest): executes the sharded-search collectives through RCCL
with a process group of ONE rank on the one GPU of the box, and compares every result with the same calls made without a process
group.  Prints one JSON line; exit code 0 = all equal."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    import torch
    import torch.distributed as dist
    from knowledge_enhanced_multimodal_retrieval_amd import dist as kd

    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n, d, nq = 6000, 256, 512
    unit = lambda x: x / x.norm(dim=1, keepdim=True)
    img = unit(torch.randn(n, d, generator=g)).to(dev)
    txt = unit(img.cpu() + 0.5 * unit(torch.randn(n, d, generator=g))).to(dev)
    q = unit(img[:nq].cpu() + 0.8 * unit(torch.randn(nq, d, generator=g))).to(dev)
    gt = torch.arange(nq, dtype=torch.int32, device=dev)
    weights = [0.5, 0.5]

    def run(gallery):
        out = {}
        out["search"] = gallery.search([q, q], weights, k=10)
        out["many"] = list(gallery.search_many(([q[i:i + 128], q[i:i + 128]] for i in range(0, nq, 128)), weights, k=10))
        out["ranks"] = gallery.ranks([q, q], gt, weights, k=10)
        out["ranks0"] = gallery.ranks([q, q], gt, weights, k=0)[0]
        torch.cuda.synchronize()
        return out

    plain = run(kd.ShardedGallery([img, txt], n, precision="bf16"))          # no process group: every helper short-circuits

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    kd.force_collectives(True)
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(), "calls": {}}
    counted = {}
    for name in ("all_gather_into_tensor", "all_reduce"):                     # count what really reaches torch.distributed
        orig = getattr(dist, name)

        def wrap(*a, _orig=orig, _name=name, **kw):
            t = a[0]
            counted.setdefault(_name, {"n": 0, "cuda": 0, "async": 0})
            counted[_name]["n"] += 1
            counted[_name]["cuda"] += int(t.is_cuda)
            counted[_name]["async"] += int(bool(kw.get("async_op", False)))
            return _orig(*a, **kw)
        setattr(dist, name, wrap)
    forced_gallery = kd.ShardedGallery([img, txt], n, precision="bf16")
    assert forced_gallery.check_rows                                          # the device branch of require_equal_rows runs
    forced = run(forced_gallery)
    # the helpers themselves, on device tensors
    x = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
    ok = torch.equal(kd.all_gather_rows(x), x)
    out, work = kd.all_gather_rows_async(x)
    work.wait()
    ok = ok and work is not None and torch.equal(out, x)
    y = x.clone()
    ok = ok and torch.equal(kd.all_reduce_sum(y), x)
    try:
        kd.require_equal_rows(7, dev)
    except ValueError:
        ok = False
    torch.cuda.synchronize()
    report["calls"] = counted

    def same(a, b):
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(same(u, v) for u, v in zip(a, b))
        return torch.equal(a, b)
    report["equal"] = {k: bool(same(plain[k], forced[k])) for k in plain}
    report["helpers_ok"] = bool(ok)
    report["top1_hit"] = float((forced["search"][1][:, 0] == gt).float().mean())
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(report))
    return 0 if ok and all(report["equal"].values()) else 1


if __name__ == "__main__":
    sys.exit(main())
