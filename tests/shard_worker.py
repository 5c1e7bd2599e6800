"""Worker for the frame-shard tests (launched by torch.distributed.run, one process per rank, gloo backend).

  exchange : CPU.  Every rank builds the same seeded "whole clip" tensor, takes its frame block, runs the FrameShard
             exchanges and checks them against plain slicing of the whole tensor.
  unet     : GPU.  Ranks share cuda:0 (the test boxes have one GPU; gloo stages the collectives through the host).  Every
             rank runs the frame-sharded engine AND the unsharded engine on the same inputs and compares.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoc_amd.frame_shard import FrameShard  # noqa: E402


def exchange(out_dir):
    rank, world = dist.get_rank(), dist.get_world_size()
    report = {"rank": rank, "cases": 0}
    for mode in ("a2a", "allgather"):
        sh = FrameShard(exchange=mode)
        for (B, F, hw, C) in ((1, 4, 6, 8), (2, 4, 6, 8), (5, 2, 10, 16), (1, 8, 2, 8)):
            g = torch.Generator().manual_seed(B * 1000 + F * 10 + hw)
            whole = torch.randn(B, F, hw, C, generator=g).half()
            f0, f1 = sh.frame_range(F)
            p0, p1 = sh.pixel_range(hw)
            mine = whole[:, f0:f1].reshape(-1, C).contiguous()
            px = sh.to_pixel_shard(mine, B, f1 - f0, hw)
            assert torch.equal(px, whole[:, :, p0:p1].reshape(-1, C)), (mode, B, F, hw, "to_pixel_shard")
            back = sh.to_frame_shard(px, B, f1 - f0, hw)
            assert torch.equal(back, mine), (mode, B, F, hw, "to_frame_shard")
            full = sh.gather_frames(whole[:, f0:f1].permute(0, 3, 1, 2).contiguous(), dim=2)  # [B, C, F, hw]
            assert torch.equal(full, whole.permute(0, 3, 1, 2)), (mode, "gather_frames")
            mom = torch.full((B, 4, 3), float(rank))
            parts = sh.all_gather(mom)
            assert parts.shape == (world, B, 4, 3) and all(float(parts[r].mean()) == r for r in range(world))
            report["cases"] += 1
        try:
            sh.check(7, 8)
            raise AssertionError("check() accepted 7 frames")
        except RuntimeError:
            pass
    json.dump(report, open(os.path.join(out_dir, f"exchange_r{rank}.json"), "w"))


def unet(out_dir, which):
    from mvoc_amd.unet import I2VGenXLUNet
    from mvoc_amd.unet_spec import UNetConfig
    from mvoc_amd import pnp_utils

    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0"
    # the 4-level toy of the oracle tests (full attribute tree the hooks address) or the production network
    cfg = UNetConfig() if which in ("full", "cfg4") else UNetConfig(
        block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=8, cross_attention_dim=64,
        attention_head_dim=64, transformer_in_heads=2, context_pool=8)
    eng = I2VGenXLUNet(cfg, device=dev).init_random(31)
    eng.prune_dead_chunks = False  # (frame-sharded forwards compute every chunk: compare like with like)
    from mvoc_amd.unet import Linear
    Linear.USE_GN_FOLD = False  # (... and apply their GroupNorms as kernels -- the fold into proj_in is a single-GPU form)
    report = {"rank": rank}

    def inputs(B, F, h, w, seed):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g).half().to(dev)
        return dict(sample=r(B, 4, F, h, w), fps=torch.full((B,), 8.0).to(dev), first=r(B, 4, F, h, w), lat=r(B, 4, F, h, w),
                    emb=r(B, F, cfg.cross_attention_dim), ehs=r(B, 7, cfg.cross_attention_dim))

    def run(x, mfg):
        return eng.forward_ext(x["sample"], torch.tensor([500.0]).to(dev), x["fps"], x["first"], x["lat"], x["emb"], x["ehs"],
                               multi_frame_guidance=mfg)[0].float()

    def compare(name, B, F, h, w, mfg, hooks=None, modes=("a2a", "allgather")):
        x = inputs(B, F, h, w, seed=B * 100 + F)
        for mode in modes:
            eng.set_frame_shard(None)
            if hooks:
                hooks()
            ref = run(x, mfg)
            eng.set_frame_shard(FrameShard(exchange=mode))
            if hooks:
                hooks()
            got = run(x, mfg)
            eng.set_frame_shard(None)
            assert got.shape == ref.shape
            d = (got - ref).abs().max().item()
            rel = ((got - ref).norm() / ref.norm()).item()
            report[f"{name}_{mode}"] = {"max_abs": d, "rel_l2": rel, "ref_max": ref.abs().max().item()}
            # every rank must hold the same full output (the loops around the UNet run replicated)
            parts = FrameShard().all_gather(got)
            assert all(torch.equal(parts[0], parts[r]) for r in range(world)), f"{name}: ranks disagree"

    def pnp_hooks(F, hw_, seed):
        g = torch.Generator().manual_seed(seed)
        hard = (torch.rand(2, 1, 1, F, hw_, hw_, generator=g) > 0.5).expand(2, 1, 4, F, hw_, hw_)
        soft = (torch.randint(0, 256, (2, 1, 1, F, hw_, hw_), generator=g).float() / 255).half().expand(2, 1, 4, F, hw_, hw_)
        masks = [(soft[j].contiguous().to(dev), hard[j].contiguous().to(dev)) for j in range(2)]

        class Pipe:
            unet = eng

        def hooks():
            sched = torch.tensor([500, 400])
            pnp_utils.register_spatial_attention_pnp(Pipe, sched, True)
            pnp_utils.register_temp_attention_pnp(Pipe, sched, True)
            pnp_utils.register_temp_conv_injection(Pipe, sched)
            pnp_utils.register_out_conv_injection(Pipe, sched)
            pnp_utils.register_resnet_injection(Pipe, sched)
            pnp_utils.register_time_all(Pipe, 500, masks)
        return hooks

    if which == "cfg4":
        # BASELINE configs[3] at its real size: 32 frames, 768x768 -> 96x96 latents (294 912 rows at L0), all-to-all form
        compare("cfg4_b1", 1, 32, 96, 96, False, modes=("a2a",))
    elif which == "full":
        compare("full_b1", 1, 16, 32, 32, False)
        compare("full_pnp_b5", 5, 8, 32, 32, False, pnp_hooks(8, 32, 6))  # composition-shaped step at production widths
    else:
        F = 4 * world
        compare("plain_b1", 1, F, 16, 16, False)
        compare("multiframe_b2", 2, F, 16, 16, True)
        # composition step: batch of 5 with every injection site live (soft temporal masks at a different resolution
        # than some feature maps -> the nearest resize of the pixel-sharded masks is exercised)
        hooks = pnp_hooks(F, 16, 5)
        compare("pnp_b5", 5, F, 16, 16, True, hooks)
    json.dump(report, open(os.path.join(out_dir, f"unet_r{rank}.json"), "w"))


def pipeline(out_dir):
    """5-step DDIM inversion (cfg 7.5 -> UNet batch 2) through the pipeline: frame-sharded vs unsharded, files by rank 0 only"""
    from mvoc_amd.unet import I2VGenXLUNet
    from mvoc_amd.unet_spec import UNetConfig
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler

    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0"
    cfg = UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=8, cross_attention_dim=64,
                     attention_head_dim=64, transformer_in_heads=2, context_pool=8)
    f, h, w = 2 * world, 16, 16
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).half().to(dev)
    x0, pe, ne, ie, il = r(1, 4, f, h, w), r(1, 7, 64), r(1, 7, 64), r(1, 1, 64), r(1, 4, f, h, w)
    seqs = []
    for sharded in (False, True):
        eng = I2VGenXLUNet(cfg, device=dev).init_random(31)
        pipe = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=not sharded)
        if sharded:
            pipe.enable_frame_shard(FrameShard())
        d = os.path.join(out_dir, f"lat_{'shard' if sharded else 'single'}_r{rank}" if not sharded else "lat_shard")
        inv = pipe.invert(height=h * 8, width=w * 8, num_frames=f, num_inference_steps=5, guidance_scale=7.5, target_fps=8,
                          latents=x0, prompt_embeds=pe, negative_prompt_embeds=ne, image_embeddings=ie, image_latents=il,
                          return_dict=False, output_dir=d)
        seqs.append(inv.float())
    dist.barrier()
    diff = (seqs[0] - seqs[1]).abs().max().item()
    files = sorted(os.listdir(os.path.join(out_dir, "lat_shard")))
    same = all(torch.equal(torch.load(os.path.join(out_dir, "lat_shard", f"ddim_latents_{t}.pt")), seqs[1][0, 4 - i][None].half().cpu())
               for i, t in enumerate((1, 201, 401, 601, 801)))
    json.dump({"rank": rank, "max_abs": diff, "files": files, "files_match": same},
              open(os.path.join(out_dir, f"pipeline_r{rank}.json"), "w"))


def rccl_native(out_dir):
    """the library's own RCCL communicator (C ABI mvoc_comm_* / mvoc_allgather_frames / mvoc_alltoall_frames): with the world
    this test box allows (one GPU -> one rank) the exchanges are identities, which checks symbol resolution, the id hand-off,
    communicator life cycle and stream semantics; the engine then runs frame-sharded on that transport"""
    from mvoc_amd.unet import I2VGenXLUNet
    from mvoc_amd.unet_spec import UNetConfig
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    sh = FrameShard(transport="rccl")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2 * 4 * 6, 16, generator=g).half().to(dev)
    px = sh.to_pixel_shard(x, 2, 4, 6)
    back = sh.to_frame_shard(px, 2, 4, 6)
    parts = sh.all_gather(x)
    torch.cuda.synchronize()
    ok = bool(torch.equal(back, x)) and tuple(parts.shape) == (world,) + tuple(x.shape) and bool(torch.equal(parts[rank], x))
    cfg = UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=8, cross_attention_dim=64,
                     attention_head_dim=64, transformer_in_heads=2, context_pool=8)
    eng = I2VGenXLUNet(cfg, device=dev).init_random(31)
    eng.prune_dead_chunks = False  # (frame-sharded forwards compute every chunk: compare like with like)
    from mvoc_amd.unet import Linear
    Linear.USE_GN_FOLD = False  # (... and apply their GroupNorms as kernels -- the fold into proj_in is a single-GPU form)
    r = lambda *s_: torch.randn(*s_, generator=g).half().to(dev)
    inp = (r(1, 4, 4, 16, 16), torch.tensor([500.0]).to(dev), torch.full((1,), 8.0).to(dev), r(1, 4, 4, 16, 16), r(1, 4, 4, 16, 16),
           r(1, 4, 64), r(1, 7, 64))
    ref = eng.forward_ext(*inp)[0]
    eng.set_frame_shard(sh)
    got = eng.forward_ext(*inp)[0]
    eng.set_frame_shard(None)
    torch.cuda.synchronize()
    sh.close()
    # (Capturing a sharded iteration on this transport into a hipGraph was tried in round 3: the capture of the RCCL calls
    # never returns at world size 1 with the RCCL copy torch ships -- the worker had to be killed by the test's timeout -- so
    # sharded iterations stay eager)
    json.dump({"rank": rank, "exchanges_ok": ok, "forward_max_abs": float((got.float() - ref.float()).abs().max())},
              open(os.path.join(out_dir, f"rccl_r{rank}.json"), "w"))


def rccl_native_multi(out_dir):
    """the NATIVE transport across real GPUs (one rank per device): every rank builds the same seeded full tensor, so the result
    of each exchange is known by slicing -- peer order, per-peer byte counts and the all-to-all / all-gather signatures are
    verified against it for both exchange forms, then a frame-sharded forward against the unsharded one.  Only reachable on a
    box with >= 2 GPUs (tests/test_frame_shard_gpu.py skips otherwise)."""
    from mvoc_amd.unet import I2VGenXLUNet
    from mvoc_amd.unet_spec import UNetConfig
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = f"cuda:{int(os.environ.get('LOCAL_RANK', rank))}"
    torch.cuda.set_device(dev)
    rep = {"rank": rank, "world": world}
    B, F, HW, C = 2, 4 * world, 6 * world, 16
    g = torch.Generator().manual_seed(7)
    full = torch.randn(B, F, HW, C, generator=g).half().to(dev)
    fl, pl = F // world, HW // world
    mine_f = full[:, rank * fl:(rank + 1) * fl].reshape(-1, C).contiguous()                 # this rank's frames, all pixels
    mine_p = full[:, :, rank * pl:(rank + 1) * pl].reshape(-1, C).contiguous()              # all frames, this rank's pixels
    for ex in ("a2a", "allgather"):
        with FrameShard(exchange=ex, transport="rccl", device=dev) as sh:
            px = sh.to_pixel_shard(mine_f, B, fl, HW)
            back = sh.to_frame_shard(px, B, fl, HW)
            parts = sh.all_gather(mine_f)
            torch.cuda.synchronize()
            want_parts = torch.stack([full[:, r * fl:(r + 1) * fl].reshape(-1, C) for r in range(world)])
            rep[ex] = {"pixel_shard": bool(torch.equal(px, mine_p)), "frame_shard": bool(torch.equal(back, mine_f)),
                       "all_gather": bool(torch.equal(parts, want_parts))}
    cfg = UNetConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=8, cross_attention_dim=64,
                     attention_head_dim=64, transformer_in_heads=2, context_pool=8)
    eng = I2VGenXLUNet(cfg, device=dev).init_random(31)
    eng.prune_dead_chunks = False
    r = lambda *s_: torch.randn(*s_, generator=g).half().to(dev)
    nf = 2 * world
    inp = (r(1, 4, nf, 16, 16), torch.tensor([500.0]).to(dev), torch.full((1,), 8.0).to(dev), r(1, 4, nf, 16, 16), r(1, 4, nf, 16, 16),
           r(1, nf, 64), r(1, 7, 64))
    ref = eng.forward_ext(*inp)[0]
    with FrameShard(transport="rccl", device=dev) as sh:
        eng.set_frame_shard(sh)
        got = eng.forward_ext(*inp)[0]
        eng.set_frame_shard(None)
        torch.cuda.synchronize()
    rep["forward"] = {"rel_l2": float((got.float() - ref.float()).norm() / ref.float().norm()),
                      "max_abs": float((got.float() - ref.float()).abs().max()), "ref_max": float(ref.float().abs().max())}
    json.dump(rep, open(os.path.join(out_dir, f"rccl2_r{rank}.json"), "w"))


if __name__ == "__main__":
    mode, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    try:
        if mode == "exchange":
            exchange(out_dir)
        elif mode == "pipeline":
            pipeline(out_dir)
        elif mode == "rccl":
            rccl_native(out_dir)
        elif mode == "rccl2":
            rccl_native_multi(out_dir)
        else:
            unet(out_dir, sys.argv[3] if len(sys.argv) > 3 else "tiny")
        dist.barrier()
    finally:
        dist.destroy_process_group()
