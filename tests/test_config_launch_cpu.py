"""CPU: OmegaConf shim known-answer tests, round-robin sharding, and the world_size-2 (gloo) launch path that
bench.py / the drivers use for N>1 (barrier + max-reduce only: the data path has no collective)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_shim_known_answers():
    from mvoc_amd.config import OmegaConf
    t = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
    t.data_dir = "/D"
    c = OmegaConf.merge(t, OmegaConf.create({"video_name": "boat", "image_size": [128, 64], "recon_config": {"enable_recon": True}}))
    # nested ${...} chains and whole-value references keeping their type (hand-derived from the YAML)
    assert c.output_dir == "/D/inversions/i2vgen-xl/boat"
    assert c.inverse_config.output_dir == "/D/inversions/i2vgen-xl/boat/ddim_latents"
    assert c.recon_config.ddim_latents_path == "/D/inversions/i2vgen-xl/boat/ddim_latents"
    assert c.inverse_config.image_size == [128, 64] and c.inverse_config.n_frames == 4
    assert c.recon_config.enable_recon is True and c.recon_config.n_steps == 5  # deep merge keeps siblings
    assert t.video_name == "ReplaceMe"  # merge does not mutate its inputs
    c.video_path = os.path.join(c.video_dir, c.video_name + ".mp4")
    assert c.video_path == "/D/demo/boat.mp4"
    assert "video_path" in c and c.get("nope", 3) == 3
    ct = OmegaConf.load(os.path.join(REPO, "tests", "data", "composite_template.yaml"))
    ct.data_dir = ".."
    cc = OmegaConf.merge(ct, OmegaConf.create({"video_name": "v", "edited_video_name": "e n", "task_name": "T", "fusion_step": [0, 1]}))
    assert cc.output_dir == "../Results/T/i2vgen-xl/v/e n/" and cc.fusion_step == [0, 1]
    assert cc.bg_ddim_latents_path == "../inversions/i2vgen-xl/v/ddim_latents"
    assert "Results" in OmegaConf.to_yaml(cc, resolve=True)


def test_round_robin_sharding():
    from mvoc_amd.launch import my_entries
    entries = [{"active": i != 2, "video_name": str(i)} for i in range(8)]
    got = [[e["video_name"] for e in my_entries(entries, f"{r}/3")] for r in range(3)]
    assert got == [["0", "4", "7"], ["1", "5"], ["3", "6"]]
    assert sorted(sum(got, [])) == ["0", "1", "3", "4", "5", "6", "7"]  # every active entry exactly once


def test_two_process_gloo_launch(tmp_path):
    """the N>1 protocol of bench.py on CPU: env rendezvous on 127.0.0.1, barrier, MAX all-reduce of the local
    times, per-rank shards with no data exchange"""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import json, os, sys, time
        sys.path.insert(0, {REPO!r})
        import torch, torch.distributed as dist
        from mvoc_amd.launch import my_entries, shard_of
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        assert (rank, world) == shard_of(None)
        mine = my_entries([{{"active": True, "video_name": str(i)}} for i in range(5)])
        dist.barrier()
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        json.dump({{"rank": rank, "mine": [e["video_name"] for e in mine], "tmax": float(t)}}, open(os.path.join({str(tmp_path)!r}, f"r{{rank}}.json"), "w"))
        dist.barrier()
        dist.destroy_process_group()
    """))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.load(open(tmp_path / f"r{i}.json")) for i in range(2)]
    assert out[0]["mine"] == ["0", "2", "4"] and out[1]["mine"] == ["1", "3"]
    assert out[0]["tmax"] == out[1]["tmax"] == pytest.approx(0.2)


def test_bench_gpus2_launched_plainly_starts_two_ranks():
    """the driver's command form `python bench.py --gpus N ...` with no torchrun around it: bench.py starts the N ranks itself
    (child processes, before any GPU call), relays rank 0's ONE JSON line, and says n_gpus = N (launcher self-test mode: gloo,
    sleeps instead of UNet steps -- this container has no GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--selftest-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ms_per_step"] >= 2.0  # the MAX over ranks (rank 1 sleeps 2 ms per step, rank 0 one)


def test_bench_self_launch_fails_when_a_rank_fails_and_refuses_too_many_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=dict(env, MVOC_BENCH_SELFTEST_FAIL_RANK="1"))
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # the real workload: more ranks than visible devices is refused before anything is launched (no GPU here: 0 visible)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64", "--steps", "4"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)
    # and a rank count that contradicts the launcher's WORLD_SIZE is an error, never a silent one-GPU run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "4", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_flop_model_matches_survey():
    """2.48 / 20.96 / 104.82 TFLOP per UNet forward"""
    from mvoc_amd.flops import unet_flops
    from mvoc_amd.unet_spec import UNetConfig
    c = UNetConfig()
    assert abs(unet_flops(c, 1, 8, 32, 32)["total"] / 1e12 - 2.48) < 0.02
    assert abs(unet_flops(c, 1, 16, 64, 64)["total"] / 1e12 - 20.96) < 0.1
    assert abs(unet_flops(c, 5, 16, 64, 64)["total"] / 1e12 - 104.82) < 0.5


def test_latent_cache_roundtrip(tmp_path):
    from mvoc_amd.latent_cache import LatentCache, latent_file
    from mvoc_amd.utils import load_ddim_latents_at_t
    c = LatentCache("cpu")
    x = torch.randn(1, 4, 3, 5, 6).half()
    c.put(str(tmp_path / "d"), 981, x)
    c.flush()
    assert os.path.basename(latent_file(str(tmp_path / "d"), 981)) == "ddim_latents_981.pt"
    y = load_ddim_latents_at_t(981, str(tmp_path / "d"))  # the reference's reader signature
    assert y.dtype == torch.float16 and torch.equal(x, y)
    c2 = LatentCache("cpu")
    assert torch.equal(c2.get(str(tmp_path / "d"), 981), x)
    with pytest.raises(AssertionError):
        c2.get(str(tmp_path / "d"), 961)


def test_committed_bench_line_has_the_contract_fields():
    """the bench.py JSON line committed with the round's profiles carries every field of the measurement contract"""
    prof = os.path.join(REPO, "profiles")
    rounds = sorted(d for d in os.listdir(prof) if os.path.isdir(os.path.join(prof, d)))
    rounds = [d for d in rounds if any(f.endswith("_bench.json") for f in os.listdir(os.path.join(prof, d)))]
    assert rounds, "no committed bench line"
    rnd = rounds[-1]  # the latest round that has one (a round's directory fills up as the round goes)
    files = sorted(f for f in os.listdir(os.path.join(prof, rnd)) if f.endswith("_bench.json"))
    j = json.load(open(os.path.join(prof, rnd, files[-1])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "steps/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f16" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"]
    assert abs(j["value"] - j["n_gpus"] * j["steps"] / (j["ms_per_step"] * j["steps"] / 1e3)) / j["value"] < 1e-3
