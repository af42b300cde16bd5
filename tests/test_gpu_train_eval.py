"""End-to-end on the GPU box: train.py (stage 1 -> stage switch -> SSL stage with dynamic threshold) and eval.py on
synthetic heads; checks the log-line tags, the 4-key checkpoint with reference-compatible state_dict names, and that a
checkpoint reloads into the agent bit-exactly."""
import glob
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_eval(tmp_path):
    env = dict(os.environ, SUHPE_SYNTH_SIZE="64", SUHPE_SYNTH_BATCHES="4", PYTHONPATH=ROOT)
    common = ["--config", os.path.join(ROOT, "settings", "300WLP_AFLW2000.yml"), "--network", "effinetv2", "--synthetic",
              "--exp_root", str(tmp_path), "--batch_size", "8", "--ulb_batch_ratio", "2"]
    cmd = [sys.executable, os.path.join(ROOT, "train.py")] + common + [
        "--stage1_iteration", "4", "--max_iteration", "8", "--log_frequency", "2", "--val_frequency", "2", "--save_frequency", "6",
        "--dynamic_thres", "--cutout_aug", "--cutmix_aug", "--rotate_aug", "--left_ratio", "0.75"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    runs = glob.glob(os.path.join(str(tmp_path), "300WLP_AFLW2000", "*", "*"))
    assert len(runs) == 1
    log = open(glob.glob(os.path.join(runs[0], "txt_*.log"))[0]).read()
    for tag in ("train/lr:", "train/loss:", "train/err_mean:", "train_SSL/unsuper_loss:", "train_SSL/entropy_mean:",
                "train_SSL/mask_ratio:", "train_SSL/err_strongSuper_pseudo:", "test/loss:", "test/err_mean:", "test_ema/loss:",
                "test/mask_ratio:", "settings/conf_thres:"):
        assert tag in log, tag
    assert "train/lr:0.000100" in log and "train/lr:0.000010" in log          # lr x0.1 at the stage switch
    ck = torch.load(os.path.join(runs[0], "best.pth"), map_location="cpu")
    assert set(ck) == {"clock", "model_state_dict", "model_state_dict_ema", "optimizer_state_dict"}
    assert set(ck["clock"]) == {"epoch", "minibatch", "iteration"}
    sd = ck["model_state_dict"]
    assert "features.0.0.weight" in sd and "classifier.7.bias" in sd and len(sd) == 796
    assert all(v.is_contiguous() for v in sd.values())
    st = ck["optimizer_state_dict"]["state"]
    assert len(st) == 460 and set(st[0]) == {"step", "exp_avg", "exp_avg_sq"}
    exp_detail, date = runs[0].split(os.sep)[-2:]
    cmd = [sys.executable, os.path.join(ROOT, "eval.py")] + common + [f"{exp_detail}/{date}/best"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Euler MAE (pitch, yaw, roll, mean):" in r.stdout and "EMA_Euler MAE" in r.stdout
    assert os.path.exists(os.path.join(str(tmp_path), "exps", "MAE_300WLP_AFLW2000_effinetv2_AFLW2000.json"))


def test_two_rank_training_with_autotune_and_a_ragged_validation_deal(tmp_path):
    """with the autotuner ON, two ranks and an odd number of validation batches (one rank meets a batch shape the other never sees) the job
    must finish: the autotuner enters no collective, so the ranks' collectives stay paired (gloo rehearsal of the two-rank flow on one GPU)."""
    env = dict(os.environ, SUHPE_SYNTH_SIZE="64", SUHPE_SYNTH_BATCHES="3", SUHPE_AUTOTUNE="1", SUHPE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29561",
           os.path.join(ROOT, "train.py"), "--config", os.path.join(ROOT, "settings", "300WLP_AFLW2000.yml"), "--network", "effinetv2", "--synthetic",
           "--exp_root", str(tmp_path), "--batch_size", "8", "--ulb_batch_ratio", "2", "--stage1_iteration", "2", "--max_iteration", "6",
           "--log_frequency", "2", "--val_frequency", "2", "--save_frequency", "6", "--dynamic_thres", "--cutout_aug", "--cutmix_aug", "--rotate_aug",
           "--left_ratio", "0.75"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    runs = glob.glob(os.path.join(str(tmp_path), "300WLP_AFLW2000", "*", "*"))
    assert len(runs) == 1 and os.path.exists(os.path.join(runs[0], "best.pth"))
    log = open(glob.glob(os.path.join(runs[0], "txt_*.log"))[0]).read()
    assert "test/err_mean:" in log and "train_SSL/unsuper_loss:" in log


def test_checkpoint_roundtrip_is_exact(tmp_path):
    from types import SimpleNamespace

    from semiuhpe_amd.src.agent import SSLAgent
    from semiuhpe_amd.synthetic import synthetic_batches
    cfg = dict(lr=1e-4, log_dir=str(tmp_path), model_dir=str(tmp_path), distribution="matrixFisher", is_full_range=False,
               rotate_aug=False, train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=False, conf_thres=1e9,
               type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network="effinetv2",
               num_classes=9, exp_name="t", date="t", left_ratio=0.95, save_feat=False)
    a = SSLAgent(SimpleNamespace(**cfg))
    data, ulb = synthetic_batches(4, 8, size=64, seed=1)
    data = {k: v.cuda() for k, v in data.items()}
    ulb = {k: v.cuda() for k, v in ulb.items()}
    for _ in range(2):
        a.train_func(data, ulb)
        a.clock.tick()
    a.save_ckpt("best")
    b = SSLAgent(SimpleNamespace(**cfg))
    b.load_ckpt("best")
    assert b.clock.iteration == 2
    for (n, p), (_, q) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert torch.equal(p, q), n
    for (n, p), (_, q) in zip(a.ema_net.state_dict().items(), b.ema_net.state_dict().items()):
        assert torch.equal(p, q), n
    fa, fb = a.optimizer.flat_buffers[0], b.optimizer.flat_buffers[0]
    assert torch.equal(fa.exp_avg, fb.exp_avg) and torch.equal(fa.exp_avg_sq, fb.exp_avg_sq) and fa.step == fb.step == 2
    with pytest.raises(ValueError):
        b.load_ckpt("does_not_exist")
    # same next step from both agents (same masks via the same device RNG state is not guaranteed -> eval-mode check)
    a.net.eval(); b.net.eval()
    with torch.no_grad():
        assert torch.equal(a.net(data["img"]), b.net(data["img"]))
    # a save must not freeze the live optimiser state (Optimizer.state_dict() hands out the live dicts): train on, save again,
    # reload -> the SECOND checkpoint carries the current Adam moments and step, and the live state entries are still device views
    for _ in range(2):
        a.train_func(data, ulb)
        a.clock.tick()
    a.save_ckpt("best")
    st0 = a.optimizer.state[fa.params[0]]
    assert st0["exp_avg"].is_cuda and st0["exp_avg"].data_ptr() == fa.exp_avg.data_ptr() and st0["step"] is fa.step_t
    c = SSLAgent(SimpleNamespace(**cfg))
    c.load_ckpt("best")
    fc = c.optimizer.flat_buffers[0]
    assert c.clock.iteration == 4 and fc.step == fa.step == 4
    assert torch.equal(fa.exp_avg, fc.exp_avg) and torch.equal(fa.exp_avg_sq, fc.exp_avg_sq)
    assert not torch.equal(fb.exp_avg, fc.exp_avg)            # (moments did move between the two saves)


def test_bench_two_rank_flow_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on a single GPU
    with the gloo backend: both ranks must reach the end (the instrumented roofline steps all-reduce too) and rank 0 must
    print exactly one JSON line with the contract's keys."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SUHPE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 320
    assert "cpu_baseline" not in out                       # reported at N=1 only


def _bench_line(cmd, env, timeout=900):
    import json, subprocess
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (how the driver's BENCH/SCALE commands look): the parent spawns the two ranks
    itself before touching the GPU and forwards rank 0's line: n_gpus 2, global batch 320.  One GPU here -> gloo rehearsal."""
    env = dict(os.environ, SUHPE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 320 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and "cpu_baseline" not in out


def test_bench_eight_rank_rehearsal():
    """`python bench.py --gpus 8` exactly as the driver's SCALE command starts it (no launcher: the parent spawns the 8 ranks), on ONE GPU
    with the gloo backend and 96-pixel images: eight children, one JSON line, n_gpus 8, global batch 1280, the bucketed gradient exchange
    under the backward, and replicas that hold the same bits after the steps."""
    env = dict(os.environ, SUHPE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SUHPE_STEPS_IN_FLIGHT="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--size", "96"], env, timeout=1200)
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 1280 and out["config"]["parallelism"] == "dp8"
    assert out["config"]["rehearsal_image_size"] == 96 and out["scaling"] == "weak"
    assert out["config"]["replicas_identical"] is True
    assert out["value"] > 0 and "cpu_baseline" not in out and "other_configs" not in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_bench_two_ranks_over_rccl():
    """the same command on the backend it is written for (nccl = RCCL over xGMI)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SUHPE_DIST_BACKEND"):
        env.pop(k, None)
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"], env)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 320


def test_bucketed_exchange_bit_identical(tmp_path):
    """Two data-parallel ranks (gloo rehearsal on one GPU): the gradient exchange in reverse-layer buckets, issued while the
    backward is still running, gives the SAME BITS as one all-reduce after the backward (SUHPE_GRAD_BUCKETS=1), and both ranks
    hold identical parameters, teacher and BatchNorm buffers after two SSL steps + one supervised step."""
    import subprocess
    res = {}
    for mode in ("4", "1"):
        out = tmp_path / f"b{mode}"
        out.mkdir()
        env = dict(os.environ, SUHPE_GRAD_BUCKETS=mode, SUHPE_AUTOTUNE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29547", os.path.join(ROOT, "tests", "ddp_worker.py"), str(out)]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[mode] = [torch.load(out / f"rank{k}.pt") for k in range(2)]
    assert res["4"][0]["buckets"] is not None and len(res["4"][0]["buckets"]) >= 4 and res["1"][0]["buckets"] is None
    for mode in res:
        a, b = res[mode]
        for k in ("params", "grads", "ema", "bn"):
            assert torch.equal(a[k], b[k]), (mode, k)               # replicas stay identical
    for k in ("params", "grads", "ema", "bn"):
        assert torch.equal(res["4"][0][k], res["1"][0][k]), k       # bucketing does not change a bit
    assert float(res["4"][0]["grads"].abs().max()) > 0


def test_ranks_agree_on_measured_tiles(tmp_path):
    """two data-parallel ranks (gloo rehearsal on one GPU), autotuner ON, EMPTY tile database -- every shape of the step is one
    the shipped database does not hold.  Rank 0 measures the menu on a dry pass at the first step of each kind and broadcasts its table
    (SSLAgent._agree_on_tiles): both ranks end with the SAME table, the library reports the same running kernel for every key on both ranks,
    measured (non-heuristic) tiles -- Winograd among them -- are in use, and the replicas hold the same bits after the steps."""
    import json, subprocess
    db = tmp_path / "empty_tiles.json"
    db.write_text(json.dumps({"tiles": {}}))
    env = dict(os.environ, SUHPE_AUTOTUNE="1", SUHPE_TILE_DB=str(db), HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, SUHPE_STEPS_IN_FLIGHT="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SUHPE_TILE_DB_SAVE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "tests", "tile_agree_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    a, b = (torch.load(tmp_path / f"rank{k}.pt") for k in range(2))
    assert a["agreed"] == b["agreed"] == ["s1", "ssl"]
    assert a["table"] == b["table"] and len(a["table"]) > 100          # identical pins by name; forward, input-gradient and filter-gradient shapes
    assert a["running"] == b["running"]                                # ... and the library of each rank resolves them to the same kernels
    names = set(a["running"].values())
    assert any(n.startswith("wino") for n in names), sorted(names)     # Winograd tiles in use
    assert any(n.startswith("gemm_nt_") for n in names), sorted(names)  # and tiles no heuristic ever picks (persistent / resident / LDS-DMA)
    for k in ("params", "ema", "bn"):
        assert torch.equal(a[k], b[k]), k                              # replicas_identical


@pytest.mark.parametrize("network", ["effinetv2", "repvgg"])
def test_foreign_checkpoint_loads_and_evaluates(tmp_path, network):
    """A checkpoint written the way the REFERENCE writes it (src/agent.py:302-331: plain torch modules -- torchvision / RepVGG key
    order, NCHW-contiguous filters, torch.optim.Adam's state) loads through SSLAgent.load_ckpt and evaluates to the oracle's
    rotation matrices (the released *.pth files are not available offline; this is their format, produced by the oracle modules)."""
    from types import SimpleNamespace

    from oracle import fisher as OF
    from oracle import networks as ON
    from parity_utils import seeded_state_dict
    from semiuhpe_amd.src.agent import SSLAgent
    torch.manual_seed(5)
    make = {"effinetv2": ON.EfficientNetV2S, "repvgg": ON.RepVggNet}[network]
    net, ema = make(), make()
    if network == "repvgg":
        net.load_state_dict(seeded_state_dict(net.state_dict(), 31))
    else:
        for m in net.classifier:
            if isinstance(m, torch.nn.Linear):
                m.weight.data.mul_(20.0)
    ema.load_state_dict(net.state_dict())
    with torch.no_grad():
        for p in ema.parameters():
            p.mul_(0.99)
    opt = torch.optim.Adam(net.parameters(), 1e-4)
    x = torch.randn(4, 3, 96, 96)
    net.train()
    net(x).square().mean().backward()
    opt.step()                                                   # Adam state with step = 1 and real moments
    assert all(v.is_contiguous() for v in net.state_dict().values())
    path = str(tmp_path / "foreign-best.pth")
    torch.save({"clock": {"epoch": 3, "minibatch": 7, "iteration": 1234}, "model_state_dict": net.state_dict(),
                "model_state_dict_ema": ema.state_dict(), "optimizer_state_dict": opt.state_dict()}, path)
    cfg = SimpleNamespace(lr=1e-4, log_dir=str(tmp_path), model_dir=str(tmp_path), distribution="matrixFisher", is_full_range=network == "repvgg",
                          rotate_aug=False, train_labeled="DAD3DHeads" if network == "repvgg" else "300WLP", train_unlabeled="300WLP",
                          cutout_aug=False, cutmix_aug=False, conf_thres=1e9, type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999,
                          eman=False, stage1_iteration=0, network=network, num_classes=9, exp_name="t", date="t", left_ratio=0.95,
                          save_feat=False, autotune=False)
    agent = SSLAgent(cfg)
    agent.load_ckpt(path)
    assert agent.clock.iteration == 1234 and agent.clock.epoch == 3
    fb = agent.optimizer.flat_buffers[0]
    assert fb.step == 1
    st = opt.state_dict()["state"]
    for i, p in enumerate(fb.params):
        assert torch.equal(fb.view_like(fb.exp_avg, i).cpu(), st[i]["exp_avg"]), i
    net.eval(); ema.eval()
    xe = torch.randn(6, 3, 96, 96, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        for ref, hip in ((net, agent.net), (ema, agent.ema_net)):
            hip.eval()
            a_ref, a = ref(xe), hip(xe.cuda())
            assert float((a.cpu() - a_ref).abs().max()) < 2e-4 * max(1.0, float(a_ref.abs().max()))
            R_ref = OF.batch_torch_A_to_R(a_ref)
            from semiuhpe_amd import kernels as K
            assert float((K.fisher_svd_project(a).cpu() - R_ref).abs().max()) < 1e-4
    # and the state dict written back has the reference's keys in the reference's order
    assert list(agent.net.state_dict().keys()) == list(net.state_dict().keys())


def test_rccl_api_on_one_rank():
    """the data-parallel plumbing on the nccl (= RCCL) backend with one rank: real RCCL calls, handles and stream hand-offs
    (tests/nccl_single_worker.py); the boxes of this pool have a single GPU, so N > 1 over RCCL stays for the driver's SCALE run"""
    import socket, subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SUHPE_AUTOTUNE="0", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_single_worker.py"), str(port)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "NCCL_SINGLE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
