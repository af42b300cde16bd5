"""Two runs of N seeded C2 steps (tile database, two streams, everything on) must end on bit-identical weights: every reduction in the library has a
fixed order, so a difference means a race (a wait counted wrong, an LDS image read before its barrier).    python tools/debug/determinism_soak.py [net] [steps]"""
import hashlib, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from semiuhpe_amd.src.agent import SSLAgent
    net, steps = sys.argv[2], int(sys.argv[3])
    torch.manual_seed(1234)
    cfg = bench.make_config(f"/tmp/suhpe_det_{os.getpid()}")
    cfg.network = net
    agent = SSLAgent(cfg)
    B = int(os.environ.get("DET_B", "32"))
    data, ulb = bench.synthetic(0, torch.device("cuda", 0), B, 4 * B, size=int(os.environ.get("DET_SIZE", "224")))
    cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
    torch.manual_seed(100)
    for i in range(steps):
        fd, fu, out = agent.train_func(data, ulb)
        agent.clock.tick()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, v in sorted(agent.net.state_dict().items()):
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    for k, v in sorted(agent.ema_net.state_dict().items()):
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    print("DIGEST", h.hexdigest(), "loss", float(out["loss_all"]), flush=True)
    sys.exit(0)
net = sys.argv[1] if len(sys.argv) > 1 else "effinetv2"
steps = sys.argv[2] if len(sys.argv) > 2 else "60"
outs = []
for r in range(2):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", net, steps], capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("DIGEST")]
    print(r, line or p.stderr[-800:], flush=True)
    outs.append(line[0] if line else None)
print("IDENTICAL" if outs[0] and outs[0] == outs[1] else "DIFFERENT")
