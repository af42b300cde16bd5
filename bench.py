#!/usr/bin/env python
"""bench.py — throughput of SemiUHPE's teacher-student training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one process per GPU over RCCL.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
this process is one rank; started bare (`python bench.py --gpus N`, no WORLD_SIZE) it spawns the N ranks itself as child
processes -- before anything in the parent touches the GPU -- and exits with their status.

Workload (BASELINE.json configs[1], "C2"): SemiUHPE 300WLP_AFLW2000, EfficientNetV2-S, per GPU B=32 labeled +
muB=128 unlabeled 224x224 synthetic heads, dynamic threshold (left_ratio 0.95) + CutOut + CutMix, fp32.
One "step" = SSLAgent.train_func: EMA update, student fwd (labeled), teacher fwd (weak), strong-aug, student fwd
(strong), Fisher NLL + entropy filter + CE, backward over 160 images, gradient all-reduce (N>1), Adam.
Inputs are resident in HBM before the timed region.  metric = counted train images/s = (B + muB) * N / step time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, live HIP-event timing) and `cpu_baseline`
(the CPU oracle timed on this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL between the ranks' processes (set before HIP starts)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_HBM_GBS = 8000.0
B_LABELED, MU = 32, 4


def make_config(log_dir):
    return SimpleNamespace(
        lr=1e-4, log_dir=log_dir, model_dir=log_dir, distribution="matrixFisher", is_full_range=False, rotate_aug=False,
        train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=-5.3,
        type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0,
        network="effinetv2", num_classes=9, exp_name="300WLP_AFLW2000", date="bench", left_ratio=0.95, save_feat=False,
        dynamic_thres=True)


def synthetic(rank, device, B, muB, size=224):
    from semiuhpe_amd.synthetic import synthetic_batches
    data, ulb = synthetic_batches(B, muB, size=size, seed=rank, rotate_aug=False)
    to = lambda d: {k: v.to(device).contiguous() for k, v in d.items()}
    return to(data), to(ulb)


def warm_up_with_live_threshold(agent, cfg, data, ulb, warm, world=1, device=None):
    """`warm` untimed SSL steps (at least one) in the regime the timed steps stand for, with the pseudo-label threshold the way a long run holds it:
      * the EMA decay at its configured 0.999 -- `min(1 - 1 / (k + 1), decay)` of the reference (src/agent.py:277-299) ramps over the first 1 000
        iterations; at iteration 5 the teacher would still move a sixth of the way to the student per step and its entropies, and with them the
        share of pseudo-labels that pass, would drift through the timed region;
      * conf_thres = the left_ratio quantile of the entropies of the teacher AS THE STEP RUNS IT (train mode: batch statistics), taken on
        the first untimed step.  The eval-mode scan (compute_dynamic_entropy_threshold, what train.py calls) uses the running statistics: on a
        random-init network its entropies sit elsewhere and every sample passed the filter (mask_ratio 1.0 in the round-5 line).
        The same scalar on every rank (max over ranks; each rank still filters its own batch)."""
    agent.clock.iteration = 100000
    out = None
    for i in range(max(warm, 1)):
        out = agent.train_func(data, ulb)
        agent.clock.tick()
        if i == 0:
            ent = out[1]["entropy"].detach().float().cpu().numpy().copy()
            ent.sort()
            thr = torch.tensor([float(ent[min(int(len(ent) * cfg.left_ratio), len(ent) - 1)])], dtype=torch.float64, device=device or agent.device)
            if world > 1:
                dist.all_reduce(thr, op=dist.ReduceOp.MAX)
            cfg.conf_thres = float(thr.item())
    return out


def gpu_leg(args, rank, world, device):
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src.agent import SSLAgent
    torch.manual_seed(1234)                                 # same init on every rank (also broadcast by the agent)
    cfg = make_config(os.path.join("/tmp", f"suhpe_bench_{os.getpid()}"))
    agent = SSLAgent(cfg)
    if args.serial:
        agent.overlap_teacher, K.WGRAD_OVERLAP = False, False
    data, ulb = synthetic(rank, device, B_LABELED, B_LABELED * MU, size=args.size)
    # dynamic threshold at warm-up: the batch's left_ratio quantile of teacher entropies, so ~95% pass the filter
    cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
    torch.manual_seed(100 + rank)
    warm_up_with_live_threshold(agent, cfg, data, ulb, args.warmup, world, device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fd, fu, out = agent.train_func(data, ulb)
        agent.clock.tick()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    agent.check_finite()
    gib = 1.0 / (1 << 30)
    info = dict(mask_ratio=float(fu["mask_ratio"]), loss_all=float(out["loss_all"]), conf_thres=float(cfg.conf_thres),
                hbm_gib={"peak_allocated": round(torch.cuda.max_memory_allocated() * gib, 2), "reserved": round(torch.cuda.memory_reserved() * gib, 2)})
    if world > 1:
        # what the step pays for the gradient exchange: mean time per timed step the compute stream waited for the communication stream at the end of the
        # backward (BucketedAllReduce.finish); everything else of the all-reduce ran under the backward.  With it the line explains its own scaling.
        bk = getattr(agent, "_buckets", None)
        ex = bk.exposed_ms(last=args.steps) if bk is not None else None
        info["allreduce_exposed_ms"] = round(ex, 3) if ex is not None else None
        info["grad_buckets"] = (len(bk.bounds) - 1) if bk is not None else 1
        info["transport"] = "suhpe_comm (library-owned RCCL communicator; torch.distributed on gloo as host channel)" if os.environ.get("SUHPE_COMM") == "native" \
            else f"torch.distributed/{dist.get_backend()}"
        # data-parallel replicas must hold the same bits after the timed steps: every rank's student / teacher parameters and
        # BatchNorm buffers as (sum, sum of squares) in float64, gathered and compared (outside the timed region)
        fbs = [fb.data for fb in agent.optimizer.flat_buffers] + [agent._ema_flat.data] + ([agent._bn_flat_all.data] if agent._bn_flat_all is not None else [])
        sig = torch.stack([torch.stack([t.double().sum(), (t.double() ** 2).sum()]) for t in fbs]).flatten()
        if dist.get_backend() != "nccl":
            sig = sig.cpu()                                 # (gloo gathers host tensors)
        sigs = [torch.empty_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        info["replicas_identical"] = bool(all(torch.equal(sigs[0], s_) for s_ in sigs))

    # ---- roofline of the dominant kernel: same step, every dense-conv launch bracketed by HIP events on the
    # launch stream (torch's current stream is the stream the C ABI launches on)
    # Every rank runs the instrumented steps (train_func all-reduces gradients: a rank stepping alone would deadlock the
    # others at their barrier); only rank 0 records and reports.
    # The instrumented steps run on ONE stream (teacher pass and filter-gradient GEMMs back on the main stream): with the
    # product's concurrent streams a kernel's start-to-end time includes the time it shares the GPU with the other stream's
    # kernels, which says nothing about the kernel.  `--serial` runs the whole bench that way (the rocprofv3 summary under
    # profiles/ that these durations must agree with is taken with it).
    roof = None
    nprof = 2
    overlap = (agent.overlap_teacher, K.WGRAD_OVERLAP)
    agent.overlap_teacher, K.WGRAD_OVERLAP = False, False
    agent.train_func(data, ulb)
    agent.clock.tick()
    torch.cuda.synchronize()
    if rank == 0:
        K.PROFILE = []
    t_i = time.perf_counter()
    for _ in range(nprof):
        agent.train_func(data, ulb)
        agent.clock.tick()
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t_i) / nprof * 1e3
    agent.overlap_teacher, K.WGRAD_OVERLAP = overlap
    if rank == 0:
        recs, K.PROFILE = K.PROFILE, None
        # FLOP of a launch = the multiply-adds its algorithm executes: 2 M N K for the direct kernels; the Winograd F(2x2, 3x3) kernel spends 16
        # multiplies per 2x2 output tile and channel pair where the direct sum spends 36, so its launches count 1 / 2.25 of the direct figure
        # (`direct_equivalent_tflops` = the rate a direct kernel would need to finish in the same time; timed with its filter transform launch)
        exec_share = lambda nm: (16.0 / 36.0) if nm.startswith("wino") else 1.0
        agg = {}
        for name, flops, nbytes, e0, e1, _shape in recs:
            a = agg.setdefault(name, [0.0, 0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += flops * exec_share(name)
            a[2] += nbytes
            a[3] += 1
        total_gemm_s = sum(a[0] for a in agg.values())
        # multiply-adds the Winograd launches do NOT execute, per step (what `executed_flop_per_step` subtracts from SURVEY 8d's direct count)
        info["wino_saved_flop_per_step"] = sum(flops * (1.0 - exec_share(nm)) for nm, flops, *_ in recs) / nprof
        name, (secs, flops, nbytes, launches) = max(agg.items(), key=lambda kv: kv[1][0])
        tf = flops / secs / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_MFMA_F32_TFLOPS, 4), "traffic": pmc_traffic(name),
                "algorithmic_bytes_per_launch": round(nbytes / launches),
                "launches_per_step": launches // nprof, "avg_launch_us": round(secs / launches * 1e6, 2),
                "avg_launch_gflop": round(flops / launches / 1e9, 3),
                "kernel_share_of_step": round(secs / nprof / (serial_ms * 1e-3), 3),
                "timing": "HIP events around every dense-conv launch, single-stream steps (with event overhead: %.1f ms/step)" % serial_ms,
                "all_gemm_kernels": {k: dict({"ms_per_step": round(v[0] / nprof * 1e3, 3), "tflops": round(v[1] / v[0] / 1e12, 2),
                                              "launches_per_step": v[3] // nprof},
                                             **({"direct_equivalent_tflops": round(v[1] / exec_share(k) / v[0] / 1e12, 2)} if exec_share(k) != 1.0 else {}))
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])},
                "gemm_ms_per_step": round(total_gemm_s / nprof * 1e3, 2)}
        if exec_share(name) != 1.0:
            roof["direct_equivalent_tflops"] = round(tf / exec_share(name), 2)
            roof["note"] = ("Winograd F(2x2,3x3): `achieved` counts the multiply-adds the algorithm executes (16 per 2x2 tile and channel pair; the direct sum "
                            "has 36), so `frac` is the share of the fp32-MFMA peak the kernel keeps busy; `direct_equivalent_tflops` = SURVEY 8d's 2*M*N*K per "
                            "launch / time, the rate a direct kernel would need (its ratio to the peak exceeds 1); launches timed with their filter-transform launch")
    agent.logs_writer.close()
    return dt, info, roof


def csrc_digest():
    """sha256 over the kernel sources (semiuhpe_amd/csrc/*.hip|*.inl|*.h + include/suhpe.h), in name order: what a counter pass was
    taken on.  tools/pmc_traffic.py stores it in profiles/pmc_traffic.json."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "semiuhpe_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inl", ".h")))
    for f in files + [os.path.join(ROOT, "include", "suhpe.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_traffic.py and profiles/README.md).  Self-checking: the passes record the digest of the kernel sources they ran on;
    when the sources have changed since, the figure is withheld (None) instead of going stale silently.  None also when that kernel
    was not in the profiled run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        doc = json.load(fh)
    if doc.get("csrc_digest") != csrc_digest():
        print(f"bench.py: profiles/pmc_traffic.json was taken on other kernel sources (digest {doc.get('csrc_digest')} != {csrc_digest()}): "
              "roofline.traffic withheld; re-run tools/profile_round.sh", file=sys.stderr)
        return None
    rec = doc.get("kernels", {}).get(kernel)
    return rec["hbm_bytes_per_launch"] if rec else None


def pmc_step_bytes(network="effinetv2"):
    """HBM bytes of one steady-state training step from the committed PMC passes (tools/pmc_traffic.py `step`), or None when the passes were
    taken on other kernel sources / are absent.  profiles/pmc_traffic.json is the C2 workload, pmc_traffic_<network>.json the others."""
    name = "pmc_traffic.json" if network == "effinetv2" else f"pmc_traffic_{network}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        doc = json.load(fh)
    if doc.get("csrc_digest") != csrc_digest() or not doc.get("step"):
        return None
    return doc["step"]["hbm_bytes"]


def flatter_proof(s, ms, wino_saved, network):
    """The three figures that cannot be flattered, beside SURVEY 8d's direct-basis `frac`: the multiply-adds the step's
    kernels EXECUTE (Winograd launches count 16 of the direct sum's 36), that work against the fp32-MFMA peak over the whole measured step, and
    the measured HBM traffic of a step over the algorithmic bytes."""
    executed = s["flop"] - wino_saved
    hb = pmc_step_bytes(network)
    return {"executed_flop_per_step": int(executed), "step_mfma_frac": round(executed / (PEAK_MFMA_F32_TFLOPS * 1e12) / (ms * 1e-3), 4),
            "hbm_bytes_per_step_pmc": hb, "hbm_traffic_ratio": (round(hb / s["bytes"], 3) if hb else None)}


def other_config_leg(network, B, muB, warm=3, steps=8):
    """The other single-GPU BASELINE configs, timed by the same process after the C2 region (never `value`): C3 = RepVGG-B1g2 64 + 256
    (DAD3DHeads, full range), C5 per GPU = TinyViT-21M 32 + 128 (300WLP, rotation augmentation); `warm` untimed + `steps` timed SSL
    steps on resident synthetic batches, with the config's own SURVEY 8d step roofline."""
    from semiuhpe_amd.src.agent import SSLAgent
    from semiuhpe_amd.synthetic import synthetic_batches
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from count_work import step_work
    cfg = make_config(os.path.join("/tmp", f"suhpe_bench_{os.getpid()}_{network}"))
    cfg.network = network
    full = network == "repvgg"
    cfg.is_full_range, cfg.rotate_aug = full, True
    cfg.train_labeled = "DAD3DHeads" if full else "300WLP"
    torch.manual_seed(1234)
    agent = SSLAgent(cfg)
    data, ulb = synthetic_batches(B, muB, size=224, seed=0, full_range=full, rotate_aug=True)
    data = {k: v.cuda() for k, v in data.items()}
    ulb = {k: v.cuda() for k, v in ulb.items()}
    cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
    warm_up_with_live_threshold(agent, cfg, data, ulb, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fd, fu, out = agent.train_func(data, ulb)
        agent.clock.tick()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    agent.check_finite()
    s = step_work(network, B, muB)
    from semiuhpe_amd import kernels as K
    K.PROFILE = []                                             # one more step, outside the timed region: which launches ran on the Winograd tiles
    agent.train_func(data, ulb)
    agent.clock.tick()
    torch.cuda.synchronize()
    recs, K.PROFILE = K.PROFILE, None
    wino_saved = sum(flops * (1.0 - 16.0 / 36.0) for nm, flops, *_ in recs if nm.startswith("wino"))
    res = {"network": network, "per_gpu_batch": f"{B}+{muB}", "warmup": warm, "steps": steps, "ms_per_step": round(ms, 2),
           "images_per_s": round((B + muB) / (ms * 1e-3), 1), "mask_ratio": float(fu["mask_ratio"]), "loss_all": float(out["loss_all"]),
           "step_roofline": {"t_roof_ms": round(s["t_roof"] * 1e3, 2), "frac": round(s["t_roof"] * 1e3 / ms, 4), "flop_per_step": s["flop"],
                             "ideal_bytes_per_step": s["bytes"], "achieved_tflops": round(s["flop"] / (ms * 1e-3) / 1e12, 2),
                             "frac_basis": "SURVEY 8d direct count (every conv at 36 multiplies per 2x2 tile): read it WITH step_mfma_frac",
                             **flatter_proof(s, ms, wino_saved, network)}}
    agent.logs_writer.close()
    del agent, data, ulb
    torch.cuda.empty_cache()
    return res


def step_roofline(ms_per_step, wino_saved=0.0):
    """Whole-step roofline of SURVEY.md §8d: T_roof = sum over layers of max(FLOP / 157.3 TF, bytes / 6.3 TB/s) for the
    B + 2 muB image-forwards and B + muB image-backwards of one step (tools/count_work.py), against the measured step."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from count_work import step_work
    s = step_work("effinetv2", B_LABELED, B_LABELED * MU)
    return {"t_roof_ms": round(s["t_roof"] * 1e3, 2), "frac": round(s["t_roof"] * 1e3 / ms_per_step, 4),
            "flop_per_step": s["flop"], "ideal_bytes_per_step": s["bytes"],
            "achieved_tflops": round(s["flop"] / (ms_per_step * 1e-3) / 1e12, 2), "formula": "tools/count_work.py (SURVEY.md 8d)",
            "frac_basis": "SURVEY 8d direct count (every conv at 36 multiplies per 2x2 tile): read it WITH step_mfma_frac",
            **flatter_proof(s, ms_per_step, wino_saved, "effinetv2")}


def cpu_leg():
    """The CPU oracle (our torch-CPU restatement of the reference step, kind="port") on this host's cores, SURVEY.md 8d:
    C1 (the reference's own CPU-runnable config: supervised train_func_s1 steps, B=32) 2 warm-up + 5 timed, and the C2 step
    shape (B=32 + muB=128 SSL step, the workload of `value`) 1 small warm-up + 2 timed -- a bounded sample (about a minute of
    CPU work on 32 threads).  `value` is the C2 rate (same workload as the GPU line).  The same leg holds the Euler-MAE parity
    check of the HIP path against this CPU path on the 1 964 images of the 8d evaluation set (tools/mae_parity.py, about 15 s of CPU
    evaluation; also asserted in tests/test_gpu_mae_parity.py)."""
    from oracle import networks as ON
    from oracle.agent import OracleAgent, synthetic_batches
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # measured on the GPU box (256 hardware threads): 32 torch threads give the best CPU step time (64: 2.3x slower)
    torch.set_num_threads(max(1, min(ncores, 32)))
    torch.manual_seed(1234)
    cfg = make_config("/tmp")
    cfg.conf_thres = 1e9
    agent = OracleAgent(cfg, lambda: ON.EfficientNetV2S())
    d0, u0 = synthetic_batches(2, 8, size=224, seed=0)
    agent.train_func(d0, u0)                                   # warm-up (allocator, thread pool)
    data, ulb = synthetic_batches(B_LABELED, B_LABELED * MU, size=224, seed=0)
    c1_times = []
    for i in range(7):                                         # C1: 2 warm-up + 5 timed supervised steps
        t0 = time.perf_counter()
        agent.train_func_s1(data)
        if i >= 2:
            c1_times.append(time.perf_counter() - t0)
    c2_times = []
    for _ in range(2):                                         # C2 step shape
        t0 = time.perf_counter()
        agent.train_func(data, ulb)
        c2_times.append(time.perf_counter() - t0)
    dt = sum(c2_times) / len(c2_times)
    dt1 = sum(c1_times) / len(c1_times)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    del agent
    mae = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import mae_parity
        mae = mae_parity.run(n_images=mae_parity.EVAL_IMAGES, warm_steps=3, train_batch=(8, 16), size=224, network="effinetv2")
    except Exception as e:                                     # the throughput line must not die with the checker
        mae = {"error": repr(e)}
    return {"value": round((B_LABELED + B_LABELED * MU) / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"C2 step shape: {len(c2_times)} SSL steps B=32+muB=128 EfficientNetV2-S 224x224 fp32 ({dt:.1f} s/step) after a "
                                      f"small warm-up step; C1: 5 supervised steps B=32 after 2 warm-up ({dt1:.2f} s/step); torch-CPU oracle",
            "c1_supervised_bs32": {"value": round(B_LABELED / dt1, 3), "unit": "images/s", "s_per_step": round(dt1, 3), "steps": "2 warm-up + 5 timed"},
            "host_threads": ncores, "cpu": model, "mae_parity": mae}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this parent never initialises the
    GPU and never exec()s), wait for them, kill the rest when one fails, exit non-zero on any failure.  Rank 0 inherits stdout,
    so its JSON line is this command's JSON line."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()                        # counting devices does not initialise HIP
    if have < n and os.environ.get("SUHPE_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (set SUHPE_DIST_BACKEND=gloo to rehearse the {n}-rank flow on fewer)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()                    # exact child PIDs only
        time.sleep(0.2)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 step timings that follow the C2 region")
    ap.add_argument("--serial", action="store_true", help="one stream: no teacher / filter-gradient overlap (per-kernel profiling)")
    ap.add_argument("--size", type=int, default=224, help="image size; anything but 224 is a rehearsal (tests), not the BASELINE workload")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local %= max(1, torch.cuda.device_count())             # (a single-GPU box can rehearse --gpus 2 with SUHPE_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        from semiuhpe_amd.src.distributed import init_from_env
        # the ranks share the host: cap each rank's torch CPU threads at its share of the cores (the hot path has no CPU arithmetic, but
        # 8 ranks x the default thread pool oversubscribe the host on the few torch-CPU ops of the setup)
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(torch.get_num_threads(), ncores // world)))
        init_from_env(os.environ.get("SUHPE_DIST_BACKEND", "nccl"))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):            # library / evaluation prints go to stderr: stdout carries the ONE JSON line
        dt, info, roof = gpu_leg(args, rank, world, device)
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "train images/sec (labeled+unlabeled)", "value": round((B_LABELED + B_LABELED * MU) * world / (dt / args.steps), 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 SemiUHPE 300WLP_AFLW2000: EfficientNetV2-S, per-GPU bs=32 labeled + 128 unlabeled (ulb_ratio 4), "
                                   "224x224 synthetic heads, dynamic_thres(left_ratio 0.95)+CutOut+CutMix, CE unsup loss, Adam lr 1e-4",
                       "per_gpu_batch": "32+128", "parallelism": f"dp{world}", "global_batch": (B_LABELED + B_LABELED * MU) * world,
                       **({} if args.size == 224 else {"rehearsal_image_size": args.size}),
                       "streams": "1 (--serial)" if args.serial else "2 (teacher pass and filter-gradient GEMMs beside the main stream)",
                       **info},
            "roofline": roof,
            "step_roofline": step_roofline(ms, info.pop("wino_saved_flop_per_step", 0.0)),
        }
        if world == 1 and not args.no_other_configs and not args.serial and args.size == 224:
            # the other single-GPU configs of BASELINE.json, after (outside) the C2 timed region: informational, never `value`
            torch.cuda.empty_cache()
            others = {}
            with contextlib.redirect_stdout(sys.stderr):
                for tag, (net, b, mu) in (("C3", ("repvgg", 64, 256)), ("C5_per_gpu", ("tinyvit", 32, 128))):
                    try:
                        others[tag] = other_config_leg(net, b, mu)
                    except Exception as e:                  # the C2 line must not die with the extra configs
                        others[tag] = {"error": repr(e)}
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_leg()
            out["mae_parity"] = out["cpu_baseline"].pop("mae_parity", None)
            # the 20-step drift of the two training trajectories at the BASELINE batch is CPU-bound (about 7 minutes of oracle steps): measured once per
            # round on the GPU box (`python tools/mae_parity.py trend profiles/drift_trend.json`) and quoted here with the kernel-source digest it ran on
            tpath = os.path.join(ROOT, "profiles", "drift_trend.json")
            if isinstance(out["mae_parity"], dict) and os.path.exists(tpath):
                with open(tpath) as fh:
                    tdoc = json.load(fh)
                out["mae_parity"]["own_trajectory_trend"] = {"source": "profiles/drift_trend.json", "same_kernel_sources": tdoc.get("csrc_digest") == csrc_digest(),
                                                             "protocol": tdoc.get("protocol"), "rows": tdoc.get("rows"),
                                                             "held_by": "tests/test_gpu_mae_parity.py::test_own_trajectory_drift_is_held_to_the_control (steps 1-8, HIP drift <= 2 x control drift)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
