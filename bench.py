#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md §8d "cfg2"): one `train.py` step per GPU =
forward + BCE loss + backward of the full model on a batch of 8 synthetic stereo 16 kHz 10-s clips
(bf16 activations, train mode with dropout), + for N > 1 one RCCL all-reduce of the flat fp32
gradient bucket, + the AdamW/clip update and the refresh of the bf16 compute copies.
Metric: audio-seconds per second, whole job (N GPUs, weak scaling: 8 clips per GPU).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live (HIP events on the launch stream around
every kernel, amt_profile_begin/end) on extra steps after the timed region; `cpu_baseline` times the
oracle's torch-CPU restatement on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIP_SECONDS = 10.0
T_SAMPLES = 160000
B_PER_GPU = 8
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec, MI355X_MICROARCH.md
FLOPS_PER_STEP = 8 * 50.75e9  # 50.75 GFLOP fwd+bwd per 10 s clip


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cpu.max / cfs_quota_us), or None when unlimited / unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, per = f.read().split()[:2]
            return None if q == "max" else max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:         # cgroup v1
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else max(1, int(round(q / per)))
    except (OSError, ValueError):
        return None


def cpu_baseline(seconds_budget: float = 30.0):
    """Oracle (torch-CPU fp32 restatement of the reference; JAX is unavailable offline) on the host cores: fwd + bwd of
    B=8 x 10 s clips (the same per-step batch as one GPU).  BASELINE.md section 2: N = ALL host cores the process may run
    on (N stated), two warm-ups, median of five.  The figure at 16 threads - the box's CPU share per GPU, what rounds 1-3
    reported - is measured beside it (`at_16_threads`).  "May run on" = min(sched_getaffinity, the cgroup's CPU quota): the
    GPU box shows 256 logical CPUs to a process whose cgroup grants 16 cores' worth of time, and 256 torch threads on that
    share did not finish ONE step in four minutes (gpurun_out/r4h, round 4)."""
    from oracle import np_model, torch_model
    import audio_to_midi_amd as amt
    conf = np_model.MODEL_CONFIG
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    Bc = 8
    params = np_model.init_params(conf, seed=1234, gamma=1e-6)
    audio, labels = amt.synth.make_batch_fast(Bc, T_SAMPLES, seed=1)
    p = torch_model.to_torch(params, torch.float32, requires_grad=True)
    a = torch.from_numpy(audio)
    y = torch.from_numpy(labels)

    def step():
        for v in p.values():
            v.grad = None
        logits, _ = torch_model.forward(p, a, conf)
        torch_model.loss_fn(logits, y).backward()

    def measure(threads, warmups, iters, budget):
        torch.set_num_threads(threads)
        log(f"cpu_baseline: torch-CPU oracle on {threads} threads ...")
        t_start = time.perf_counter()
        warm = []
        for _ in range(warmups):
            t0 = time.perf_counter(); step(); warm.append(time.perf_counter() - t0)
            log(f"cpu_baseline: warm-up step {warm[-1]:.2f} s")
        times = []
        for _ in range(iters):
            if times and time.perf_counter() - t_start + times[-1] > budget:
                break                                   # bounded sample: the default bench run finishes within minutes
            t0 = time.perf_counter(); step(); times.append(time.perf_counter() - t0)
            log(f"cpu_baseline: step {times[-1]:.2f} s")
        med = float(np.median(times)) if times else warm[-1]
        return med, len(times), int(torch.get_num_threads())

    quota = cgroup_cpu_quota()
    usable = min(avail, quota) if quota else avail
    if quota is None and avail > 64:
        usable = 16      # no readable quota on a many-core box: the documented CPU share per GPU (an oversubscribed run would not end)
    cap = int(os.environ.get("AMT_CPU_BASELINE_THREADS", "0"))      # (0: every usable core)
    n_all = max(1, min(usable, cap) if cap > 0 else usable)
    med, n, thr = measure(n_all, 2, 5, seconds_budget * 0.6)
    out = {"value": round(Bc * CLIP_SECONDS / med, 2), "unit": "audio-sec/s", "cores": thr, "kind": "port",
           "sample": f"fwd+bwd of B={Bc} x 10 s clips, fp32, full model, median of {n} after 2 warm-ups ({med:.2f} s/step); "
                     f"torch-CPU restatement of the reference (JAX unavailable offline); torch threads = {thr} = every core the "
                     f"process may run on (sched_getaffinity: {avail}, cgroup CPU quota: {quota}, os.cpu_count(): {os.cpu_count()})"}
    if thr != 16 and avail >= 16:
        med16, n16, thr16 = measure(16, 1, 3, seconds_budget * 0.4)
        out["at_16_threads"] = {"value": round(Bc * CLIP_SECONDS / med16, 2), "cores": thr16,
                                "sample": f"median of {n16} after 1 warm-up ({med16:.2f} s/step): the box's CPU share per GPU, the figure of rounds 1-3"}
    return out


def named_rooflines(breakdown, profile_steps):
    """north_star: achieved fraction of the attention-MFMA and the conv-frontend-HBM rooflines.
    attention: QK^T + PV flops credited by the launchers (4 B H S^2 64 forward, 2x backward) over the time of
    the four attention kernels.  conv frontend: SURVEY.md 8(d) algorithmic bytes (23.94 MB bf16 per 10-s clip
    forward, backward counted 2x) over the time of the CNN kernel classes (the Block-backward launches also
    carry weight-gradient workgroups, and the 6 downsample LayerNorm/GEMM launches are not separable from
    the transformer's: the first overstates, the second understates the time by a few percent)."""
    att = [b for b in breakdown if b["kernel"].startswith("attn_")]
    cnn = [b for b in breakdown if b["kernel"].split("+")[0] in ("stem_fwd", "stem_bwd", "block_small_fwd", "block_small_bwd",
                                                                  "block_fwd_fused", "block_bwd_fused", "dwblock_bwd", "dwln_fwd")]
    out = {}
    t_att = sum(b["avg_us"] * b["launches_per_step"] for b in att) * 1e-6
    if t_att > 0:
        fl = sum((b["tflops"] or 0) * 1e12 * b["avg_us"] * 1e-6 * b["launches_per_step"] for b in att if b["kernel"].count("+") == 0)
        out["attention_mfma"] = {"achieved_tflops": round(fl / t_att / 1e12, 2), "peak_tflops": PEAK_BF16_TFLOPS,
                                 "frac": round(fl / t_att / 1e12 / PEAK_BF16_TFLOPS, 5), "us_per_step": round(t_att * 1e6, 1)}
    t_cnn = sum(b["avg_us"] * b["launches_per_step"] for b in cnn) * 1e-6
    if t_cnn > 0:
        by = B_PER_GPU * 23.94e6 * 3.0
        out["conv_frontend_hbm"] = {"achieved_gbs": round(by / t_cnn / 1e9, 1), "peak_gbs": PEAK_HBM_GBS,
                                    "frac": round(by / t_cnn / 1e9 / PEAK_HBM_GBS, 5), "us_per_step": round(t_cnn * 1e6, 1),
                                    "algorithmic_bytes_per_step": by}
    return out


def build_id():
    """sha-256 (first 16 hex digits) over the kernel sources the library was built from: profile summaries under
    profiles/ carry it, and a summary taken from another build is not quoted (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "audio-to-midi_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def survey_8d_bytes(kernel: str, B: int, es: int = 2):
    """SURVEY.md 8(d) "Algorithmic bytes, conv frontend", per LAUNCH of a conv-frontend class at 10-s clips: a Stem /
    Block reads its input once and writes its output once (C * L = 128 000 elements per 10-s clip at every stage, the
    2C hidden expansion never leaves the chip), the backward is counted 2x the forward.  None for classes 8(d) prices
    by flops (transformer GEMMs, attention)."""
    base = kernel.split("+")[0]
    per_clip = 128000
    if base in ("block_fwd_fused", "block_small_fwd", "block_run_fwd", "dwln_fwd"):
        return 2.0 * B * per_clip * es
    if base in ("block_bwd_fused", "block_small_bwd_full", "block_small_bwd", "dwblock_bwd"):
        return 4.0 * B * per_clip * es
    if base == "stem_fwd":
        return B * (320000 + per_clip) * es
    if base == "stem_bwd":
        return 2.0 * B * (320000 + per_clip) * es
    return None


def dominant_roofline(top, profile_steps):
    """`roofline` of the kernel class with the largest share of the step.

    achieved = ALGORITHMIC bytes (conv frontend: SURVEY.md 8(d)'s numerator, NOT the bytes this design moves) or
    algorithmic flops (2 N M K per GEMM, 4 B H S^2 64 per attention forward) per launch / the class's average launch
    duration, measured here with HIP events on the launch stream around every EAGER launch (`timing`); the duration of
    the same class under hipGraph replay is in the committed rocprofv3 summary named by `rocprof_stats` and is quoted
    as `avg_launch_us_graph` when that summary was taken from this build.  `design_bytes_per_launch` is what the launch
    sites account (chain reads + writes + the operands of the weight-gradient riders): the design's own traffic model.
    `traffic` = HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 corrections of
    MI355X_MICROARCH.md), read from profiles/pmc_traffic_by_class.json ONLY if that file was taken from this build."""
    name = top["name"]
    avg_s = 1e-3 * top["total_ms"] / top["launches"]
    flops_l = top["flops"] / top["launches"]
    design_l = top["bytes"] / top["launches"]
    alg_bytes = survey_8d_bytes(name, B_PER_GPU)
    bid = build_id()
    traffic, graph_us, stats_file = None, None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_by_class.json")) as f:
            pj = json.load(f)
        if pj.get("_build_id") == bid:
            traffic = pj.get(name, {}).get("hbm_bytes_per_launch")
            graph_us = pj.get(name, {}).get("avg_launch_us_graph")
            stats_file = pj.get("_rocprof_stats")
        else:
            log(f"roofline: profiles/pmc_traffic_by_class.json is from build {pj.get('_build_id')}, this is {bid}: traffic = null")
    except OSError:
        pass
    # matrix-pipe duty cycle of the class from a committed rocprofv3 SQ pass of THIS build (profiles/make_mfma.py)
    mfma_busy = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_mfma_by_class.json")) as f:
            mj = json.load(f)
        if mj.get("_build_id") == bid:
            mfma_busy = mj.get(name, {}).get("mfma_busy_of_chip")
    except OSError:
        pass
    # The step is timed under hipGraph replay, where a class runs shorter than its eager launches (no host-paced gaps between
    # dependent launches): `frac` is priced with the graph-replay duration whenever the committed rocprofv3 summary was taken
    # from THIS build, the live eager HIP-event figure stays beside it (`avg_launch_us_eager`, `frac_eager`);
    # without a matching summary the eager figure is the only one there is.
    eager_s = avg_s
    if graph_us is not None:
        avg_s = graph_us * 1e-6
    common = {"kernel": name, "traffic": traffic, "mfma_busy": mfma_busy, "avg_launch_us": round(avg_s * 1e6, 2),
              "timing": ("rocprofv3 --kernel-trace --stats under hipGraph replay, this build (" + str(stats_file) + ")") if graph_us is not None
                        else "HIP events around eager launches (no rocprofv3 summary of this build under profiles/)",
              "avg_launch_us_eager": round(eager_s * 1e6, 2), "avg_launch_us_graph": graph_us, "rocprof_stats": stats_file, "build_id": bid,
              "flops_per_launch": flops_l, "design_bytes_per_launch": design_l,
              "frac_mfma": round(flops_l / avg_s / 1e12 / PEAK_BF16_TFLOPS, 5)}
    ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    if alg_bytes is not None:       # conv frontend: HBM roof, SURVEY 8(d) numerator
        gbs = alg_bytes / avg_s / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 5), "frac_eager": round(alg_bytes / eager_s / 1e9 / PEAK_HBM_GBS, 5),
                "algorithmic_bytes_per_launch": alg_bytes,
                "numerator": "SURVEY.md 8(d): 2 x (read input + write output) of a Block backward, bf16", **common}
    tf = flops_l / avg_s / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_BF16_TFLOPS, 5), "frac_eager": round(flops_l / eager_s / 1e12 / PEAK_BF16_TFLOPS, 5),
            "algorithmic_bytes_per_launch": None,
            "numerator": "SURVEY.md 8(d): 2 N M K per GEMM / 4 B H S^2 64 per attention forward", "ridge": round(ridge, 1), **common}


def dp_report_fields(dp):
    """What the bench line says about the gradient exchange, from the DataParallelStep that ran it: the form actually used -
    `dp.overlap` / `dp.signal` AFTER prepare(), whose trial step may have fallen back to the one-collective form - and why."""
    fused = bool(getattr(dp, "in_place", False))
    form = (("two collectives, the first overlapped with the CNN backward"
             + (" behind a device-side signal of ONE launch sequence" if dp.signal else " (step cut in two host calls)")) if dp.overlap else "one bucket")
    form += " (SUM, in place on the context's accumulator; 1 / world in the optimizer)" if fused else " (ReduceOp.AVG on a separate bucket)"
    if dp.wire == "bf16":
        form += "; bf16 on the wire, fp32 accumulation on arrival"
    return {"grad_allreduce": form, "grad_allreduce_fallback": dp.fallback}


def launch_ranks(args):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as child processes through
    torch.distributed.run (the reference gets its device mesh from one process, train.py:238-244, 834-842; here it is
    one process per GPU over RCCL), relay their output, exit with their worst return code.  Nothing in this process has
    initialised the GPU (no HIP call, no torch.cuda.is_available()): it only spawns and waits."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("bench: no WORLD_SIZE in the environment, starting the ranks:", " ".join(cmd))
    env = dict(os.environ)
    # the pool's host driver supports dmabuf IPC only: without this RCCL / device-tensor sharing across processes fails with
    # `hipIpcGetMemHandle: invalid argument` (the image exports it already; kept for an environment that was built by hand)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = subprocess.call(cmd, env=env)
    sys.exit(rc)


def main_dry_run(args):
    """World-size-N rendezvous + one all-reduce on CPU tensors (gloo): what a driver sees of the launch path, without a GPU."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(4)
    dist.all_reduce(t)
    ok = bool((t == world).all())
    if rank == 0:
        print(json.dumps({"metric": "launcher dry run", "n_gpus": args.gpus, "ranks_seen": dist.get_world_size(),
                          "backend": dist.get_backend(), "allreduce_ok": ok}), flush=True)
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def main_infer(args):
    """BASELINE configs[3]: 32 clips x 60 s, overlap 0.5 s -> 14 windows of 5 s per clip =
    448 windows (448, 2, 80000), forward only (bf16), then stitch_probs per clip on the host (C++)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import audio_to_midi_amd as amt
    from audio_to_midi_amd import infer as amt_infer, lib as amtlib
    model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
    clips, seconds, overlap = 32, 60.0, 0.5
    rng = np.random.default_rng(0)
    full = rng.standard_normal((2, int(seconds * 16000))).astype(np.float32)
    windows, _ = amt_infer.slice_full_audio(amt_infer.normalize_audio(full), overlap)
    n_win = windows.shape[0]
    batch = torch.from_numpy(np.tile(windows, (clips, 1, 1))).to(dev, model.audio_dtype)     # resident in HBM
    B = batch.shape[0]

    def step():
        return model.predict(None, batch)

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits, probs = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = 1e3 * elapsed / args.steps
    # host part: stitch every clip (device -> host copy + C++ stitch), timed separately
    t1 = time.perf_counter()
    p = probs.float().cpu().numpy().reshape(clips, n_win, -1, probs.shape[-1])
    stitched = [amt_infer.stitch_output_probs(p[i], 5.0 / p.shape[2], overlap) for i in range(clips)]
    host_ms = 1e3 * (time.perf_counter() - t1)
    breakdown = []
    if args.profile_steps > 0:
        amtlib.profile_begin()
        for _ in range(args.profile_steps):
            step()
        ents = amtlib.profile_end()
        tot = sum(e["total_ms"] for e in ents) or 1.0
        ents.sort(key=lambda e: -e["total_ms"])
        for e in ents:
            avg_us = 1e3 * e["total_ms"] / e["launches"]
            breakdown.append({"kernel": e["name"], "launches_per_step": e["launches"] // args.profile_steps,
                              "avg_us": round(avg_us, 2), "share": round(e["total_ms"] / tot, 4),
                              "tflops": round(e["flops"] / e["total_ms"] / 1e9, 2) if e["flops"] else None,
                              "gbs": round(e["bytes"] / e["total_ms"] / 1e6, 1)})
        log("kernel class breakdown (HIP events, eager), inference:")
        for b in breakdown:
            log(f"  {b['kernel']:32s} x{b['launches_per_step']:4d}  avg {b['avg_us']:9.2f} us  share {100*b['share']:5.1f}%"
                f"  {b['tflops']} TFLOP/s  {b['gbs']} GB/s")
    cnn = [b for b in breakdown if b["kernel"] in ("stem_fwd", "block_small_fwd", "block_fwd_fused", "dwln_fwd")]
    t_cnn = sum(b["avg_us"] * b["launches_per_step"] for b in cnn) * 1e-6
    conv = None
    if t_cnn > 0:
        by = B * 11.97e6        # 11.97 MB bf16 per 5-s clip forward
        conv = {"achieved_gbs": round(by / t_cnn / 1e9, 1), "peak_gbs": PEAK_HBM_GBS, "frac": round(by / t_cnn / 1e9 / PEAK_HBM_GBS, 5),
                "us_per_step": round(t_cnn * 1e6, 1), "algorithmic_bytes_per_step": by}
    out = {"metric": "audio-sec/s (inference fwd, 5 s windows)", "value": round(clips * seconds / (elapsed / args.steps), 1),
           "unit": "audio-sec/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "streaming inference (BASELINE configs[3]): 32 clips x 60 s -> 448 windows x 5 s, "
                                  "forward (bf16) resident in HBM; stitch_probs on the host timed separately",
                      "windows": int(B), "frames_per_window": int(probs.shape[1]), "stitched_frames_per_clip": int(stitched[0].shape[0])},
           "model_tflops": round(B * 7.946e9 / (elapsed / args.steps) / 1e12, 2),
           "host_copy_and_stitch_ms": round(host_ms, 1),
           "rooflines": {"conv_frontend_hbm": conv}, "breakdown": breakdown[:12]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resident-inputs", action="store_true",
                    help="replay the same HBM-resident batch every step (rounds 1-3) instead of uploading a fresh batch per step "
                         "from pinned host memory on a copy stream (train.py:280 uploads every minibatch)")
    ap.add_argument("--labels-fp32", action="store_true",
                    help="labels as float32 (default: float16, the dtype the reference's loader hands over, grain_loader.py:88)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no-fused-optimizer", action="store_true",
                    help="N = 1: write the unscaled gradients to the bucket and run the three-pass optimizer (the N > 1 form)")
    ap.add_argument("--wire", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: gradient wire format - fp32 all-reduce (default) or bfloat16 shards with fp32 accumulation on "
                         "arrival (the reference's message size; audio_to_midi_amd.train.DataParallelStep)")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: one all-reduce after the whole backward pass instead of the two-bucket overlap")
    ap.add_argument("--force-overlap", action="store_true", help="use the two-phase step even at N = 1 (testing)")
    ap.add_argument("--no-signal", action="store_true",
                    help="overlap as two host calls / two graphs (amt_forward_backward_phase) instead of ONE launch sequence that "
                         "publishes the first half's completion to the host (amt_first_half_wait)")
    ap.add_argument("--workload", choices=["train", "infer"], default="train",
                    help="train = BASELINE configs[1] (default, the headline metric); infer = configs[3]: streaming "
                         "inference, 32 x 60 s clips -> 448 windows of 5 s, forward + stitch (reported, not the headline)")
    ap.add_argument("--clips-per-gpu", type=int, default=8,
                    help="10 s clips per GPU and step (default 8 = configs[1]; 32 = configs[4]'s 256 over 8 GPUs)")
    ap.add_argument("--attn-fp8", action="store_true",
                    help="configs[4]: global-attention forward on fp8 MFMA (amt_set_attn_fp8); reported, not the headline")
    ap.add_argument("--dry-run-launcher", action="store_true",
                    help="plumbing test of the N > 1 launch path: gloo on CPU tensors, one all-reduce, no GPU work")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)           # before anything touches the GPU: the ranks are CHILD processes
    if args.dry_run_launcher:
        return main_dry_run(args)
    if args.workload == "infer":
        return main_infer(args)
    global B_PER_GPU, FLOPS_PER_STEP
    B_PER_GPU = args.clips_per_gpu
    FLOPS_PER_STEP = B_PER_GPU * 50.75e9

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # AMT_BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses cuda:0 and the ranks exchange through gloo (RCCL refuses two ranks on
    # one device).  A FUNCTIONAL rehearsal of the N > 1 program on a one-GPU box - launcher, rank set-up, signalled overlap,
    # in-place collectives, max-over-ranks timing, the JSON line - not a scaling number: the line says so ("rehearsal").
    rehearse = world > 1 and os.environ.get("AMT_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # a collective that never completes fails the run after two minutes (non-zero exit, no silent hang); what CAN be
            # recovered in-process - a missing first-half signal, a collective that raises - is DataParallelStep's fall-back
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=120))

    import audio_to_midi_amd as amt
    from audio_to_midi_amd import lib as amtlib

    model = amt.OutputSequenceGenerator(amt.model_config, key=1234, dtype="bfloat16", device=dev)
    if args.attn_fp8:
        model.set_attn_fp8(True)
    audio_np, labels_np = amt.synth.make_batch_fast(B_PER_GPU, T_SAMPLES, seed=100 + rank)
    audio = torch.from_numpy(audio_np).to(dev, model.audio_dtype)    # inputs resident in HBM
    # labels in the reference loader's own dtype (`midi_events.astype(np.float16)`, grain_loader.py:88); --labels-fp32: float32
    label_dtype = torch.float32 if args.labels_fp32 else torch.float16
    labels = torch.from_numpy(labels_np).to(dev, label_dtype)
    # One flat fp32 bucket (audio_to_midi_amd.train.GradBucket): [finite flag of the whole step, 3 pad | gradients
    # (amt layout) | loss, finite flag of the first half].  With N > 1 it is averaged over the ranks (ReduceOp.AVG:
    # mean over the global batch, train.py:62; a flag stays 1.0 iff every rank's is 1).  The step itself - two-phase
    # overlap, graph capture, collectives, device-gated AdamW - is audio_to_midi_amd.train.DataParallelStep, the same
    # code the gloo world-2 test drives (tests/test_dp_cpu.py).
    from audio_to_midi_amd import train as amt_train
    # The fused protocol (include/amt.h) at every N: the bucket IS the context's gradient accumulator
    # (GradBucket.of_model), the backward pass leaves its raw fp32 sums there, the collectives (SUM) run on it in place and
    # one optimizer launch reads sums * 1 / world, tests finiteness, clips, commits and zeroes it.  The N > 1 rank runs the
    # N = 1 program plus the collectives.  --no-fused-optimizer: the round-3 form (unscale pass into a separate bucket,
    # ReduceOp.AVG, three-pass optimizer).
    fused = not args.no_optimizer and not args.no_fused_optimizer
    bucket = amt_train.GradBucket.of_model(model) if fused else amt_train.GradBucket(model.n_params, dev)
    # N > 1: the step runs in two halves (include/amt.h: amt_forward_backward_phase).  The gradients of the
    # decoder + transformer + final CNN norm (37 of the 46 MB) are complete after the first half and their
    # all-reduce (with the loss and the first half's finite flag behind them) runs on a side stream while the CNN
    # stages are still in their backward pass; the CNN range, preceded by the whole step's finite flag, follows:
    # two collectives per step.  Per-GPU work is identical to the one-call step.
    overlap = (world > 1 and not args.no_overlap) or args.force_overlap
    if overlap and dist is None and os.environ.get("AMT_BENCH_NO_PG"):
        pass      # diagnostic: the two-phase per-rank program with NO collective calls at all (what the split itself costs)
    elif overlap and dist is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    # (AdamW reads the gradient sums where the backward pass - and the collectives - left them: grads / scale, the
    # finite check and the clip are all still computed, train.py:319-325)
    backend = amt_train.HipTrainBackend(model, audio, labels, bucket, scale=1.0, use_optimizer=not args.no_optimizer, fused=fused)
    dp = amt_train.DataParallelStep(backend, bucket, backend.split, dist=dist, overlap=overlap, graph=not args.no_graph,
                                    lr_base=1e-4, signal=False if args.no_signal else None, wire=args.wire)
    if not args.resident_inputs:
        backend.add_input_set()  # a second input set: the step is captured once per set, uploads land where a step reads them
    dp.prepare(seed=12345)       # every allocation (optimizer state, workspace plan) happens before graph capture
    graph = dp._graphs
    # A fresh batch per step (train.py:280: the reference uploads every minibatch): four distinct pinned host batches in
    # rotation, batch k + 1 on its way over PCIe on a copy stream while step k computes (audio_to_midi_amd.train.InputFeeder)
    feeder, host_pool = None, []
    if not args.resident_inputs:
        feeder = amt_train.InputFeeder(backend=backend)
        host_pool = [amt_train.InputFeeder.pin(audio_np, labels_np, model.audio_dtype, label_dtype)]      # one pinned block per batch: one upload
        for j in range(1, 4):
            an, yn = amt.synth.make_batch_fast(B_PER_GPU, T_SAMPLES, seed=1000 * j + rank)
            host_pool.append(amt_train.InputFeeder.pin(an, yn, model.audio_dtype, label_dtype))
        feeder.prefetch(*host_pool[0])
    step_i = [0]

    def one_step():
        if feeder is not None:
            feeder.take()
            dp.step()
            step_i[0] += 1
            feeder.prefetch(*host_pool[step_i[0] % len(host_pool)])
        else:
            dp.step()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    host_issue_ms = 1e3 * (time.perf_counter() - t0) / args.steps     # how long the HOST needs to issue a step (diagnostic)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B_PER_GPU * CLIP_SECONDS / (elapsed / args.steps)
    loss_val, grads_valid = dp.result()
    # a non-finite step would have been skipped on the device (gated AdamW); the timed run must not contain one
    if not (math.isfinite(loss_val) and grads_valid):
        raise SystemExit(f"bench: non-finite step in the timed run (loss {loss_val}, grads_valid {grads_valid})")
    if model.team_status() != 0:
        raise SystemExit("bench: an XCD-team launch timed out (amt_team_status != 0)")
    if dp.optimizer_timeouts and not rehearse:      # (ranks sharing one device hold each other's compute units: expected there)
        raise SystemExit("bench: the one-launch optimizer timed out at its grid barrier in the timed run (something else holds compute units)")
    # the same loop on the batch that is already in HBM: what the uploads cost on top (0 when they are hidden)
    resident_ms = None
    if feeder is not None:
        feeder.take()
        for _ in range(3):
            dp.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(5, args.steps // 2)):
            dp.step()
        torch.cuda.synchronize()
        resident_ms = 1e3 * (time.perf_counter() - t1) / max(5, args.steps // 2)
        dp.result()

    # ---- roofline leg: per-kernel HIP-event timing on the launch stream (eager, extra steps)
    roofline, breakdown = None, []
    if rank == 0 and args.profile_steps > 0:
        amtlib.profile_begin()
        for i in range(args.profile_steps):
            backend.full(1000 + i)
            backend.apply(dp.step_no + 1 + i, 1e-4)
        ents = amtlib.profile_end()
        tot = sum(e["total_ms"] for e in ents) or 1.0
        ents.sort(key=lambda e: -e["total_ms"])
        for e in ents:
            avg_us = 1e3 * e["total_ms"] / e["launches"]
            breakdown.append({"kernel": e["name"], "launches_per_step": e["launches"] // args.profile_steps,
                              "avg_us": round(avg_us, 2), "share": round(e["total_ms"] / tot, 4),
                              "tflops": round(e["flops"] / e["total_ms"] / 1e9, 2) if e["flops"] else None,
                              "gbs": round(e["bytes"] / e["total_ms"] / 1e6, 1)})
        log("kernel class breakdown (HIP events, eager):")
        for b in breakdown:
            log(f"  {b['kernel']:32s} x{b['launches_per_step']:4d}  avg {b['avg_us']:9.2f} us  share {100*b['share']:5.1f}%"
                f"  {b['tflops']} TFLOP/s  {b['gbs']} GB/s")
        log(f"  sum of kernel time per step: {tot / args.profile_steps:.3f} ms (timed step {ms_per_step:.3f} ms)")
        top = ents[0]
        roofline = dominant_roofline(top, args.profile_steps)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "audio-sec/s (fwd+bwd, 10 s clips)", "value": round(value, 1), "unit": "audio-sec/s",
            "n_gpus": world, "ranks_seen": dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1,
            "comm_backend": (dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else " (host-staged)")) if (dist is not None and dist.is_initialized()) else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (fp8 e4m3 QK^T / PV in the global-attention forward)" if args.attn_fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"train.py step: fwd + BCE + bwd of the full model, {B_PER_GPU} clips x 10 s @ 16 kHz stereo per "
                                   "GPU, train mode (dropout 0.1)" + (", global-attention forward on fp8 MFMA" if args.attn_fp8 else "") + (", RCCL all-reduce of the fp32 grad bucket" if world > 1 else "")
                                   + ("" if args.no_optimizer else ", AdamW + global-norm clip + bf16 weight refresh"),
                       "global_batch": world * B_PER_GPU, "clip_seconds": CLIP_SECONDS, "frames": T_SAMPLES // 320,
                       "parallelism": f"dp{world}", "hipgraph": graph is not None,
                       **(dp_report_fields(dp) if world > 1 or overlap else {"grad_allreduce": None}),
                       "optimizer": None if args.no_optimizer else ("fused: norm pass + commit pass over the context's gradient accumulator" if fused else "grads_finalize + adamw_compute + adamw_apply over the bucket"),
                       "inputs": "audio float16, labels " + ("float32" if args.labels_fp32 else "float16 (grain_loader.py:88)") + ", one contiguous block per batch",
                       "weights": "random init (Equinox distributions), 11606230 params"},
            "model_tflops": round(world * FLOPS_PER_STEP / (elapsed / args.steps) / 1e12, 2),
            "host_issue_ms_per_step": round(host_issue_ms, 4),
            **({"rehearsal": f"{world} ranks share ONE GPU and exchange through gloo: a functional rehearsal of the N > 1 program, "
                             "not a scaling measurement", "optimizer_timeouts": int(dp.optimizer_timeouts)} if rehearse else {}),
            "inputs": ("resident in HBM (same batch every step)" if feeder is None else
                       f"fresh batch per step: {feeder.bytes_per_batch / 1e6:.2f} MB from pinned host memory over PCIe on a copy stream, "
                       "straight into the input set the next step reads (two input sets, the step captured on each; train.py:280)"),
            "h2d_overlapped": None if feeder is None else True,
            "ms_per_step_resident_inputs": None if resident_ms is None else round(resident_ms, 4),
            "loss": round(loss_val, 3), "grads_valid": bool(grads_valid), "loss_finite": True,
            "roofline": roofline,
            "rooflines": named_rooflines(breakdown, args.profile_steps) if breakdown else None,
            "cpu_baseline": cpu,
            "breakdown": breakdown[:12],
        }
        print(json.dumps(out), flush=True)
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
