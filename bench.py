#!/usr/bin/env python
"""bench.py -- Mpixels/s of (learned lifting DWT encode + entropy-model forward) at 512x512 RGB on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): learned 4-level lifting (k=5, 16 ch)
+ SubbandAutoEncoder + conditioned2ZTsepSubbands context model, 8 x 3 x 512 x 512 per GPU, fp32, eval mode, synthetic
inputs resident in HBM, deterministic by-name weights.  One step = RGB->YCbCr, encode (lifting + subband AE), entropy
model forward (all context CNNs + Gaussian rate) for 3 planes, sum of bits.  N > 1: the image batch is sharded across
ranks (weak scaling, no data-path collective on the forward path).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA / f32 vector peak
HBM_PEAK_GBS = 8000.0


def build_model(levels, device):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import \
        LiftingBasedDWTNetWrapper
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(dwtlevels=levels, mode="validate")
    torch.manual_seed(1337)                       # random-init weights of the architecture (PyTorch default initialisers)
    net = LiftingBasedDWTNetWrapper(cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}    # host copy, handed to the cpu_baseline leg only
    return net.to(device).eval(), sd, cfg


def cpu_baseline(sd, cfg, size, seconds_budget=25.0):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload: 1 x 3 x size x size."""
    from oracle import model as omodel
    from oracle.entropy import ENTROPY_LAYERS
    cores = min(16, len(os.sched_getaffinity(0)))     # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    x = torch.rand(1, 3, size, size, generator=torch.Generator().manual_seed(1337))

    def run():
        y = omodel.rgb2ycbcr(x) - omodel._YSHIFT
        tot = 0.0
        for c in range(3):
            s = omodel.sub(sd, "model%d." % c)
            oxe, oxo = omodel.encode(y[:, c:c + 1], omodel.sub(s, "autoencoder."), dict(cfg))
            si_xe, si_xo, _, _ = ENTROPY_LAYERS[cfg["entropy_layer"]](oxe, oxo, omodel.sub(s, "entropymodel."), dict(cfg), False)
            tot += float(si_xe.sum()) + sum(float(t.sum()) for t in si_xo)
        return tot
    with torch.no_grad():
        run()
        t0 = time.perf_counter()
        n = 0
        while True:
            run()
            n += 1
            if time.perf_counter() - t0 > seconds_budget / 2 or n >= 5:
                break
        dt = (time.perf_counter() - t0) / n
    return {"value": size * size / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch-CPU restatement of the reference path), 1x3x%dx%d, %d timed runs after 1 warm-up" % (size, size, n)}


def train_leg(a, dev, rank, world, x):
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents.liftingDWT_agent import LiftingBasedDWTAgent
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(dwtlevels=a.levels, mode="train", batch_size=a.batch, patch_size=a.size, seed=1337)
    if x.shape[2] != x.shape[3] and x.numel() > 3 * 1024 * 1024 * 8:
        return {"skipped": "training leg is sized for the square BASELINE crops"}
    torch.manual_seed(1337)
    agent = LiftingBasedDWTAgent(cfg)                     # random-init weights (same on every rank: replicated model)
    agent.model.train()
    torch.manual_seed(parallel.rank_seed(1337, rank))
    torch.cuda.empty_cache()
    agent.train_step(x)                                   # warm-up (allocator, packs)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(a.train_steps):
        loss, mse, r1, r2 = agent.train_step(x)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    log("train leg done: %.3f s for %d steps" % (dt, a.train_steps))
    return {"ms_per_step": dt / a.train_steps * 1e3,
            "Mpixels/s": x.shape[0] * x.shape[2] * x.shape[3] * world * a.train_steps / dt / 1e6,
            "steps": a.train_steps, "loss": float(loss.detach()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
            "what": "forward (noise) + hand-written backward + flat-bucket gradient all-reduce (mean over ranks) + Adam, "
                    "same workload, batch sharded over ranks"}


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--height", type=int, default=0, help="non-square input (e.g. 2160 x 3840 for BASELINE configs[4]); default: --size")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=2,
                    help="extra (not part of `value`): time this many full training steps (fwd + hand-written bwd + "
                         "gradient all-reduce + Adam) on the same workload; 0 disables")
    a = ap.parse_args()
    H_, W_ = (a.height or a.size), (a.width or a.size)

    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel
    rank, world, local = parallel.env_rank()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    parallel.init("nccl", dev)                               # backend "nccl" == RCCL over xGMI; no-op for N == 1

    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import rate_planes
    log("building model")
    net, sd, cfg = build_model(a.levels, dev)
    log("model ready")
    nets = net.nets()
    x = torch.rand(a.batch, 3, H_, W_, device=dev, generator=torch.Generator(device=dev).manual_seed(parallel.rank_seed(1337, rank)))
    bit_acc = torch.zeros(1, dtype=torch.float64, device=dev)

    # HIP events around the dominant kernel (plc second conv, 243 -> 243 3x3, LiftingBasedDWT_net.py:271-272): the
    # kernels run on torch's current stream, which is the stream handed to the C-ABI.
    dom = {"events": [], "flops": 0.0, "on": False}
    conv2d_orig = ops.conv2d

    def conv2d_timed(x_, w, bias, K, **kw):
        is_dom = dom["on"] and K == 3 and w.shape[1] == 243 and w.shape[2] == 243 and not kw.get("transposed")
        if not is_dom:
            return conv2d_orig(x_, w, bias, K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = conv2d_orig(x_, w, bias, K, **kw)
        e1.record()
        P, B, _, h, wd = x_.shape
        dom["events"].append((e0, e1))
        dom["flops"] += 2.0 * 243 * 243 * 9 * P * B * h * wd
        return out
    ops.conv2d = conv2d_timed
    import imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net as M
    M.ops.conv2d = conv2d_timed

    def step():
        with torch.no_grad():
            y = ops.rgb_to_ycc(x)
            si_xe, si_xo = rate_planes(nets, y, False)
            ops.sum_into(si_xe, bit_acc)
            for t in si_xo:
                ops.sum_into(t, bit_acc)

    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        log("warmup %d done" % i)
    parallel.barrier()
    torch.cuda.synchronize()
    dom["on"] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom["on"] = False
    log("timed region done: %.3f s for %d steps" % (dt, a.steps))
    dt = parallel.max_over_ranks(dt, dev)                    # the slowest rank defines the step time

    traffic, traffic_src = None, None
    tpath = os.path.join(REPO, "profiles", "r01_traffic.json")
    if os.path.exists(tpath) and a.batch == 8 and a.size == 512 and a.levels == 4:
        with open(tpath) as f:
            tj = json.load(f)
        traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same command)"
    dom_ms = sum(e0.elapsed_time(e1) for e0, e1 in dom["events"])
    n_launch = max(len(dom["events"]), 1)
    achieved = dom["flops"] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    pixels = a.batch * H_ * W_ * world * a.steps
    out = {
        "metric": "Mpixels/sec (lifting DWT + entropy-model fwd) at 512x512 RGB",
        "value": pixels / dt / 1e6, "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: learned %d-level lifting (k=5,16ch) + SubbandAutoEncoder + "
                               "conditioned2ZTsepSubbands, %dx3x%dx%d per GPU, eval" % (a.levels, a.batch, H_, W_),
                   "per_gpu_batch": a.batch, "sharding": "batch over ranks, no data-path collective"},
        "roofline": {"bound": "mfma", "kernel": "plc conv 243->243 3x3 (tree context, 61% of the step's FLOPs)",
                     "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": dom_ms / n_launch, "launches": len(dom["events"]),
                     "algorithmic_flop_per_launch": dom["flops"] / n_launch},
    }
    # ---- extra, reported beside the metric: the training step of the same workload (north_star: fwd/bwd path, batch
    # sharded over ranks, ONE flat-bucket gradient all-reduce over RCCL per step)
    if a.train_steps > 0:
        try:
            out["train"] = train_leg(a, dev, rank, world, x)
        except Exception as e:       # never lose the metric line because of the extra leg
            out["train"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, cfg, min(a.size, 256))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
