#!/usr/bin/env python
"""bench.py -- Mpixels/s of (lifting DWT encode + entropy-model forward) on MI355X; default = BASELINE.json configs[2].

    python bench.py                          # N=1, configs[2] (the configuration the metric is quoted on)
    python bench.py --gpus 8                 # self-launching: spawns 8 rank processes (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --config {0,1,2,3,4}     # the other BASELINE.json configs (parity-test cases; extra bench lines)

One step = RGB->YCbCr, encode (DWT + subband auto-encoder), entropy-model forward (context CNNs + likelihood + -log2)
for the 3 colour planes, sum of bits; eval mode, fp32, synthetic inputs resident in HBM, random-init weights.
N > 1: the image batch is sharded across ranks (weak scaling, one process per GPU, no data-path collective on the
forward path); the training leg adds ONE flat-bucket gradient all-reduce (RCCL) per step.

A run with --gpus N either sees N ranks or FAILS: when WORLD_SIZE is unset and N > 1 this process (which never touches
the GPU) launches the ranks itself; when the launcher's WORLD_SIZE differs from --gpus it exits non-zero.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32-input MFMA / f32 vector peak
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak (the split-fp16 conv issues 3 fp16 MFMA products per fp32 MAC)
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs[i] -> the workload this repo runs for it (what differs from the BASELINE wording is stated)
CONFIGS = {
    0: dict(netType="CDF97", entropy_layer="factorized", levels=4, batch=1, H=256, W=256, strips=0,
            note="configs[0] is the reference's CPU plumbing case (liftingDWT.json); the product has no CPU path, so the "
                 "same model runs on the GPU here (HBM / launch bound)"),
    1: dict(netType="LiftingBasedNeuralWaveletv4", entropy_layer="factorized", levels=3, batch=16, H=256, W=256, strips=0,
            precision="bf16",
            note="configs[1] names bf16: the matrix kernels run ONE bf16 MFMA product per MAC (LLDWT_PRECISION=bf16; fp32 accumulate, "
                 "fp32 tensors in HBM; tolerance class 1e-2, tests/test_gpu_precision.py); --precision f16x3 gives the fp32-accurate line"),
    2: dict(netType="LiftingBasedNeuralWaveletv4", entropy_layer="conditioned2ZTsepSubbands", levels=4, batch=8, H=512,
            W=512, strips=0, note="the configuration the metric is quoted on"),
    3: dict(netType="LiftingBasedNeuralWaveletv4", entropy_layer="onlyEZWT", levels=4, batch=4, H=1024, W=1024, strips=0,
            note="configs[3]: batch 32 over 8 ranks = 4 images of 1024x1024 per GPU; the training leg is the DP step"),
    4: dict(netType="LiftingBasedNeuralWaveletv4", entropy_layer="onlyEZWT", levels=4, batch=1, H=2160, W=3840, strips=8,
            precision="fp16",
            note="configs[4]: one 3840x2160 frame per GPU cut into 8 independent 480x2160 strips (tiling.split_strips); names fp16: the "
                 "matrix kernels run ONE fp16 MFMA product per MAC (LLDWT_PRECISION=fp16; fp32 accumulate, fp32 tensors in HBM; "
                 "tolerance class 1e-2, tests/test_gpu_precision.py); --precision f16x3 gives the fp32-accurate line"),
}


def source_hashes():
    """git blob hashes (sha1 of "blob <n>\\0" + bytes: what `git hash-object` prints) of the kernel sources a PMC traffic
    figure was measured on: profiles/traffic_current.json carries them, and a figure whose source has changed since is
    dropped from the line instead of being reported stale."""
    import hashlib
    out = {}
    d = os.path.join(REPO, "imagecompressionlearnedliftingandlearnedtreebasedmodels_amd", "csrc")
    for name in ("conv_f16x3.hip", "lifting_f16.hip", "split_f16.h", "lifting_f16.h"):
        try:
            with open(os.path.join(d, name), "rb") as f:
                b = f.read()
            out[name] = hashlib.sha1(b"blob %d\0" % len(b) + b).hexdigest()
        except OSError:
            out[name] = None
    return out


def traffic_is_current(tj, names):
    have, now = tj.get("source_hashes") or {}, source_hashes()
    return all(have.get(n) is not None and have.get(n) == now.get(n) for n in names)


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[i]")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of the config")
    ap.add_argument("--size", type=int, default=0, help="override: square input size")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--entropy", default="", help="override entropy_layer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-kernels", action="store_true", help="skip the secondary roofline_hbm block")
    ap.add_argument("--train-steps", type=int, default=2,
                    help="extra (not part of `value`): time this many full training steps (fwd + hand-written bwd + "
                         "gradient all-reduce + Adam) on the same workload; 0 disables")
    ap.add_argument("--precision", default="", choices=["", "f16x3", "fp16", "bf16"],
                    help="arithmetic of the eval path's matrix kernels; default: what the BASELINE config names (configs[1] bf16, "
                         "configs[4] fp16, otherwise f16x3 = three fp16 products per fp32 MAC, fp32-level accuracy)")
    ap.add_argument("--plc-mode", default="", help="override LLDWT_PLC_MODE (f16x3 | f32) for the dominant conv")
    ap.add_argument("--storage", default="", help="override LLDWT_STORAGE (fp32 | fp16): storage type of the tree-context tensor "
                                                  "(BASELINE configs[4] names fp16; its own tolerance class, never the headline)")
    return ap.parse_args()


# LLDWT_BENCH_REHEARSE=1: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks -- the ranks share the visible
# GPU(s) and exchange over gloo instead of RCCL.  The line says so ("backend": "gloo", "rehearsal": true): its value is the
# throughput of N processes time-slicing one GPU, NOT a scaling measurement.
REHEARSE = os.environ.get("LLDWT_BENCH_REHEARSE", "") not in ("", "0")


def self_launch(a):
    """--gpus N > 1 without a launcher: start N rank processes BEFORE anything touches the GPU, pass their output
    through and exit with their status.  torch.cuda.device_count() does not initialise the GPU."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and not REHEARSE:
        print("bench.py: --gpus %d requested but only %d GPU(s) are visible; refusing to run fewer ranks than asked"
              % (a.gpus, n_dev), file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("self-launch: %s" % " ".join(cmd))
    sys.exit(subprocess.call(cmd, env=env))


def resolve_workload(a):
    c = dict(CONFIGS[a.config])
    explicit = []
    if a.batch:
        c["batch"] = a.batch
        explicit.append("batch")
    if a.size:
        c["H"] = c["W"] = a.size
        c["strips"] = 0
        explicit.append("size")
    if a.height:
        c["H"] = a.height
        explicit.append("height")
    if a.width:
        c["W"] = a.width
        explicit.append("width")
    if a.levels:
        c["levels"] = a.levels
        explicit.append("levels")
    if a.entropy:
        c["entropy_layer"] = a.entropy
        explicit.append("entropy")
    c["precision_run"] = a.precision or c.get("precision", "f16x3")
    if a.precision and a.precision != c.get("precision", "f16x3"):
        explicit.append("precision")
    c["overrides"] = explicit
    return c


def workload_string(c, a):
    tr = "fixed CDF 9/7 (bior4.4, periodization)" if c["netType"] == "CDF97" else \
        "learned %d-level lifting (k=5, 16 ch)" % c["levels"]
    if c["netType"] == "CDF97":
        tr += " %d-level" % c["levels"]
    if c["strips"]:
        shape = "%d frame(s) of 3x%dx%d per GPU as %d strips of 3x%dx%d" % (
            c["batch"], c["H"], c["W"], c["batch"] * c["strips"], c["H"], c["W"] // c["strips"])
    else:
        shape = "%dx3x%dx%d per GPU" % (c["batch"], c["H"], c["W"])
    st = os.environ.get("LLDWT_STORAGE", "fp32")
    prec = c.get("precision_run", "f16x3")
    s = "BASELINE configs[%d]%s: %s + SubbandAutoEncoder + %s, %s, %s, eval" % (
        a.config, " with overrides (%s)" % ",".join(c["overrides"]) if c["overrides"] else "", tr, c["entropy_layer"], shape,
        ("fp32 tensors, %s matrix arithmetic" % prec) if st == "fp32" else "fp32 with fp16 STORAGE of the tree-context tensor")
    return s


def build_model(c, device):
    import torch
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import \
        LiftingBasedDWTNetWrapper
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(dwtlevels=c["levels"], mode="validate", netType=c["netType"], entropy_layer=c["entropy_layer"])
    torch.manual_seed(1337)                       # random-init weights of the architecture (PyTorch default initialisers)
    net = LiftingBasedDWTNetWrapper(cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}    # host copy, handed to the cpu_baseline leg only
    return net.to(device).eval(), sd, cfg


def cpu_baseline(sd, cfg, size, seconds_budget=25.0):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload: 1 x 3 x size x size."""
    import torch
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
            if time.perf_counter() - t0 > seconds_budget / 2 or n >= 12:
                break
        dt = (time.perf_counter() - t0) / n
    return {"value": size * size / dt / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch-CPU restatement of the reference path), same model, one image of the batch "
                      "(1x3x%dx%d), %d timed runs after 1 warm-up, %.1f s of CPU work" % (size, size, n, dt * n)}


def train_leg(a, c, dev, rank, world, x):
    import torch
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.agents.liftingDWT_agent import LiftingBasedDWTAgent
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.utils.config import make_config
    cfg = make_config(dwtlevels=c["levels"], mode="train", batch_size=x.shape[0], patch_size=x.shape[2], seed=1337,
                      netType=c["netType"], entropy_layer=c["entropy_layer"])
    if x.shape[0] * x.shape[2] * x.shape[3] > 8 * 1024 * 1024:
        return {"skipped": "training leg is sized for <= 8 Mpixel per GPU per step (saved activations)"}
    # pre-flight, rank-local (no collective): build the agent and take one step.  Then the ranks agree whether ALL of them got
    # here; a rank that failed must not leave the others waiting in the gradient all-reduce
    err = None
    try:
        torch.manual_seed(1337)
        agent = LiftingBasedDWTAgent(cfg)                 # random-init weights (same on every rank: replicated model)
        agent.model.train()
        torch.manual_seed(parallel.rank_seed(1337, rank))
        torch.cuda.empty_cache()
        agent.train_step(x, allreduce=False)              # warm-up (allocator, packs)
        torch.cuda.synchronize()
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, str(e)[:200])
        log("train leg pre-flight failed on rank %d: %s" % (rank, err))
    if not parallel.all_ranks_ok(err is None, dev):
        return {"error": err or "the pre-flight step failed on another rank"}
    parallel.broadcast_parameters(agent.model)            # the local step used rank-local noise: replicas identical again
    agent.train_step(x)                                   # warm-up of the exchange (first RCCL all-reduce)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(a.train_steps):
        loss, mse, r1, r2 = agent.train_step(x)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    log("train leg done: %.3f s for %d steps" % (dt, a.train_steps))
    # the gradient exchange alone (it runs after backward, not overlapped with it: this is what an N > 1 step pays for it)
    ar_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        parallel.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            agent._bucket.all_reduce_mean()
        torch.cuda.synchronize()
        ar_ms = parallel.max_over_ranks(time.perf_counter() - t1, dev) / 5 * 1e3
    return {"ms_per_step": dt / a.train_steps * 1e3,
            "Mpixels/s": x.shape[0] * x.shape[2] * x.shape[3] * world * a.train_steps / dt / 1e6,
            "steps": a.train_steps, "loss": float(loss.detach()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
            "grad_allreduce": "one flat fp32 bucket of %d floats, backend %s, world %d" % (
                agent._bucket.flat.numel(), parallel.backend_name(), world),
            "allreduce_ms": ar_ms, "allreduce_exposed": "not overlapped with backward (issued after it); 0 for one rank",
            "what": "forward (noise) + hand-written backward + flat-bucket gradient all-reduce (mean over ranks) + Adam, "
                    "same workload, batch sharded over ranks"}


def hbm_kernels(dev, B, S):
    """Secondary block: the HBM-bound kernels of the path at the BASELINE batch (north_star's HBM-roofline target applies
    to these, not to the learned transform): achieved algorithmic GB/s from HIP events on the launch stream."""
    import torch
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.entropy_models import EntropyBottleneck

    def timeit(fn, iters=20):
        """-> (seconds per call of a HIP-graph replay of `iters` back-to-back calls, seconds per call launched eagerly).  The
        calls take 10-30 us on the GPU at the BASELINE batch, the Python + ctypes launch path about as long: the graph replay
        times the kernels, the eager loop what a Python caller sees.  No graph (capture refused): both are the eager time."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / iters * 1e-3
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (3 * iters) * 1e-3, eager
        except Exception as e:
            log("hbm_kernels: graph capture refused (%s: %s), eager timing" % (type(e).__name__, str(e)[:100]))
            torch.cuda.synchronize()
            return eager, eager
    x = torch.rand(B, 3, S, S, device=dev)
    npx = B * 3 * S * S
    y = ops.rgb_to_ycc(x).reshape(1, B, 3, S, S).contiguous()
    out = []

    def add(name, nbytes, tt, what):
        t, eager = tt
        out.append({"kernel": name, "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nbytes / t / 1e9 / HBM_PEAK_GBS, "ms": t * 1e3, "ms_eager": eager * 1e3,
                    "timing": "HIP-graph replay of 20 back-to-back calls (ms); ms_eager = the same calls through Python + ctypes",
                    "algorithmic_bytes": nbytes, "what": what})
    t = timeit(lambda: ops.cdf97_forward(y, 4))
    add("cdf97_forward L=4 (k_cdf97_*)", 2 * npx * 4, t, "%dx3x%dx%d: read the image, write all subbands (8 B/sample)" % (B, S, S))
    ll, yh = ops.cdf97_forward(y, 4)
    t = timeit(lambda: ops.cdf97_inverse(ll, yh))
    add("cdf97_inverse L=4", 2 * npx * 4, t, "same volume, inverse")
    cf = torch.randn(3, B, 3, S // 2, S // 2, device=dev) * 3
    prm = torch.rand(3, B, 6, S // 2, S // 2, device=dev) * 2
    t = timeit(lambda: ops.gauss_rate(cf, prm))
    add("k_gauss_rate", cf.numel() * 16, t, "level-0 subbands: read x, sigma, mu; write bits (16 B/coefficient)")
    eb = torch.stack([EntropyBottleneck(3).packed() for _ in range(3)], 0).to(dev)
    t = timeit(lambda: ops.factorized_rate(cf, eb))
    add("k_factorized_rate", cf.numel() * 12, t, "level-0 subbands: read x; write bits and q (12 B/coefficient)")
    t = timeit(lambda: ops.rgb_to_ycc(x))
    add("k_rgb_to_ycc", 2 * npx * 4, t, "read RGB, write YCbCr planes")
    return out


def main():
    a = parse_args()
    # stdout carries exactly ONE line, the JSON result: anything a library or the agent prints on the way goes to stderr
    json_out = sys.stdout
    sys.stdout = sys.stderr
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                                   # does not return
    env_world = int(os.environ.get("WORLD_SIZE", 1))
    if env_world != a.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a mislabelled line"
              % (a.gpus, env_world), file=sys.stderr)
        sys.exit(2)
    wd = float(os.environ.get("LLDWT_BENCH_WATCHDOG", "0") or 0)
    if wd > 0:                                            # diagnosis: after wd seconds dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True, file=sys.__stderr__)
    if a.plc_mode:
        os.environ["LLDWT_PLC_MODE"] = a.plc_mode
    if a.storage:
        os.environ["LLDWT_STORAGE"] = a.storage

    import torch
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel
    rank, world, local = parallel.env_rank()
    if REHEARSE:                                             # ranks share the visible GPU(s), exchange over gloo
        local %= max(torch.cuda.device_count(), 1)
        os.environ["LOCAL_RANK"] = str(local)                # the agent picks cuda:LOCAL_RANK
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    parallel.init("gloo" if REHEARSE else "nccl", dev)       # backend "nccl" == RCCL over xGMI; no-op for N == 1
    world_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    assert world_seen == a.gpus, (world_seen, a.gpus)

    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops, tiling
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net import rate_planes
    c = resolve_workload(a)
    ops.set_precision(c["precision_run"])                    # eval path only; the training leg always runs f16x3 / fp32
    nprod_mode = 3.0 if c["precision_run"] == "f16x3" else 1.0
    log("building model: %s" % workload_string(c, a))
    net, sd, cfg = build_model(c, dev)
    nets = net.nets()
    gen = torch.Generator(device=dev).manual_seed(parallel.rank_seed(1337, rank))
    x = torch.rand(c["batch"], 3, c["H"], c["W"], device=dev, generator=gen)
    if c["strips"]:
        x = tiling.split_strips(x, c["strips"])              # (batch*strips, 3, H, W/strips): independent images
    Bx, _, Hx, Wx = x.shape
    bit_acc = torch.zeros(1, dtype=torch.float64, device=dev)

    # ---- HIP events around the dominant kernel; the kernels run on torch's current stream = the stream handed to the C-ABI
    dom = {"events": [], "work": 0.0, "on": False}
    lifting = c["netType"] != "CDF97"
    has_plc = c["entropy_layer"] in ("conditioned2ZTsepSubbands", "onlyEZWT")
    import imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.models.LiftingBasedDWT_net as M
    import imagecompressionlearnedliftingandlearnedtreebasedmodels_amd.graphs.layers.lifting_dwt_nets as LN

    dom2 = {"events": [], "work": 0.0}                       # the second kernel family of the step (lifting, when plc leads)

    def timed(fn, work_of, acc=None):
        acc = dom if acc is None else acc

        def wrapper(*args, **kw):
            w = work_of(*args, **kw) if dom["on"] else None
            if not w:
                return fn(*args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            acc["events"].append((e0, e1))
            acc["work"] += w
            return out
        return wrapper
    if has_plc:
        # plc second conv, 243 -> 243 3x3 (LiftingBasedDWT_net.py:271-272,793-795): 2*243*243*9 FLOP per output pixel
        def plc_work(x_, w, bias, K, **kw):
            if K == 3 and w.shape[1] == 243 and w.shape[2] == 243 and not kw.get("transposed"):
                P, B, _, h, wd = x_.shape
                return 2.0 * 243 * 243 * 9 * P * B * h * wd
            return 0.0
        def plc16_work(x_, packed, bias, cout, *args, **kw):
            P, B, cin_, h, wd = x_.shape
            return 2.0 * cin_ * cout * 9 * P * B * h * wd if (cin_ == 243 and cout == 243) else 0.0
        ops.conv2d = timed(ops.conv2d, plc_work)                  # mode f32: the fp32 MFMA engine
        ops.conv3x3_f16x3 = timed(ops.conv3x3_f16x3, plc16_work)  # mode f16x3: split-fp16 on the fp16 matrix cores
        ops.conv3x3_f16in = timed(ops.conv3x3_f16in, plc16_work)  # fp16 storage: two products per MAC

        def fused_work(parent, packed1, packed2, bias2, cmid, cout, **kw):
            # the pair in one launch (default): both convs' algorithmic MACs, the halo recomputation of the first NOT counted
            P, B, _, hp, wp = parent.shape
            return 2.0 * (cmid * cout * 9 + 3 * cmid * 9) * P * B * 4 * hp * wp if (cmid == 243 and cout == 243) else 0.0
        ops.plc_fused = timed(ops.plc_fused, fused_work)
        roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS,
                "kernel": "plc conv 243->243 3x3 (tree context model; the largest share of the step's FLOPs)"}
        if lifting:                                          # the other large kernel family: reported beside the dominant one
            mac2 = {3: 71416, 4: 72266}.get(c["levels"], 72266)

            def lift_work2(x_, *args, **kw):
                P, B, _, H, W = x_.shape
                return 2.0 * mac2 * P * B * H * W
            wrapped2 = timed(ops.lifting_forward, lift_work2, dom2)
            ops.lifting_forward = wrapped2
            LN.ops.lifting_forward = wrapped2
    elif lifting:
        # whole learned-lifting forward (all levels, the k_lift_* launches): SURVEY 8d MAC / plane-pixel x 2
        mac = {3: 71416, 4: 72266}.get(c["levels"], 72266)

        def lift_work(x_, *args, **kw):
            P, B, _, H, W = x_.shape
            return 2.0 * mac * P * B * H * W
        wrapped = timed(ops.lifting_forward, lift_work)
        ops.lifting_forward = wrapped
        LN.ops.lifting_forward = wrapped
        roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS,
                "kernel": "learned lifting forward, %d levels (k_lift_* launches of one lldwt_lifting_forward call)" % c["levels"]}
    else:
        def cdf_work(x_, levels, **kw):
            return 2.0 * 4 * x_.numel()                  # read the image + write all subbands
        wrapped = timed(ops.cdf97_forward, cdf_work)
        ops.cdf97_forward = wrapped
        LN.ops.cdf97_forward = wrapped
        roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "kernel": "fixed CDF 9/7 %d-level forward (k_cdf97_* launches of one call)" % c["levels"]}

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
    tpath = os.path.join(REPO, "profiles", "traffic_current.json")
    if os.path.exists(tpath) and a.config == 2 and not c["overrides"]:
        with open(tpath) as f:
            tj = json.load(f)
        fused_now = ops.plc_mode() == "f16x3" and ops.plc_fuse() and ops.storage_dtype() != "fp16"
        if tj.get("plc_mode", "f32") == ops.plc_mode() and bool(tj.get("fused", False)) == fused_now:
            if traffic_is_current(tj, ("conv_f16x3.hip", "split_f16.h")):
                traffic, traffic_src = tj["traffic_bytes_per_launch"], tj.get("source")
            else:
                traffic_src = "dropped: csrc/conv_f16x3.hip changed since profiles/traffic_current.json was measured"
    dom_ms = sum(e0.elapsed_time(e1) for e0, e1 in dom["events"])
    n_launch = max(len(dom["events"]), 1)
    scale = 1e12 if roof["unit"] == "TFLOP/s" else 1e9
    achieved = dom["work"] / (dom_ms * 1e-3) / scale if dom_ms > 0 else 0.0
    pixels = Bx * Hx * Wx * world_seen * a.steps
    roof.update({"achieved": achieved, "frac": achieved / roof["peak"], "traffic": traffic, "traffic_source": traffic_src,
                 "avg_launch_ms": dom_ms / n_launch, "launches": len(dom["events"]),
                 ("algorithmic_flop_per_launch" if roof["unit"] == "TFLOP/s" else "algorithmic_bytes_per_launch"):
                     dom["work"] / n_launch})
    if lifting and not has_plc:
        from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib as _L
        if _L.load().lldwt_get_lift_mode() == 1:
            # fused split-fp16 lifting step (k_lift_fused_f16): three fp16 MFMA products per fp32 MAC, bounded by the
            # dense fp16 MFMA peak (halo recomputation NOT counted as work)
            roof["arithmetic"] = c["precision_run"]
            roof["kernel"] = ("learned lifting forward, %d levels: k_lift_fused_f16 (one persistent launch per lifting step, "
                              "%s MFMA) inside one lldwt_lifting_forward call" % (c["levels"], c["precision_run"]))
            roof["fp32_equivalent_tflops"] = achieved
            roof["frac_of_fp32_mfma_peak"] = achieved / F32_MFMA_PEAK_TFLOPS
            roof["peak"] = F16_MFMA_PEAK_TFLOPS
            roof["achieved"] = nprod_mode * achieved
            roof["frac"] = nprod_mode * achieved / F16_MFMA_PEAK_TFLOPS
            lpf = 8 * c["levels"]             # kernel launches per forward call (4 row + 4 paired column steps per level)
            roof["launches_per_forward"] = lpf
            roof["launches"] = lpf * n_launch
            roof["avg_launch_ms"] = dom_ms / roof["launches"]
            roof["algorithmic_flop_per_launch"] = nprod_mode * dom["work"] / roof["launches"]
            roof["peak_note"] = ("peak = dense fp16 / bf16 MFMA (2.5 PFLOP/s); achieved = %d MFMA product(s) per MAC x the "
                                 "transform's algorithmic MACs (SURVEY 8d) / HIP-event time of the whole forward call "
                                 "(avg_launch_ms = that time / the call's kernel launches)" % int(nprod_mode))
    if has_plc:
        mode = ops.plc_mode()
        roof["arithmetic"] = mode
        if mode == "f16x3" and ops.plc_fuse() and ops.storage_dtype() != "fp16":
            roof["kernel"] = ("tree-context pair conv 3->243 (on the fly) + conv 243->243 3x3 in one launch, k_conv3_f16x3<2> "
                              "(lldwt_plc_fused; the largest share of the step's FLOPs)")
        nprod = 2.0 if ops.storage_dtype() == "fp16" else nprod_mode
        if ops.storage_dtype() == "fp16":
            roof["storage"] = "fp16 (tree-context tensor stored as fp16; 2 MFMA products per MAC; tolerance class 1e-2)"
        if mode == "f16x3":
            # split-fp16: every fp32 MAC is three fp16 MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate), so the
            # roof that bounds the kernel is the dense fp16 MFMA peak and its algorithmic work is 3 x the conv's FLOPs
            # (padding 243 -> 256 channels NOT counted).  The fp32-equivalent rate is given beside it, against the fp32
            # MFMA peak that bounds the reference arithmetic (mode f32).
            roof["fp32_equivalent_tflops"] = achieved
            roof["frac_of_fp32_mfma_peak"] = achieved / F32_MFMA_PEAK_TFLOPS
            roof["peak"] = F16_MFMA_PEAK_TFLOPS
            roof["achieved"] = nprod * achieved
            roof["frac"] = nprod * achieved / F16_MFMA_PEAK_TFLOPS
            roof["algorithmic_flop_per_launch"] = nprod * dom["work"] / n_launch
            roof["arithmetic"] = c["precision_run"] if ops.storage_dtype() != "fp16" else "f16x2 (fp16 storage)"
            roof["peak_note"] = ("peak = dense fp16 / bf16 MFMA (2.5 PFLOP/s); achieved = %g MFMA product(s) per MAC x the conv's "
                                 "algorithmic MACs / HIP-event time; fp32_equivalent_tflops / frac_of_fp32_mfma_peak compare "
                                 "the same launches with the fp32 MFMA roof (157.3 TF) of the reference arithmetic" % nprod)
        else:
            roof["peak_note"] = "peak = dense fp32-input MFMA (157.3 TFLOP/s); exact fp32 arithmetic (LLDWT_PLC_MODE=f32)"
    # what executes: every tensor in HBM is fp32; the matrix work of the eval path runs as three fp16 MFMA products per fp32 MAC
    # (split-fp16, fp32 accumulate) unless the fp32 kernels were selected
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib as _L0
    f16_lift = lifting and _L0.load().lldwt_get_lift_mode() == 1
    f16_plc = has_plc and ops.plc_mode() == "f16x3"
    if not (f16_lift or f16_plc):
        dtype = "f32"
    elif c["precision_run"] == "f16x3":
        dtype = "f32 storage, f16x3 arithmetic (3 fp16 MFMA products per fp32 MAC, fp32 accumulate)"
    else:
        dtype = "%s (one %s MFMA product per MAC, fp32 accumulate; fp32 storage)" % (c["precision_run"], c["precision_run"])
    if roof.get("fp32_equivalent_tflops") is not None and roof["unit"] == "TFLOP/s":
        roof["frac_algorithmic"] = roof["fp32_equivalent_tflops"] / roof["peak"]     # the conv's own MACs only, same roof
    out = {
        "metric": "Mpixels/sec (lifting DWT + entropy-model fwd) at %dx%d RGB" % (c["H"], c["W"]),
        "value": pixels / dt / 1e6, "unit": "Mpixels/s", "n_gpus": world_seen, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload_string(c, a), "note": c["note"], "per_gpu_images": Bx, "image_hw": [Hx, Wx],
                   "sharding": "batch over ranks, no data-path collective",
                   "backend": parallel.backend_name(), "world_size_seen": world_seen,
                   **({"rehearsal": True} if REHEARSE else {})},
        "roofline": roof,
    }
    if dom2["events"]:
        ms2 = sum(e0.elapsed_time(e1) for e0, e1 in dom2["events"])
        tf2 = dom2["work"] / (ms2 * 1e-3) / 1e12
        from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import _lib as _L2
        f16l = _L2.load().lldwt_get_lift_mode() == 1
        lift_roof = {
            "bound": "mfma", "unit": "TFLOP/s",
            "kernel": "learned lifting forward, %d levels (one lldwt_lifting_forward call per step: %s)" % (
                c["levels"], "persistent k_lift_fused_f16 launches (the L / H column passes of a level share a launch), split-fp16" if f16l else "fp32 MFMA launches"),
            "ms_per_step": ms2 / len(dom2["events"]), "calls": len(dom2["events"]),
            "fp32_equivalent_tflops": tf2, "frac_of_fp32_mfma_peak": tf2 / F32_MFMA_PEAK_TFLOPS,
            "achieved": (nprod_mode if f16l else 1.0) * tf2, "peak": F16_MFMA_PEAK_TFLOPS if f16l else F32_MFMA_PEAK_TFLOPS,
            "frac": (nprod_mode if f16l else 1.0) * tf2 / (F16_MFMA_PEAK_TFLOPS if f16l else F32_MFMA_PEAK_TFLOPS),
            "note": "algorithmic MACs of the transform (SURVEY 8d), halo recomputation not counted; %d MFMA product(s) per MAC (%s)"
                    % (int(nprod_mode), c["precision_run"])}
        lift_roof["frac_algorithmic"] = tf2 / lift_roof["peak"]
        # kernel launches of one forward call: per level 4 row-pass steps + 4 column steps (the L and H column passes share
        # a launch); the fp32 path takes 3 launches for each of the 12 steps of a level.  HIP events bracket the whole call.
        lpf = (8 if f16l else 36) * c["levels"]
        lift_roof["launches_per_forward"] = lpf
        lift_roof["launches"] = lpf * len(dom2["events"])
        lift_roof["avg_launch_ms"] = ms2 / lift_roof["launches"]
        lift_roof["algorithmic_flop_per_launch"] = (nprod_mode if f16l else 1.0) * dom2["work"] / lift_roof["launches"]
        lift_roof["traffic"], lift_roof["traffic_source"] = None, None
        try:
            with open(tpath) as f:
                tl = json.load(f).get("lifting")
            if tl and a.config == 2 and not c["overrides"] and f16l:
                with open(tpath) as f:
                    tj2 = json.load(f)
                if traffic_is_current(tj2, ("lifting_f16.hip", "lifting_f16.h", "split_f16.h")):
                    lift_roof["traffic"] = tl["traffic_bytes_per_forward"] / lpf
                    lift_roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), average over the "
                                                   "launches of one forward call; " + str(tj2.get("csv", "profiles/")))
                else:
                    lift_roof["traffic_source"] = "dropped: csrc/lifting_f16.hip changed since profiles/traffic_current.json was measured"
        except (OSError, ValueError, KeyError):
            pass
        # `roofline` is the kernel family that took more of the timed region; the other one is reported beside it
        if ms2 > dom_ms:
            out["roofline"], out["roofline_second"] = lift_roof, roof
        else:
            out["roofline_second"] = lift_roof
    ops.set_precision("f16x3")
    if a.train_steps > 0:
        try:
            out["train"] = train_leg(a, c, dev, rank, world_seen, x)
        except Exception as e:       # never lose the metric line because of the extra leg
            out["train"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0 and not a.no_hbm_kernels:
        try:
            # the BASELINE batch (25 MB per call: latency-bound) and batch 96 (bandwidth-bound) side by side
            out["roofline_hbm"] = hbm_kernels(dev, 8, 512) + [dict(r, kernel=r["kernel"] + " @batch96") for r in hbm_kernels(dev, 96, 512)]
        except Exception as e:
            out["roofline_hbm"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if rank == 0 and world_seen == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, cfg, min(Hx, Wx, 512))      # one image of the batch (cropped to 512): ~12 s of CPU
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if world_seen > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
