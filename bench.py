#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W            (N = 1, or N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1 under a launcher)

A "step" is one YOLOv1-448 train step (S=7, B=2, C=20, per-GPU batch 64, bf16 conv on MFMA): zero_grad,
forward (24 Conv2D blocks + 4 pools + fc head), fused composite loss, backward, RCCL gradient exchange
(N > 1), clip_grad_norm_(1.0) + SGD(momentum 0.9, wd 5e-4).  Inputs are resident in HBM before the timed
region.  `value` = global images/s.  The JSON line also carries:
  roofline      the conv kernel set (MFMA-bound), achieved = algorithmic FLOPs / HIP-event time of the conv launches
                inside the timed region; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference step, oracle/model_ref.py) timed on the
                host cores, rank 0, N = 1 only, bounded sample
  nms           BASELINE config 4 (decode + IoU + class-wise NMS, 1024 images x 845 boxes): images/s and HBM roofline
"""
import argparse
import gc
import json
import os
import sys
import time

# The gradient exchange owns no stream (dp.GradComm): a rank under RCCL drives three (compute, weight gradients, RCCL's own),
# within the box's GPU_MAX_HW_QUEUES=4 hardware queues, so the driver's multi-GPU runs leave that setting alone.  Only the
# one-GPU REHEARSAL of the N > 1 path (YFS_BENCH_ONE_GPU=1: gloo on CUDA tensors, which takes a fresh pool stream for every
# collective) raises a smaller setting to 16, as in round 2; YFS_KEEP_HW_QUEUES=1 leaves it alone.
# Read at HIP initialisation, hence set before torch is imported.
if (os.environ.get("YFS_BENCH_ONE_GPU") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1
        and os.environ.get("YFS_KEEP_HW_QUEUES") != "1"):
    try:
        _q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        _q = 4
    if _q < 16:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# ---- `python bench.py --gpus N` without a launcher: this process becomes a parent that never touches the GPU ----------
def rank_environments(n, port, base=None):
    """Environment of each of the n rank processes (what torch.distributed.run would export), one rank per GPU."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
        envs.append(e)
    return envs


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, child=None, watchdog_s=None):
    """Start n fresh rank processes of this script (never an exec of this one, never a fork after HIP initialisation:
    the parent imports neither torch nor the library), relay rank 0's stdout, forward every rank's stderr, and return a
    non-zero code if any rank fails or the watchdog fires (the remaining ranks are then killed by PID)."""
    import subprocess
    import threading
    child = child or [sys.executable, "-u", os.path.abspath(__file__)]
    watchdog_s = watchdog_s or int(os.environ.get("YFS_BENCH_WATCHDOG_S", "600")) + 60
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r, env in enumerate(rank_environments(n, port)):
        procs.append(subprocess.Popen(child + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    deadline = time.monotonic() + watchdog_s
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr, flush=True)
        if pending and (rc != 0 or time.monotonic() > deadline):
            if rc == 0:
                rc = 124
                print(f"bench.py: ranks {sorted(pending)} still running after {watchdog_s} s; killing them", file=sys.stderr,
                      flush=True)
            for r in pending:
                procs[r].kill()
            for r in pending:
                procs[r].wait()
            break
        time.sleep(0.2)
    reader.join(timeout=5)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return rc


def _wants_parent(argv):
    """--gpus N > 1 given to a process no launcher has prepared (no WORLD_SIZE in the environment)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n if (n > 1 and "WORLD_SIZE" not in os.environ) else 0


if __name__ == "__main__" and _wants_parent(sys.argv[1:]):
    sys.exit(launch_ranks(_wants_parent(sys.argv[1:]), sys.argv[1:]))

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same table
BATCH = 64
LOSS_BYTES_PER_IMG = 17640    # read pred + read target + write grad, 3 x 1470 x 4
ASSIGN_BYTES_PER_IMG = 32750  # gts 640 + decoded preds 13,520 + masks 1,690 + targets 16,900
NMS_BYTES_PER_IMG = 108160    # raw head 84,500 + boxes/score/class 20,280 + kept indices 3,380
# HBM bytes per launch (roofline.traffic of the HBM-bound sets) and MFMA utilisation of the conv kernels come from the PMC
# counters.  bench.py cannot run rocprofv3 on itself, so they are COMMITTED measurements at the sub-benchmarks' own sizes:
# the "#json" line that tools/pmc_traffic.py / tools/pmc_summarize.py write at the end of profiles/rNN/
# pmc_traffic_hbm_kernels.txt and conv_pmc_summary.txt (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, FETCH_SIZE
# doubled per the gfx950 note of MI355X_MICROARCH.md; SQ_VALU_MFMA_BUSY_CYCLES over the active cycles of all SIMDs).  The
# newest round's file is read at run time and the record names it with its git blob hash, so a stale file shows.
def committed_pmc(name):
    """-> (values from the file's #json line, {"file", "git_blob"}) of the newest profiles/rNN/<name>, or (None, None)."""
    import glob
    import hashlib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", name)), reverse=True):
        data = open(path, "rb").read()
        for line in reversed(data.decode("utf-8", "replace").splitlines()):
            if line.startswith("#json "):
                blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()      # = git hash-object
                return json.loads(line[6:]), {"file": os.path.relpath(path, ROOT), "git_blob": blob}
    return None, None


def conv_flops_table(model, batch, hw=448):
    """Algorithmic FLOPs per conv launch (2*MACs), keyed like the timers in conv_ops."""
    from yolo_from_scratch_amd import model as ym
    from yolo_from_scratch_amd import conv_ops as co
    fwd = dgrad = wgrad = 0.0
    h = hw
    first = True
    for m in model.feat:
        if isinstance(m, ym.HipMaxPool2d):
            h //= 2
            continue
        c = m[0]
        oh, _ = co.out_hw(h, h, c.kernel_size, c.stride)
        macs = batch * oh * oh * c.out_channels * c.in_channels * c.kernel_size ** 2
        fwd += 2 * macs
        wgrad += 2 * macs
        if not first:
            dgrad += 2 * macs
        first = False
        h = oh
    return {"conv_fwd": fwd, "conv_dgrad": dgrad, "conv_wgrad": wgrad}


class ConvTimers:
    """HIP-event brackets around every conv launch (they run on torch's current stream)."""

    def __init__(self):
        self.events = {"conv_fwd": [], "conv_dgrad": [], "conv_wgrad": []}

    def install(self):
        from yolo_from_scratch_amd import conv_ops as co
        self._orig = {}
        for key, name in (("conv_fwd", "conv_fwd"), ("conv_dgrad", "conv_dgrad"), ("conv_wgrad", "conv_wgrad"),
                          ("conv_fwd", "stem_fwd"), ("conv_wgrad", "stem_wgrad"), ("conv_fwd", "conv_fwd_pool"),
                          ("conv_fwd", "stem_fwd_pool"), ("conv_wgrad", "stem_wgrad_pool")):
            fn = getattr(co, name)
            self._orig[name] = fn

            def wrapped(*a, _fn=fn, _key=key, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _fn(*a, **kw)
                e1.record()
                self.events[_key].append((e0, e1))
                return out
            setattr(co, name, wrapped)

    def remove(self):
        from yolo_from_scratch_amd import conv_ops as co
        for name, fn in self._orig.items():
            setattr(co, name, fn)

    def totals_ms(self):
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.events.items()}


def cpu_baseline_train(sample_batch=4, steps=2):
    """The oracle's train step (torch-CPU fp32 restatement of the reference step + C composite loss) on the host:
    once on the threads this job may use (OMP_NUM_THREADS on the GPU box = its CPU share) and once on 8 threads, the
    count SURVEY.md 8(d) anchors its own measurement on."""
    import oracle
    from oracle import model_ref
    from yolo_from_scratch_amd import synth
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    params = model_ref.init_params(seed=0, head_out=1470)
    x = torch.rand(sample_batch, 3, 448, 448, generator=torch.Generator().manual_seed(0))
    tgt = synth.v1_targets(sample_batch, seed=1)

    class _OracleLoss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, out):
            lo, g = oracle.yolov1_loss(out.detach().numpy().reshape(-1, 7, 7, 30), tgt)
            ctx.g = torch.from_numpy(g.reshape(out.shape))
            return torch.tensor(float(lo[0]))

        @staticmethod
        def backward(ctx, go):
            return ctx.g * go

    loss_fn = lambda out, _t: _OracleLoss.apply(out)   # noqa: E731

    def timed(threads):
        torch.set_num_threads(threads)
        bufs = {}
        model_ref.train_step(params, bufs, x, None, loss_fn)          # warm-up (allocations, oneDNN primitive cache)
        t0 = time.perf_counter()
        for _ in range(steps):
            model_ref.train_step(params, bufs, x, None, loss_fn)
        return sample_batch * steps / (time.perf_counter() - t0)
    v = timed(cores)
    v8 = timed(min(8, cores)) if cores != 8 else v
    return {"value": round(v, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "value_8_threads": round(v8, 3), "host_cpu_count": os.cpu_count(), "host_cpu": _cpu_model(),
            "sample": f"{steps} train steps at batch {sample_batch}, 448x448, fp32 torch-CPU restatement "
                      f"(oracle/model_ref.py + oracle yolov1_loss), 1 warm-up step untimed, on {cores} threads "
                      f"(the job's CPU share; os.cpu_count() = {os.cpu_count()}) and again on 8 threads"}


def cpu_baseline_loss(sample=2048):
    import oracle
    from yolo_from_scratch_amd import synth
    t = synth.v1_targets(sample, seed=1)
    p = np.random.default_rng(3).uniform(0, 1, t.shape).astype(np.float32)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.0:
        oracle.yolov1_loss(p, t)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample * reps / dt, 1), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes over {sample} images (C oracle orc_yolov1_loss, scalar) on {_cpu_model()}"}


def cpu_baseline_assign(sample=1024, S=13, G=32):
    import oracle
    from yolo_from_scratch_amd import synth
    gt = synth.gt_boxes(sample, G=G, seed=5)
    pred = np.random.default_rng(6).uniform(0.05, 0.9, (sample, S * S * 5, 4)).astype(np.float32)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.0:
        oracle.assign_targets_v2(gt, synth.VOC_ANCHORS, S, pred, 0.6)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample * reps / dt, 1), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes over {sample} images (C oracle orc_assign_targets_v2, scalar) on {_cpu_model()}"}


def nms_bench(device, iters=20):
    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import synth
    N = 1024
    raw = torch.from_numpy(synth.raw_head_v2(N, 13, 5, 20, seed=1234)).to(device)
    anchors = torch.from_numpy(synth.VOC_ANCHORS).to(device)
    op = y.DecodeNms(N, 13, 5, 20, device)
    for _ in range(3):
        op(raw, anchors)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op(raw, anchors)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    out = {"metric": "NMS images/sec (decode + IoU + class-wise NMS, 13x13x5 anchors, C=20, batch 1024)",
           "value": round(N / (ms * 1e-3), 1), "unit": "images/s", "ms_per_batch": round(ms, 4), "dtype": "f32",
           "roofline": _hbm_record(N * NMS_BYTES_PER_IMG, ms, "nms")}
    return out


def _time_ms(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _hbm_record(nbytes, ms, traffic_key=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    vals, src = committed_pmc("pmc_traffic_hbm_kernels.txt") if traffic_key else (None, None)
    return {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 5), "traffic": float(vals[traffic_key]) if vals and vals.get(traffic_key) else None,
            "traffic_source": src}


def loss_bench(device, N=65536):
    """Kernel set 2 at the batch where its HBM fraction means something (N=64 moves 1.1 MB and is
    latency-bound; N=65,536 moves 1.16 GB).  Calls the C ABI directly: forward + gradient in one launch."""
    from yolo_from_scratch_amd import _lib as L
    g = torch.Generator(device="cpu").manual_seed(3)
    pred = torch.rand(4096, 7, 7, 30, generator=g).repeat(N // 4096, 1, 1, 1).to(device)
    tgt = torch.from_numpy(np.tile(__import__("yolo_from_scratch_amd.synth", fromlist=["x"]).v1_targets(4096, seed=1),
                                   (N // 4096, 1, 1, 1))).to(device)
    ws_bytes = L.lib().yfs_yolov1_loss_workspace_bytes(L.i64(N), L.c_int(7))
    ws = torch.empty(ws_bytes // 4 + 1, device=device, dtype=torch.float32)
    out = torch.empty(5, device=device, dtype=torch.float32)
    grad = torch.empty_like(pred)

    def run():
        L.check(L.lib().yfs_yolov1_loss_fwd_bwd_f32(L.ptr(pred), L.ptr(tgt), L.i64(N), L.c_int(7), L.c_int(2), L.c_int(20),
                                                    L.c_float(5.0), L.c_float(0.5), L.ptr(out), L.ptr(grad), L.ptr(ws),
                                                    L.c_size(ws_bytes), L.stream()), "yfs_yolov1_loss_fwd_bwd_f32")
    ms = _time_ms(run, 20)
    return {"metric": "fused YOLOv1 loss + gradient images/sec (S=7 B=2 C=20, batch %d)" % N,
            "value": round(N / (ms * 1e-3), 1), "unit": "images/s", "ms_per_batch": round(ms, 4), "dtype": "f32",
            "roofline": _hbm_record(N * LOSS_BYTES_PER_IMG, ms, "loss" if N == 65536 else None)}


def assign_bench(device, N=8192, S=13, G=32):
    """Kernel set 4: YOLOv2 target assignment + ignore mask, S=13, A=5, G=32 padded gts per image."""
    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import synth
    gt = torch.from_numpy(synth.gt_boxes(N, G=G, seed=5)).to(device)
    anchors = torch.from_numpy(synth.VOC_ANCHORS).to(device)
    pred = torch.rand(N, S * S * 5, 4, generator=torch.Generator().manual_seed(6)).to(device)
    ms = _time_ms(lambda: y.assign_targets_v2(gt, anchors, S, pred, 0.6), 20)
    return {"metric": "YOLOv2 target assignment images/sec (S=13, A=5, G=32, batch %d)" % N,
            "value": round(N / (ms * 1e-3), 1), "unit": "images/s", "ms_per_batch": round(ms, 4), "dtype": "f32",
            "roofline": _hbm_record(N * ASSIGN_BYTES_PER_IMG, ms, "assign" if (N, S, G) == (8192, 13, 32) else None)}


def mask_boxes_bench(device, N=64, H=375, W=500, cpu_sample=6):
    """The reference's mask -> component boxes -> largest box step on VOC-sized synthetic masks:
    a few filled shapes plus speckle.  Integer work bound by dependent L2 round trips and atomics (DESIGN.md §4 set 5), so
    it carries no HBM fraction; the oracle's flood fill on the host is timed beside it on a few masks."""
    from yolo_from_scratch_amd import data as yd
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[:H, :W]
    masks = np.zeros((N, H, W), np.uint8)
    for i in range(N):
        m = rng.uniform(size=(H, W)) < 0.002
        for _ in range(6):
            cy, cx, ry, rx = rng.integers(0, H), rng.integers(0, W), rng.integers(3, H // 4), rng.integers(3, W // 4)
            m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        masks[i] = m
    dev = torch.from_numpy(masks).to(device)
    out = {}
    ms = _time_ms(lambda: out.__setitem__("r", yd.masks_to_boxes(dev, max_boxes=1024)), 20)
    rec = {"metric": "mask -> boxes -> largest box, masks/sec (%d x %d x 1 u8, batch %d)" % (H, W, N),
           "value": round(N / (ms * 1e-3), 1), "unit": "masks/s", "ms_per_batch": round(ms, 4), "dtype": "u8/int32",
           "components_per_mask": round(float(out["r"][1].float().mean()), 1)}
    if cpu_sample:
        from oracle import data_ref
        t0 = time.perf_counter()
        for i in range(cpu_sample):
            data_ref.target_box(masks[i])
        dt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": round(cpu_sample / dt, 2), "unit": "masks/s", "cores": 1, "kind": "port",
                               "sample": f"{cpu_sample} of the same masks (oracle/data_ref.py flood fill, pure Python) on {_cpu_model()}"}
    return rec


def measured_ceilings(device):
    """What this box sustains, next to the vendor peaks the roofline fractions are priced against."""
    import ctypes
    from yolo_from_scratch_amd import _lib as L
    a = torch.empty(1 << 30, device=device, dtype=torch.uint8)
    b = torch.empty_like(a)
    ms = _time_ms(lambda: b.copy_(a), 10)
    hbm = 2 * a.numel() / (ms * 1e-3) / 1e9
    del a, b
    flop = ctypes.c_double(0.0)
    scratch = torch.zeros(4, device=device)
    blocks, iters = 256 * 8, 4096

    def run():
        L.check(L.lib().yfs_probe_mfma_bf16(L.c_int(iters), L.c_int(blocks), ctypes.byref(flop), L.ptr(scratch), L.stream()),
                "yfs_probe_mfma_bf16")
    ms = _time_ms(run, 5, warm=2)
    ones = flop.value / (ms * 1e-3) / 1e12
    rnd = {}
    for shape, label in ((0, "16x16x32"), (1, "32x32x16")):
        def run_rnd(shape=shape):
            L.check(L.lib().yfs_probe_mfma_bf16_ex(L.c_int(shape), L.c_int(1), L.c_int(20000), L.c_int(1024), ctypes.byref(flop),
                                                   L.ptr(scratch), L.stream()), "yfs_probe_mfma_bf16_ex")
        ms = _time_ms(run_rnd, 15, warm=15)      # ~0.2 s of load before the timed launches: the clock has settled
        rnd[label] = round(flop.value / (ms * 1e-3) / 1e12, 1)
    # the vendor's plain GEMM on its friendliest shape, as a yardstick for the conv kernels: torch.mm -> hipBLASLt
    gemm = {}
    ga = (torch.rand(8192, 8192, device=device) - 0.5).to(torch.bfloat16)
    gb = (torch.rand(8192, 8192, device=device) - 0.5).to(torch.bfloat16)
    gout = torch.empty(8192, 8192, device=device, dtype=torch.bfloat16)
    for label, (pa, pb) in (("random", (ga, gb)), ("zeros", (torch.zeros_like(ga), torch.zeros_like(gb)))):
        ms = _time_ms(lambda pa=pa, pb=pb: torch.mm(pa, pb, out=gout), 100, warm=150)
        gemm[label] = round(2.0 * 8192 ** 3 / (ms * 1e-3) / 1e12, 1)
    del ga, gb, gout
    return {"hbm_copy_GBps": round(hbm, 1), "mfma_bf16_TFLOPs": round(ones, 1),
            "mfma_bf16_random_operands_TFLOPs": rnd, "hipblaslt_gemm_8192_bf16_TFLOPs": gemm,
            "note": "device-to-device copy of 1 GiB (read + write bytes); register-only v_mfma_f32_16x16x32_bf16 loop, "
                    "8 workgroups x 4 waves per CU, at the clock the chip holds under that load; the same loop with "
                    "operands that change from one MFMA to the next (pseudo-random bf16 in +-[0.5, 2)): the matrix pipe "
                    "alone, fed from registers, is power-limited on switching data (DESIGN.md set 1, "
                    "profiles/r02/clock_under_load.txt); torch.mm (hipBLASLt) on 8192^3 bf16 with random and with zero operands"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline_nms(sample=64):
    import oracle
    from yolo_from_scratch_amd import synth
    raw = synth.raw_head_v2(sample, 13, 5, 20, seed=1234)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        b, s, c = oracle.decode_grid(raw, synth.VOC_ANCHORS)
        oracle.nms(b, s, c, 0.05, 0.5)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(sample * reps / dt, 1), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes over {sample} images of config 4 (C oracle, scalar) on {_cpu_model()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--compress-bf16", action="store_true", help="bf16 gradient buckets for the all-reduce")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: libraries that print there (RCCL writes a version banner at
    # communicator set-up) are sent to stderr for the whole run, and the record goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # (without WORLD_SIZE in the environment `--gpus N` starts its own ranks: launch_ranks above)
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} set by the launcher")
    # Rehearsal of the N > 1 code path on a one-GPU box (YFS_BENCH_ONE_GPU=1): every rank on device 0, gloo instead of
    # RCCL (two RCCL ranks cannot share a device).  Never used by the driver; its numbers mean nothing.
    one_gpu = os.environ.get("YFS_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    # YFS_BENCH_FORCE_DP=1 (rehearsal, never used by the driver): take the N > 1 code path with a group of ONE rank on real
    # RCCL: process-group set-up as below, the full gradient exchange (dp.GradComm(force=True)), barriers, the MAX
    # all-reduce of the time.  The exchange of one rank changes no value; its cost is the machinery itself.
    force_dp = os.environ.get("YFS_BENCH_FORCE_DP") == "1" and world == 1
    multi = world > 1 or force_dp
    if force_dp:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if multi:
        # A stalled collective must end the run with a diagnosis, not a silent driver timeout: every rank dumps the
        # stacks of all its threads to stderr and exits non-zero if the whole run takes longer than this.
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("YFS_BENCH_WATCHDOG_S", "600")), exit=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import yolo_from_scratch_amd as y
    from yolo_from_scratch_amd import dp, synth
    from yolo_from_scratch_amd import model as ym
    from yolo_from_scratch_amd import train as yt

    torch.manual_seed(0)
    model = ym.YoloRegressorModel(head_out=7 * 7 * 30).to(device)
    model.train()                                    # Dropout(0.2) active, as in the reference's train loop
    comm = dp.GradComm(world, compress_bf16=args.compress_bf16, force=force_dp) if multi else None
    opt = yt.FlatSGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, max_norm=1.0, comm=comm)
    # as train_one_epoch runs its loop: fc.1's fused weight-gradient + SGD kernel (HBM-bound) is left running on the
    # weight-gradient stream when opt.step() returns and overlaps the next step's first convolutions; the next fc.1
    # forward waits for it, and the device-wide synchronize that ends the timed region includes the last one.
    opt.overlap_tail = os.environ.get("YFS_OVERLAP_TAIL", "1") != "0"
    dp.broadcast_parameters(opt)
    crit = y.YoloV1Loss(7, 2, 20)
    g = torch.Generator().manual_seed(rank)
    x = torch.rand(args.batch, 3, 448, 448, generator=g).to(device).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    target = torch.from_numpy(synth.v1_targets(args.batch, seed=1 + rank)).to(device)

    def note(msg):
        if one_gpu:
            print(f"[rehearsal rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("model, optimiser and data ready")

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = crit(out, target)
        loss.backward()
        opt.step()
        return loss

    # A full (generation-2) Python garbage collection walks every object torch created at import: ~75 ms, which
    # the GPU spends idle.  It fired once every ~30 steps and cost 2-4 ms/step in short runs; the timed loop
    # allocates no reference cycles, so collect once and keep the collector off while timing.
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
        note("warm-up step done")
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
        if one_gpu:
            torch.cuda.synchronize()
            note("timed step done")
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    final_loss = float(loss.detach())
    # Roofline pass: the same step a few more times with HIP-event brackets around every conv launch.  The timed
    # loop above runs the weight gradients on a second stream, where brackets of concurrent kernels overlap and
    # each kernel is stretched by its neighbour; for per-kernel durations the second stream is switched off here.
    rsteps = max(3, min(10, args.steps))
    two = ym._TWO_STREAMS
    ym._TWO_STREAMS = False
    opt.overlap_tail = False
    step()
    timers = ConvTimers()
    timers.install()
    torch.cuda.synchronize()
    for _ in range(rsteps):
        step()
    torch.cuda.synchronize()
    timers.remove()
    ym._TWO_STREAMS = two
    gc.enable()
    if multi:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    if rank == 0:
        flops = conv_flops_table(model, args.batch)
        tot = timers.totals_ms()
        conv_ms = sum(v[0] for v in tot.values()) / rsteps
        conv_flops = sum(flops.values())
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12
        kernels = {k: {"launches_per_step": tot[k][1] // rsteps, "ms_per_step": round(tot[k][0] / rsteps, 4),
                       "tflops": round(flops[k] / (tot[k][0] / rsteps * 1e-3) / 1e12, 1)} for k in tot}
        rec = {
            "metric": "train images/sec YOLOv1-448", "value": round(args.batch * world * args.steps / dt, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "YOLOv1 448x448 S=7 B=2 C=20 train step, per-GPU batch %d, bf16 conv MFMA + fused "
                                   "composite loss + clip + SGD (BASELINE configs[1]; configs[2] when n_gpus=8)" % args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "grad_allreduce": (("bf16" if args.compress_bf16 else "fp32") +
                                          (" buckets; fc.1 as all-gathered gradient factors" if opt._defer else ""))
                       if multi else None},
            "final_loss": round(final_loss, 5),
            "roofline": {"bound": "mfma", "kernel": "conv implicit-GEMM set (igemm_kernel / igemm8_kernel / stem_pool_kernel fwd+dgrad, wgrad_kernel / wgrad8_kernel)",
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         # memory-side bytes of the conv kernels per step (committed PMC passes, like mfma_busy_pmc below)
                         "traffic": (lambda v, src: float(v["conv"]) if v and v.get("conv") else None)(
                             *committed_pmc("pmc_traffic_conv_step.txt")),
                         "traffic_source": committed_pmc("pmc_traffic_conv_step.txt")[1],
                         "algorithmic_tflop_per_step": round(conv_flops / 1e12, 4), "conv_ms_per_step": round(conv_ms, 3),
                         "mfma_busy_pmc": dict(zip(("kernels", "source"), committed_pmc("conv_pmc_summary.txt"))),
                         "measured_over": "%d extra steps with the weight-gradient stream off (non-overlapping "
                                          "brackets); the headline value runs with it on" % rsteps,
                         "kernels": kernels},
        }
        if world == 1 and not args.no_nms:
            rec["nms"] = nms_bench(device)
            if not args.no_cpu_baseline:
                rec["nms"]["cpu_baseline"] = cpu_baseline_nms()
            rec["loss"] = loss_bench(device)
            rec["assign"] = assign_bench(device)
            rec["mask_boxes"] = mask_boxes_bench(device, cpu_sample=0 if args.no_cpu_baseline else 6)
            if not args.no_cpu_baseline:
                rec["loss"]["cpu_baseline"] = cpu_baseline_loss()
                rec["assign"]["cpu_baseline"] = cpu_baseline_assign()
            rec["roofline"]["measured_ceilings"] = measured_ceilings(device)
        if world == 1 and not args.no_cpu_baseline:
            del model, opt
            torch.cuda.empty_cache()
            rec["cpu_baseline"] = cpu_baseline_train()
        if force_dp:
            rec["rehearsal"] = "YFS_BENCH_FORCE_DP=1: one-rank RCCL group with the full gradient exchange forced on"
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        import faulthandler
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
