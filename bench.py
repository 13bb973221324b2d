#!/usr/bin/env python3
"""bench.py -- images/sec of the PyraPose hot path (fwd + losses + bwd + clipnorm-Adam) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  Prints ONE JSON line on rank 0.

Workload at N=1 = BASELINE.json configs[1]: LineMOD 13-class training, batch 8, 640x480, ResNet-50 PFPN
+ heads, synthetic data resident in HBM before the timed region (inputs: uint8 U[0,255] minus caffe
BGR means; targets: the HIP target-assignment kernel on seeded synthetic annotations).  N > 1 is weak
scaling: batch 8 per GPU, gradient all-reduce over RCCL (pyrapose_amd/parallel.py).

`roofline`: the dominant kernel family is the implicit-GEMM MFMA convolution; every launch of it is
bracketed with HIP events inside the timed region (torch.cuda.Event on the ctx stream, which IS the
stream the kernels are launched on) and achieved = sum(algorithmic 2*MAC flops) / sum(durations).
Peak = 157.3 TFLOP/s dense f32 MFMA (MI355X_MICROARCH.md).  `cpu_baseline` (N=1, rank 0): the oracle's
PyTorch-CPU float32 restatement of the same train step on a 1-image sample ("port": the Keras/TF
reference cannot run here or on the GPU box -- tensorflow/keras are not installed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# numpy / torch are imported by the worker (main_worker); the N > 1 parent (spawn_ranks) imports neither and makes no
# GPU call: it only starts one child process per rank and relays rank 0's line.

PEAK_F32_MFMA_TFLOPS = 157.3   # dense f32 MFMA, MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (never the 2:1-sparsity figure)
ALGO_GFLOP_PER_IMAGE = {(13, 480, 640): 683.2}  # BASELINE.md §3 (fwd 234.2 + bwd 449.1)


DTYPE_NOTES = {
    "f32": "exact f32 MFMA (v_mfma_f32_32x32x2_f32) everywhere",
    "bf16x3": "every conv product = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on bf16 MFMA with f32 accumulation (2^-16 relative); master weights, "
              "losses and Adam in float32; head outputs within 1e-3 of the float64 oracle (tests/test_gpu_model.py, test_gpu_parity.py)",
    "f16c8": "every conv product = x_hi*w_hi on f16 MFMA + (x_hi8*w_lo8 + x_lo8*w_hi8) * 2^-12 on block-scaled e5m2 MFMA, f32 accumulation "
             "(2^-15 relative); gradients travel times a power of two; master weights, losses and Adam in float32",
    "mixed": "NOT a reduced-precision number: float32-equivalent emulated products with f32 accumulation in both halves.  Backbone (conv1, "
             "res2-res5): x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on bf16 MFMA (2^-16 relative).  FPN + heads: x_hi*w_hi on f16 MFMA + both "
             "cross terms on block-scaled e5m2 MFMA (2^-15 relative; gradients travel times a power of two).  Master weights, losses "
             "and Adam in float32; head outputs and weight gradients within 1e-3 of the float64 oracle at full size "
             "(tests/test_gpu_parity.py, test_gpu_model.py)",
}

EVENT_EVERY = 4  # per-launch HIP events on steps 0, 4, 8, ... of the timed region
P16_AUDIT = {}           # Engine.p16_stats() of the most recent run() (rank 0)
P16_AUDIT_HEADLINE = {}  # ... of the headline run


def measured_peak(mode, achieved):  # achieved: MFMA-unit TFLOP/s (executed flops x units per product)
    """the box-measured denominator next to the spec peak (profiles/r01_peaks.json, tools/ubench/peaks.hip)"""
    path = os.path.join(ROOT, "profiles", "r01_peaks.json")
    if mode == "f32" or not os.path.exists(path):
        return None
    with open(path) as f:
        pk = json.load(f)
    m = pk["bf16_mfma_tflops_with_lds_reads"]
    return {"bf16_mfma_tflops_sustained_random_data": pk["bf16_mfma_tflops_registers"], "with_lds_fragment_reads": m,
            "mfma_issue_frac_of_measured": achieved / m, "source": "profiles/r01_peaks.json (tools/ubench/peaks.hip, 20 ms launches)"}


def synth_batch(B, H, W, C, seed, side=(40, 160), boxes=None):
    """images uint8 U[0,255] - caffe means; K~U{1..3} boxes, sides U[40,160].
    boxes: exactly that many objects per image instead of U{1..3} (--boxes-per-image: LineMOD has 1, T-LESS 8-15)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, size=(B, H, W, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)
    anns, images = [], []
    fx, fy, cx, cy = 572.4114, 573.57043, 325.2611, 242.04899
    for b in range(B):
        K = int(rng.integers(1, 4)) if boxes is None else int(boxes)
        mask = np.zeros((H, W), np.uint8)
        a = {"mask": [mask], "labels": np.empty((0,)), "bboxes": np.empty((0, 4)), "poses": np.empty((0, 7)),
             "segmentations": np.empty((0, 8, 3)), "cam_params": np.empty((0, 4)), "mask_ids": np.empty((0,))}
        for k in range(K):
            w, h = rng.uniform(side[0], side[1], 2)
            x1, y1 = rng.uniform(0, W - w), rng.uniform(0, H - h)
            mask[int(y1):int(y1 + h), int(x1):int(x1 + w)] = k + 1
            z = 800.0
            sx, sy, sz = w * z / fx, h * z / fy, rng.uniform(40, 120)
            box = np.array([[sx / 2, sy / 2, sz / 2], [sx / 2, sy / 2, -sz / 2], [sx / 2, -sy / 2, -sz / 2], [sx / 2, -sy / 2, sz / 2],
                            [-sx / 2, sy / 2, sz / 2], [-sx / 2, sy / 2, -sz / 2], [-sx / 2, -sy / 2, -sz / 2], [-sx / 2, -sy / 2, sz / 2]], np.float32)
            tx, ty = ((x1 + w / 2) - cx) * z / fx, ((y1 + h / 2) - cy) * z / fy
            a["labels"] = np.concatenate([a["labels"], [float(rng.integers(0, C))]])
            a["bboxes"] = np.concatenate([a["bboxes"], [[x1, y1, x1 + w, y1 + h]]])
            a["poses"] = np.concatenate([a["poses"], [[tx, ty, z, 1.0, 0.0, 0.0, 0.0]]])
            a["segmentations"] = np.concatenate([a["segmentations"], [box]])
            a["cam_params"] = np.concatenate([a["cam_params"], [[fx, fy, cx, cy]]])
            a["mask_ids"] = np.concatenate([a["mask_ids"], [float(k + 1)]])
        anns.append(a)
        images.append(np.zeros((H, W, 3), np.float32))
    return x, images, anns


def cpu_baseline(weights, x, anns, targets, C, H, W, n_timed=2):
    """The CPU leg (rank 0, N=1): the oracle's PyTorch-CPU float32 train step on the WHOLE batch of the timed configuration
    (batch 8; weights converted to contiguous OIHW once, outside the timed region; one warm-up step, then n_timed timed steps,
    the fastest counts), plus the oracle's anchors_for_shape + anchor_targets_bbox on the whole batch pinned to ONE thread
    (the reference's numpy / Cython path holds the GIL: SURVEY 8d; min of 5 repetitions).  kind = "port": the Keras/TF
    reference cannot run here (tensorflow / keras are not installed)."""
    import platform
    from oracle import anchors_np as AN
    from oracle import model_torch as MT
    nb = len(anns)
    yb, yc, ym = (t[:nb].cpu().numpy() for t in targets)
    # conv kernels as contiguous OIHW views of HWIO storage, made ONCE: oracle.conv2d's permute(3, 2, 0, 1) of these is a
    # contiguous tensor, so no step re-lays 42 M weights out (the r02 line did, on every call)
    wt = {}
    for k, v in weights.items():
        a = np.asarray(v, np.float32)
        if k.endswith("/kernel") and a.ndim == 4:
            a = np.ascontiguousarray(a.transpose(3, 2, 0, 1)).transpose(2, 3, 1, 0)  # HWIO view of OIHW-contiguous memory
        wt[k] = a
    nthr = torch.get_num_threads()
    try:
        phys = len({tuple(sorted(int(c) for c in open(f).read().replace("-", ",").split(",") if c.strip().isdigit()))
                    for f in __import__("glob").glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list")}) or None
    except (OSError, ValueError):
        phys = None

    def step():
        _, grads, _ = MT.loss_and_grads(wt, x[:nb], yb, yc, ym, C, torch.float32)
        m = {k: torch.zeros_like(g) for k, g in grads.items()}
        v = {k: torch.zeros_like(g) for k, g in grads.items()}
        MT.adam_clipnorm_step(wt, grads, m, v, 1)
    step()  # warm-up: first-touch, autograd graph construction, thread pool
    times = []
    for _ in range(n_timed):
        t1 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t1)
    cpu_dt = min(times)
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # anchor / target leg: ONE thread (torch / BLAS pools pinned), min of 5
    B = len(anns)
    shapes = [(H, W, 3)] * B
    torch.set_num_threads(1)
    try:
        AN.anchors_for_shape((H, W))
        ta = []
        for _ in range(5):
            t1 = time.perf_counter()
            anc = AN.anchors_for_shape((H, W))
            ta.append(time.perf_counter() - t1)
        AN.anchor_targets_bbox(anc, shapes, anns, C)
        tt = []
        for _ in range(5):
            t1 = time.perf_counter()
            AN.anchor_targets_bbox(anc, shapes, anns, C)
            tt.append(time.perf_counter() - t1)
    finally:
        torch.set_num_threads(nthr)
    return {"value": nb / cpu_dt, "unit": "images/sec", "cores": nthr, "physical_cores": phys, "kind": "port", "cpu_model": cpu_model,
            "step_seconds": times,
            "sample": "train step (fwd+loss+bwd+Adam) on the whole synthetic batch of %d images, PyTorch-CPU float32 restatement "
                      "(oracle/model_torch.py; weights contiguous OIHW, converted outside the timed region), not Keras: 1 warm-up + "
                      "%d timed steps, fastest; torch.get_num_threads() = %d" % (nb, n_timed, nthr),
            "anchors_targets": {"anchors_for_shape_ms": 1e3 * min(ta), "anchor_targets_bbox_ms_per_batch": 1e3 * min(tt),
                                "batch": B, "threads": 1, "repetitions": 5, "statistic": "min",
                                "kind": "port (oracle/anchors_np.py, numpy float64; pinned bit-exact "
                                "to the reference's utils/anchors.py by tests/golden)"}}


def inference_leg(ctx, mode, B=32, C=8, H=480, W=640, steps=6, warmup=2):
    """BASELINE configs[2] / SURVEY 8d config 3: Occlusion-style inference, batch 32, 8 classes: forward + device anchors (D1) +
    box3D decode (D2) + score > 0.5 compaction (D3), and the same plus filter_detections (D4: NMS / top-300) -- timed
    with HIP events on the ctx stream.  The final cls bias is shifted so that ~1 % of the scores pass 0.5 (seeded)."""
    from pyrapose_amd import arch, ops
    from pyrapose_amd.engine import Engine
    Wt = arch.init_weights(C, seed=0)
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.integers(0, 256, (B, H, W, 3)).astype(np.float32) - np.array([103.939, 116.779, 123.68], np.float32)).cuda()
    probe = Engine(ctx, C, 1, H, W, weights=Wt, train=False, conv_mode=mode)
    _, sc, _ = probe.predict_on_batch(x[:1])
    q = float(torch.quantile(torch.logit(sc.flatten()[:: 7].double().clamp(1e-7, 1 - 1e-7)), 0.99))
    Wt["cls_out/bias"] = (np.asarray(Wt["cls_out/bias"]) - q).astype(np.float32)
    del probe
    eng = Engine(ctx, C, B, H, W, weights=Wt, train=False, conv_mode=mode)
    N = eng.anchors_device_f32().shape[0]
    res = {}

    def step(nms):
        boxes3d, scores, mask = eng.predict_on_batch(x)
        idx = ops.score_threshold_compact(ctx, scores, 0.5)
        if nms:
            xs, ys = boxes3d[..., 0::2], boxes3d[..., 1::2]
            boxes = torch.stack([xs.amin(-1), ys.amin(-1), xs.amax(-1), ys.amax(-1)], -1).contiguous()
            ops.filter_detections_batch(ctx, boxes, boxes3d, scores, 0.05, 0.5, 300)
        return scores
    for nms in (False, True):
        for _ in range(warmup):
            scores = step(nms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            scores = step(nms)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        res["decode_compact_nms" if nms else "decode_compact"] = {"value": B * 1e3 / ms, "unit": "images/sec", "ms_per_batch": ms,
                                                                 "anchors_per_sec": B * N * 1e3 / ms}
    res.update({"workload": "Occlusion-style inference, batch %d, %d classes, %dx%d, forward + anchors + box3D decode + score>0.5 "
                            "compaction [+ filter_detections] (BASELINE configs[2])" % (B, C, W, H),
                "frac_scores_over_0.5": float((scores > 0.5).float().mean()), "steps": steps, "warmup": warmup,
                "dtype": "f32" if eng.conv_mode == "f32" else {"mixed": "bf16x3+f16c8"}.get(eng.arith, eng.arith),
                "n_gpus": 1, "data": "synthetic"})
    del eng
    torch.cuda.empty_cache()
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


DEFAULT_RANK_TIMEOUT_S = 900.0


def _tail(f, n=40):
    f.flush()
    f.seek(0)
    lines = f.read().splitlines()
    return lines[-n:]


def spawn_ranks(n, argv, worker=None, timeout=DEFAULT_RANK_TIMEOUT_S, env_extra=None):
    """`bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): start the N ranks ourselves.

    This process has made no GPU call (torch is not even imported) and makes none: it starts one child per rank --
    `python bench.py <same argv>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the
    environment, which is exactly what torch.distributed.run would export -- waits for them, relays rank 0's JSON line
    and returns non-zero if any rank failed (the others are then terminated by PID; nothing is ever re-exec'd, a failed rank
    is never restarted in place).  `timeout` (seconds, `--rank-timeout`, default 900; None / 0 = none): a rank stuck in RCCL
    initialisation ends the run with a message instead of hanging the parent.  Every rank's stderr goes to a temporary file
    and is passed on when the rank ends; on failure the failing rank's last 40 lines are repeated under a header, so the
    first real multi-GPU run can be diagnosed from its log.  `worker` (tests): the argv prefix to run instead of
    [python, bench.py]."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    procs, errs = [], []
    out0 = tempfile.TemporaryFile(mode="w+")  # rank 0's stdout (a file, so a long line can never block on a pipe)
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "PP_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(env_extra or {})
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=(out0 if r == 0 else subprocess.DEVNULL), stderr=errs[r]))
    t_end = None if not timeout else time.time() + float(timeout)
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if t_end is not None and time.time() > t_end:
            running = [r for r, c in enumerate(codes) if c is None]
            failed = (running[0] if running else -1, "no exit within %.0f s (ranks still running: %s)" % (float(timeout), running))
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()          # by PID: only the children started above
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        sys.stderr.write("bench.py: rank %s failed (%s); %d-rank run aborted\n" % (failed[0], failed[1], n))
        if 0 <= failed[0] < n:
            sys.stderr.write("bench.py: ---- last lines of rank %d's stderr ----\n" % failed[0])
            for ln in _tail(errs[failed[0]]):
                sys.stderr.write("  [rank %d] %s\n" % (failed[0], ln))
            sys.stderr.write("bench.py: ---- end of rank %d's stderr ----\n" % failed[0])
        sys.stderr.flush()
        return 1
    for r in range(n):  # a clean run: pass the ranks' stderr on (warnings, PP_DP_DEBUG bucket plans)
        for ln in _tail(errs[r], 200):
            sys.stderr.write("[rank %d] %s\n" % (r, ln))
    out0.seek(0)
    out = out0.read()
    line = None
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            line = ln
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    try:
        got = json.loads(line).get("n_gpus")
    except ValueError:
        got = None
    if got != n:
        sys.stderr.write("bench.py: rank 0 reports n_gpus=%r, expected %d\n" % (got, n))
        return 1
    print(line)
    sys.stdout.flush()
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--classes", type=int, default=13)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--backbone", default="resnet50", choices=["resnet50", "resnet101"],
                    help="resnet101 = the [3,4,23,3] variant of BASELINE configs[4] (with --height 540 --width 720 --classes 30)")
    ap.add_argument("--conv-mode", default=None, choices=["f32", "bf16x3", "f16c8", "mixed"],
                    help="arithmetic of the convolutions (engine.py): default env PP_CONV_MODE or mixed = bf16x3 backbone, f16c8 FPN + heads")
    ap.add_argument("--boxes-per-image", type=int, default=None,
                    help="objects per synthetic image (default: U{1..3}, SURVEY 8d config 2).  The sparse backward of the 3D-box head makes "
                         "`value` depend on the annotation density: LineMOD has 1 object per image, YCB-V / T-LESS 5-15")
    ap.add_argument("--rank-timeout", type=float, default=DEFAULT_RANK_TIMEOUT_S,
                    help="--gpus N without a launcher: seconds after which ranks that have not exited are terminated (0 = none)")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the short run of the other conv mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write a per-launch table (name, kind, GFLOP, avg us, TFLOP/s)")
    ap.add_argument("--no-inference", action="store_true", help="skip the config-3 inference leg (B=32, C=8)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="with PP_PREFETCH=1 (engine built with a prefix lane): do not use the look-ahead -- the frozen prefix (conv1 + res2) "
                         "of a batch runs inside its own step instead of beside the previous one.  Without PP_PREFETCH=1 (default) there is no look-ahead.")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of N ranks (before torch is imported or any GPU call is made)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], timeout=args.rank_timeout))
    main_worker(args)


def main_worker(args):
    global np, torch
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it spawns the ranks) or under "
                         "torch.distributed.run --nproc-per-node N with the same N" % (args.gpus, world))
    import torch.distributed as dist
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PP_DIST_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode (several ranks share a card); the driver's
        # scaling runs use the default: RCCL ("nccl"), one rank per GPU.
        backend = os.environ.get("PP_DIST_BACKEND", "nccl")
        n_dev = torch.cuda.device_count()
        if backend == "nccl" and n_dev < world:
            raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, %d visible (PP_DIST_BACKEND=gloo rehearses the N-rank "
                             "path with several ranks per card)" % (world, world, n_dev))
        local_rank = local_rank % max(n_dev, 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    else:
        torch.cuda.set_device(0)
        local_rank = 0

    from pyrapose_amd import arch
    from pyrapose_amd.engine import Engine
    from pyrapose_amd.parallel import DataParallel
    from pyrapose_amd.runtime import default_context
    from pyrapose_amd.utils import anchors as UA

    B, H, W, C = args.batch, args.height, args.width, args.classes
    ctx = default_context(local_rank)
    weights = arch.init_weights(C, seed=0, backbone=args.backbone)
    x, images, anns = synth_batch(B, H, W, C, seed=1000 + rank, boxes=args.boxes_per_image)
    anchors = UA.anchors_for_shape_device((H, W))
    y_box, y_cls, y_mask = UA.anchor_targets_bbox_device(anchors, images, anns, C)
    x_dev = torch.from_numpy(x).cuda()
    fwd_fl, bwd_fl = arch.conv_flops(C, H, W, args.backbone)
    algo_gflop = ALGO_GFLOP_PER_IMAGE.get((C, H, W), (fwd_fl + bwd_fl) / 1e9) if args.backbone == "resnet50" else (fwd_fl + bwd_fl) / 1e9

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def sparse_fractions(eng):
        """Share of the reduction that the row-block skip of the 3D-box head's backward actually executes, per layer:
        weight gradient = 32-row blocks of dy that hold a non-zero / all blocks; data gradient = 32-row blocks of dx that such
        a block can reach (the second half of the flags buffer, written by the bwd-data launch) / all blocks."""
        out = {}
        for op in eng.graph_ops:
            sk = op.get("skip")
            if sk is None:
                continue
            nb = sk[0].numel() // 2
            both = sk[0].cpu().numpy().astype(bool)
            out[op["spec"].name] = {"wgrad": float(both[:nb].mean()), "dgrad": float(both[nb:].mean())}
        return out

    pf_flags = []  # per run(): did the steps use the look-ahead?

    def run(mode, steps, warmup, events, dump_ops=None, sparse=None, sparse_fwd=False):
        """Build an engine in `mode`, run warmup + timed steps, return (dt, images, losses, roofline dict)."""
        if sparse is not None:
            os.environ["PP_SPARSE_BWD"] = sparse
        if sparse_fwd:
            os.environ["PP_SPARSE_FWD"] = "1"
        eng = Engine(ctx, C, B, H, W, backbone=args.backbone, weights=weights, train=True, conv_mode=mode)
        os.environ.pop("PP_SPARSE_BWD", None) if sparse is not None else None
        os.environ.pop("PP_SPARSE_FWD", None) if sparse_fwd else None
        if world > 1:
            DataParallel(eng)
        eng.set_targets(y_box, y_cls, y_mask)
        eng.x_in.copy_(x_dev)
        torch.cuda.synchronize()
        # look-ahead (opt-in: PP_PREFETCH=1; engine.py, prefix lane): conv1 + res2 are frozen (bin/train.py:72-78, models/resnet.py:87-110), so their
        # output for batch i+1 does not depend on step i's update and is computed beside step i's trunk on a stream of its
        # own -- ONE prefix per step either way (the last timed step computes the one of a batch that never comes)
        nx = eng.RESIDENT if (eng.prefix_lane is not None and not args.no_prefetch) else None
        pf_flags.append(nx is not None)
        for _ in range(warmup):
            eng.train_step(next_x=nx)
        records = []
        units = {}  # MFMA issue slots per product of each conv layer: f32 1, bf16x3 3, f16c8 2 (csrc/planes_fmt.h)
        # per-launch events cost CPU time (the launch loop must stay ahead of the GPU): they are recorded on every
        # EVENT_EVERY-th timed step only, from pools created before the timed region
        sample = [-1]  # index of the sampled step, or -1
        every = 1 if dump_ops else EVENT_EVERY
        n_pool = (steps + every - 1) // every
        if events:
            def wrap(op):
                inner = op.fn
                st = eng.streams[op.lane]
                pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_pool)]
                def fn():
                    k = sample[0]
                    if k < 0:
                        return inner()
                    s, e = pool[k]
                    s.record(st)
                    inner()
                    e.record(st)
                    records.append((op.kind, op.name, op.flops, s, e))
                units[op.name] = 1 if eng.conv_mode == "f32" else (2 if eng._fmt(op.name) == 1 else 3)

                op.fn = fn
            for op in eng.fwd_ops + eng.bwd_ops:
                if op.kind in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
                    wrap(op)
        barrier()
        base_ev = torch.cuda.Event(enable_timing=True)
        base_ev.record(eng.streams[0])
        n_sampled = 0
        t0 = time.perf_counter()
        for i in range(steps):
            sample[0] = (i // every) if (events and i % every == 0) else -1
            n_sampled += int(sample[0] >= 0)
            eng.train_step(next_x=nx)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        losses = eng.losses()
        roof = None
        sp_frac = sparse_fractions(eng) if rank == 0 else {}
        if records and rank == 0:
            # Launches of independent chains run on different lanes (streams) and overlap, so the family rate is
            # algorithmic flops / length of the UNION of the launch intervals (time during which >= 1 conv kernel
            # ran); per-kernel figures use each launch's own event-bracketed interval (inflated where launches overlap).
            agg, spans = {}, []
            dense_flops = unit_flops = 0.0
            for kind, name, flops, s_ev, e_ev in records:
                t_s, t_e = base_ev.elapsed_time(s_ev) * 1e-3, base_ev.elapsed_time(e_ev) * 1e-3
                spans.append((t_s, t_e))
                a = agg.setdefault(kind, [0.0, 0.0, 0])
                dense_flops += flops
                # `achieved` follows the contract (SURVEY 8d algorithmic flops); `executed` credits the launches that skip zero
                # blocks of the gradient with the share of the reduction they run: a statement about the kernels, not the data
                flops *= sp_frac.get(name, {}).get({"conv_wgrad": "wgrad", "conv_dgrad": "dgrad"}.get(kind, ""), 1.0)
                a[0] += flops
                a[1] += t_e - t_s
                a[2] += 1
                unit_flops += flops * units[name]
            spans.sort()
            union, cur_s, cur_e = 0.0, spans[0][0], spans[0][1]
            for t_s, t_e in spans[1:]:
                if t_s > cur_e:
                    union += cur_e - cur_s
                    cur_s, cur_e = t_s, t_e
                else:
                    cur_e = max(cur_e, t_e)
            union += cur_e - cur_s
            # the largest single launch (most algorithmic flops), by its own event-bracketed interval (launches of the other lane
            # overlap it, so this is a lower bound on the kernel's stand-alone rate; profiles/ holds the rocprofv3 durations)
            big = {}
            for kind, name, flops, s_ev, e_ev in records:
                b = big.setdefault((kind, name), [flops, 0.0, 0])
                b[1] += s_ev.elapsed_time(e_ev) * 1e-3
                b[2] += 1
            (bk, bn), (bfl, bsec, bcnt) = max(((k, v) for k, v in big.items() if k[0] == "conv_fwd"), key=lambda kv: kv[1][0])
            largest = {"launch": "%s %s" % (bk, bn), "gflop": bfl / 1e9, "avg_us": 1e6 * bsec / bcnt, "tflops": bfl / (bsec / bcnt) / 1e12,
                       "units": units[bn]}
            names = {"conv_fwd": "igemm fwd", "conv_dgrad": "igemm bwd-data", "conv_wgrad": "wgrad"}
            kernels = [{"kernel": names[k], "launches": n, "avg_ms": 1e3 * sec / n, "tflops_own_interval": fl / sec / 1e12}
                       for k, (fl, sec, n) in agg.items()]
            fl = sum(v[0] for v in agg.values())
            roof = {"achieved": dense_flops / union / 1e12, "executed": fl / union / 1e12, "executed_units": unit_flops / union / 1e12, "sparse": sp_frac,
                    "conv_share_of_step": union / (dt * n_sampled / steps), "lanes": eng.n_lanes,
                    "sampled_steps": n_sampled, "largest_launch": largest,
                    "dominant": max(agg.items(), key=lambda kv: kv[1][1])[0], "per_kernel": kernels}
            if dump_ops:
                per, order = {}, []
                for kind, name, flops, s_ev, e_ev in records:
                    key = (kind, name)
                    if key not in per:
                        per[key] = [flops, 0.0, 0]
                        order.append(key)
                    per[key][1] += s_ev.elapsed_time(e_ev) * 1e-3
                    per[key][2] += 1
                with open(dump_ops, "w") as f:
                    f.write("kind,name,gflop,avg_us,tflops\n")
                    for key in order:
                        fl_, sec, n = per[key]
                        f.write("%s,%s,%.3f,%.1f,%.1f\n" % (key[0], key[1], fl_ / 1e9, 1e6 * sec / n, fl_ / (sec / n) / 1e12))
        mode_used = "f32" if eng.conv_mode == "f32" else eng.arith  # the arithmetic that ran: f32 | bf16x3 | f16c8 | mixed
        if roof is None and sp_frac:
            roof = {"sparse": sp_frac}
        # the P16 tensors of the last step (outside the timed region): nothing at the encode's clamp, largest |half| per group
        # (Engine.p16_stats / pp_planes_stats: the format saturates and underflows silently, this is where it would show)
        if rank == 0 and mode_used in ("mixed", "f16c8"):
            try:
                P16_AUDIT.clear()
                P16_AUDIT.update(eng.p16_stats())
            except Exception as e:  # noqa: BLE001 -- an audit, never a reason to lose the measurement
                P16_AUDIT["error"] = repr(e)
        del eng
        torch.cuda.empty_cache()
        return dt, steps * B * world, losses, roof, mode_used

    dt, images_total, losses, roof, mode = run(args.conv_mode, args.steps, args.warmup, not args.no_kernel_events, args.dump_ops)
    P16_AUDIT_HEADLINE.update(P16_AUDIT)
    prefetch_used = pf_flags[0]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = images_total / dt
    roofline = None
    if roof and "achieved" in roof:
        # HBM traffic of the dominant launch: rocprofv3 PMC counter data (separate --pmc passes, tools/pmc_traffic.sh) kept under
        # profiles/ -- the newest r*_traffic.json.  It is a measurement of a BUILD: the file names the sha256 of the kernel
        # source it was taken on, and the line says whether that is still the source of the library that just ran
        traffic = None
        import glob
        import hashlib
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
        if tfiles and (C, H, W, B) == (13, 480, 640, 8):
            with open(tfiles[-1]) as f:
                tj = json.load(f)
            traffic = tj.get(mode)
            if traffic is not None:
                try:  # (the plane-format kernels = conv3.hip + the arithmetic header it is compiled against)
                    hsh = hashlib.sha256()
                    for src in (["conv.hip"] if mode == "f32" else ["conv3.hip", "conv3_shared.h", "planes_fmt.h", "p16.h"]):
                        with open(os.path.join(ROOT, "pyrapose_amd", "csrc", src), "rb") as f:
                            hsh.update(f.read())
                    now = hsh.hexdigest()[:16]
                except OSError:
                    now = None
                traffic = dict(traffic, file=os.path.relpath(tfiles[-1], ROOT), kernel_source_sha16_now=now,
                               stale=(traffic.get("kernel_source_sha16") != now))
        if mode != "f32":
            peak = PEAK_BF16_MFMA_TFLOPS  # (dense f16 = dense bf16 = 2.5 PFLOP/s; the e5m2 cross terms are priced in f16-MFMA units, below)
            kname = {"bf16x3": "conv implicit-GEMM family (igemm3x / igemm3f fwd + bwd-data, wgrad3f): 3 x v_mfma_f32_32x32x16_bf16 per product, f32 accumulate",
                     "f16c8": "conv implicit-GEMM family (igemm3x / igemm3f fwd + bwd-data, wgrad3f) on P16 planes: per 32-deep step 2 x "
                              "v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 (e5m2 cross terms) = 2 MFMA units per product"}.get(
                mode, "conv implicit-GEMM family (igemm4x = the multi-stage LDS-DMA form of the 512-wide head launches, igemm3x / igemm3f fwd + "
                      "bwd-data, wgrad3f; wgrad3w = the producer / consumer-wave form of the dense 512-channel weight gradients), one "
                      "source compiled per plane format: backbone = 3 x v_mfma_f32_32x32x16_bf16 per product "
                      "(bf16x3); FPN + heads = v_mfma_f32_32x32x16_f16 + half a v_mfma_scale_f32_32x32x64_f8f6f4 per 16-deep step "
                      "(f16c8: 2 MFMA units per product)")
        else:
            peak, kname = PEAK_F32_MFMA_TFLOPS, "conv implicit-GEMM family (igemm fwd/bwd-data + wgrad): v_mfma_f32_32x32x2_f32"
        mm = roof["executed_units"] / roof["executed"]  # MFMA issue slots (16-deep f16/bf16 MFMA equivalents) per executed product
        roofline = {"bound": "mfma", "achieved": roof["executed"], "peak": peak, "unit": "TFLOP/s", "frac": roof["executed"] / peak,
                    "achieved_algorithmic": roof["achieved"], "frac_algorithmic": roof["achieved"] / peak,
                    "traffic": (traffic or {}).get("hbm_bytes_per_launch"), "traffic_detail": traffic, "kernel": kname,
                    "method": "achieved / frac = EXECUTED 2*MAC flops of every conv launch / union of their HIP-event intervals (events on the "
                              "launch streams), on every %d-th step of the timed region (%d of %d steps; per-launch events on every step "
                              "cost ~2.5 %% of the step time).  Executed = algorithmic, except that the backward launches of the 3D-box "
                              "head, which skip the zero blocks of their sparse gradient, are credited only with the share of the "
                              "reduction they run (sparse_backward.executed_share).  achieved_algorithmic / frac_algorithmic count the "
                              "skipped flops too (SURVEY 8d work per image): a statement about the step, not about the kernels"
                              % (EVENT_EVERY, roof["sampled_steps"], args.steps),
                    "mfma_flops_per_algorithmic_flop": mm,
                    "mfma_issue_frac": roof["executed"] * mm / peak,
                    "vs_f32_mfma_peak_157.3": roof["executed"] / PEAK_F32_MFMA_TFLOPS,
                    "measured_peak": measured_peak(mode, roof["executed_units"]),
                    "dominant": roof["dominant"], "conv_share_of_step": roof["conv_share_of_step"], "lanes": roof["lanes"],
                    "largest_launch": dict(roof["largest_launch"], frac=roof["largest_launch"]["tflops"] / peak,
                                           mfma_issue_frac=roof["largest_launch"]["tflops"] * roof["largest_launch"]["units"] / peak),
                    "per_kernel": roof["per_kernel"]}

    other = None
    if world == 1 and not args.no_alt_mode:
        # the other arithmetics on the same box, same process (short runs): the exact f32 MFMA path, and -- when the step ran mixed --
        # the all-bf16x3 step it replaced (3 MFMA units per product everywhere)
        other = []
        for alt in (["f32"] if mode == "bf16x3" else (["bf16x3"] if mode == "f32" else ["bf16x3", "f32"])):
            dt2, img2, losses2, roof2, _ = run(alt, max(3, args.steps // 2), 2, True)
            pk = PEAK_F32_MFMA_TFLOPS if alt == "f32" else PEAK_BF16_MFMA_TFLOPS
            other.append({"conv_mode": alt, "value": img2 / dt2, "unit": "images/sec", "ms_per_step": 1e3 * dt2 / max(3, args.steps // 2),
                          "roofline_achieved_tflops": roof2["achieved"] if roof2 else None, "roofline_peak": pk,
                          "roofline_frac": (roof2["achieved"] / pk) if roof2 else None, "losses": losses2})

    sparse = None
    if roof and roof.get("sparse"):
        sparse = {"what": "the gradient of the 3D-box loss is exactly zero away from the positive anchors (orthogonal_l1 keeps state == 1 rows, "
                          "losses.py:332-333); the bwd-weight / bwd-data launches of that head reduce over the 32-row blocks that hold a "
                          "non-zero only (pp_row_block_list + pp_ctx_set_row_block_skip).  Exact, data-dependent; targets = the synthetic "
                          "annotations of SURVEY.md 8d config 2 (1-3 boxes of 40-160 px per image).  PP_SPARSE_BWD=0 runs every launch dense.",
                  "executed_share": roof["sparse"]}
        if world == 1 and not args.no_alt_mode:
            dt3, img3, _, _, _ = run(mode, max(3, args.steps // 2), 2, False, sparse="0")
            sparse["dense_backward"] = {"value": img3 / dt3, "unit": "images/sec", "ms_per_step": 1e3 * dt3 / max(3, args.steps // 2)}
            if mode != "f32" and os.environ.get("PP_SPARSE_FWD", "0") != "1":
                # NOT part of `value`: the opt-in PP_SPARSE_FWD=1 also skips, in the FORWARD pass of a training step, the rows of the
                # 3D-box head that its loss never reads (dead outputs in train_on_batch; losses / gradients / weights unchanged,
                # tests/test_gpu_pipeline.py) -- reported so that the number exists, kept out of the headline because the skipped
                # rows are not zeros but unread values
                try:  # (an extra: it must never cost the line its headline)
                    dt4, img4, losses4, _, _ = run(mode, max(3, args.steps // 2), 2, False, sparse_fwd=True)
                    sparse["sparse_forward_opt_in"] = {"value": img4 / dt4, "unit": "images/sec", "ms_per_step": 1e3 * dt4 / max(3, args.steps // 2),
                                                       "losses": losses4, "env": "PP_SPARSE_FWD=1"}
                except Exception as e:  # noqa: BLE001
                    os.environ.pop("PP_SPARSE_FWD", None)
                    sparse["sparse_forward_opt_in"] = {"error": "%s: %s" % (type(e).__name__, e)}

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(weights, x, anns, (y_box, y_cls, y_mask), C, H, W)

    inference = None
    if world == 1 and not args.no_inference:
        inference = inference_leg(ctx, mode)

    out = {
        "metric": "images/sec 640x480 fwd+bwd (train step: fwd + losses + bwd + clipnorm-Adam)",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "rccl_ranks": (world if backend == "nccl" else 0), "dist_backend": backend,
        "value_dense_backward": ((sparse or {}).get("dense_backward") or {}).get("value"),
        "dtype": {"mixed": "bf16x3+f16c8"}.get(mode, mode), "data": "synthetic", "sparse_backward": sparse,
        "pipeline": ("software-pipelined over steps: the frozen prefix (conv1 + res2, no trainable weight) of batch i+1 runs on its own "
                     "stream beside batch i; every step executes one prefix, one trunk + heads forward, one backward, one optimizer "
                     "update (--no-prefetch: everything of a batch inside its own step)") if prefetch_used else "none",
        "dtype_note": DTYPE_NOTES[mode],
        "config": {"workload": "%s %d-class training, batch %d/GPU, %dx%d, %s PFPN + heads (BASELINE configs[%s]), %s"
                               % ("LineMOD" if C == 13 else ("YCB-Video" if C == 21 else ("T-LESS" if C == 30 else "synthetic")), C, B, W, H,
                                  {"resnet50": "ResNet-50", "resnet101": "ResNet-101"}[args.backbone],
                                  "1" if (C, H, W, args.backbone) == (13, 480, 640, "resnet50") else
                                  ("3" if (C, H, W, args.backbone) == (21, 480, 640, "resnet50") else
                                   ("4" if (C, H, W, args.backbone) == (30, 540, 720, "resnet101") else "-")),
                                  ("%d boxes per image" % args.boxes_per_image) if args.boxes_per_image is not None else "1-3 boxes per image"),
                   "boxes_per_image": args.boxes_per_image if args.boxes_per_image is not None else "U{1..3}",
                   "global_batch": B * world, "parallelism": "dp%d" % world,
                   "algorithmic_gflop_per_image": algo_gflop},
        "step_tflops": images_total * algo_gflop / dt / 1e3,
        "losses": losses,
        "roofline": roofline, "cpu_baseline": cpu, "other_mode": other, "inference": inference,
        "p16_audit": (dict(P16_AUDIT_HEADLINE) or None),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
