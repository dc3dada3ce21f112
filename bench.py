#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chamfer|fps|ball_group]

Default workload = BASELINE.json configs[1]: Chamfer forward+backward, B=32 (per GPU), N=M=16384,
C=3, fp32, synthetic area-uniform unit-sphere clouds, through the public autograd API
(pytorch_points_amd.network.model_loss.nndistance -> _ext.losses -> C ABI -> HIP kernels).
A "step" is one forward + one backward over that batch, inputs resident in HBM.

Metric: point-pairs/s = n_gpus * 2*B*N*M / t(step) (both directions counted).

N > 1 (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL): the batch is
sharded, B=32 per rank (weak scaling); each step also all-gathers the per-shard (dist, idx) over
xGMI as BASELINE.json's north_star specifies.  Timing = barrier + synchronize on both sides, MAX
over ranks; rank 0 prints ONE JSON line.

The JSON line also carries
  roofline      HBM roofline of the dominant kernel (the forward scan): algorithmic bytes per
                launch / its average duration, HIP events on the launch stream inside the timed loop
  valu          the roof that actually binds that kernel (fp32 VALU issue)
  cpu_baseline  the CPU oracle (a port of the reference semantics; the reference has no CPU
                path) timed on this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
VALU_PEAK_LANEOPS = 78.6e12    # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz fp32 lane-ops/s (same table)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="chamfer", choices=["chamfer", "fps", "ball_group"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--variant", type=int, default=0, help="forward kernel variant (0 = automatic)")
    ap.add_argument("--search", default="auto", choices=["auto", "bruteforce"],
                    help="chamfer: auto = the operator's default (exact grid search, brute-force "
                         "fallback); bruteforce = evaluate every pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "ext", "eager"],
                    help="chamfer: how the step's kernels are issued.  graph = hipGraph replay; ext = two calls "
                         "of the _ext.losses functions (forward, backward) on static buffers; eager = through "
                         "torch.autograd.Function; auto (default) = graph or ext, whichever a short calibration "
                         "after the warm-up finds faster on this host")
    ap.add_argument("--with-backward", action="store_true", help="ball_group: also time group_points_grad")
    return ap.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("PP_BENCH_FORCE_DIST") == "1":
        # PP_BENCH_FORCE_DIST=1: run the distributed code path (RCCL process group, packed all-gather,
        # graph capture beside the RCCL watchdog) with a single rank -- a logic check on a 1-GPU box
        import torch.distributed as dist
        if os.environ.get("PP_BENCH_DEBUG_GLOO") == "1":
            # logic check on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL (not a measurement)
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            return dist, world, rank, 0
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        return dist, world, rank, local
    if args.gpus > 1:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                         "(one rank per GPU)" % args.gpus)
    torch.cuda.set_device(0)
    return None, 1, 0, 0


def timed_region(dist, fn, steps, warmup, device):
    """W untimed steps, then exactly K steps between barrier+synchronize; MAX over ranks (seconds)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ------------------------------------------------------------------------------------- chamfer
def cpu_baseline_chamfer(N, C):
    """Oracle fwd+bwd on a bounded sample of the workload (same N=M, fewer batch elements)."""
    import oracle
    from pytorch_points_amd import synthetic as S
    oracle.build()
    x1 = S.unit_sphere(0, 1, N, C)
    x2 = S.unit_sphere(1, 1, N, C)

    def run(a, b):
        bb = a.shape[0]
        d1, i1, d2, i2 = oracle.chamfer_forward(a, b)
        g = np.full((bb, N), 1.0 / (bb * N), np.float32)
        oracle.chamfer_backward(a, b, g, g, i1, i2)

    run(x1[:, :256], x2[:, :256])  # thread-pool warm-up
    t0 = time.perf_counter()
    run(x1, x2)
    t1 = time.perf_counter() - t0
    # about 20 core-seconds of CPU work per repetition, at most the full batch of 32
    cores = oracle.num_threads()
    bs = int(max(1, min(32, round(20.0 / max(t1 * cores, 1e-3)))))
    a = np.ascontiguousarray(np.repeat(x1, bs, 0))
    b = np.ascontiguousarray(np.repeat(x2, bs, 0))
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(a, b)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    return {"value": 2.0 * bs * N * N / dt, "unit": "pairs/s", "cores": cores,
            "kind": "port", "sample": "Chamfer fwd+bwd B=%d N=M=%d C=%d, best of 3 (oracle/pp_oracle.c: "
            "OpenMP, AVX2+FMA, same canonical arithmetic), %.2f s = %.0f core-seconds"
            % (bs, N, C, dt, dt * cores)}


def bench_chamfer(args, dist, world, rank, device):
    from pytorch_points_amd import _lib, synthetic as S
    from pytorch_points_amd.network.model_loss import nndistance
    import ctypes
    B = args.batch or 32
    N = args.points or 16384
    M, C = N, 3
    if args.search == "bruteforce":
        fs = _lib.lib().pp_debug_set_nmdistance_search
        fs.argtypes = [ctypes.c_int]
        fs.restype = None
        fs(1)
    if args.variant:
        fn = _lib.lib().pp_debug_set_nmdistance_variant
        fn.argtypes = [ctypes.c_int]
        fn.restype = None
        fn(args.variant)
    # rank r owns batch elements [r*B, (r+1)*B) of the global batch; seeds 0 / 1 as SURVEY.md §8d
    x1 = torch.from_numpy(S.unit_sphere(0, B, N, C, batch_offset=rank * B)).to(device)
    x2 = torch.from_numpy(S.unit_sphere(1, B, M, C, batch_offset=rank * B)).to(device)
    x1.requires_grad_(True)
    x2.requires_grad_(True)
    g1 = torch.full((B, N), 1.0 / (B * N), device=device)   # gradient of dist.mean()
    g2 = torch.full((B, M), 1.0 / (B * M), device=device)
    exchange = None
    if dist is not None:
        # one asynchronous collective per step: (dist1 | dist2 | idx1 | idx2) of the shard, packed (idx as
        # 16-bit words: 6 MiB per rank at B=32, N=M=16384), double-buffered (pytorch_points_amd/sharded.py)
        from pytorch_points_amd.sharded import PackedShardGather
        exchange = PackedShardGather(B, N, M, device)
    fwd_events = []
    pending = []   # slot of the previous step's all-gather

    instrument = [True]   # HIP events around the forward (they cost host time: off for the eager timing)

    def step():
        x1.grad = None
        x2.grad = None
        if instrument[0]:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        d1, d2, i1, i2 = nndistance(x1, x2)
        if instrument[0]:
            e1.record()
            fwd_events.append((e0, e1))
        if exchange is not None:
            # all-gather of the per-shard (dist, idx) over xGMI (RCCL), asynchronous: it runs on the
            # collective stream beside this step's backward and the next step's forward; the previous
            # step's gathered result is consumed (unpacked to the global-batch tensors) first
            if pending:
                exchange.wait(pending.pop())
            pending.append(exchange.launch(d1, d2, i1, i2))
        torch.autograd.backward([d1, d2], [g1, g2])

    def drain():
        if exchange is not None:
            while pending:
                exchange.wait(pending.pop())
            exchange.drain()

    def run_timed(fn, warmup, steps):
        """W untimed steps, then exactly K steps between barrier+synchronize; MAX over ranks (seconds).
        The last step's gather is part of the timed work: drained before the closing synchronize."""
        for _ in range(warmup):
            fn()
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([t], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    # hipGraph replay of the same step (pytorch_points_amd/graphs.py): the kernels of one step take
    # less time than the Python/autograd work that launches them, so the eager loop is host-bound
    gstep = None
    graph_note = None
    if args.launch in ("auto", "graph"):
        try:
            from pytorch_points_amd.graphs import GraphedChamferStep
            gstep = GraphedChamferStep(B, N, M, device)
            with torch.no_grad():
                gstep.xyz1.copy_(x1)
                gstep.xyz2.copy_(x2)
                gstep.grad_dist1.copy_(g1)
                gstep.grad_dist2.copy_(g2)
            gstep.capture()
        except Exception as exc:  # capture not available: report the eager loop and say so
            gstep = None
            graph_note = "graph capture failed (%s: %s); eager launches timed instead" % (type(exc).__name__, exc)
    if dist is not None:  # every rank must take the same path
        flag = torch.tensor([1 if gstep is not None else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            gstep = None

    def graph_step():
        d1, d2, i1, i2, _, _ = gstep.replay()
        if exchange is not None:
            if pending:
                exchange.wait(pending.pop())
            pending.append(exchange.launch(d1, d2, i1, i2))   # packs on this stream, before the next replay

    # the step through the reference's extension-module API (pytorch_points._ext.losses: nmdistance_forward,
    # nmdistance_backward -- what the reference's autograd.Function calls, _ext/nmdistance.cpp:30-34) on
    # static buffers: two Python calls per step, no autograd bookkeeping, plain stream launches
    from pytorch_points_amd._ext import losses as ext_losses
    sx1, sx2 = x1.detach(), x2.detach()
    od1, od2 = torch.empty(B, N, device=device), torch.empty(B, M, device=device)
    oi1 = torch.empty(B, N, dtype=torch.int32, device=device)
    oi2 = torch.empty(B, M, dtype=torch.int32, device=device)
    ogx1, ogx2 = torch.empty_like(sx1), torch.empty_like(sx2)

    def ext_step():
        ext_losses.nmdistance_forward(sx1, sx2, od1, od2, oi1, oi2)
        if exchange is not None:
            if pending:
                exchange.wait(pending.pop())
            pending.append(exchange.launch(od1, od2, oi1, oi2))
        ext_losses.nmdistance_backward(sx1, sx2, ogx1, ogx2, g1, g2, oi1, oi2)

    instrument[0] = False
    modes = {}   # launch mode -> ms per step over a short calibration run (reported; "auto" picks from it)
    n_cal = 50
    if gstep is not None:
        modes["graph"] = run_timed(graph_step, 5, n_cal) / n_cal * 1e3
    modes["ext"] = run_timed(ext_step, 5, n_cal) / n_cal * 1e3
    n_eager = 200   # enough steps that the closing synchronize does not weigh on the per-step time
    eager_dt = run_timed(step, 3, n_eager) / n_eager
    modes["eager"] = eager_dt * 1e3
    want = args.launch
    if want == "graph" and gstep is None:
        want = "eager"
    if want == "auto":
        want = "graph" if (gstep is not None and modes["graph"] <= modes["ext"]) else "ext"
    timed_fn = {"graph": graph_step, "ext": ext_step, "eager": step}[want]
    dt = run_timed(timed_fn, args.warmup, args.steps)

    # forward duration: the forward's launches alone, issued back to back on the launch stream between two
    # HIP events (the _ext.losses call: the same launches as the operator's forward)
    for _ in range(3):
        ext_losses.nmdistance_forward(sx1, sx2, od1, od2, oi1, oi2)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n_eager):
        ext_losses.nmdistance_forward(sx1, sx2, od1, od2, oi1, oi2)
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / n_eager

    # the same step with the search forced to the brute-force kernel (every pair evaluated)
    brute = None
    if args.search == "auto":
        setter = _lib.lib().pp_debug_set_nmdistance_search
        setter.argtypes = [ctypes.c_int]
        setter.restype = None
        setter(1)
        try:
            bev = []

            def bstep():
                x1.grad = None
                x2.grad = None
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                d1, d2, i1, i2 = nndistance(x1, x2)
                e1.record()
                bev.append((e0, e1))
                torch.autograd.backward([d1, d2], [g1, g2])

            for _ in range(3):
                bstep()
            torch.cuda.synchronize()
            bev.clear()
            tb = time.perf_counter()
            nb = max(5, min(args.steps, 20))
            for _ in range(nb):
                bstep()
            torch.cuda.synchronize()
            tb = (time.perf_counter() - tb) / nb
            bfwd = float(np.mean([a.elapsed_time(b) for a, b in bev]))
            brute = {"ms_per_step": tb * 1e3, "fwd_ms": bfwd, "pairs_per_s_per_gpu": 2.0 * B * N * M / tb}
        finally:
            setter(0)

    pairs_per_step = 2.0 * B * N * M * world
    ms = dt / args.steps * 1e3
    alg_bytes_fwd = 4.0 * C * B * (N + M) + 8.0 * B * (N + M)
    hbm_gbs = alg_bytes_fwd / (fwd_ms * 1e-3) / 1e9
    # VALU lane-ops the brute-force kernel issues per pair: 3 sub + 1 mul + 2 fma + 1/2 min3 + 13/64
    # per-group bookkeeping (DESIGN.md "nmdist_fwd_c3_kernel")
    laneops = 2.0 * B * N * M * 6.703125
    grid = args.search == "auto" and int(_lib.lib().pp_nmdistance_forward_workspace_bytes(B, N, M, C)) > 0
    out = {
        "metric": "chamfer_fwd_bwd_point_pairs_per_s", "value": pairs_per_step / (dt / args.steps),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Chamfer fwd+bwd B=%d/GPU N=M=%d C=3 fp32, area-uniform unit sphere"
                               % (B, N), "global_batch": B * world,
                   "search": ("exact uniform-grid search with brute-force fallback (operator default): "
                              "outputs bit-identical to the brute force, most pairs pruned, value = "
                              "2*B*N*M/t" if grid else "brute force: every pair evaluated"),
                   "parallelism": "batch-shard x%d%s" % (world, " + RCCL all-gather(dist,idx), async" if world > 1 else "")},
        "fwd_ms": fwd_ms,
    }
    out["config"]["launch"] = {
        "graph": "hipGraph replay of the step's launches (same kernels as the eager operator)",
        "ext": "two calls per step of the extension-module API (_ext.losses.nmdistance_forward / _backward) on "
               "static buffers: plain stream launches, same kernels as the autograd operator",
        "eager": "eager (torch.autograd.Function, one Python call per operator)"}[want]
    out["launch_modes_ms_per_step"] = dict(modes, note="same kernels in every mode; calibration runs of %d steps "
                                           "(eager: %d); --launch auto times the faster of graph / ext" % (n_cal, n_eager))
    out["eager"] = {"ms_per_step": eager_dt * 1e3, "pairs_per_s": pairs_per_step / eager_dt,
                    "note": "same step issued through torch.autograd.Function calls, one Python call per operator"}
    if graph_note:
        out["config"]["launch_note"] = graph_note
    if grid:
        # forward = grid_build_kernel + grid_query_kernel (the dominant one, ~2/3 of the forward; stage A and
        # the wide stages) + the brute-force kernel over the unresolved list; timed together by the HIP events
        out["roofline"] = {"bound": "hbm", "kernel": "grid_build_kernel + grid_query_kernel + list fallback",
                           "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": hbm_gbs / HBM_PEAK_GBS,
                           # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, summed over the forward's kernels
                           # (profiles/r1/pmc_summary.txt: build 6.2 + 24.2 MB, search 11.6 + 8.0); most of it
                           # is the search structure itself (sorted clouds + cell tables,
                           # 25 MB, written with scattered 16-byte stores), not re-reads of the inputs
                           "traffic": 50.0e6 if (B, N, M) == (32, 16384, 16384) else None,
                           "note": "VALU-issue / L2-latency-bound search over a 42 MB workspace (25 MB of it touched per call), not HBM-bound; "
                                   "'bruteforce' carries the every-pair kernel and its VALU roofline"}
        if brute is not None:
            bg = alg_bytes_fwd / (brute["fwd_ms"] * 1e-3) / 1e9
            brute.update({
                "roofline": {"bound": "hbm", "kernel": "nmdist_fwd_c3_kernel", "achieved": bg,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bg / HBM_PEAK_GBS,
                             "traffic": 14.4e6 if (B, N, M) == (32, 16384, 16384) else None,
                             "note": "6500 flop/B: VALU-bound, see valu"},
                "valu": {"achieved": laneops / (brute["fwd_ms"] * 1e-3), "peak": VALU_PEAK_LANEOPS,
                         "unit": "lane-ops/s", "frac": laneops / (brute["fwd_ms"] * 1e-3) / VALU_PEAK_LANEOPS}})
            out["bruteforce"] = brute
    else:
        out["roofline"] = {"bound": "hbm", "kernel": "nmdist_fwd_c3_kernel", "achieved": hbm_gbs,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                           # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r1/pmc_summary.txt):
                           # 6.1 MB + 8.0 MB per launch; reads are scalar/dword loads (uncalibrated on
                           # gfx950) -- at most the algorithmic 20.97 MB either way: no re-reads beyond L2
                           "traffic": 14.4e6 if (B, N, M) == (32, 16384, 16384) else None,
                           "note": "exact brute force is fp32-VALU-bound (6500 flop/B); see 'valu'"}
        out["valu"] = {"achieved": laneops / (fwd_ms * 1e-3), "peak": VALU_PEAK_LANEOPS,
                       "unit": "lane-ops/s", "frac": laneops / (fwd_ms * 1e-3) / VALU_PEAK_LANEOPS,
                       "pairs_per_s_fwd": 2.0 * B * N * M / (fwd_ms * 1e-3)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU leg runs at N=1 only
        out["cpu_baseline"] = cpu_baseline_chamfer(N, C)
    return out


# ------------------------------------------------------------------------ fps + gather (config 3)
def bench_fps(args, dist, world, rank, device):
    from pytorch_points_amd import synthetic as S
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    B = args.batch or 16
    N = args.points or 65536
    npoint = 4096
    x = torch.from_numpy(S.unit_sphere(0, B, N)).to(device)

    def step():
        furthest_point_sample(x, npoint, NCHW=False, seedIdx=0)

    dt = timed_region(dist, step, args.steps, args.warmup, device)
    ms = dt / args.steps * 1e3
    updates = float(B) * (npoint - 1) * N * world
    alg_bytes = 12.0 * B * N + 4.0 * B * npoint
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"metric": "fps_point_updates_per_s", "value": updates / (dt / args.steps), "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "furthest_point_sample + gather_points B=%d N=%d npoint=%d" % (B, N, npoint),
                       "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "fps_cluster_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "note": "serial chain of npoint-1 dependent steps: latency-bound, not HBM-bound"},
            "us_per_pick": ms * 1e3 / (npoint - 1)}


# ------------------------------------------------------------- ball_query + group_points (config 4)
def bench_ball_group(args, dist, world, rank, device):
    from pytorch_points_amd import synthetic as S
    from pytorch_points_amd.network.operations import ball_query, grouping_operation
    B = args.batch or 32
    N = args.points or 16384
    C, ns, r = 128, 64, 0.1
    x = torch.from_numpy(S.unit_sphere(0, B, N)).to(device)
    centres = x[:, ::4].contiguous()                       # npoint = N/4 = 4096
    npoint = centres.shape[1]
    feats = torch.from_numpy(S.normal(2, (B, C, N))).to(device)
    ev = []

    from pytorch_points_amd._ext import sampling as _sampling
    grad_out = None

    def step():
        nonlocal grad_out
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        idx = ball_query(r, ns, x, centres)
        e[1].record()
        out = grouping_operation(feats, idx)
        e[2].record()
        if args.with_backward:
            _sampling.group_points_grad(out, idx, N)     # dL/dout := out (any dense gradient)
        e[3].record()
        ev.append(e)

    dt = timed_region(dist, step, args.steps, args.warmup, device)
    torch.cuda.synchronize()
    bq_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev[-args.steps:]]))
    gp_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev[-args.steps:]]))
    gpg_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev[-args.steps:]])) if args.with_backward else None
    ms = dt / args.steps * 1e3
    # the caller one level up (network/operations.py:162-213): fused (one output tensor) vs the
    # reference's composition (group, group, subtract, torch.cat)
    from pytorch_points_amd.network.operations import QueryAndGroup
    qg = QueryAndGroup(r, ns, use_xyz=True)

    def time_it(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b_.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b_) / n

    with torch.no_grad():
        qg_fused_ms = time_it(lambda: qg(x, centres, feats))
        qg_unfused_ms = time_it(lambda: qg.forward_unfused(x, centres, feats))
    gp_bytes = 4.0 * B * C * N + 4.0 * B * npoint * ns + 4.0 * B * C * npoint * ns
    gbs = gp_bytes / (gp_ms * 1e-3) / 1e9
    return {"metric": "group_points_output_bytes_per_s", "value": 4.0 * B * C * npoint * ns * world / (gp_ms * 1e-3),
            "unit": "B/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ball_query r=%.2f nsample=%d + group_points B=%d N=%d npoint=%d C=%d"
                                   % (r, ns, B, N, npoint, C), "parallelism": "batch-shard x%d" % world},
            "ball_query_ms": bq_ms, "group_points_ms": gp_ms, "group_points_grad_ms": gpg_ms,
            "query_and_group_fused_ms": qg_fused_ms, "query_and_group_composed_ms": qg_unfused_ms,
            "ball_query_pairs_per_s": float(B) * npoint * N / (bq_ms * 1e-3),
            "roofline": {"bound": "hbm", "kernel": "group_points_dma_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         # PMC passes (profiles/r1/pmc_summary.txt): WRITE_SIZE 4096 MB exact, FETCH_SIZE
                         # 144 MB x2 (16-B loads read 1/2 on gfx950) = 288 MB
                         "traffic": 4583e6 if (B, N, C, ns) == (32, 16384, 128, 64) else None}}


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dist, world, rank, local = init_dist(args)
    device = torch.device("cuda", local)
    fn = {"chamfer": bench_chamfer, "fps": bench_fps, "ball_group": bench_ball_group}[args.workload]
    out = fn(args, dist, world, rank, device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
