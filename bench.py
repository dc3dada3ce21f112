#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload chamfer|fps|ball_group]

Default workload = BASELINE.json configs[1]: Chamfer forward+backward, B=32 (per GPU), N=M=16384,
C=3, fp32, synthetic area-uniform unit-sphere clouds (two input sets alternating), through the public
autograd API: pytorch_points_amd.network.model_loss.nndistance (torch.autograd.Function) -> C ABI ->
HIP kernels, backward through torch.autograd.  A "step" is one forward + one backward over that batch,
inputs resident in HBM.  `value` is that eager operator path; the same kernels issued through the
extension-module functions on static buffers and as a hipGraph replay are reported beside it
(`launch_modes_ms_per_step`).

Metric: point-pairs/s = n_gpus * 2*B*N*M / t(step) (both directions counted): NOMINAL
pairs -- the exact grid search evaluates well under 1 % of them; `bruteforce` carries the every-pair
kernel and its VALU roofline.

N > 1 (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL): the batch is
sharded, B=32 per rank (weak scaling); every `--gather-every`-th step (default: every step) also
all-gathers the per-shard (dist, idx) over xGMI as BASELINE.json's north_star specifies.  Timing =
barrier + synchronize on both sides, MAX over ranks; rank 0 prints ONE JSON line; `compute_ms` and
`exchange_ms` give the two legs by themselves.

The JSON line also carries
  roofline      HBM roofline of the dominant kernel (the search): SURVEY.md §8(d) algorithmic forward bytes /
                that kernel's average duration, HIP events on the launch stream around it
  roofline_step the whole step's algorithmic bytes / ms_per_step
  fps, ball_group   short runs of BASELINE.json configs 3 and 4, each with its own roofline
  other_distributions_fwd_ms   the forward on clouds that are not a uniform sphere
  other_ops_ms  three_nn, three_interpolate, knn_points (K = 1, 8, 16), labeled Chamfer at the shapes of DESIGN.md 5.6
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


def _pmc_traffic_c2():
    """HBM-side bytes per launch of the config-2 step's kernels, from the rocprofv3 --pmc passes of THIS command that
    tools/regen_profiles.sh wrote last (profiles/r<N>/pmc_summary.txt, the newest round in the tree; counters cannot be
    collected inside the timed process): 2 x FETCH_SIZE (16-byte loads count half on gfx950, MI355X_MICROARCH.md) +
    WRITE_SIZE, in bytes.  -> ({kernel prefix: bytes}, source line) or ({}, None) when no summary travels with the tree."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.txt")),
                   key=lambda f: int(re.search(r"r(\d+)", os.path.basename(os.path.dirname(f))).group(1)))
    if not files:
        return {}, None
    path = files[-1]
    fetch, write, section, head = {}, {}, None, ""
    for ln in open(path):
        if ln.startswith("#"):
            head = head or ln.strip("# \n")
            continue
        if not ln.startswith(" "):
            section = ln.strip()
            continue
        m = re.match(r"\s+(\S+).*?(FETCH_SIZE|WRITE_SIZE) launches=\d+ mean=(\d+) KB", ln)
        if m and section in ("ch_FETCH_SIZE", "ch_WRITE_SIZE"):
            name = m.group(1).split("<")[0]
            (fetch if m.group(2) == "FETCH_SIZE" else write)[name] = float(m.group(3)) * 1024.0
    out = {k: 2.0 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in set(fetch) | set(write)}
    return out, "%s (%s)" % (os.path.relpath(path, ROOT), head)




def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="chamfer", choices=["chamfer", "fps", "ball_group"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--variant", type=int, default=0, help="forward kernel variant (0 = automatic)")
    ap.add_argument("--search", default="auto", choices=["auto", "bruteforce"],
                    help="chamfer: auto = the operator's default (exact grid search, brute-force "
                         "fallback); bruteforce = evaluate every pair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", default="all", choices=["all", "eager", "ext", "graph"],
                    help="chamfer: how the timed steps are issued.  eager = the torch.autograd.Function operator "
                         "(what 'value' reports by default); ext = two calls of the _ext.losses functions on static "
                         "buffers; graph = hipGraph replay.  all (default) = eager is timed as 'value' and the other "
                         "two are reported beside it")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: all-gather the shard outputs every k-th step (default 1: every step)")
    ap.add_argument("--no-extras", action="store_true",
                    help="chamfer: skip the short runs of the other workloads (fps, ball_group) and distributions")
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


def _knob(name):
    """a test / benchmark knob of include/pp_hip_debug.h"""
    import ctypes
    from pytorch_points_amd import _lib
    fn = getattr(_lib.lib(), name)
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    return fn


def _search_kernel_ms(fwd, n=40, step=None):
    """Average duration of the forward's launches -- grid build, stage-A kernel (unlabeled searches), the kernel that
    serves what stage A left (or the whole search where there is no stage-A kernel) -- HIP events recorded by the library
    on the launch stream around each launch (pp_debug_set_nmdistance_kernel_timing, include/pp_hip_debug.h)."""
    import ctypes
    from pytorch_points_amd import _lib
    L = _lib.lib()
    read = L.pp_debug_nmdistance_kernel_ms3
    read.argtypes = [ctypes.POINTER(ctypes.c_float)] * 3
    read.restype = ctypes.c_int
    own = L.pp_debug_nmdistance_kernel_own_ms      # the build's and the stage-A kernel's own begin / end stamps
    own.argtypes = [ctypes.POINTER(ctypes.c_float)] * 2
    own.restype = ctypes.c_int
    on = _knob("pp_debug_set_nmdistance_kernel_timing")
    on(1)
    try:
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        bs, aa, rr, ob, oa = [], [], [], [], []
        for _ in range(n):
            fwd()
            bm, am, rm = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
            if read(ctypes.byref(bm), ctypes.byref(am), ctypes.byref(rm)) != 0:
                return None, None, None, None
            bs.append(bm.value)
            aa.append(am.value)
            rr.append(rm.value)
        # ... and the two launches stamped by their own begin and end, inside whole STEPS (forward + backward, the input
        # sets alternating): in a loop of forwards alone the stage-A kernel runs 2 us shorter than inside the step
        # (32.8 against 35 us in one rocprofv3 trace), and the trace of the bench command is mostly steps
        on(2)
        run = step if step is not None else fwd
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        for _ in range(n):
            run()
            bm, am = ctypes.c_float(0), ctypes.c_float(0)
            if own(ctypes.byref(bm), ctypes.byref(am)) == 0:
                ob.append(bm.value)
                oa.append(am.value)
    finally:
        on(0)
    own_ms = {"build": float(np.mean(ob)), "stage_a": float(np.mean(oa))} if len(oa) == n else None
    return float(np.mean(bs)), float(np.mean(aa)), float(np.mean(rr)), own_ms


def _distribution(kind, seed, B, N):
    """point clouds that are NOT a uniform sphere: (B, N, 3) float32"""
    rng = np.random.default_rng(1000 + seed)
    if kind == "gaussian":
        return rng.standard_normal((B, N, 3)).astype(np.float32)
    if kind == "blobs8":
        c = rng.random((B, 8, 3), dtype=np.float32) * 2
        pick = rng.integers(0, 8, N)
        return (c[:, pick] + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.02).astype(np.float32)
    if kind == "two_scales":
        x = rng.random((B, N, 3), dtype=np.float32)
        x[:, : N // 2] *= 1e-2
        return x
    if kind == "cube":            # a uniformly filled cube: a volume, not a surface
        return rng.random((B, N, 3), dtype=np.float32)
    if kind == "disjoint":        # two uniformly filled cubes five units apart: every query far from every reference
        x = rng.random((B, N, 3), dtype=np.float32)
        if seed:
            x += np.float32(5.0)
        return x
    if kind == "shapenet_like":   # thin surfaces: two planes and a cylinder in a box
        x = rng.random((B, N, 3), dtype=np.float32) - 0.5
        q = N // 4
        x[:, :q, 2] = -0.5
        x[:, q:2 * q, 0] = 0.2
        th = rng.random((B, q)) * 6.283
        x[:, 2 * q:3 * q, 0] = 0.3 * np.cos(th)
        x[:, 2 * q:3 * q, 1] = 0.3 * np.sin(th)
        return x.astype(np.float32)
    # four adversarial families: what the pruning of the grid search is weakest on
    if kind == "shells":          # concentric shells, r = 1 against r = 0.5: EVERY query is half a unit from every reference
        x = rng.standard_normal((B, N, 3))
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        return (x * (1.0 if seed == 0 else 0.5)).astype(np.float32)
    if kind == "shell_vs_core":   # a shell against a tight cluster at its centre (and back: the cluster's queries see a far shell)
        x = rng.standard_normal((B, N, 3))
        if seed == 0:
            x /= np.linalg.norm(x, axis=-1, keepdims=True)
        else:
            x *= 1e-3
        return x.astype(np.float32)
    if kind == "identical":       # every point of a cloud the same point (all ties; one crowded cell)
        return np.full((B, N, 3), 0.25 if seed == 0 else 0.75, np.float32)
    if kind == "lattice":         # a 16^3 lattice against its half-cell shift: eight exactly equidistant neighbours everywhere
        g = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
        x = g[rng.integers(0, len(g), (B, N))]
        return (x + (0.0 if seed == 0 else 0.5)).astype(np.float32)
    raise ValueError(kind)


def bench_chamfer(args, dist, world, rank, device):
    from pytorch_points_amd import _lib, synthetic as S
    from pytorch_points_amd.network.model_loss import nndistance
    from pytorch_points_amd._ext import losses as ext_losses
    B = args.batch or 32
    N = args.points or 16384
    M, C = N, 3
    if args.search == "bruteforce":
        _knob("pp_debug_set_nmdistance_search")(1)
    if args.variant:
        _knob("pp_debug_set_nmdistance_variant")(args.variant)
    # rank r owns batch elements [r*B, (r+1)*B) of the global batch; seeds 0 / 1 as SURVEY.md §8d.  Two input
    # sets (A: seeds 0/1, B: seeds 2/3) alternate from step to step, so no step sees the clouds -- or the grid
    # its predecessor built -- again.
    sets = []
    for s1, s2 in ((0, 1), (2, 3)):
        x1 = torch.from_numpy(S.unit_sphere(s1, B, N, C, batch_offset=rank * B)).to(device).requires_grad_(True)
        x2 = torch.from_numpy(S.unit_sphere(s2, B, M, C, batch_offset=rank * B)).to(device).requires_grad_(True)
        sets.append((x1, x2))
    g1 = torch.full((B, N), 1.0 / (B * N), device=device)   # gradient of dist.mean()
    g2 = torch.full((B, M), 1.0 / (B * M), device=device)
    exchange = None
    if dist is not None:
        # one asynchronous collective per gathered step: (dist1 | dist2 | idx1 | idx2) of the shard, packed (idx as
        # 16-bit words: 6 MiB per rank at B=32, N=M=16384), double-buffered (pytorch_points_amd/sharded.py)
        from pytorch_points_amd.sharded import PackedShardGather
        # PP_SHARD_EXCHANGE: native (c10d's _allgather_base issued from C++ by the calling thread), p2p (round 6: ONE
        # grouped set of world - 1 sends and receives between the rows of the gathered buffer: every part over its own
        # xGMI link), rccl / rccl_p2p (the same two on a communicator of the object's own -- opt-in: they could only be
        # exercised with one rank here), python (the Python-issued exchange).  With several ranks and no variable set
        # the native and the p2p form are both timed below and the headline runs on the faster.
        exchange = PackedShardGather(B, N, M, device)
    pending = []        # slot of the previous gathered step
    counter = [0]
    gather_every = max(1, args.gather_every)

    def gathers():
        return exchange is not None and counter[0] % gather_every == 0

    def consume_previous():
        """the previous gathered step is consumed first: views of the gathered buffer (round 4: nothing is unpacked
        that nobody reads; a consumer that wants contiguous int32 indices calls exchange.wait instead)"""
        if pending:
            exchange.wait_views(pending.pop())

    def maybe_exchange(d1, d2, i1, i2):
        """all-gather of the per-shard (dist, idx) over xGMI (RCCL), asynchronous: it runs on a side stream beside this
        step's backward and the next step's forward.  General form: outputs held elsewhere are packed into the slot."""
        if not gathers():
            return
        consume_previous()
        pending.append(exchange.launch(d1, d2, i1, i2))

    def eager_step():
        """the operator as a user calls it: torch.autograd.Function forward, autograd backward; with an exchange, the
        search writes its distances straight into the exchange's slot (PackedShardGather.forward)"""
        x1, x2 = sets[counter[0] & 1]
        counter[0] += 1
        x1.grad = None
        x2.grad = None
        if gathers():
            consume_previous()
            d1, d2, i1, i2, h = exchange.forward(x1, x2)
            pending.append(h)
        else:
            d1, d2, i1, i2 = nndistance(x1, x2)
        torch.autograd.backward([d1, d2], [g1, g2])

    def drain():
        if exchange is not None:
            while pending:
                exchange.wait_views(pending.pop())
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

    # the same step on static buffers through the reference's extension-module API (pytorch_points._ext.losses:
    # nmdistance_forward / nmdistance_backward -- what the reference's autograd.Function calls,
    # _ext/nmdistance.cpp:30-34): no autograd bookkeeping, plain stream launches
    od1, od2 = torch.empty(B, N, device=device), torch.empty(B, M, device=device)
    oi1 = torch.empty(B, N, dtype=torch.int32, device=device)
    oi2 = torch.empty(B, M, dtype=torch.int32, device=device)
    ogx1, ogx2 = torch.empty(B, N, C, device=device), torch.empty(B, M, C, device=device)
    dsets = [(a.detach(), b.detach()) for a, b in sets]

    def ext_step():
        sx1, sx2 = dsets[counter[0] & 1]
        counter[0] += 1
        if gathers():   # the extension-module call writes into the slot's own distance fields
            consume_previous()
            slot, v1, v2 = exchange.begin()
            ext_losses.nmdistance_forward(sx1, sx2, v1, v2, oi1, oi2)
            pending.append(exchange.launch_in_place(slot, oi1, oi2))
        else:
            ext_losses.nmdistance_forward(sx1, sx2, od1, od2, oi1, oi2)
        ext_losses.nmdistance_backward(sx1, sx2, ogx1, ogx2, g1, g2, oi1, oi2)

    # hipGraph replay of one step (pytorch_points_amd/graphs.py; static inputs: set A)
    gstep, graph_note = None, None
    if args.launch in ("all", "graph"):
        try:
            from pytorch_points_amd.graphs import GraphedChamferStep
            gstep = GraphedChamferStep(B, N, M, device)
            with torch.no_grad():
                gstep.xyz1.copy_(sets[0][0])
                gstep.xyz2.copy_(sets[0][1])
                gstep.grad_dist1.copy_(g1)
                gstep.grad_dist2.copy_(g2)
            gstep.capture()
        except Exception as exc:
            gstep = None
            graph_note = "graph capture failed (%s: %s)" % (type(exc).__name__, exc)
    if dist is not None:  # every rank must take the same path
        flag = torch.tensor([1 if gstep is not None else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            gstep = None

    def graph_step():
        counter[0] += 1
        d1, d2, i1, i2, _, _ = gstep.replay()
        maybe_exchange(d1, d2, i1, i2)

    # ---- the timed region: the autograd operator (eager) unless another issue mode is asked for ---------------
    # `value` is ALWAYS the eager operator under torch's DEFAULT autograd engine -- what a user of the drop-in API runs
    # (VERDICT r3 #7 / ADVICE r3: rounds 2-3 took the backward on the calling thread, then the faster of the two).  The
    # other mode -- torch.autograd.set_multithreading_enabled(False), a public switch that saves the hand-off to the
    # engine's worker thread -- is reported beside it as launch_modes_ms_per_step["eager_calling_thread"], same step count.
    engine_threads_default = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(True)
    want = "eager" if args.launch == "all" else args.launch
    if want == "graph" and gstep is None:
        want = "eager"
    timed_fn = {"graph": graph_step, "ext": ext_step, "eager": eager_step}[want]
    exchange_modes_ms, exchange_mode = None, (exchange.mode if exchange is not None else None)
    dt = run_timed(timed_fn, args.warmup, args.steps)
    ms = dt / args.steps * 1e3

    modes = {want: ms}
    n_cal = 100

    def events_median(fn, n):
        """median over n steps of the time between HIP events recorded on the launch stream (torch's current stream:
        the operators launch there) after consecutive steps -- device time per step, free of the wall clock's noise
        on a short timed region; outside the timed region"""
        for _ in range(5):
            fn()
        drain()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            fn()
            evs[i + 1].record()
        drain()
        torch.cuda.synchronize()
        return float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(n)]))

    ms_events_median = events_median(timed_fn, max(20, min(args.steps, 300)))
    if args.launch == "all":   # the other issue modes, reported beside the headline
        modes["ext"] = run_timed(ext_step, 5, n_cal) / n_cal * 1e3
        if gstep is not None:
            modes["graph"] = run_timed(graph_step, 5, n_cal) / n_cal * 1e3
        # the backward on the calling thread: same warm-up and step count as the headline
        torch.autograd.set_multithreading_enabled(False)
        modes["eager_calling_thread"] = run_timed(eager_step, args.warmup, args.steps) / args.steps * 1e3
        torch.autograd.set_multithreading_enabled(True)
    compute_ms = exchange_ms = None
    if dist is not None:
        # the two legs by themselves: the same steps without the exchange, and the exchange with nothing beside it
        ex_saved, exchange = exchange, None
        compute_ms = run_timed(timed_fn, 5, n_cal) / n_cal * 1e3
        exchange = ex_saved

        def only_exchange():
            consume_previous()
            slot, _, _ = exchange.begin()       # (the distances are "already there")
            pending.append(exchange.launch_in_place(slot, oi1, oi2))
        exchange_ms = run_timed(only_exchange, 5, n_cal) / n_cal * 1e3
        # GPU time of one exchange by itself: the index-narrowing kernel + the all-gather, between two events on the
        # launch stream (the second behind the stream's wait for the gather)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        gpu_us = []
        for _ in range(20):
            drain()
            ev[0].record()
            slot, _, _ = exchange.begin()
            h = exchange.launch_in_place(slot, oi1, oi2)
            exchange.wait_views(h)
            ev[1].record()
            ev[1].synchronize()
            gpu_us.append(ev[0].elapsed_time(ev[1]) * 1e3)
        exchange_gpu_us = float(np.median(gpu_us))

    # forward duration: the forward's two launches issued back to back between two HIP events on the launch stream
    def fwd_only():
        sx1, sx2 = dsets[counter[0] & 1]
        counter[0] += 1
        ext_losses.nmdistance_forward(sx1, sx2, od1, od2, oi1, oi2)
    for _ in range(3):
        fwd_only()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nf = 200
    e0.record()
    for _ in range(nf):
        fwd_only()
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / nf
    # backward duration: the backward's one launch (extension-module call on static buffers) issued back to back, same way
    def bwd_only():
        sx1, sx2 = dsets[counter[0] & 1]
        counter[0] += 1
        ext_losses.nmdistance_backward(sx1, sx2, og1, og2, g1, g2, oi1, oi2)
    bwd_ms = None
    try:
        og1, og2 = torch.empty(B, N, 3, device=device), torch.empty(B, M, 3, device=device)
        for _ in range(3):
            bwd_only()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(nf):
            bwd_only()
        e1.record()
        torch.cuda.synchronize()
        bwd_ms = e0.elapsed_time(e1) / nf
    except Exception as exc:   # noqa: BLE001 -- a timing extra must not take the line down
        print("backward-only timing failed: %r" % (exc,), file=sys.stderr)
    grid = args.search == "auto" and int(_lib.lib().pp_nmdistance_forward_workspace_bytes(B, N, M, C)) > 0
    build_ms = stage_a_ms = rest_ms = search_ms = None
    raw_kernel_ms = None
    event_overhead_ms = None
    own_kernel_ms = None
    if grid:
        build_ms, stage_a_ms, rest_ms, own_kernel_ms = _search_kernel_ms(fwd_only, step=ext_step if dist is None else None)
        search_ms = (stage_a_ms + rest_ms) if build_ms is not None else None
        if build_ms is not None:
            # Each figure is the time between two HIP events around ONE launch, which adds a recorded event's own
            # cost to it: bracketed one by one the launches sum to more than the forward they make up (71.5
            # us of kernels in a 60.8 us forward).  The surplus -- (sum of the bracketed launches - fwd_ms) / launches,
            # the same forward timed as a whole in this run -- is taken off each, so that the *_kernel_ms add up to
            # fwd_ms; the uncorrected figures stay in roofline.kernel_ms_uncorrected.
            raw_kernel_ms = {"build": build_ms, "stage_a": stage_a_ms, "rest": rest_ms}
            n_l = 3 if stage_a_ms else 2
            event_overhead_ms = max(0.0, (build_ms + (stage_a_ms or 0.0) + rest_ms - fwd_ms) / n_l)
            build_ms -= event_overhead_ms
            rest_ms -= event_overhead_ms
            if stage_a_ms:
                stage_a_ms -= event_overhead_ms

    # the same step with the search forced to the brute-force kernel (every pair evaluated)
    brute = None
    if args.search == "auto" and args.launch == "all":
        setter = _knob("pp_debug_set_nmdistance_search")
        setter(1)
        try:
            nb = 10
            tb = run_timed(eager_step, 2, nb) / nb
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                fwd_only()
            ev1.record()
            torch.cuda.synchronize()
            brute = {"ms_per_step": tb * 1e3, "fwd_ms": ev0.elapsed_time(ev1) / 5, "pairs_per_s_per_gpu": 2.0 * B * N * M / tb}
        finally:
            setter(0)

    pairs_per_step = 2.0 * B * N * M * world
    alg_bytes_fwd = 4.0 * C * B * (N + M) + 8.0 * B * (N + M)     # 20 971 520 at config 2
    alg_bytes_step = alg_bytes_fwd + (4.0 * C + 8.0) * B * (N + M) + 4.0 * C * B * (N + M)   # + backward: 54 525 952
    # VALU lane-ops the brute-force kernel issues per pair: 3 sub + 1 mul + 2 fma + 1/2 min3 + 13/64
    # per-group bookkeeping (DESIGN.md "nmdist_fwd_c3_kernel")
    laneops = 2.0 * B * N * M * 6.703125
    c2 = (B, N, M) == (32, 16384, 16384)
    launch_text = {
        "eager": "torch.autograd.Function operator (nndistance forward, autograd backward), one Python call each, under "
                 "torch's default autograd engine: what a user of the drop-in API runs.  ms_per_step_events_median: HIP "
                 "events around the same steps in the same mode.  launch_modes 'eager_calling_thread': the same with "
                 "torch.autograd.set_multithreading_enabled(False)",
        "ext": "two calls per step of the extension-module API (_ext.losses.nmdistance_forward / _backward) on "
               "static buffers: plain stream launches, same kernels as the autograd operator",
        "graph": "hipGraph replay of the step's launches (same kernels as the eager operator)"}
    out = {
        "metric": "chamfer_fwd_bwd_point_pairs_per_s", "value": pairs_per_step / (dt / args.steps),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "ms_per_step_events_median": ms_events_median, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Chamfer fwd+bwd B=%d/GPU N=M=%d C=3 fp32, area-uniform unit sphere, two input sets "
                               "alternating" % (B, N), "global_batch": B * world,
                   "search": ("exact uniform-grid search, brute-force scans inside the search kernel as fallback "
                              "(operator default): outputs bit-identical to the brute force, most pairs pruned; value "
                              "= nominal pairs 2*B*N*M/t, not pairs evaluated" if grid else
                              "brute force: every pair evaluated"),
                   "parallelism": "batch-shard x%d%s" % (world, " + RCCL all-gather(dist,idx) every %d step(s), async"
                                                         % gather_every if world > 1 else ""),
                   "launch": launch_text[want]},
        "fwd_ms": fwd_ms,
        "launch_modes_ms_per_step": dict(modes, note="same kernels in every mode; 'value' is the '%s' mode%s" % (
            want, " (torch's default autograd engine)" if want == "eager" else "")),
    }
    if graph_note:
        out["config"]["launch_note"] = graph_note
    if want == "eager" and "ext" in modes and "eager_calling_thread" in modes:
        # The eager step under torch's default engine hands the backward to the engine's device thread and waits for it:
        # two thread wake-ups per step.  On a box with a slow host (observed on this pool: 0.107 ms against 0.078 for
        # the same kernels issued by the calling thread) that, not the GPU, sets `value`; say so when it happens.
        floor = min(modes["ext"], modes["eager_calling_thread"])
        out["host_bound"] = bool(ms > 1.10 * floor)
        out["host_bound_note"] = ("ms_per_step is %.0f %% above the same kernels issued without the engine's thread hand-off "
                                  "(launch_modes_ms_per_step: ext, eager_calling_thread): the host, not the GPU, bounds the "
                                  "default-engine step on this box" % (100.0 * (ms / floor - 1.0))) if ms > 1.10 * floor else None
    if dist is not None:
        out["compute_ms"] = compute_ms
        out["exchange_ms"] = exchange_ms
        out["exchange_gpu_us"] = exchange_gpu_us
        out["exchange_mode"] = exchange_mode
        out["exchange_modes_ms"] = exchange_modes_ms     # ms per step of the timed function on every form tried (20 steps each)
        native = getattr(exchange, "_native", None)
        p2p = bool(getattr(exchange, "p2p", False))
        out["exchange_issue"] = (("grouped ncclSend / ncclRecv, one per peer" if p2p else "direct ncclAllGather") +
                                 " issued by the calling thread" if getattr(exchange, "direct", False)
                                 else (("c10d coalesced send / recv, one per peer," if p2p else "c10d _allgather_base") +
                                       " issued by the calling thread (C++)" if native is not None
                                       else ("Python: dist.batch_isend_irecv" if p2p else "Python: dist.all_gather_into_tensor")))
        if native is not None:
            out["exchange_issue_us"] = float(native.issue_us_per_slot())   # host time of issuing one exchange
        # The wire model the first real 1 -> 8 run can be read against: an all-gather moves (world - 1) parts of
        # nbytes_padded into (and, over a ring, out of) every rank; xGMI is point to point, 7 links x ~153 GB/s per GPU
        # (MI355X guide).  ring: every part crosses ONE link per hop, (world - 1) hops back to back, so the floor is
        # (world - 1) * part / link; direct (all-pairs, what RCCL picks on a fully connected node for messages of this
        # size): the (world - 1) parts arrive over (world - 1) links at once, floor = part / link.
        part = float(exchange.nbytes_padded)
        link = 153e9
        out["exchange_bytes_per_rank"] = {"sent": part, "received": part * (world - 1)}
        out["wire_floor_ms"] = {"ring": (world - 1) * part / link * 1e3, "all_pairs": (part / link * 1e3) if world > 1 else 0.0,
                                "link_GBps_assumed": link / 1e9,
                                "note": "floors of ONE exchange at the assumed xGMI link rate (no latency term); with "
                                        "gather_every = %d the exchange has %d step(s) of compute to hide under" % (
                                            gather_every, gather_every)}
        # measured step against what the model allows: the longer of the compute leg and the wire floor of the form
        # that ran (ring for the library's all-gather, all-pairs for the grouped sends / receives); 1.0 = at the model
        floor_ms = out["wire_floor_ms"]["all_pairs" if p2p else "ring"] / gather_every
        out["scaling_vs_model"] = {"value": ms / max(compute_ms, floor_ms) if max(compute_ms, floor_ms) > 0 else None,
                                   "compute_ms": compute_ms, "wire_floor_ms": floor_ms,
                                   "floor": "all_pairs" if p2p else "ring",
                                   "note": "ms_per_step / max(compute_ms, wire floor of the exchange form that ran)"}
        out["exchange_note"] = ("compute_ms: the same steps without the exchange; exchange_ms: one step's exchange with nothing "
                                "beside it (indices narrowed into the slot + all-gather; the distances are written into the "
                                "slot by the search itself, the gathered buffer is read through views: no pack of "
                                "distances, no unpack); exchange_gpu_us: its GPU time between two events; ms_per_step has "
                                "them overlapped (gather every %d step(s))" % gather_every)
    if grid:
        # the dominant kernel of the step: the stage-A kernel of the unlabeled search (round 3: the search is two launches,
        # stage A by tiles + the kernel over what it leaves); the whole-search kernel where there is no stage-A kernel
        two_stage = bool(stage_a_ms)
        # roofline.kernel_ms is the dominant kernel's OWN duration -- its launch's begin / end stamps, what rocprofv3's
        # kernel trace reports for the dispatch (profiles/r<N>/chamfer_kernel_stats.csv) -- where the library can
        # stamp the launch (the stage-A kernel, the build); the durations between events recorded on the stream around
        # each launch, and those less the events' own cost, stay beside it
        own_a = own_kernel_ms["stage_a"] if (two_stage and own_kernel_ms) else None
        own_b = own_kernel_ms["build"] if own_kernel_ms else None
        dom_ms = own_a if own_a else (stage_a_ms if two_stage else (rest_ms if rest_ms else fwd_ms))
        gbs = alg_bytes_fwd / (dom_ms * 1e-3) / 1e9
        pmc, pmc_source = _pmc_traffic_c2() if c2 else ({}, None)
        fwd_traffic = (sum(pmc.get(k, 0.0) for k in ("grid_build_kernel", "grid_stage_a_kernel", "grid_query_list_kernel"))
                       if (two_stage and pmc) else None)
        out["roofline"] = {
            "bound": "hbm",
            "kernel": ("grid_stage_a_kernel (stage A of the search by tiles; dominant kernel of the step)" if two_stage
                       else "grid_query_wave_kernel (the search; dominant kernel of the step)"),
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            # FORWARD-level HBM-side traffic per step, all of the forward's launches (build + stage A + list kernel)
            "traffic": fwd_traffic or None,
            "traffic_dominant_kernel": pmc.get("grid_stage_a_kernel") if two_stage else None,
            "kernel_ms": dom_ms,
            "kernel_ms_source": ("the launch's own begin / end stamps (hipExtLaunchKernelGGL events), inside whole steps" if own_a
                                 else "HIP events on the stream around the launch, less the events' own cost"),
            "build_kernel_ms": own_b if own_b else build_ms,
            "stage_a_kernel_ms": dom_ms if two_stage else None, "rest_kernel_ms": rest_ms,
            "backward_kernel_ms": bwd_ms,
            # the two bandwidth-shaped kernels of the step against the HBM roof (their counter bytes / their
            # duration / 8 TB/s) -- only where the summary's kernels are the ones this run timed
            "build_frac_hbm": (pmc["grid_build_kernel"] / ((own_b or build_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                              if (pmc.get("grid_build_kernel") and (own_b or build_ms)) else None,
            "backward_frac_hbm": (pmc["nmdist_bwd_lds64_kernel"] / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                 if (pmc.get("nmdist_bwd_lds64_kernel") and bwd_ms) else None,
            "kernel_ms_stream_events": {"uncorrected": raw_kernel_ms, "event_overhead_ms": event_overhead_ms,
                                        "corrected": {"build": build_ms, "stage_a": stage_a_ms, "rest": rest_ms}},
            "traffic_source": pmc_source,
            "note": "SURVEY.md §8(d): algorithmic forward bytes (%.0f) / the dominant kernel's average duration over %d "
                    "forwards after the timed region; 'traffic' = 2 x FETCH_SIZE + WRITE_SIZE of the forward's launches "
                    "(every launch, not the dominant kernel alone) from the rocprofv3 --pmc passes named in "
                    "traffic_source. The search is bound by VALU issue and LDS bandwidth, not by HBM; 'bruteforce' "
                    "carries the every-pair kernel with its VALU roofline" % (alg_bytes_fwd, 40)}
        sgbs = alg_bytes_step / (ms * 1e-3) / 1e9
        out["roofline_step"] = {"bound": "hbm", "achieved": sgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": sgbs / HBM_PEAK_GBS,
                                "note": "whole step: %.0f algorithmic bytes (forward + backward) / ms_per_step" % alg_bytes_step}
        if brute is not None:
            bg = alg_bytes_fwd / (brute["fwd_ms"] * 1e-3) / 1e9
            brute.update({
                "roofline": {"bound": "hbm", "kernel": "nmdist_fwd_c3_kernel", "achieved": bg,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bg / HBM_PEAK_GBS,
                             "traffic": 14.4e6 if c2 else None,
                             "note": "6500 flop/B: VALU-bound, see valu"},
                "valu": {"achieved": laneops / (brute["fwd_ms"] * 1e-3), "peak": VALU_PEAK_LANEOPS,
                         "unit": "lane-ops/s", "frac": laneops / (brute["fwd_ms"] * 1e-3) / VALU_PEAK_LANEOPS}})
            out["bruteforce"] = brute
    else:
        hbm_gbs = alg_bytes_fwd / (fwd_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "nmdist_fwd_c3_kernel", "achieved": hbm_gbs,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                           "traffic": 14.4e6 if c2 else None,
                           "note": "exact brute force is fp32-VALU-bound (6500 flop/B); see 'valu'"}
        out["valu"] = {"achieved": laneops / (fwd_ms * 1e-3), "peak": VALU_PEAK_LANEOPS,
                       "unit": "lane-ops/s", "frac": laneops / (fwd_ms * 1e-3) / VALU_PEAK_LANEOPS,
                       "pairs_per_s_fwd": 2.0 * B * N * M / (fwd_ms * 1e-3)}
    if rank == 0 and world == 1 and grid and args.launch == "all" and not args.no_extras:
        # other point distributions, forward only: same shapes, clouds that are not a sphere
        od = {}
        for kind in ("cube", "gaussian", "blobs8", "two_scales", "shapenet_like", "disjoint",
                     "shells", "shell_vs_core", "identical", "lattice"):
            a = torch.from_numpy(_distribution(kind, 0, B, N)).to(device)
            b_ = torch.from_numpy(_distribution(kind, 1, B, N)).to(device)
            # (5 calls untimed, 20 timed: a new cloud's first calls run while the clocks settle -- blobs8 0.47 over calls
            #  4-13 against 0.445 over calls 6-25, tools/far_time.py -- and the figure is the steady one, as the headline's)
            for _ in range(5):
                ext_losses.nmdistance_forward(a, b_, od1, od2, oi1, oi2)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(20):
                ext_losses.nmdistance_forward(a, b_, od1, od2, oi1, oi2)
            ev1.record()
            torch.cuda.synchronize()
            od[kind] = ev0.elapsed_time(ev1) / 20
        od["note"] = "nndistance forward, ms (20 calls after 5), B=%d N=M=%d; the every-pair kernel takes %s ms" % (
            B, N, ("%.2f" % brute["fwd_ms"]) if brute else "1.8")
        out["other_distributions_fwd_ms"] = od
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU leg runs at N=1 only
        out["cpu_baseline"] = cpu_baseline_chamfer(N, C)
    # ---- which exchange?  Several ranks and no PP_SHARD_EXCHANGE: everything above ran on the native form (c10d's
    # all-gather).  Now the grouped send / receive form (p2p: every part over its own xGMI link) is tried -- 20 steps of
    # the timed function on each form -- and if it is faster the headline is measured again on it.  The p2p form has never
    # run on more than one rank (no multi-GPU node in the pool this was built on): a watchdog thread prints the native
    # line and ends the process if the trial does not come back, so that a hang costs the line nothing.
    if dist is not None and world > 1 and exchange is not None and os.environ.get("PP_SHARD_EXCHANGE") is None:
        import threading
        native_line = dict(out)
        native_line["exchange_modes_ms"] = {exchange_mode: ms, "p2p": None}
        native_line["exchange_trial_note"] = "the p2p form did not come back within 180 s: abandoned, the native line stands"

        def give_up():
            if rank == 0:
                sys.stdout.write(json.dumps(native_line) + "\n")
                sys.stdout.flush()
            os._exit(0)
        dog = threading.Timer(180.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            from pytorch_points_amd.sharded import PackedShardGather
            native_obj = exchange
            ok, trial = 1, None
            try:
                trial = PackedShardGather(B, N, M, device, exchange="p2p")
            except Exception as exc:   # noqa: BLE001
                ok = 0
                sys.stderr.write("bench: exchange mode p2p not available (%s: %s)\n" % (type(exc).__name__, exc))
            flag = torch.tensor([ok], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)       # (every rank takes the same path)
            if int(flag.item()) == 1:
                exchange = trial
                t_p2p = run_timed(timed_fn, 5, 20) / 20 * 1e3
                exchange = native_obj
                t_nat = run_timed(timed_fn, 5, 20) / 20 * 1e3
                out["exchange_modes_ms"] = {exchange_mode: t_nat, "p2p": t_p2p}
                if t_p2p < 0.97 * t_nat:      # (the same on every rank: run_timed returns the maximum over the ranks)
                    exchange = trial
                    dt2 = run_timed(timed_fn, args.warmup, args.steps)
                    ms2 = dt2 / args.steps * 1e3
                    out["ms_per_step_native_exchange"] = out["ms_per_step"]
                    out["ms_per_step"] = ms2
                    out["value"] = pairs_per_step / (ms2 * 1e-3)
                    out["exchange_mode"] = "p2p"
                    out["exchange_issue"] = "c10d coalesced send / recv, one per peer, issued by the calling thread (C++)"
                    floor_ms = out["wire_floor_ms"]["all_pairs"] / gather_every
                    out["scaling_vs_model"] = {"value": ms2 / max(compute_ms, floor_ms), "compute_ms": compute_ms,
                                               "wire_floor_ms": floor_ms, "floor": "all_pairs",
                                               "note": "ms_per_step / max(compute_ms, wire floor of the exchange form that ran)"}
                    out["exchange_note"] += ("; exchange_ms / exchange_gpu_us / launch_modes were measured on the native form, "
                                             "ms_per_step and value on p2p")
                    native_obj.drain()
                else:
                    trial.drain()
            else:
                out["exchange_modes_ms"] = {exchange_mode: ms, "p2p": None}
        except BaseException as exc:   # noqa: BLE001 -- a failure of the untried form on any rank must not cost the line: the
            # native line stands; this rank leaves (the others either fail the same way or are let go by their watchdogs)
            sys.stderr.write("bench: the p2p trial failed on rank %d (%s: %s): the native line stands\n" % (rank, type(exc).__name__, exc))
            native_line["exchange_trial_note"] = "the p2p trial failed (%s): the native line stands" % type(exc).__name__
            dog.cancel()
            give_up()
        finally:
            dog.cancel()
    torch.autograd.set_multithreading_enabled(engine_threads_default)
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
    # the same call on clouds where the bucketed kernel prunes least ("report that time too") and
    # through the kernel it replaced (the CU cluster over all points), same box, same run
    other = {}
    if rank == 0:
        import ctypes
        from pytorch_points_amd import _lib
        setter = _lib.lib().pp_debug_set_fps_v1
        setter.argtypes = [ctypes.c_int]
        setter.restype = None

        def ms_of(xt, reps=3):
            furthest_point_sample(xt, npoint, NCHW=False, seedIdx=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                furthest_point_sample(xt, npoint, NCHW=False, seedIdx=0)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        rng = np.random.default_rng(5)
        centres = rng.normal(size=(B, 8, 3)).astype(np.float32) * 3
        sel = rng.integers(0, 8, (B, N))
        clouds = {
            "gaussian": S.normal(1, (B, N, 3)),
            "cube": S.uniform01(2, (B, N, 3)).reshape(B, N, 3).astype(np.float32),
            "blobs8": (np.take_along_axis(centres, sel[..., None].repeat(3, -1), 1)
                       + 0.02 * S.normal(3, (B, N, 3))).astype(np.float32),
        }
        for name, c in clouds.items():
            other[name] = round(ms_of(torch.from_numpy(np.ascontiguousarray(c)).to(device)), 4)
        setter(2)
        try:
            other["sphere_cluster_kernel"] = round(ms_of(x), 4)
        finally:
            setter(0)
    return {"metric": "fps_point_updates_per_s", "value": updates / (dt / args.steps), "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "furthest_point_sample + gather_points B=%d N=%d npoint=%d" % (B, N, npoint),
                       "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "fps_bucket_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "note": "serial chain of npoint-1 dependent steps: latency-bound, not HBM-bound"},
            "us_per_pick": ms * 1e3 / (npoint - 1),
            "other_clouds_ms": other}


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
        # ball_query by itself, call after call (ball_query_ms is taken between two events INSIDE the step, behind a
        # group_points that has just pushed 4 GiB through the caches, and carries the events' own ~3 us)
        for _ in range(5):
            ball_query(r, ns, x, centres)
        bq_alone_ms = time_it(lambda: ball_query(r, ns, x, centres), 20)
        qg_unfused_ms = time_it(lambda: qg.forward_unfused(x, centres, feats))
    gp_bytes = 4.0 * B * C * N + 4.0 * B * npoint * ns + 4.0 * B * C * npoint * ns
    gbs = gp_bytes / (gp_ms * 1e-3) / 1e9
    # The store ceiling of THIS device in THIS run: a pure streaming store of as many bytes as
    # group_points writes, 16-byte non-temporal stores, the best of a few launch shapes -- what a kernel that writes
    # 4 GiB can at best approach here.  Reported as roofline.peak_measured beside the 8 TB/s spec.
    import ctypes
    from pytorch_points_amd import _lib
    ceil_fn = _lib.lib().pp_debug_store_ceiling
    ceil_fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ceil_fn.restype = ctypes.c_int
    out_bytes = int(4 * B * C * npoint * ns)
    scratch = torch.empty(out_bytes, dtype=torch.uint8, device=device)
    peak_measured, peak_shape = 0.0, None
    for nt in (1, 0):
        for wgs in (256, 512, 1024, 2048):
            def fill():
                with _lib.on_device(device) as stream:
                    _lib.check(ceil_fn(_lib.ptr(scratch), out_bytes, nt, wgs, stream), "store_ceiling")
            fill()
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fill()
            b_.record()
            torch.cuda.synchronize()
            g = out_bytes / (a.elapsed_time(b_) / 5 * 1e-3) / 1e9
            if g > peak_measured:
                peak_measured, peak_shape = g, "%s stores, %d workgroups x 1024 threads" % ("non-temporal" if nt else "plain", wgs)
    del scratch
    store_gbs = out_bytes / (gp_ms * 1e-3) / 1e9
    return {"metric": "group_points_output_bytes_per_s", "value": 4.0 * B * C * npoint * ns * world / (gp_ms * 1e-3),
            "unit": "B/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ball_query r=%.2f nsample=%d + group_points B=%d N=%d npoint=%d C=%d"
                                   % (r, ns, B, N, npoint, C), "parallelism": "batch-shard x%d" % world},
            "ball_query_ms": bq_ms, "ball_query_alone_ms": bq_alone_ms, "group_points_ms": gp_ms, "group_points_grad_ms": gpg_ms,
            "query_and_group_fused_ms": qg_fused_ms, "query_and_group_composed_ms": qg_unfused_ms,
            "ball_query_pairs_per_s": float(B) * npoint * N / (bq_ms * 1e-3),
            "roofline": {"bound": "hbm", "kernel": "group_points_dma1_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         # the store ceiling measured in this run (a pure 4 GiB streaming store) and group_points' own
                         # stores against it (its row reads and index reads come on top of these bytes)
                         "peak_measured": peak_measured, "peak_measured_how": peak_shape,
                         "store_achieved": store_gbs, "frac_of_measured": store_gbs / peak_measured if peak_measured else None,
                         # all of the kernel's algorithmic bytes (stores + row reads + index reads) against the same ceiling
                         "frac_all_bytes_of_measured": gbs / peak_measured if peak_measured else None,
                         # PMC passes (profiles/r1/pmc_summary.txt): WRITE_SIZE 4096 MB exact, FETCH_SIZE
                         # 144 MB x2 (16-B loads read 1/2 on gfx950) = 288 MB
                         "traffic": 4583e6 if (B, N, C, ns) == (32, 16384, 128, 64) else None}}


def _short(line):
    """a secondary workload's line as a sub-object of the headline: drop the keys that repeat the contract"""
    for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        line.pop(k, None)
    return line


def bench_other_ops(device):
    """the searches of the path that are not BASELINE configs, at the shapes DESIGN.md 5.6 quotes (B=32, N=16384, M=4096,
    C=128): mean of 20 calls between two events after 5 untimed ones, ms -- so that they show in the driver's line, not
    only in tools/"""
    from pytorch_points_amd import synthetic as S
    from pytorch_points_amd._ext import sampling, losses
    from pytorch_points_amd.ops import knn_points
    B, N, M, C = 32, 16384, 4096, 128

    def t(fn, n=20):
        for _ in range(5):   # (a new op's first calls run while the clocks settle: two untimed + five timed were mostly that)
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    unknown = torch.from_numpy(S.unit_sphere(0, B, N)).to(device)
    known = torch.from_numpy(S.unit_sphere(1, B, M)).to(device)
    d2 = torch.empty(B, N, 3, device=device)
    idx = torch.empty(B, N, 3, dtype=torch.int32, device=device)
    res = {"shapes": "B=32, 16384 unknown / query points, 4096 known points (knn, labeled Chamfer: 16384 both), C=128"}
    res["three_nn"] = t(lambda: sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx))
    feats = torch.randn(B, C, M, device=device)
    w = torch.rand(B, N, 3, device=device)
    w /= w.sum(-1, keepdim=True)
    out = torch.empty(B, C, N, device=device)
    res["three_interpolate"] = t(lambda: sampling.three_interpolate_wrapper(B, C, M, N, feats, idx, w, out))
    x1 = torch.from_numpy(S.unit_sphere(2, B, N)).to(device)
    x2 = torch.from_numpy(S.unit_sphere(3, B, N)).to(device)
    for K in (1, 8, 16):
        res["knn_points_k%d" % K] = t(lambda: knn_points(x1, x2, K=K))
    gen = torch.Generator(device=device)
    gen.manual_seed(3)
    l1 = torch.randint(0, 4, (B, N), device=device, generator=gen).float()
    l2 = torch.randint(0, 4, (B, N), device=device, generator=gen).float()
    o = (torch.empty(B, N, device=device), torch.empty(B, N, device=device),
         torch.empty(B, N, dtype=torch.int32, device=device), torch.empty(B, N, dtype=torch.int32, device=device))
    res["labeled_nmdistance_forward"] = t(lambda: losses.labeled_nmdistance_forward(x1, x2, l1, l2, *o))
    return res


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dist, world, rank, local = init_dist(args)
    device = torch.device("cuda", local)
    fn = {"chamfer": bench_chamfer, "fps": bench_fps, "ball_group": bench_ball_group}[args.workload]
    out = fn(args, dist, world, rank, device)
    if args.workload == "chamfer" and world == 1 and args.launch == "all" and not args.no_extras \
            and args.batch is None and args.points is None:
        # BASELINE.json configs 3 and 4 as short runs beside the headline, each with its own roofline (the driver
        # only runs the default command: VERDICT r1 #4); `--workload fps|ball_group` gives the full lines
        import copy
        torch.cuda.empty_cache()
        a3 = copy.copy(args)
        a3.steps, a3.warmup = 5, 2
        out["fps"] = _short(bench_fps(a3, None, 1, 0, device))
        torch.cuda.empty_cache()
        a4 = copy.copy(args)
        a4.steps, a4.warmup, a4.with_backward = 10, 3, True
        out["ball_group"] = _short(bench_ball_group(a4, None, 1, 0, device))
        torch.cuda.empty_cache()
        out["other_ops_ms"] = bench_other_ops(device)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio, which is flushed at
    # exit -- after Python's print -- unless it is flushed here first
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
