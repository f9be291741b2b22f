#!/usr/bin/env python3
"""Benchmark of the datapath evaluator on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one pass of the hot path over one resident batch of synthetic
operands.  Headline workload (N = 1): 2^24 random ray-vs-4-AABB box ops, SoA
fp32 operands already in HBM.  With N > 1 every rank runs the same-size batch on
its own GPU (independent op streams, no collective on the data path: weak
scaling) and rank 0 prints the whole-job rate.  `--gpus N` from a plain shell
starts the N ranks itself; under torchrun the ranks are taken as given.

One JSON line is printed by rank 0.  Besides the contract fields it carries
  roofline      -- the dominant kernel against the HBM roof (algorithmic bytes)
  cpu_baseline  -- the scalar C golden model (oracle B) timed on the host cores
  per_opcode    -- the other three opcodes and a mixed batch (2^24 units per GPU, weak), the box and triangle
                   streams with >= 1 % of every edge class, the 2^28-unit mixed stream cut into one pair-aligned
                   shard per GPU (strong scaling) and the k-NN configuration (strong scaling, one all-gather),
                   each with its own roofline / VALU block

Other workloads, one JSON line each:
  --only tri|euclid|angular|mixed     one opcode (or the mixed batch) as the headline
  --only mixed --strong               the 2^--strong-log2-ops mixed stream sharded over the GPUs as the headline
  --only stream [--stream-mode 1|2]   record-at-a-time streams resident in HBM
  --only knn [--knn-tree 0|1|2] [--knn-metric angular] [--knn-data clustered]
                                      exact top-16 search, 2^20 points x 2^16 queries (strong scaling over GPUs)
  --only trace [--trace-wavefront 1]  closest-hit traversal of a 4-wide hierarchy
  --sweep                             A/B of the kernel shapes
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks():
    """`python bench.py --gpus N` from a plain shell (no torchrun environment): start the N ranks as a CHILD
    `python -m torch.distributed.run` and relay its output and exit code.  This runs before torch or the HIP
    library is imported, so the parent never touches a GPU; it is a subprocess, never an exec."""
    import socket
    import subprocess
    n = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ or "-h" in argv or "--help" in argv:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd, env=dict(os.environ, RT_DP_BENCH_SPAWNED="1")))


if __name__ == "__main__":
    spawn_ranks()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import capi, dist_util, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak
# algorithmic bytes per unit of work: operands in + results out
BYTES = {"box": 156, "tri": 76, "euclid": 136, "angular": 76}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--log2-ops", type=int, default=24, help="ops per GPU of the headline box batch")
    p.add_argument("--ops", type=int, default=0, help="ops per GPU when not a power of two (overrides --log2-ops)")
    p.add_argument("--stagger-kib", type=int, default=0, help="box workload: operand arrays live in one allocation, array k shifted by k x this many KiB "
                                                              "beyond its natural place, so that the arrays do not cross page boundaries together")
    p.add_argument("--launch-ops", type=int, default=0, help="box workload, diagnostic: launch only the first this-many ops of the resident arrays "
                                                              "(the arrays keep their size and spacing)")
    p.add_argument("--ops-per-lane", type=int, default=0)
    p.add_argument("--nontemporal", type=int, default=0)
    p.add_argument("--vec-tile", type=int, default=0)
    p.add_argument("--variant", type=int, default=0)
    p.add_argument("--vec-passes", type=int, default=0)
    p.add_argument("--wgs-per-cu", type=int, default=0, help="resident workgroups per CU of the op kernels: 0 library default, 3..16, 99 no cap")
    p.add_argument("--only", default="", help="restrict to one workload: box|tri|euclid|angular|mixed|box_edge|tri_edge|stream|knn|trace")
    # tuning: 0 = library default; nontemporal 1 = on, 2 = off
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=10.0)
    p.add_argument("--knn-log2-points", type=int, default=20)
    p.add_argument("--knn-log2-queries", type=int, default=16)
    p.add_argument("--stream-mode", type=int, default=0, help="record streams: 0 automatic, 1 partition by opcode first, 2 evaluate in place then fold the beats, 3 one pass")
    p.add_argument("--trace-wavefront", type=int, default=0, help="1: breadth-first walk with queue compaction; 0: depth-first walk")
    p.add_argument("--trace-sorted", type=int, default=0, help="1: rays sorted by origin / target cell (coherent waves)")
    p.add_argument("--trace-build", default="host", help="host: median-split builder on the CPU; device: rt_dp_bvh4_build_device (Morton order) on the GPU")
    p.add_argument("--trace-log2-tris", type=int, default=18)
    p.add_argument("--trace-log2-rays", type=int, default=22)
    p.add_argument("--knn-tree", type=int, default=2, help="1: walk the k-d tree index with box pruning; 2: leaf-major search on the same index; 0: exhaustive scan")
    p.add_argument("--knn-walk", type=int, default=128, help="leaf-major search: leaves walked per query for the first bound")
    p.add_argument("--knn-metric", default="euclid", help="euclid, or angular (ranked by dot / sqrt(norm); --knn-tree 0 exhaustive, 1 / 2 the tree walk with score bounds)")
    p.add_argument("--knn-data", default="uniform", help="uniform: U(-1,1)^16 (the named configuration); clustered: 256 Gaussian clusters, sigma 0.05")
    p.add_argument("--knn-leaf", type=int, default=128, help="points per tree leaf")
    p.add_argument("--knn-build", default="torch", help="torch: index built by torch sort / gather per level; device: rt_dp_knn_tree_build_device")
    p.add_argument("--knn-splits", type=int, default=0, help="point splits per query block (0 = chosen by the library front-end)")
    p.add_argument("--derive", type=int, default=0, help="1: box / tri workloads derive inv / shear / k from dir inside the kernels "
                   "(RT_DP_F_DERIVE_RAY: 144 and 72 algorithmic bytes per op); not the headline format")
    p.add_argument("--sweep", action="store_true", help="A/B the kernel shapes instead of the timed run")
    p.add_argument("--strong", action="store_true", help="with --only mixed: one 2^--strong-log2-ops stream cut into one shard per GPU (strong scaling)")
    p.add_argument("--strong-log2-ops", type=int, default=28, help="units of the sharded mixed stream (whole job)")
    p.add_argument("--edge-per-class", type=float, default=0.01, help="fraction of the ops of the edge streams in each edge class")
    p.add_argument("--no-extras", action="store_true", help="default line without the edge streams, the sharded 2^28 stream and k-NN")
    p.add_argument("--layout", default="tiled", choices=["plain", "tiled"],
                   help="box / triangle segments as one tiled buffer per segment ([tile][word][8192 ops], RT_DP_F_BOX_TILED / "
                        "RT_DP_F_TRI_TILED: the layout the library recommends) or as one array per operand (plain); the default "
                        "line reports both")
    p.add_argument("--ballast-gib", type=float, default=0.0, help="diagnostic: hold this much other device memory, allocated before the workload's arrays")
    p.add_argument("--verify", action="store_true", help="sharded mixed stream: also evaluate the whole stream on rank 0 and compare it with the "
                                                         "shards' results put together (small sizes only: results travel through host memory)")
    a = p.parse_args()
    if a.stagger_kib or a.sweep or a.launch_ops:
        a.layout = "plain"      # array spacing, ops per lane and partial launches are properties of the plain layout only
    return a


def tiled_box(ctx, batch, fields, n, dev):
    """Puts the box segment of `batch` into tiled form: the operands are packed from `fields` (which the caller may free)."""
    operands = torch.empty(capi.tiled_words(n, capi.BOX_TILE_IN_WORDS), dtype=torch.float32, device=dev)
    results = torch.empty(capi.tiled_words(n, capi.BOX_TILE_OUT_WORDS), dtype=torch.int32, device=dev)
    src = [f if f is not None else fields[0] for f in fields]     # derive mode: the inv words are never read
    torch.cuda.synchronize()                                      # the fields were produced on torch's stream
    ctx.tile_pack(src, operands, n)
    ctx.synchronize()
    batch.set_box_tiled(operands, results, n)
    batch.box_tiled = (operands, results)


def tiled_tri(ctx, batch, fields, k, n, dev):
    operands = torch.empty(capi.tiled_words(n, capi.TRI_TILE_IN_WORDS), dtype=torch.float32, device=dev)
    results = torch.empty(capi.tiled_words(n, capi.TRI_TILE_OUT_WORDS), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.tile_pack(list(fields) + [k if k is not None else fields[0]], operands, n)
    ctx.synchronize()
    batch.set_tri_tiled(operands, results, n)
    batch.tri_tiled = (operands, results)


def make_batches(n, dev, which, stagger_kib=0, derive=False, launch_ops=0, edge=0.0, layout="plain", ctx=None):
    """Resident operand / result arrays and one capi.Batch per workload."""
    out = {}
    i32, f32 = torch.int32, torch.float32
    tiled = layout == "tiled"

    def box(n_):
        fields = synth.box_stream(n_, 20241008, dev, edge)
        tm = torch.empty(4, n_, dtype=f32, device=dev)
        tmin = [tm[0], tm[1], tm[2], tm[3]]
        flags = torch.empty(n_, dtype=i32, device=dev)
        if stagger_kib:
            pitch = n_ + (0 if stagger_kib < 0 else stagger_kib * 256)  # floats; a negative value: one pool, natural spacing
            pool = torch.empty(39 * pitch + 64, dtype=f32, device=dev)
            def place(k, t):
                v = pool[k * pitch:k * pitch + n_]
                v.copy_(t.view(f32) if t.dtype != f32 else t)
                return v
            fields = [place(k, f) for k, f in enumerate(fields)]
            tmin = [pool[(34 + k) * pitch:(34 + k) * pitch + n_] for k in range(4)]
            flags = pool[38 * pitch:38 * pitch + n_].view(i32)
        return fields, tmin, flags

    def tri(n_):
        fields, k = synth.tri_stream(n_, 20241009, dev, edge)
        return fields, k, torch.empty(n_, dtype=f32, device=dev), torch.empty(n_, dtype=f32, device=dev), torch.empty(n_, dtype=i32, device=dev)

    def euc(n_):
        a, b, ctl = synth.vec_stream(n_, 16, 20241010, dev)
        return a, b, ctl, torch.empty(ctl.shape[0], dtype=f32, device=dev)

    def ang(n_):
        q, c, ctl = synth.vec_stream(n_, 8, 20241011, dev)
        return q, c, ctl, torch.empty(ctl.shape[0], dtype=f32, device=dev), torch.empty(ctl.shape[0], dtype=f32, device=dev)

    if "box" in which:
        b = capi.Batch().derive_ray(derive)
        fields, tmin, flags = box(n)
        if tiled:
            del tmin, flags
            tiled_box(ctx, b, fields, launch_ops or n, dev)
        else:
            b.set_box(fields, tmin, flags, launch_ops or None)
        b.box_fields = fields
        out["box"] = (b.finalize(), {"box": launch_ops or n})
    if "tri" in which:
        b = capi.Batch().derive_ray(derive)   # derive: the three shear arrays stand in for the ray direction
        t = tri(n)
        if tiled:
            tiled_tri(ctx, b, t[0], t[1], n, dev)
        else:
            b.set_tri(*t)
        b.tri_fields, b.tri_k = t[0], t[1]
        out["tri"] = (b.finalize(), {"tri": n})
    if "euclid" in which:
        b = capi.Batch()
        e = euc(n)
        b.set_euclid(*e)
        out["euclid"] = (b.finalize(), {"euclid": int(e[2].shape[0])})
    if "angular" in which:
        b = capi.Batch()
        a = ang(n)
        b.set_angular(*a)
        out["angular"] = (b.finalize(), {"angular": int(a[2].shape[0])})
    if "mixed" in which:
        # opcode ~ U{0..3}: a quarter of the stream per opcode, opcode-sorted segments
        q = n // 4
        b = capi.Batch().derive_ray(derive)
        if tiled:
            tiled_box(ctx, b, box(q)[0], q, dev)
            t = tri(q)
            tiled_tri(ctx, b, t[0], t[1], q, dev)
            del t
        else:
            b.set_box(*box(q))
            b.set_tri(*tri(q))
        e, a = euc(q), ang(q)
        b.set_euclid(*e)
        b.set_angular(*a)
        out["mixed"] = (b.finalize(), {"box": q, "tri": q, "euclid": int(e[2].shape[0]), "angular": int(a[2].shape[0])})
    return out


_STREAM = None


def bench_stream():
    """A dedicated HIP stream: kernels are launched on it and the timing events are
    recorded on it (a NULL handle would select the context's private stream)."""
    global _STREAM
    if _STREAM is None:
        _STREAM = torch.cuda.Stream()
    return _STREAM


def sweep(ctx, batches, which, rounds=7, launches=10):
    """Interleaved A/B of kernel shapes in one process: median / min device ms per launch."""
    stream = bench_stream()
    def V(ops_per_lane=0, nontemporal=1, vec_tile=0, variant=0, vec_passes=0):
        return (ops_per_lane, nontemporal, vec_tile, variant, vec_passes)

    ops = [V(v, nt) for v in (1, 2, 4) for nt in (1, 2)]
    if os.environ.get("RT_DP_DIAG"):
        ops = [V(2, 1), V(2, 1, 0, 10)]
    variants = {"box": ops, "tri": ops,
                "euclid": [V(vec_tile=t, vec_passes=p) for p in (2, 4, 8) for t in (64 * p, 128 * p)],
                "angular": [V(vec_tile=t, vec_passes=p) for p in (2, 4, 8) for t in (128 * p, 256 * p)]}
    rows = []
    for name in which:
        if name not in variants:
            continue
        batch, counts = batches[name]
        algo = sum(BYTES[k] * v for k, v in counts.items())
        times = {v: [] for v in variants[name]}
        for _ in range(rounds):
            for v in variants[name]:
                ctx.set_tuning(*v)
                ctx.launch(batch, stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _i in range(launches):
                    ctx.launch(batch, stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
                times[v].append(e0.elapsed_time(e1) / launches)
        for v, ts in times.items():
            ts = sorted(ts)
            med, mn = ts[len(ts) // 2], ts[0]
            rows.append({"workload": name, "ops_per_lane": v[0], "nontemporal": v[1], "vec_tile": v[2], "variant": v[3], "vec_passes": v[4],
                         "median_ms": round(med, 4), "min_ms": round(mn, 4),
                         "median_GBs": round(algo / med / 1e6, 1), "frac_of_8TBs": round(algo / med / 1e6 / HBM_PEAK_GBS, 4)})
    return rows


def run(ctx, batch, steps, warmup, dist):
    """Returns (wall seconds for `steps` launches, max over ranks; device ms per launch on this rank)."""
    stream = bench_stream()
    for _ in range(warmup):
        ctx.launch(batch, stream.cuda_stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(steps):
        ctx.launch(batch, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / steps
    wall = dist_util.max_over_ranks(wall)
    return wall, dev_ms


def run_pack(ctx, arrays, tiled, n, steps, warmup, dist):
    """rt_dp_tile_pack of `arrays` into `tiled`, timed like a batch: (wall seconds for `steps` calls, max over ranks; device ms per call)."""
    stream = bench_stream()
    for _ in range(warmup):
        ctx.tile_pack(arrays, tiled, n, stream.cuda_stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(steps):
        ctx.tile_pack(arrays, tiled, n, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = dist_util.max_over_ranks(time.perf_counter() - t0)
    return wall, e0.elapsed_time(e1) / steps


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


OPCODE = {"tri": 0, "box": 1, "euclid": 2, "angular": 3}


def kernel_names(ctx, batch, counts):
    """The kernel the library runs for every segment of `batch`, as rocprofv3 names it (rt_dp_kernel_name)."""
    return {k: ctx.kernel_name(batch, OPCODE[k]) for k in counts}


def roofline(counts, dev_ms, measured_format=True, layout="plain", names=None):
    """The launch against the HBM roof.  `achieved` is live (algorithmic bytes / HIP-event time of this run); `traffic` and
    `valu_issue_fraction` are NOT measured in this run: they are the per-unit figures of the committed rocprofv3 --pmc passes
    (profiles/traffic.json) scaled to this launch, and `traffic_source` says so.  The passes are looked up by KERNEL NAME:
    `names` holds the kernel the library resolved for each segment of this launch, and a segment whose kernel has no
    committed pass (the library picks another kernel than the one that was measured) makes traffic null and the source
    "stale", naming the kernel."""
    algo = sum(BYTES[k] * v for k, v in counts.items())
    gbs = algo / (dev_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
         "traffic": None, "valu_issue_fraction": None, "traffic_source": None}
    if names:
        r["kernels"] = names
    if measured_format and names and os.path.exists(TRAFFIC_FILE):   # the committed passes measured the operand format without derived fields
        try:
            per_unit = json.load(open(TRAFFIC_FILE))
            by_kernel = {v["kernel"]: v for k, v in per_unit.items() if isinstance(v, dict) and "kernel" in v}
            missing = [names[k] for k in counts if names[k] not in by_kernel]
            if missing:
                r["traffic_source"] = "stale: profiles/traffic.json holds no counter pass for %s" % ", ".join(sorted(set(missing)))
                return r
            r["traffic"] = int(sum(by_kernel[names[k]]["hbm_bytes_per_unit"] * v for k, v in counts.items()))
            if len(counts) == 1:
                (k,) = counts.keys()
                r["valu_issue_fraction"] = by_kernel[names[k]].get("valu_issue_fraction")
            r["traffic_source"] = "profiles/traffic.json (%s), per-unit counter figures of the same kernels x the units of this launch; not re-measured in this run" % per_unit.get(
                "_source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes")
        except Exception as e:  # noqa: BLE001
            r["traffic_source"] = "unavailable: %s" % e
    return r


def cpu_baseline(batch, n, seconds):
    """Oracle B (scalar C, one thread) on a bounded sample of the box batch."""
    from oracle import dp_oracle as O  # the checker, used here only as the timed CPU port
    O.build()
    sample = min(n, 1 << 20)
    host = [f[:sample].cpu().numpy() for f in batch.box_fields]  # first `sample` ops of each operand array
    O.box_soa(host)  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        O.box_soa(host)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    one = {"value": round(sample * reps / el, 1), "unit": "ops/s", "cores": 1, "kind": "port",
           "sample": "first 2^%d box ops of the batch, %d passes, %.1f s, scalar C golden model (oracle B), 1 thread" % (
               int(np.log2(sample)), reps, el)}
    # the same model over the host cores this job may use: one slice of the sample per thread
    # (the foreign call releases the interpreter lock)
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 16))
    cuts = [sample * t // threads for t in range(threads + 1)]
    slices = [[f[cuts[t]:cuts[t + 1]] for f in host] for t in range(threads)]
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(O.box_soa, slices))
        reps, t0 = 0, time.perf_counter()
        while True:
            list(pool.map(O.box_soa, slices))
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds / 2:
                break
    one["all_cores"] = {"value": round(sample * reps / el, 1), "unit": "ops/s", "cores": threads,
                        "sample": "same 2^%d ops split over %d threads, %d passes, %.1f s" % (int(np.log2(sample)), threads, reps, el)}
    one["reference_simulation"] = "unavailable: Verilator / JVM / mill toolchain missing on this host, no generated Verilog in the reference tree"
    return one


def stream_measure(a, world, rank, local, dev, log2_n, mode, steps, warmup):
    """Record-at-a-time stream resident in HBM (EnhancedInputBundle images, opcode ~ U{0..3}, reset bit ~ U{0, 1}) through
    rt_dp_eval_bundles_device; every call includes the library's own synchronisation (the entry point is synchronous)."""
    n = 1 << log2_n
    g = torch.Generator(device=dev)
    g.manual_seed(20241013)
    rec = torch.zeros(n, 84, dtype=torch.int32, device=dev)
    op = torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.int32)
    rec[:, 0] = op
    vals = (torch.rand(n, 81, generator=g, device=dev) * 200 - 100)
    rec[:, 1:82] = vals.view(torch.int32)
    del vals
    rec[:, 11:14] = torch.randint(0, 3, (n, 3), generator=g, device=dev, dtype=torch.int32)
    rec[:, 82] = 0xFFFF
    rec[:, 83] = torch.randint(0, 2, (n,), generator=g, device=dev, dtype=torch.int32)
    out = torch.empty(n, 21, dtype=torch.int32, device=dev)
    ctx = capi.Context(device=local)
    ctx.set_option("stream_mode", mode)
    torch.cuda.synchronize()
    for _ in range(warmup):
        ctx.eval_bundles_device(rec, out)
    dist_util.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.eval_bundles_device(rec, out)
    torch.cuda.synchronize()
    dist_util.barrier()
    wall = dist_util.max_over_ranks(time.perf_counter() - t0)
    ctx.close()
    del rec, out
    torch.cuda.empty_cache()
    per = wall / steps
    kernels = "rtdp::k_stream_bits + rtdp::k_stream_eval + rtdp::k_stream_fold" if mode in (0, 2) else "mode %d" % mode
    try:
        passes = json.load(open(TRAFFIC_FILE))
    except Exception:  # noqa: BLE001
        passes = {}
    measured = passes.get("record_stream") if mode in (0, 2) else None
    fresh = bool(measured) and measured.get("kernel") == kernels
    return {"metric": "datapath records/s (AoS record stream, mixed opcodes)", "value": round(n * world / per, 1), "unit": "records/s",
            "ms_per_step": round(per * 1e3, 4), "steps": steps, "warmup": warmup, "scaling": "weak",
            "workload": "stream: 2^%d EnhancedInputBundle records per GPU resident in HBM, opcode ~ U{0..3}" % log2_n,
            "roofline": {"bound": "hbm", "achieved": round(n * 420 / per / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(n * 420 / per / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": int(measured["hbm_bytes_per_unit"] * n) if fresh else None,
                         "traffic_source": ("profiles/traffic.json (%s)" % passes.get("_source", "") + ", per-record counter figures x the records of this call; not re-measured in this run") if fresh
                                           else "stale or absent: no counter pass of " + kernels,
                         "kernels": {"stream": kernels},
                         "note": "420 B per record = 336 B record in + 84 B record out; wall time per synchronous call; mode %d (1: partition first, moves every live operand twice more; 0/2: evaluate in place, then fold the beats; 3: one pass)" % mode}}


def run_stream(a, world, rank, local, dev):
    r = stream_measure(a, world, rank, local, dev, a.log2_ops, a.stream_mode, a.steps, a.warmup)
    if rank == 0:
        emit({"metric": r["metric"], "value": r["value"], "unit": r["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": r["workload"]}, "roofline": r["roofline"]})
    dist_util.shutdown()


def run_trace(a, world, rank, local, dev):
    """Closest-hit traversal: 2^18 random triangles in a 4-wide hierarchy, 2^22 rays per GPU aimed at triangle
    centroids (the reference's genRandomRayGivenPoint pattern); every rank traces its own rays (weak scaling)."""
    from raytracer_amd import bvh
    n_tris, n_rays = 1 << a.trace_log2_tris, 1 << a.trace_log2_rays
    tris = bvh.random_scene(n_tris)
    ctx = capi.Context(device=local)
    if a.trace_build == "device":
        T9 = torch.from_numpy(tris).to(dev)
        ctx.bvh4_build_device(T9, 4)          # first call: workspace allocation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N, T, _ = ctx.bvh4_build_device(T9, 4)
        build_s = time.perf_counter() - t0
    else:
        t0 = time.perf_counter()
        nodes, tris12, _ = bvh.build_bvh4(tris, leaf_size=4)
        build_s = time.perf_counter() - t0
        N, T = torch.from_numpy(nodes).to(dev), torch.from_numpy(tris12).to(dev)
    org, dirs = bvh.rays_towards(tris, n_rays, seed=20241015 + rank)
    if a.trace_sorted:  # neighbouring lanes get neighbouring rays: sort by the cell of the origin, then of the target
        cell = lambda p: np.clip(((p + 1000.0) / 2000.0 * 16).astype(np.int64), 0, 15)
        co, ct = cell(org), cell(org + dirs)
        key = ((co[:, 0] * 16 + co[:, 1]) * 16 + co[:, 2]) * 4096 + (ct[:, 0] * 16 + ct[:, 1]) * 16 + ct[:, 2]
        order = np.argsort(key, kind="stable")
        org, dirs = org[order], dirs[order]
    o = [torch.from_numpy(np.ascontiguousarray(org[:, c])).to(dev) for c in range(3)]
    d = [torch.from_numpy(np.ascontiguousarray(dirs[:, c])).to(dev) for c in range(3)]
    out_t = torch.empty(n_rays, dtype=torch.float32, device=dev)
    out_i = torch.empty(n_rays, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        stats = ctx.trace_closest(N, T, o, d, 4000.0, out_t, out_i, wavefront=bool(a.trace_wavefront))
    dist_util.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stats = ctx.trace_closest(N, T, o, d, 4000.0, out_t, out_i, wavefront=bool(a.trace_wavefront))
    torch.cuda.synchronize()
    dist_util.barrier()
    wall = dist_util.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        per = wall / a.steps
        hits = float((out_i != -1).float().mean().item())
        emit({"metric": "rays/s (closest hit, BVH4 %s traversal on the datapath ops)" % ("wavefront" if a.trace_wavefront else "depth-first"), "value": round(n_rays * world / per, 1),
                          "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(per * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "trace: 2^%d triangles, 2^%d %s rays per GPU, leaf size 4, hierarchy built on the %s in %.4f s (not timed)" % (
                              a.trace_log2_tris, a.trace_log2_rays, "cell-sorted" if a.trace_sorted else "unsorted", a.trace_build, build_s)},
                          "per_call": {k: int(v) for k, v in stats.items()}, "hit_fraction": round(hits, 4),
                          "box_ops_per_s": round(stats["box_ops"] / per, 1), "tri_ops_per_s": round(stats["tri_ops"] / per, 1)})
    ctx.close()
    dist_util.shutdown()


def knn_measure(a, ctx, world, rank, dev, tree_mode, steps, warmup):
    """k-NN configuration: 2^20 points (sharded over the ranks), 2^16 queries (replicated), dim 16, k = 16;
    per-GPU partial lists merged after one all-gather.  Total work is fixed: strong scaling.  Returns the result
    block (same on every rank)."""
    from raytracer_amd import knn
    n_pts, n_qry = 1 << a.knn_log2_points, 1 << a.knn_log2_queries
    g = torch.Generator(device=dev)
    g.manual_seed(20241012)
    pts = torch.rand(n_pts, 16, generator=g, device=dev) * 2 - 1   # same on every rank
    qry = torch.rand(n_qry, 16, generator=g, device=dev) * 2 - 1
    if a.knn_data == "clustered":
        centres = torch.rand(256, 16, generator=g, device=dev) * 2 - 1
        pts = centres[torch.randint(0, 256, (n_pts,), generator=g, device=dev)] + 0.05 * torch.randn(n_pts, 16, generator=g, device=dev)
        qry = centres[torch.randint(0, 256, (n_qry,), generator=g, device=dev)] + 0.05 * torch.randn(n_qry, 16, generator=g, device=dev)
    lo, hi = (rank * n_pts) // world, ((rank + 1) * n_pts) // world
    shard_pts = pts[lo:hi].contiguous()
    stream = bench_stream()
    times = []
    tree, build_ms = None, None
    tree_stats = {}
    angular = a.knn_metric == "angular"
    if angular and tree_mode == 2:
        tree_mode = 1        # the leaf-major form exists for the Euclidean op only
    group = dist_util.data_group()   # RCCL if it works on every rank, else None: the exchange is staged through the host over gloo
    if tree_mode:
        from raytracer_amd import knn_tree
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if a.knn_build == "device":
            knn_tree.KnnTree.build_device(ctx, shard_pts, leaf_size=a.knn_leaf, index_base=lo)   # first call: workspace allocation
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tree = knn_tree.KnnTree.build_device(ctx, shard_pts, leaf_size=a.knn_leaf, index_base=lo)
        else:
            tree = knn_tree.KnnTree(shard_pts, leaf_size=a.knn_leaf, index_base=lo)  # index construction: outside the timed queries
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
    with torch.cuda.stream(stream):
        for it in range(warmup + steps):
            torch.cuda.synchronize()
            dist_util.barrier()
            t0 = time.perf_counter()
            if tree is not None and angular:
                d, i = knn.unpack_signed_keys(knn.all_gather_merge(knn_tree.local_topk_angular(ctx, tree, qry, splits=a.knn_splits or None, stats=tree_stats), group=group))
            elif tree is not None:
                d, i = knn_tree.search(ctx, tree, qry, group=group, splits=a.knn_splits or None, grouped=(tree_mode == 2), walk_leaves=a.knn_walk)
            elif angular:
                d, i = knn.unpack_signed_keys(knn.all_gather_merge(knn.local_topk_angular(ctx, shard_pts, qry, lo, splits=a.knn_splits or None), group=group))
            else:
                d, i = knn.search(ctx, shard_pts, qry, lo, group=group, splits=a.knn_splits or None)
            torch.cuda.synchronize()
            dist_util.barrier()
            if it >= warmup:
                times.append(time.perf_counter() - t0)
    wall = dist_util.max_over_ranks(sum(times))
    per = wall / steps
    pairs = float(n_pts) * n_qry
    if angular:   # the candidate's norm is computed once per point, not per pair
        lane_ops, note = 35, "16 mul + 16 add (dot, two beats) + 1 mul, 1 add, 1 compare (filter) per pair, none fusable"
    else:
        lane_ops, note = 48, "16 sub + 16 mul + 15 add + 1 compare per pair, none fusable"
    note += "; peak = 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz"
    return {"metric": "k-NN queries/s (exact, %s, k=16)" % ("angular op score" if angular else "Euclidean beat distance"),
            "value": round(n_qry / per, 1), "unit": "queries/s", "ms_per_step": round(per * 1e3, 3), "steps": steps, "warmup": warmup, "scaling": "strong",
            "workload": "knn: 2^%d points sharded over %d GPU(s), 2^%d queries replicated, dim 16, k 16, %s data, one all-gather merge" % (
                a.knn_log2_points, world, a.knn_log2_queries, a.knn_data),
            "search": ("k-d tree %s, leaf %d, index build %.1f ms (not timed)" % ("leaf-major search" if tree_mode == 2 else "walk with box pruning", a.knn_leaf, build_ms)) if tree_mode else "exhaustive",
            "pair_rate_equivalent_per_s": round(pairs / per, 1), "exchange": dist_util.data_backend(), **tree_stats,
            "valu": None if tree_mode else {"lane_ops_per_pair": lane_ops, "achieved_T_lane_ops": round(pairs * lane_ops / per / 1e12, 2), "peak_T_lane_ops": 78.6,
                                            "frac": round(pairs * lane_ops / per / 78.6e12, 4), "note": note}}


def run_knn(a, world, rank, local, dev):
    ctx = capi.Context(device=local)
    r = knn_measure(a, ctx, world, rank, dev, a.knn_tree, a.steps, a.warmup)
    if rank == 0:
        emit({"metric": r["metric"], "value": r["value"], "unit": r["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": r["workload"]}, "search": r["search"],
                          "pair_rate_equivalent_per_s": r["pair_rate_equivalent_per_s"], "valu": r["valu"], "exchange": r["exchange"],
                          "leaves_scanned_fraction": r.get("leaves_scanned_fraction")})
    ctx.close()
    dist_util.shutdown()


def make_strong_mixed(total, world, rank, dev, keep_whole=False, layout="plain", ctx=None):
    """The config-4 stream: `total` units, a quarter per opcode, opcode-sorted; this rank keeps the shard that
    shard.plan gives it (contiguous ranges, beat segments cut on vector-pair boundaries: every shard starts from +0
    accumulators and no collective touches the data path).  Every rank draws the same whole stream and keeps its part.
    Returns (batch of the shard, unit counts of the shard, unit counts of the whole stream, whole-stream arrays or None)."""
    from raytracer_amd import shard
    q = total // 4
    f32, i32 = torch.float32, torch.int32
    ea, eb, ectl = synth.vec_stream(q, 16, 20241010, dev)
    aq, ac, actl = synth.vec_stream(q, 8, 20241011, dev)
    ne, na = int(ectl.shape[0]), int(actl.shape[0])
    plan = shard.plan(q, q, ectl.cpu().numpy().view(np.uint32), actl.cpu().numpy().view(np.uint32), world)[rank]
    whole = {} if keep_whole else None

    def cut(t, rng):
        if world == 1:
            return t
        return t[rng[0]:rng[1]].clone()

    if keep_whole:
        whole.update(ea=ea, eb=eb, ectl=ectl, aq=aq, ac=ac, actl=actl)
    e = (cut(ea, plan["euclid"]), cut(eb, plan["euclid"]), cut(ectl, plan["euclid"]))
    g = (cut(aq, plan["angular"]), cut(ac, plan["angular"]), cut(actl, plan["angular"]))
    del ea, eb, ectl, aq, ac, actl
    bf = synth.box_stream(q, 20241008, dev)
    if keep_whole:
        whole["box"] = bf
    box = [cut(f, plan["box"]) for f in bf]
    del bf
    tf, tk = synth.tri_stream(q, 20241009, dev)
    if keep_whole:
        whole["tri"], whole["trik"] = tf, tk
    tri, trik = [cut(f, plan["tri"]) for f in tf], cut(tk, plan["tri"])
    del tf, tk
    torch.cuda.empty_cache()
    counts = {"box": int(box[0].shape[0]), "tri": int(tri[0].shape[0]), "euclid": int(e[2].shape[0]), "angular": int(g[2].shape[0])}
    b = batch_from(box, tri, trik, e, g, dev, layout, ctx)
    del box, tri, trik
    torch.cuda.empty_cache()
    return b, counts, {"box": q, "tri": q, "euclid": ne, "angular": na}, whole


def batch_from(box, tri, trik, euc, ang, dev, layout="plain", ctx=None):
    """One mixed batch over the given operand arrays with freshly allocated result arrays (kept in batch.results).
    layout "tiled": the box and triangle segments are packed into tiled buffers (the given arrays are not referenced)."""
    f32, i32 = torch.float32, torch.int32
    nb, nt, ne, na = int(box[0].shape[0]), int(tri[0].shape[0]), int(euc[2].shape[0]), int(ang[2].shape[0])
    res = {"acc": torch.empty(ne, dtype=f32, device=dev), "dot": torch.empty(na, dtype=f32, device=dev), "norm": torch.empty(na, dtype=f32, device=dev)}
    b = capi.Batch()
    if layout == "tiled":
        tiled_box(ctx, b, box, nb, dev)
        tiled_tri(ctx, b, tri, trik, nt, dev)
        res["tiled"] = (nb, nt)
    else:
        res.update({"tmin": [torch.empty(nb, dtype=f32, device=dev) for _ in range(4)], "flags": torch.empty(nb, dtype=i32, device=dev),
                    "t_num": torch.empty(nt, dtype=f32, device=dev), "t_denom": torch.empty(nt, dtype=f32, device=dev),
                    "hit": torch.empty(nt, dtype=i32, device=dev)})
        b.set_box(box, res["tmin"], res["flags"])
        b.set_tri(tri, trik, res["t_num"], res["t_denom"], res["hit"])
    b.set_euclid(euc[0], euc[1], euc[2], res["acc"])
    b.set_angular(ang[0], ang[1], ang[2], res["dot"], res["norm"])
    b.finalize()
    b.results = res
    return b


def result_list(batch, ctx):
    """The result arrays of a batch_from batch in one fixed order; tiled segments are unpacked into plain arrays first."""
    res = batch.results
    if "tiled" in res:
        nb, nt = res["tiled"]
        dev = res["acc"].device
        box = [torch.empty(nb, dtype=torch.int32, device=dev) for _ in range(5)]
        tri = [torch.empty(nt, dtype=torch.int32, device=dev) for _ in range(3)]
        ctx.tile_unpack(batch.box_tiled[1], box, nb)
        ctx.tile_unpack(batch.tri_tiled[1], tri, nt)
        ctx.synchronize()
        return box + tri + [res["acc"], res["dot"], res["norm"]]
    return res["tmin"] + [res["flags"], res["t_num"], res["t_denom"], res["hit"], res["acc"], res["dot"], res["norm"]]


def strong_mixed_measure(a, ctx, world, rank, dev, dist, log2_total, steps, warmup, layout=None):
    """Runs the sharded mixed stream; returns the result block (identical on every rank)."""
    total = 1 << log2_total
    layout = layout or a.layout
    batch, counts, whole_counts, whole = make_strong_mixed(total, world, rank, dev, keep_whole=a.verify, layout=layout, ctx=ctx)
    torch.cuda.synchronize()
    ctx.reset()
    wall, dev_ms = run(ctx, batch, steps, warmup, dist)
    units = sum(whole_counts.values())
    block = {"value": round(units * steps / wall, 1), "unit": "ops/s", "ms_per_step": round(wall / steps * 1e3, 4), "kernel_ms": round(dev_ms, 4),
             "steps": steps, "warmup": warmup, "scaling": "strong", "units_whole_job": units, "units_this_gpu": sum(counts.values()),
             "workload": "mixed: 2^%d-unit opcode-sorted stream (a quarter per opcode), cut by shard.plan into %d pair-aligned shard(s), no collective" % (log2_total, world),
             "layout": layout, "roofline": roofline(counts, dev_ms, True, layout, kernel_names(ctx, batch, counts))}
    if a.verify:
        # the shards' results, put together in rank order, must be the whole stream's results (computed on rank 0 in one batch)
        ctx.reset()
        ctx.launch(batch, bench_stream().cuda_stream)
        torch.cuda.synchronize()
        mine = [t.cpu().numpy().view(np.uint32) for t in result_list(batch, ctx)]
        parts = dist_util.gather_objects(mine)
        ok = None
        if rank == 0:
            # the whole stream always in the plain layout: a tiled run is thereby also checked against the plain kernels
            wb = batch_from(whole["box"], whole["tri"], whole["trik"], (whole["ea"], whole["eb"], whole["ectl"]),
                            (whole["aq"], whole["ac"], whole["actl"]), dev)
            c2 = capi.Context(device=torch.cuda.current_device())
            c2.launch(wb, bench_stream().cuda_stream)
            torch.cuda.synchronize()
            want = [t.cpu().numpy().view(np.uint32) for t in result_list(wb, c2)]
            c2.close()
            ok = all(np.array_equal(np.concatenate([p[k] for p in parts]), want[k]) for k in range(len(want)))
        block["verified_against_whole_stream"] = ok
    del batch
    torch.cuda.empty_cache()
    return block


def resolved_tuning(ctx, a):
    """The kernel shapes actually in effect (library defaults filled in), for the JSON line."""
    t = {"requested": {"ops_per_lane": a.ops_per_lane, "nontemporal": a.nontemporal, "vec_tile": a.vec_tile, "variant": a.variant,
                       "vec_passes": a.vec_passes, "derive": a.derive, "wgs_per_cu": a.wgs_per_cu}}
    for key in ("ops_per_lane", "nontemporal", "vec_tile", "vec_passes", "box_wgs_per_cu", "tri_wgs_per_cu"):
        try:
            t[key] = ctx.get_option(key)
        except capi.RtDpError:
            t[key] = None
    if t.get("vec_passes") == 0:
        t["vec_passes"] = {"euclid": 8, "angular": 4}
    if not a.vec_tile:
        t["vec_tile"] = {"euclid": 512, "angular": 504}
    t["beat_kernel_wgs_per_cu"] = "uncapped"
    return t


_RESULT_OUT = None


def claim_stdout():
    """Keeps the process's standard output for the result line alone: file descriptor 1 is pointed at standard error for
    everything else (libraries below Python print there too: gloo announces its connections on stdout), and emit() writes
    the JSON line to the descriptor that was standard output when the process started."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    line = json.dumps(obj)
    out = _RESULT_OUT if _RESULT_OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    a = parse()
    claim_stdout()
    world, rank, local = dist_util.env_world()
    if a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launch environment has %d rank(s); reporting n_gpus = %d" % (a.gpus, world, world), file=sys.stderr)
    if os.environ.get("RT_DP_REHEARSE_ONE_GPU") == "1":
        # rehearsal of the N > 1 path on a one-GPU box: every rank uses GPU 0 (RCCL refuses two ranks on one
        # device, so the k-NN exchange is staged through the host as well); timings are meaningless
        local = 0
    torch.cuda.set_device(local)
    # Control (barriers, max of the wall times, small objects) always goes over gloo with host tensors, so the sharded
    # op streams, which have no collective, report their numbers whether or not RCCL comes up; RCCL is used for the
    # k-NN all-gather only (dist_util.data_group, created and verified on first use, host staging if it fails).
    dist_util.init("gloo")
    dist = torch.distributed if world > 1 else None
    dev = torch.device("cuda", local)
    n = a.ops if a.ops else 1 << a.log2_ops

    if a.only == "stream":
        run_stream(a, world, rank, local, dev)
        return
    if a.only == "knn":
        run_knn(a, world, rank, local, dev)
        return
    if a.only == "trace":
        run_trace(a, world, rank, local, dev)
        return

    ctx = capi.Context(device=local)
    ctx.set_tuning(a.ops_per_lane, a.nontemporal, a.vec_tile, a.variant, a.vec_passes)
    ctx.set_option("wgs_per_cu", a.wgs_per_cu)
    tuning = resolved_tuning(ctx, a)

    if a.only == "mixed" and a.strong:
        r = strong_mixed_measure(a, ctx, world, rank, dev, dist, a.strong_log2_ops, a.steps, a.warmup)
        if rank == 0:
            emit({"metric": "datapath ops/s (mixed opcodes, one stream sharded over the GPUs)", "value": r["value"], "unit": "ops/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": r["workload"], "units_whole_job": r["units_whole_job"], "units_this_gpu": r["units_this_gpu"],
                                         "sharding": "shard.plan: contiguous ranges per opcode segment, beat segments cut on vector-pair boundaries", "tuning": tuning},
                              "roofline": r["roofline"], "cpu_baseline": None,
                              "verified_against_whole_stream": r.get("verified_against_whole_stream")})
        ctx.close()
        dist_util.shutdown()
        return

    if a.only == "pack":   # rt_dp_tile_pack alone: what a caller with separate arrays pays for the tiled layout
        fields = synth.box_stream(n, 20241008, dev)
        tiled_buf = torch.empty(capi.tiled_words(n, capi.BOX_TILE_IN_WORDS), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        wall, dev_ms = run_pack(ctx, fields, tiled_buf, n, a.steps, a.warmup, dist)
        moved = 2 * 4 * len(fields) * n
        if rank == 0:
            emit({"metric": "ops/s packed (rt_dp_tile_pack, 34 operand arrays -> tiled box segment)", "value": round(n * world * a.steps / wall, 1), "unit": "ops/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "pack: %d box ops per GPU" % n},
                              "roofline": {"bound": "hbm", "achieved": round(moved / dev_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(moved / dev_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(dev_ms, 4)}, "cpu_baseline": None})
        ctx.close()
        dist_util.shutdown()
        return

    which = [a.only] if a.only else ["box", "tri", "euclid", "angular", "mixed"]
    only_edge = 0.0
    if a.only in ("box_edge", "tri_edge"):      # the edge-class stream of one opcode as the headline
        which, only_edge = [a.only.split("_")[0]], a.edge_per_class
    if a.derive:
        BYTES.update(box=144, tri=72)
    ballast = torch.empty(int(a.ballast_gib * 2 ** 30), dtype=torch.uint8, device=dev).fill_(1) if a.ballast_gib > 0 else None  # noqa: F841
    batches = make_batches(n, dev, which, a.stagger_kib, bool(a.derive), a.launch_ops, only_edge, a.layout, ctx)
    torch.cuda.synchronize()

    if a.sweep:
        for row in sweep(ctx, batches, which):
            emit(row)
        ctx.close()
        return

    plain = not a.derive      # the committed counter passes measured the operand format without derived fields
    results = {}
    for name in which:
        batch, counts = batches[name]
        wall, dev_ms = run(ctx, batch, a.steps, a.warmup, dist)
        units = sum(counts.values())
        results[name] = {"value": units * world * a.steps / wall, "ms_per_step": wall / a.steps * 1e3, "scaling": "weak",
                         "kernel_ms": dev_ms, "units_per_gpu": units, "layout": a.layout if name in ("box", "tri", "mixed") else "plain",
                         "roofline": roofline(counts, dev_ms, plain, a.layout, kernel_names(ctx, batch, counts))}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and "box" in batches:
        cpu = cpu_baseline(batches["box"][0], n, a.cpu_seconds)
    extras = {}
    if not a.only and not a.no_extras:
        # config 2 / 3 with their edge classes: >= 1 % of the ops in every class, same size as the headline.  The edge
        # streams are written INTO the arrays of the plain batches, so the two runs differ in their operands only and
        # not in where the 39 (19) arrays lie against each other (which alone moves the rate by several per cent)
        other = "tiled" if a.layout == "plain" else "plain"
        for name in ("box", "tri"):
            # the same ops in the other layout (the operands are packed from / kept in the arrays of the batch above)
            batch, counts = batches[name]
            ob = capi.Batch().derive_ray(bool(a.derive))
            if other == "tiled":
                if name == "box":
                    tiled_box(ctx, ob, batch.box_fields, n, dev)
                else:
                    tiled_tri(ctx, ob, batch.tri_fields, batch.tri_k, n, dev)
            elif name == "box":
                ob.set_box(batch.box_fields, [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4)], torch.empty(n, dtype=torch.int32, device=dev), n)
            else:
                ob.set_tri(batch.tri_fields, batch.tri_k, torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
                           torch.empty(n, dtype=torch.int32, device=dev), n)
            ob.finalize()
            wall, dev_ms = run(ctx, ob, a.steps, a.warmup, dist)
            results[name + "_" + other] = {"value": counts[name] * world * a.steps / wall, "ms_per_step": wall / a.steps * 1e3, "scaling": "weak",
                                           "kernel_ms": dev_ms, "units_per_gpu": counts[name], "layout": other,
                                           "roofline": roofline(counts, dev_ms, plain, other, kernel_names(ctx, ob, counts))}
            del ob
            torch.cuda.empty_cache()
        if a.layout == "tiled":
            # what a caller with one array per operand pays to get the tiled layout: rt_dp_tile_pack, one pass, operand bytes in + out
            for name, arrays, tiled_buf in (("box", batches["box"][0].box_fields, batches["box"][0].box_tiled[0]),
                                            ("tri", list(batches["tri"][0].tri_fields) + [batches["tri"][0].tri_k], batches["tri"][0].tri_tiled[0])):
                wall, dev_ms = run_pack(ctx, arrays, tiled_buf, n, a.steps, a.warmup, dist)
                moved = 2 * 4 * len(arrays) * n
                results["pack_" + name] = {"value": n * world * a.steps / wall, "ms_per_step": wall / a.steps * 1e3, "scaling": "weak", "kernel_ms": dev_ms,
                                           "units_per_gpu": n, "layout": "plain -> tiled",
                                           "roofline": {"bound": "hbm", "achieved": round(moved / dev_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                        "frac": round(moved / dev_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
                                                        "bytes_per_op": 2 * 4 * len(arrays), "kernels": {name: "rtdp::k_tile_copy16<true>"},
                                                        "note": "rt_dp_tile_pack: %d operand words per op read from separate arrays and written to the tiled buffer" % len(arrays)}}
        for name in ("box", "tri"):
            batch, counts = batches[name]
            if name == "box":
                fresh = synth.box_stream(n, 20241008, dev, a.edge_per_class)
                for dst, src in zip(batch.box_fields, fresh):
                    dst.copy_(src)
                if a.layout == "tiled":
                    torch.cuda.synchronize()
                    ctx.tile_pack(batch.box_fields, batch.box_tiled[0], n)
            else:
                fresh, fresh_k = synth.tri_stream(n, 20241009, dev, a.edge_per_class)
                for dst, src in zip(batch.tri_fields, fresh):
                    dst.copy_(src)
                batch.tri_k.copy_(fresh_k)
                if a.layout == "tiled":
                    torch.cuda.synchronize()
                    ctx.tile_pack(list(batch.tri_fields) + [batch.tri_k], batch.tri_tiled[0], n)
                del fresh_k
            del fresh
            ctx.synchronize()
            torch.cuda.synchronize()
            wall, dev_ms = run(ctx, batch, a.steps, a.warmup, dist)
            results[name + "_edge"] = {"value": counts[name] * world * a.steps / wall, "ms_per_step": wall / a.steps * 1e3, "scaling": "weak",
                                       "kernel_ms": dev_ms, "units_per_gpu": counts[name], "layout": a.layout, "roofline": roofline(counts, dev_ms, plain, a.layout, kernel_names(ctx, batch, counts)),
                                       "edge_classes": {"per_class": a.edge_per_class, "same_arrays_as": name,
                                                        "classes": synth.BOX_EDGE_CLASSES if name == "box" else synth.TRI_EDGE_CLASSES}}
        del batches, batch
        torch.cuda.empty_cache()
        # config 4: the 2^28-unit mixed stream, one shard per GPU
        def guarded(fn, *args):
            # an entry that fails (the same way on every rank: a shape or argument error) is reported as such instead of
            # taking the headline measured above down with it
            try:
                return fn(*args)
            except Exception as e:  # noqa: BLE001
                torch.cuda.empty_cache()
                return {"error": "%s: %s" % (type(e).__name__, e)}

        extras["mixed_2p%d" % a.strong_log2_ops] = guarded(strong_mixed_measure, a, ctx, world, rank, dev, dist, a.strong_log2_ops, a.steps, a.warmup)
        extras["mixed_2p%d_%s" % (a.strong_log2_ops, other)] = guarded(strong_mixed_measure, a, ctx, world, rank, dev, dist, a.strong_log2_ops, a.steps, a.warmup, other)
        # the record-at-a-time front end (what a harness that holds EnhancedInputBundle images binds), at the headline's size
        extras["record_stream"] = guarded(stream_measure, a, world, rank, local, dev, a.log2_ops, 0, max(1, min(a.steps, 10)), max(1, min(a.warmup, 3)))
        # config 5: k-NN, exhaustive and leaf-major
        ksteps, kwarm = max(1, min(a.steps, 5)), max(1, min(a.warmup, 2))
        extras["knn"] = {"exhaustive": guarded(knn_measure, a, ctx, world, rank, dev, 0, ksteps, kwarm),
                         "leaf_major": guarded(knn_measure, a, ctx, world, rank, dev, 2, ksteps, kwarm)}

    head = which[0]
    r = results[head]
    size_text = ("%d" % a.ops) if a.ops else ("2^%d" % a.log2_ops)
    line = {
        "metric": "datapath ops/s per opcode (whole node)" if head == "box" else "datapath ops/s (%s)" % head,
        "value": round(r["value"], 1), "unit": "ops/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(r["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s random ray-vs-4-AABB box ops per GPU, SoA fp32 operands resident in HBM%s" % (
                       head, size_text, " in tiles of 8192 ops" if a.layout == "tiled" else "")
                   if head == "box" else "%s, %s units per GPU" % (head, size_text),
                   "ops_per_gpu": r["units_per_gpu"], "bytes_per_op": BYTES.get(head),
                   "layout": "one array per operand" if a.layout == "plain" or head not in ("box", "tri", "mixed") else "tiled: one buffer per segment, [tile][word][8192 ops]", "sharding": "independent op streams per GPU, no collective",
                   "tuning": tuning},
        "roofline": r["roofline"],
    }
    if rank == 0:
        line["cpu_baseline"] = cpu
        line["per_opcode"] = {k: {"value": round(v["value"], 1), "unit": "ops/s" if k.split("_")[0] in ("box", "tri", "mixed", "pack") else "beats/s",
                                  "scaling": v["scaling"], "kernel_ms": round(v["kernel_ms"], 4), "units_per_gpu": v["units_per_gpu"],
                                  "layout": v.get("layout", "plain"), "roofline": v["roofline"], **({"edge_classes": v["edge_classes"]} if "edge_classes" in v else {})}
                              for k, v in results.items()}
        line["per_opcode"].update(extras)
        emit(line)
    ctx.close()
    dist_util.shutdown()


if __name__ == "__main__":
    main()
