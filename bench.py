#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman database search on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (swg_search: int16 fill, saturation re-score when
possible, top-K) over one resident synthetic database; for N > 1 every rank owns an
independent shard of the same shape (weak scaling, no data-path collective) and the only
exchange is one RCCL max-all-reduce of the n*K top-K keys per step.

The JSON line carries BASELINE.json's metric (GCUPS = lq * sum(len) / t / 1e9 over real
residues), the HBM roofline of the fill kernel, and the reference's own AVX2+OpenMP fill
timed on this host (oracle/_ref, built from the reference's sources) as `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default; the library's
# two fill streams plus torch's current, copy and RCCL streams are more than that, and streams
# that share a queue serialise (measured: the top-K all-reduce waits behind a whole fill and the
# step grows by 0.4 ms).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Several ranks on one node share its CPUs: without a setting (torchrun exports OMP_NUM_THREADS=1 by
# itself) every rank's host loops would take the whole machine.  Setup work only (packing a shard).
if int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) > 1:
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16) //
                          int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import swg_loader  # noqa: E402

# (query length, sequences, matrix); gaps are the tool defaults -2/-1
CONFIGS = {
    1: dict(lq=128, n=1024, matrix="BLOSUM62"),
    2: dict(lq=367, n=100000, matrix="PAM250"),
    3: dict(lq=500, n=570000, matrix="BLOSUM62"),
    4: dict(lq=3000, n=1250000, matrix="BLOSUM62"),   # one GPU's eighth of the 10M-sequence DB
    5: dict(lq=8192, n=100000, matrix="BLOSUM62", similar=0.01),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--max-waves", type=int, default=0)
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 systolic, 2 diagonal")
    ap.add_argument("--group", type=int, default=0, help="diagonal engine: lanes per sequence pair")
    ap.add_argument("--long-split", type=int, default=0, help="-1 off, 0 auto, else rows threshold of the long class")
    ap.add_argument("--long-cols", type=int, default=0, help="experiment: columns per lane of the long class")
    ap.add_argument("--static-streams", action="store_true", help="experiment: diagonal engine without the work queue")
    ap.add_argument("--depth", type=int, default=2, help="searches in flight (1..4)")
    ap.add_argument("--side-readout", type=int, default=-1, help="experiment: 0 = top-K and read-out on the fill stream")
    ap.add_argument("--long-helps", action="store_true", help="experiment: long-class lane groups go on with the bulk's pairs")
    ap.add_argument("--no-long-helps", action="store_true", help="(the default; kept for the sweep scripts)")
    ap.add_argument("--prio-share", type=int, default=-1, help="experiment: priority threshold, percent of a lane group's mean share")
    ap.add_argument("--long-group", type=int, default=0, help="experiment: lanes per pair of the long class")
    ap.add_argument("--lq", type=int, default=0, help="experiment: override the query length of the config")
    ap.add_argument("--nseq", type=int, default=0, help="experiment: override the sequence count of the config")
    ap.add_argument("--autotune", action="store_true",
                    help="time the best-ranked geometries on the device at the first search (default: the cost model alone, "
                         "so that every rank of a multi-GPU run uses the same plan)")
    ap.add_argument("--no-autotune", action="store_true", help="(the default; kept for the sweep scripts)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every step before queuing the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="skip the one-off search from host buffers (PCIe-inclusive figure, reported beside value)")
    ap.add_argument("--uniform-len", type=int, default=0,
                    help="diagnostic: every sequence gets this length (no length tail)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SWG_BENCH_FORCE_DIST=1: take the collective path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("SWG_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    lib_path = os.path.join(ROOT, "seq-align-gpu_amd", "libswg.so")
    if not os.path.exists(lib_path):
        if rank == 0:
            swg_loader.build_module().build()
        if use_dist:
            dist.barrier()
    swg = swg_loader.load()

    cfg = dict(CONFIGS[args.config])
    if args.lq:
        cfg["lq"] = args.lq
    if args.nseq:
        cfg["n"] = args.nseq
    lq, n = cfg["lq"], cfg["n"]
    sc = swg.load_scoring(cfg["matrix"])
    seed = 0x5EED0000 + args.config
    q = swg.synth_query(seed, lq)
    shard_seed = seed + 0x10000 * rank          # every rank: an independent shard of the same shape
    if cfg.get("similar"):
        flat, off, _ = swg.synth_db(shard_seed, n, query=q, fraction=cfg["similar"], subst=0.05)
    elif args.uniform_len:
        flat, off = swg.synth_db(shard_seed, n, min_len=args.uniform_len, max_len=args.uniform_len)
    else:
        flat, off = swg.synth_db(shard_seed, n)

    ctx = swg.Context(local_rank)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    ctx.set_option("cols_per_wave", args.cols)
    ctx.set_option("max_waves", args.max_waves)
    ctx.set_option("workgroups", args.workgroups)
    ctx.set_option("engine", args.engine)
    ctx.set_option("group_lanes", args.group)
    ctx.set_option("long_split", args.long_split)
    ctx.set_option("long_cols", args.long_cols)
    ctx.set_option("long_group", args.long_group)
    ctx.set_option("autotune", 1 if args.autotune and not args.no_autotune else 0)
    ctx.set_option("work_queue", 0 if args.static_streams else 1)
    if args.side_readout >= 0:
        ctx.set_option("side_readout", args.side_readout)
    if args.long_helps:
        ctx.set_option("long_helps", 1)
    if args.prio_share >= 0:
        ctx.set_option("prio_share", args.prio_share)
    db = swg.Database(flat, off).upload(ctx)
    residues = int(db.residues)
    # setup, untimed like the upload: the first search of a query length plans the kernel geometry
    # for this database (and with --autotune times the best-ranked plans on this device)
    ctx.search(db, want_scores=False, k=args.topk)

    K = args.topk
    merger = TopKMerger(swg, K, rank, world, "cuda") if use_dist else None

    # Steps are software-pipelined two deep: search i+1 is queued on the GPU before the host finishes
    # search i (top-K read-out, and for N > 1 the all-reduce merge).  The library runs a search's
    # top-K kernels and read-out on a stream of their own, beside the start of the next fill; deeper
    # queues measured slower.  Every step still does all of its work inside the timed region.
    depth = 1 if args.no_pipeline else args.depth

    def finish(ticket):
        if use_dist:
            keys, st = ctx.search_end_keys(ticket)
            return merger.merge_keys(keys), st
        _, hits, st = ctx.search_end(ticket)
        return hits, st

    def run_steps(n, record):
        pending = []
        for _ in range(n):
            pending.append(ctx.search_begin(db, K))
            if len(pending) >= depth:
                record(*finish(pending.pop(0)))
        while pending:
            record(*finish(pending.pop(0)))

    run_steps(args.warmup, lambda hits, st: None)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fill_ms, total_ms, lasts = [], [], []

    def record(hits, st):
        fill_ms.append(st["fill_ms"])
        total_ms.append(st["total_ms"])
        lasts.append(st)

    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, record)
    fence()
    elapsed = time.perf_counter() - t0
    last = lasts[-1]

    cells_local = lq * residues
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_total = float(c.item())
    else:
        cells_total = float(cells_local)

    if rank == 0:
        gcups = cells_total * args.steps / elapsed / 1e9
        k_ms = float(np.mean(fill_ms))
        bytes_alg = int(last["bytes_alg"])
        achieved = bytes_alg / (k_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                traffic = tj.get("config%d" % args.config, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # The binding roof is integer VALU issue, reported beside the (by construction tiny) HBM
        # fraction.  5 packed instructions per cell (10 per two cells).  The hardware issues one wave64
        # packed instruction per SIMD every 4 cycles (16 lanes per cycle) = the peak used for the
        # fraction; an isolated instruction stream measures 4.4-4.56 cycles with 4 waves per SIMD
        # (tools/valu_rate.hip, profiles/r01_valu_issue_rates.txt), the fill kernels get to 4.26.
        ops_per_cell = 5.0 if last["path_bits"] == 16 else 12.0
        simds = 256 * 4
        kernel_gcups = cells_local / (k_ms * 1e-3) / 1e9
        peak_issue = simds * 64 / 4.0 * 2.4e9 / ops_per_cell / 1e9
        peak_microbench = simds * 64 / 4.56 * 2.35e9 / ops_per_cell / 1e9
        out = {
            "metric": "GCUPS (DP cell updates/s) at 1/2/4/8 MI355X; bit-exact max scores vs CPU ref",
            "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16" if last["path_bits"] == 16 else "int32", "data": "synthetic",
            "config": {
                "workload": "config %d: 1 query (%d aa) vs %d-seq synthetic protein DB per GPU, %s, gaps -2/-1, top-%d"
                            % (args.config, lq, n, cfg["matrix"], K),
                "lq": lq, "n_seqs_per_gpu": n, "residues_per_gpu": residues, "matrix": cfg["matrix"],
                "cols_per_wave": last["cols_per_wave"], "waves": last["waves"], "passes": last["passes"],
                "workgroups": last["workgroups"], "n_rescored": last["n_rescored"],
                "engine": {1: "systolic", 2: "diagonal"}.get(last["engine"]), "group_lanes": last["group_lanes"],
                "streams": last["streams"], "long_pairs": last["long_pairs"],
                "long_cols_per_lane": last["long_cols_per_lane"], "long_streams": last["long_streams"],
                "work_queue": bool(last["work_queue"]),
                "cells_padded_over_real": round(last["cells_padded"] / max(1, last["cells"]), 4),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "kernel": (("swg_diag_dyn_kernel<%d>" if last["work_queue"] else "swg_diag_kernel<%d>") % last["cols_per_wave"])
                if last["engine"] == 2
                else "swg_fill_kernel<CellsI%d>" % last["path_bits"], "kernel_ms": round(k_ms, 4),
                "bytes_alg_per_launch": bytes_alg,
                "binding_roof": {"bound": "valu_issue", "kernel_gcups": round(kernel_gcups, 2),
                                 "instr_per_cell": ops_per_cell,
                                 "cycles_per_wave_instr": 4.0, "clock_ghz": 2.4,
                                 "peak_gcups_issue": round(peak_issue, 1),
                                 "frac_of_issue_peak": round(kernel_gcups / peak_issue, 4),
                                 "peak_gcups_microbenchmark": round(peak_microbench, 1)},
            },
            "kernel_ms": {"fill": round(k_ms, 4), "search_total": round(float(np.mean(total_ms)), 4),
                          "rescore": round(float(last["rescore_ms"]), 4), "topk_host": round(float(last["topk_ms"]), 4)},
        }
        if world == 1 and not args.no_host_inclusive:
            out["host_inclusive"] = host_inclusive(swg, ctx, flat, off, K, cells_local)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(swg, q, flat, off, sc, lq)
        print(json.dumps(out), flush=True)

    db.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def host_inclusive(swg, ctx, flat, off, k, cells):
    """The same search once from HOST buffers, outside the timed region and never `value`: pack the
    sequences (host work), copy the packed shard over PCIe, build its per-database device tables,
    fill, and copy every score back -- what a caller pays who searches a database exactly once."""
    import torch
    t0 = time.perf_counter()
    db = swg.Database(flat, off)
    t1 = time.perf_counter()
    db.upload(ctx)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx.search(db, want_scores=True, k=k)
    t3 = time.perf_counter()
    ctx.search(db, want_scores=True, k=k)
    t4 = time.perf_counter()
    nbytes = int(db.packed_bytes)
    db.close()
    return {"pack_ms": round((t1 - t0) * 1e3, 2), "upload_ms": round((t2 - t1) * 1e3, 3),
            "upload_bytes": nbytes, "first_search_ms": round((t3 - t2) * 1e3, 3),
            "next_search_ms": round((t4 - t3) * 1e3, 3),
            "gcups_upload_and_first_search": round(cells / (t3 - t1) / 1e9, 1),
            "note": "the first search of a database builds its pair tokens on the host and uploads them; all scores are copied back"}


class TopKMerger:
    """Global top-K over the ranks' shards: every rank writes the 64-bit keys of its own K hits
    (score << 32 | ~index: larger = better, total order) into its K-slot segment of a zeroed
    n*K buffer; ONE max-all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) leaves the
    union on every rank, which then keeps the best K.  n*K*8 bytes: latency-bound."""

    def __init__(self, swg, k, rank, world, device):
        import torch
        self.swg, self.k, self.rank, self.world = swg, k, rank, world
        self.buf = torch.zeros(world * k, dtype=torch.int64, device=device)
        self.stage = torch.zeros(world * k, dtype=torch.int64)          # host staging (pinned on GPU runs)
        if device != "cpu":
            self.stage = self.stage.pin_memory()

    def merge(self, hits):
        keys = np.array([self.swg.hit_key(s, i) for s, i in hits] + [0] * (self.k - len(hits)), dtype=np.uint64)
        return self.merge_keys(keys)

    def merge_keys(self, keys):
        """keys: uint64[k] of this rank (zeros = no hit).  One all-reduce, then the best k of n*k."""
        import torch
        import torch.distributed as dist
        k = self.k
        self.stage.zero_()
        self.stage[self.rank * k:(self.rank + 1) * k] = torch.from_numpy(keys.view(np.int64))
        self.buf.copy_(self.stage, non_blocking=True)
        dist.all_reduce(self.buf, op=dist.ReduceOp.MAX)
        self.stage.copy_(self.buf)                      # D2H, synchronises
        return self.swg.topk_merge_keys(self.stage.numpy().view(np.uint64), k)


def cpu_baseline(swg, q, flat, off, sc, lq):
    """The reference's own fill (oracle/_ref: its alignment.c compiled from its sources,
    dispatched as its driver does) on this host's cores, over a bounded sample of the
    same database: whole 16-record groups, evenly spaced, about 3e10 cells."""
    orc = swg_loader.oracle()
    n = len(off) - 1
    groups = n // 16
    budget_cells = 3.0e10
    total_cells = float(lq) * float(off[-1])
    take = max(1, min(groups, int(groups * budget_cells / max(total_cells, 1.0))))
    sel = np.unique(np.linspace(0, groups - 1, take).astype(np.int64))
    lens = np.diff(off.astype(np.int64))
    table = sc.table()
    if orc.have_ref():
        batches = []
        cells = 0
        for g in sel:
            seqs = [flat[int(off[i]):int(off[i + 1])] for i in range(g * 16, g * 16 + 16)]
            batches.append(orc.make_batch16(seqs))
            cells += lq * int(lens[g * 16:g * 16 + 16].sum())
        # The reference takes omp_get_max_threads() threads (src/alignment_cmdline.c:341-347): all
        # hardware threads of the host.  This process may own fewer CPUs (cgroup quota, cpuset), so it is
        # timed with that many threads too and the better rate is the baseline.
        hw = int(orc.rlib().swref_max_threads())
        share = int(swg.lib.swg_host_threads())
        orc.ref_batches(q, batches[:min(len(batches), 64)], table, -2, -1)      # warm the pages
        runs = {}
        for t in sorted({hw, min(hw, share)}):
            _, secs = orc.ref_batches(q, batches, table, -2, -1, threads=t)
            runs[t] = cells / secs / 1e9
        best = max(runs, key=runs.get)
        return {"value": round(runs[best], 3), "unit": "GCUPS", "cores": int(best), "kind": "reference",
                "sample": "%d of %d 16-record batches of the same DB (%.3g real cells), reference "
                          "alignment_fill_matrices under its OpenMP dynamic dispatch, fill region only; "
                          "this process may use %d CPUs of the host's %d hardware threads; GCUPS by threads: %s"
                          % (len(batches), groups, cells, share, hw,
                             ", ".join("%d: %.1f" % (t, v) for t, v in sorted(runs.items())))}
    # no reference build on this box: time the scalar oracle instead (a port, much slower)
    idx = np.concatenate([np.arange(g * 16, g * 16 + 16) for g in sel[:max(1, len(sel) // 16)]])
    sub_off = np.zeros(len(idx) + 1, dtype=np.uint64)
    sub_off[1:] = np.cumsum(lens[idx])
    sub_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in idx])
    t0 = time.perf_counter()
    orc.score_db(q, sub_flat, sub_off, table, -2, -1)
    secs = time.perf_counter() - t0
    return {"value": round(lq * float(sub_off[-1]) / secs / 1e9, 3), "unit": "GCUPS",
            "cores": os.cpu_count(), "kind": "port",
            "sample": "%d sequences of the same DB, scalar int32 oracle with OpenMP" % len(idx)}


if __name__ == "__main__":
    main()
