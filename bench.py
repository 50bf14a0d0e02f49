#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman database search on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` (N > 1) started WITHOUT a launcher's environment starts its N ranks itself: before anything touches
torch or HIP it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free> bench.py ...` as a child process, passes the child's one JSON line through and exits with
its return code (`--spawn-dry-run` prints that command instead of running it).

A step = one pass of the hot path (swg_search: int16 fill, saturation re-score when possible,
top-K) over one resident synthetic database of the shapes of SURVEY 8d.

One GPU, no --config: the JSON line's headline (`value`, `config`, `roofline`, `cpu_baseline`, exactly
--steps timed steps after --warmup untimed ones) is config 4's ONE 10M-sequence database (3000 aa query) searched
WHOLE on this GPU through the sharded path with one shard -- the very workload `--gpus N` deals over N ranks, so
value(N) / (N * value(1)) compares one workload with itself (it is the configuration BASELINE.json quotes its scaling
target on, and it fits one GPU: 3.8 GB of residues).  The `configs` object holds one block of the same shape for each
of configs 3 (round 3's headline), 2, 4 (one GPU's eighth of the 10M-sequence database), "4_relatives" (that share
with a family of 30-70 %-identity relatives of the query: the f16 flag-and-re-run route, first search and steady
state), 5 and 5's stress variant, "peptides" (not a BASELINE shape: 2 million sequences of 20-40 residues, which the
cost model hands to the systolic engine), and "4_whole" repeats the headline's block.

N > 1 (one rank per GPU, torch.distributed over RCCL; SWG_BENCH_FORCE_DIST=1 rehearses the path with
one rank): config 4 as ONE 10M-sequence database.  Every rank derives the same global length order,
generates only the residues of its own bins (round-robin by bin: swg_synth_db_shard), packs them
(swg_db_pack_shard) and searches them with no data-path collective; the only exchange is one RCCL
max-all-reduce of the n*K top-K keys per step.  Total work is fixed as N grows: `scaling: "strong"`,
value = cells of the whole database * steps / max-over-ranks time.  Once, outside the timed region,
the merged top-K is checked: every candidate's score against the int32 oracle, a seeded sample of
each shard against the K-th key, and the all-reduce merge against a plain gather-and-sort.

The JSON line carries BASELINE.json's metric (GCUPS = lq * sum(len) / t / 1e9 over real residues),
the HBM roofline of the fill kernel beside the roof that binds it (VALU issue), and the reference's
own AVX2+OpenMP fill timed on this host (oracle/_ref, built from the reference's sources) as
`cpu_baseline`.  The oracle is used as the checker only, never inside a timed region.
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
    4: dict(lq=3000, n=1250000, matrix="BLOSUM62", n_full=10000000),   # n: one GPU's eighth of the 10M-sequence DB
    # one GPU's eighth of SURVEY 8d's 10M sequences, 1 % of them near-copies of the query
    5: dict(lq=8192, n=1250000, matrix="BLOSUM62", similar=0.01, n_full=10000000),
    # SURVEY 8d's stress variant of config 5: 100 000 sequences, every one a near-copy of the query (block "5_stress")
    6: dict(lq=8192, n=100000, matrix="BLOSUM62", similar=1.0),
    # config 4's share with a family of relatives of the query in it (block "4_relatives"): a seeded 0.5 % of the
    # sequences are copies of the 3000-aa query at 30-70 % identity, scores about 3 800 .. 10 000 -- above the f16
    # cells' ceiling (4096), below int16's: the database on which the flag-and-re-run route is really taken
    7: dict(lq=3000, n=1250000, matrix="BLOSUM62", similar=0.005, subst=0.3, subst_hi=0.7),
    # not a BASELINE configuration: a database of very short sequences (2 million peptides of 20-40 residues, block
    # "peptides") -- the shape the lane groups are worst at and the cost model hands to the systolic engine
    8: dict(lq=128, n=2000000, matrix="BLOSUM62", shape=dict(median=29.0, sigma_ln=0.25, min_len=20, max_len=40)),
}
# BASELINE.json names config 5 "forcing 16->32-bit rescore": its block carries, beside the library's own choice
# (the wide int16 form, exact to 65535: nothing left to re-score), the same database with plain int16 cells,
# every flagged sequence re-scored by the int32 work-queue kernel.
CONFIG_LEGS = {5: (("rescore", {"wide16": 0}),), 6: (("rescore", {"wide16": 0}),)}
SHARDED = 4             # the configuration every --gpus N runs as ONE database dealt by bins (N = 1: the headline)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak
METRIC = "GCUPS (DP cell updates/s) at 1/2/4/8 MI355X; bit-exact max scores vs CPU ref"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0,
                    help="0 (default): config 4 as ONE 10M-sequence database dealt over the N GPUs (N = 1: whole on the one GPU, "
                         "the headline, followed by blocks for configs 3, 2, 4-share, 4 with relatives, 5).  C: that configuration alone")
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
    ap.add_argument("--force-bits", type=int, default=0, help="experiment: 32 = the exact int32 path for everything")
    ap.add_argument("--autotune", action="store_true",
                    help="time the best-ranked geometries on the device at the first search (default: the cost model alone, "
                         "so that every rank of a multi-GPU run uses the same plan)")
    ap.add_argument("--no-autotune", action="store_true", help="(the default; kept for the sweep scripts)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every step before queuing the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true",
                    help="skip the one-off search from host buffers (PCIe-inclusive figure, reported beside value)")
    ap.add_argument("--no-verify", action="store_true", help="skip the top-K check against the oracle (outside the timed region)")
    ap.add_argument("--no-scaling-reference", action="store_true", help="(accepted and ignored: the 10M-sequence database is the headline now)")
    ap.add_argument("--only-headline", action="store_true", help="one GPU: the headline configuration alone")
    ap.add_argument("--whole", action="store_true",
                    help="one GPU, with --config 4 or 5: the configuration's WHOLE 10M-sequence database through the sharded path "
                         "with one shard -- the N = 1 point of the curve `--gpus N --config C` continues")
    ap.add_argument("--max-len", type=int, default=0, help="diagnostic: clamp the sequence lengths here (default 5000)")
    ap.add_argument("--uniform-len", type=int, default=0,
                    help="diagnostic: every sequence gets this length (no length tail)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--last-pass", type=int, default=-1, help="experiment: 0 = the last pass of a long query with the other passes' geometry")
    ap.add_argument("--f16", type=int, default=-1, help="experiment: 0 = int16 cells only, 2 = f16 cells whenever the gap scores allow")
    ap.add_argument("--wide16", type=int, default=-1, help="experiment: 0 = plain int16 cells + int32 re-score instead of the wide form")
    ap.add_argument("--gapopen", type=int, default=-2, help="experiment: gap_open (the configurations use the reference's default -2)")
    ap.add_argument("--gapextend", type=int, default=-1, help="experiment: gap_extend (default -1)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="experiment: any swg_set_option key (repeatable), e.g. --opt batch=0")
    ap.add_argument("--spawn-dry-run", action="store_true",
                    help="--gpus N without a launcher: print the launch command as JSON instead of running it")
    return ap.parse_args()


class Env:
    """rank / world, torch.distributed (only when the collective path is taken) and the library."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # SWG_BENCH_FORCE_DIST=1: take the collective path even with one rank (rehearsal on a 1-GPU box)
        self.use_dist = self.world > 1 or os.environ.get("SWG_BENCH_FORCE_DIST") == "1"
        if self.world != args.gpus:
            args.gpus = self.world  # (the launcher's world size rules; main() spawns the ranks when there is no launcher)
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        # SWG_BENCH_BACKEND=gloo (rehearsal only): the ranks' collectives go through gloo on host tensors and the ranks
        # share the GPUs that are there (rank r on device r mod the device count) -- N ranks of the real sharded
        # search on a one-GPU box, everything but RCCL itself (which the one-rank rehearsal covers)
        self.backend = os.environ.get("SWG_BENCH_BACKEND", "nccl")
        self.coll_device = "cuda" if self.backend == "nccl" else "cpu"
        self.device_index = self.local_rank if self.backend == "nccl" else self.local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.device_index)
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        lib_path = os.path.join(ROOT, "seq-align-gpu_amd", "libswg.so")
        if not os.path.exists(lib_path):
            if self.rank == 0:
                swg_loader.build_module().build()
            if self.use_dist:
                dist.barrier()
        self.swg = swg_loader.load()

    def fence(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if self.use_dist:
            self.dist.destroy_process_group()


def make_context(env, q, sc):
    a = env.args
    ctx = env.swg.Context(env.device_index)
    ctx.set_scoring(sc, a.gapopen, a.gapextend)
    ctx.set_query(q)
    ctx.set_option("cols_per_wave", a.cols)
    ctx.set_option("max_waves", a.max_waves)
    ctx.set_option("workgroups", a.workgroups)
    ctx.set_option("engine", a.engine)
    ctx.set_option("group_lanes", a.group)
    ctx.set_option("long_split", a.long_split)
    ctx.set_option("long_cols", a.long_cols)
    ctx.set_option("long_group", a.long_group)
    ctx.set_option("force_bits", a.force_bits)
    ctx.set_option("autotune", 1 if a.autotune and not a.no_autotune else 0)
    ctx.set_option("work_queue", 0 if a.static_streams else 1)
    if a.side_readout >= 0:
        ctx.set_option("side_readout", a.side_readout)
    if a.long_helps:
        ctx.set_option("long_helps", 1)
    if a.prio_share >= 0:
        ctx.set_option("prio_share", a.prio_share)
    if a.f16 >= 0:
        ctx.set_option("f16", a.f16)
    if a.last_pass >= 0:
        ctx.set_option("last_pass", a.last_pass)
    if a.wide16 >= 0:
        ctx.set_option("wide16", a.wide16)
    for kv in a.opt:
        k, v = kv.split("=", 1)
        ctx.set_option(k, int(v))
    return ctx


def run_config(env, cnum, steps, warmup, sharded=False, n_override=0, host_inclusive_leg=False, cpu_leg=False, legs=(),
               first_search_leg=False):
    """One configuration: generate, pack, upload, warm up, time exactly `steps` steps between fences.
    sharded: the database is ONE global database dealt by bins over the ranks (strong scaling), else an
    independent database on this rank.  legs: ((name, {option: value}), ...): the same resident database timed
    again with those options set (block[name]).  Returns the block dict (rank 0) or None."""
    a, swg = env.args, env.swg
    cfg = dict(CONFIGS[cnum])
    if a.lq:
        cfg["lq"] = a.lq
    if a.nseq:
        cfg["n"] = a.nseq
    lq = cfg["lq"]
    n = n_override or cfg["n"]
    K = a.topk
    sc = swg.load_scoring(cfg["matrix"])
    seed = 0x5EED0000 + cnum
    q = swg.synth_query(seed, lq)
    t_gen0 = time.perf_counter()
    if sharded:
        sh = swg.synth_db_shard(seed, n, env.rank, env.world, query=q if cfg.get("similar") else None,
                                fraction=cfg.get("similar", 0.0), subst=0.05)
        flat, off, index = sh["flat"], sh["offsets"], sh["index"]
        residues_total = sh["residues_total"]
    else:
        if cfg.get("similar"):
            flat, off, _ = swg.synth_db(seed, n, query=q, fraction=cfg["similar"], subst=cfg.get("subst", 0.05),
                                        subst_hi=cfg.get("subst_hi"))
        elif a.uniform_len:
            flat, off = swg.synth_db(seed, n, min_len=a.uniform_len, max_len=a.uniform_len)
        elif a.max_len:
            flat, off = swg.synth_db(seed, n, max_len=a.max_len)
        elif cfg.get("shape"):
            flat, off = swg.synth_db(seed, n, **cfg["shape"])
        else:
            flat, off = swg.synth_db(seed, n)
        index = None
        residues_total = int(off[-1])
    t_gen = time.perf_counter() - t_gen0

    ctx = make_context(env, q, sc)
    t0 = time.perf_counter()
    hdb = swg.Database(flat, off, index=index, n_total=n) if sharded else swg.Database(flat, off)
    t_pack = time.perf_counter() - t0
    db = hdb.upload(ctx)
    residues = int(db.residues)
    # setup, untimed like the upload: the first search of a query length plans the kernel geometry
    # for this database (and with --autotune times the best-ranked plans on this device)
    t0 = time.perf_counter()
    _, _, first_st = ctx.search(db, want_scores=False, k=K)
    first_wall_ms = (time.perf_counter() - t0) * 1e3

    merger = TopKMerger(swg, K, env.rank, env.world, env.coll_device) if env.use_dist else None
    # Steps are software-pipelined two deep: search i+1 is queued on the GPU before the host finishes
    # search i (top-K read-out, and for N > 1 the all-reduce merge).  The library runs a search's
    # top-K kernels and read-out on a stream of their own, beside the start of the next fill; deeper
    # queues measured slower.  Every step still does all of its work inside the timed region.
    depth = 1 if a.no_pipeline else a.depth

    def finish(ticket):
        if env.use_dist:
            keys, st = ctx.search_end_keys(ticket)
            return merger.merge_keys(keys), st
        _, hits, st = ctx.search_end(ticket)
        return hits, st

    def run_steps(count, record):
        pending = []
        for _ in range(count):
            pending.append(ctx.search_begin(db, K))
            if len(pending) >= depth:
                record(*finish(pending.pop(0)))
        while pending:
            record(*finish(pending.pop(0)))

    def timed(n_steps, n_warm):
        """n_warm untimed steps, then exactly n_steps between fences: (elapsed s, fill ms, total ms, stats, last hits)"""
        run_steps(n_warm, lambda hits, st: None)
        f_ms, t_ms, sts, hits_box = [], [], [], []

        def record(hits, st):
            f_ms.append(st["fill_ms"])
            t_ms.append(st["total_ms"])
            sts.append(st)
            hits_box[:] = [hits]

        env.fence()
        t0 = time.perf_counter()
        run_steps(n_steps, record)
        env.fence()
        return time.perf_counter() - t0, f_ms, t_ms, sts, hits_box

    elapsed, fill_ms, total_ms, lasts, last_hits = timed(steps, warmup)
    last = lasts[-1]
    elapsed_own = elapsed

    torch, dist = env.torch, env.dist
    per_rank = None
    if env.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=env.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own timed region and mean fill time, so that an uneven deal of the bins would show
        mine = torch.tensor([elapsed_own * 1e3 / steps, float(np.mean(fill_ms)), float(db.residues)], dtype=torch.float64, device=env.coll_device)
        allr = [torch.zeros(3, dtype=torch.float64, device=env.coll_device) for _ in range(env.world)]
        dist.all_gather(allr, mine)
        allr = np.array([r.cpu().numpy() for r in allr])
        per_rank = {"ms_per_step": {"min": round(float(allr[:, 0].min()), 4), "max": round(float(allr[:, 0].max()), 4)},
                    "fill_ms": {"min": round(float(allr[:, 1].min()), 4), "max": round(float(allr[:, 1].max()), 4),
                                "by_rank": [round(float(v), 4) for v in allr[:, 1]]},
                    "residues": {"min": int(allr[:, 2].min()), "max": int(allr[:, 2].max())}}
    if sharded:
        cells_total = float(lq) * float(residues_total)          # ONE database, whatever the number of ranks
    elif env.use_dist:
        c = torch.tensor([float(lq) * residues], dtype=torch.float64, device=env.coll_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells_total = float(c.item())
    else:
        cells_total = float(lq) * residues
    cells_local = float(lq) * residues

    verify = None
    if not a.no_verify:
        verify = verify_topk(env, ctx, db, q, sc, flat, off, index, K, merger, last_hits[0])

    block = None
    if env.rank == 0:
        gcups = cells_total * steps / elapsed / 1e9
        whole = sharded or n == cfg.get("n_full")   # (the whole database of a configuration whose usual block is one GPU's share)
        roofline, dtype = roofline_of(a, last, fill_ms, cells_local, cnum, whole)
        step_fill_ms = float(np.mean(fill_ms))
        if sharded:
            workload = ("config %d: 1 query (%d aa) vs ONE %d-seq synthetic protein DB dealt by bins over %d GPU(s), "
                        "%s, gaps %d/%d, global top-%d by one RCCL all-reduce" % (cnum, lq, n, env.world, cfg["matrix"], a.gapopen, a.gapextend, K))
        else:
            workload = ("config %s: 1 query (%d aa) vs %d-seq synthetic protein DB%s, %s, gaps %d/%d, top-%d"
                        % ({6: "5 (stress variant)", 7: "4 (relatives)", 8: "peptides"}.get(cnum, str(cnum)), lq, n, " per GPU" if env.world > 1 else "", cfg["matrix"], a.gapopen, a.gapextend, K))
            if cnum == 4 and not n_override:
                workload += " (one GPU's eighth of the 10M-sequence database)"
            if cfg.get("shape"):
                workload += ", sequences of %d-%d residues" % (cfg["shape"]["min_len"], cfg["shape"]["max_len"])
            if cfg.get("subst_hi"):
                workload += ", %g %% of the sequences relatives of the query (%d-%d %% identity)" % (
                    100 * cfg["similar"], round(100 * (1 - cfg["subst_hi"])), round(100 * (1 - cfg["subst"])))
            elif cfg.get("similar"):
                workload += ", %g %% of the sequences near-copies of the query (5 %% substitutions)" % (100 * cfg["similar"])
        block = {
            "value": round(gcups, 3), "unit": "GCUPS", "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "dtype": dtype,
            "config": dict({
                "workload": workload, "lq": lq, "n_seqs": n, "n_seqs_this_gpu": int(db.count),
                "residues_total": int(residues_total), "residues_this_gpu": residues, "matrix": cfg["matrix"]},
                **plan_of(last)),
            "roofline": roofline,
            "kernel_ms": {"fill": round(step_fill_ms, 4), "search_total": round(float(np.mean(total_ms)), 4),
                          "rescore": round(float(np.mean([st["rescore_ms"] for st in lasts])), 4),
                          "topk_host": round(float(last["topk_ms"]), 4)},
            "setup_s": {"generate": round(t_gen, 3), "pack": round(t_pack, 3)},
        }
        if first_search_leg:
            # The first search of a (query, database): it plans, builds the pair tokens, and -- on a database whose
            # relatives the f16 cells flag -- learns what the later searches' plans assume (how long the flagged list
            # is; whether the f16 cells pay at all).  Device times of that one search beside the steady state above.
            block["first_search"] = {"fill_ms": round(float(first_st["fill_ms"]), 4), "rescore_ms": round(float(first_st["rescore_ms"]), 4),
                                     "device_total_ms": round(float(first_st["total_ms"]), 4), "wall_ms": round(first_wall_ms, 3),
                                     "n_rescored": int(first_st["n_rescored"]), "cell_form": int(first_st["cell_form"]),
                                     "gcups_device": round(cells_local / (float(first_st["total_ms"]) * 1e-3) / 1e9, 1)}
            block["steady_state"] = {"n_rescored": int(last["n_rescored"]), "cell_form": int(last["cell_form"]),
                                     "fill_ms": round(step_fill_ms, 4),
                                     "rescore_ms": round(float(np.mean([st["rescore_ms"] for st in lasts])), 4)}
        if per_rank is not None:
            block["per_rank"] = per_rank
        if verify is not None:
            block["verify"] = verify
    # the same resident database again under other options (e.g. config 5 with the flagged-int32 re-score instead
    # of the wide form): timed like the main leg, every score checked against the main leg's through the top-K
    for name, opts in legs:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.search(db, want_scores=False, k=K)           # plans (and, with hints unknown, learns) outside the timed region
        l_steps = max(2, steps)
        l_elapsed, l_fill, l_total, l_sts, l_hits = timed(l_steps, min(warmup, 2))
        if env.rank == 0:
            l_roof, l_dtype = roofline_of(a, l_sts[-1], l_fill, cells_local, cnum, sharded or n == cfg.get("n_full"))
            block[name] = {"options": dict(opts), "value": round(cells_total * l_steps / l_elapsed / 1e9, 3), "unit": "GCUPS",
                           "steps": l_steps, "ms_per_step": round(l_elapsed / l_steps * 1e3, 4), "dtype": l_dtype,
                           "n_rescored": int(l_sts[-1]["n_rescored"]), "plan": plan_of(l_sts[-1]), "roofline": l_roof,
                           "kernel_ms": {"fill": round(float(np.mean(l_fill)), 4),
                                         "rescore": round(float(np.mean([st["rescore_ms"] for st in l_sts])), 4),
                                         "search_total": round(float(np.mean(l_total)), 4)},
                           "top_k_equals_main_leg": list(l_hits[0]) == list(last_hits[0])}
            if not block[name]["top_k_equals_main_leg"]:
                raise SystemExit("bench.py: leg %s returned another top-K than the main leg" % name)
    if env.rank == 0:
        if host_inclusive_leg and env.world == 1 and not a.no_host_inclusive:
            block["host_inclusive"] = host_inclusive(env, ctx, flat, off, K, cells_local)
        if cpu_leg and not a.no_cpu_baseline:
            block["cpu_baseline"] = cpu_baseline(swg, q, flat, off, sc, lq)
    db.close()
    ctx.close()
    return block


def plan_of(last):
    """What the library chose for a search, from its stats record."""
    return {"cols_per_wave": last["cols_per_wave"], "waves": last["waves"], "passes": last["passes"],
            "last_pass_cols": last["last_pass_cols"],
            "workgroups": last["workgroups"], "n_rescored": last["n_rescored"],
            "engine": {1: "systolic", 2: "diagonal"}.get(last["engine"]), "group_lanes": last["group_lanes"],
            "cells": "packed f16, three-operand maxima (systolic engine: taken where no score can reach 4096)"
                     if last["engine"] == 1 and last["path_bits"] == 16 and last["cell_form"] == 2 else
                     {0: "packed int16", 1: "packed int16, wide form (to 65535)",
                      2: "packed f16, three-operand maxima (exact below 4096; flagged pairs run again on int16 cells, int32 only beyond those)",
                      4: "packed f16 for sequences under %d rows, wide int16 form for the longer ones "
                         "(what the f16 cells flag all the same: the wide form again)" % last["split_rows"],
                      5: "packed f16 for sequences under %d rows, packed int16 for the longer ones "
                         "(from 32767 up: int32 re-score)" % last["split_rows"]}.get(last["cell_form"])
                     if last["path_bits"] == 16 else "int32",
            "streams": last["streams"], "long_pairs": last["long_pairs"],
            "long_cols_per_lane": last["long_cols_per_lane"], "long_streams": last["long_streams"],
            "work_queue": bool(last["work_queue"]), "classes_overlapped": last["classes_overlapped"],
            "cells_padded_over_real": round(last["cells_padded"] / max(1, last["cells"]), 4)}


def _traffic_table(path):
    """profiles/traffic.json (HBM bytes per launch from the PMC counters of an earlier run of the same command,
    tools/profile_bench.sh) and a short digest of the file, so that the line says where the figure comes from."""
    try:
        raw = open(path, "rb").read()
        import hashlib
        return json.loads(raw), "%s@sha256:%s" % (os.path.relpath(path, ROOT), hashlib.sha256(raw).hexdigest()[:16])
    except Exception:
        return {}, None


def roofline_of(a, last, fill_ms, cells_local, cnum, sharded):
    """The `roofline` object of one leg, and its dtype.  A query of several passes is one launch of the fill
    kernel per pass (times the segments of a very large database): the figures are per launch -- what rocprofv3's
    per-kernel average and the PMC counters are -- so the step's fill time and its algorithmic bytes are divided
    by the number of launches."""
    q16 = last["engine"] == 2 and last["path_bits"] == 16
    launches = max(1, int(last["fill_launches"])) if last["engine"] == 2 and last["work_queue"] else 1
    step_fill_ms = float(np.mean(fill_ms))
    k_ms = step_fill_ms / launches
    bytes_alg = int(last["bytes_alg"]) // launches
    achieved = bytes_alg / (k_ms * 1e-3) / 1e9
    form = int(last["cell_form"]) if q16 else -1
    # (counters are collected per configuration and cell form as bench.py --config C runs it; a leg without a
    # measurement of its own gets null rather than a neighbour's figure)
    table, source = _traffic_table(a.traffic_json)
    sys16 = last["engine"] != 2 and last["path_bits"] == 16     # the systolic engine's 16-bit cells
    key = "config%d%s%s" % (cnum, "_whole" if sharded else "", {1: "_wide", 2: "_f16", 4: "_split", 5: "_split16"}.get(form, "") if q16
                            else ("_systolic_f16" if int(last["cell_form"]) == 2 else "_systolic") if sys16 else "_int32")
    traffic = table.get(key, {}).get("hbm_bytes_per_launch")
    # The binding roof is VALU issue, reported beside the (by construction tiny) HBM fraction: one wave64 packed
    # instruction per SIMD every 4 cycles (16 lanes per cycle) at 2.4 GHz.  Instructions per cell: 5 for the packed
    # int16 cells (10 per column pair), 4.25 for the packed f16 cells (8.5), 8 for the int32 work-queue kernel,
    # 12 for the term-by-term int32 kernels.  An isolated stream of packed instructions measures 4.4-4.56 cycles
    # with 4 waves per SIMD (tools/valu_rate.hip); the fill kernels get to 4.05.
    split = None
    if q16 and form in (4, 5):
        # both 16-bit forms in one search: half of the launches ran the long sequences on the wide form, the other
        # half -- the dominant kernel, which this object describes -- the rest on the f16 cells; the library times
        # the two parts apart (swg_stats.fill_f16_ms) and says how many cells each took
        share = float(last["fill_f16_ms"]) / max(1e-9, float(last["fill_ms"]))
        cells_f16 = int(last["cells_f16"])
        wide_ms, wide_cells = step_fill_ms * (1.0 - share), cells_local - cells_f16
        n_f16 = max(1, int(last["fill_f16_launches"]))
        n_wide = max(1, launches - n_f16)
        split = {"kernel": "swg_diag_dyn_kernel<K=%d,%s>" % (last["cols_per_wave"], "wide" if form == 4 else "int16"), "launches_per_step": n_wide,
                 "kernel_ms": round(wide_ms / n_wide, 4), "cells_share": round(wide_cells / max(1, cells_local), 4),
                 "kernel_gcups": round(wide_cells / (wide_ms * 1e-3) / 1e9, 2), "instr_per_cell": 5.0}
        bytes_alg = int(int(last["bytes_alg"]) * (cells_f16 / max(1, cells_local))) // n_f16
        step_fill_ms *= share
        cells_local = cells_f16
        launches = n_f16
        k_ms = step_fill_ms / launches
        achieved = bytes_alg / (k_ms * 1e-3) / 1e9
        ops_per_cell = 4.25
        kname = "swg_diag_dyn_kernel<K=%d,f16>" % last["cols_per_wave"]
    elif q16:
        ops_per_cell = 4.25 if form == 2 else 5.0
        kname = ("swg_diag_dyn_kernel<K=%d,%s>" if last["work_queue"] else "swg_diag_kernel<K=%d,%s>") % (
            last["cols_per_wave"], {0: "int16", 1: "wide", 2: "f16"}[form])
    elif last["engine"] == 2 and last["work_queue"]:
        exact = a.gapopen > 0 or a.gapextend > 0    # (gap scores the reduced algebra cannot express: the exact cells)
        ops_per_cell, kname = (12.0 if exact else 8.0), "swg_diag32q_kernel<K=%d%s>" % (last["cols_per_wave"], ",exact" if exact else "")
    elif last["engine"] == 2:
        ops_per_cell, kname = 12.0, "swg_diag32_kernel"
    else:
        sys_f16 = last["path_bits"] == 16 and int(last["cell_form"]) == 2       # the systolic engine on packed-f16 cells
        ops_per_cell = 4.25 if sys_f16 else 5.0 if last["path_bits"] == 16 else 12.0
        kname = "swg_fill_kernel<CellsSF16<%d>>" % last["cols_per_wave"] if sys_f16 else "swg_fill_kernel<CellsI%d<%d>>" % (last["path_bits"], last["cols_per_wave"])
        if sys_f16:
            form = 2
    if q16 and form not in (4, 5) and int(last["last_pass_cols"]) > 0 and int(last["passes"]) > 1:
        # the last pass of a long query runs an instantiation with fewer columns per lane: kernel_ms is the mean over
        # all the launches of one search's fill, i.e. rocprofv3's two per-kernel averages weighted by their calls
        per_pass = launches // int(last["passes"])
        kname += " x%d + <K=%d> x%d (the last pass) per search" % (launches - per_pass, last["last_pass_cols"], per_pass)
    simds = 256 * 4
    kernel_gcups = cells_local / (step_fill_ms * 1e-3) / 1e9
    peak_issue = simds * 64 / 4.0 * 2.4e9 / ops_per_cell / 1e9
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "traffic_source": (source + "#" + key) if traffic is not None else None,
        "kernel": kname, "kernel_ms": round(k_ms, 4), "bytes_alg_per_launch": bytes_alg,
        "launches_per_step": launches,
        "binding_roof": {"bound": "valu_issue", "kernel_gcups": round(kernel_gcups, 2),
                         "instr_per_cell": ops_per_cell, "cycles_per_wave_instr": 4.0, "clock_ghz": 2.4,
                         "peak_gcups_issue": round(peak_issue, 1),
                         "frac_of_issue_peak": round(kernel_gcups / peak_issue, 4)},
    }
    if split is not None:
        split["peak_gcups_issue"] = round(simds * 64 / 4.0 * 2.4e9 / split["instr_per_cell"] / 1e9, 1)
        split["frac_of_issue_peak"] = round(split["kernel_gcups"] / split["peak_gcups_issue"], 4)
        roofline["binding_roof"]["cells_share"] = round(1.0 - split["cells_share"], 4)
        roofline["binding_roof"]["other_kernel"] = split
    dtype = ("f16" if form == 2 else "f16+int16" if form in (4, 5) else "int16") if last["path_bits"] == 16 else "int32"
    return roofline, dtype


def verify_topk(env, ctx, db, q, sc, flat, off, index, K, merger, timed_hits):
    """Outside the timed region, once: the (merged) top-K against the int32 oracle.
    (a) every candidate this rank contributed: oracle score == GPU score;
    (b) a seeded sample of this rank's sequences: none of them beats the rank's K-th key without being in
        its list (the list really is the top of the shard);
    (c) N > 1: the all-reduce-max merge == gathering every rank's keys and sorting them, identical on all
        ranks, and identical to what the timed steps returned."""
    swg = env.swg
    orc = swg_loader.oracle()
    table = sc.table()
    keys, _ = ctx.search_keys(db, K)
    keys = keys[keys != 0]
    local = [swg.key_hit(int(k)) for k in keys]                   # (score, global index), best first
    n_local = len(off) - 1
    pos_of = (lambda g: int(np.searchsorted(index, g))) if index is not None else (lambda g: int(g))

    def oracle_scores(positions):
        lens = [int(off[p + 1]) - int(off[p]) for p in positions]
        sub_off = np.zeros(len(positions) + 1, dtype=np.uint64)
        sub_off[1:] = np.cumsum(lens)
        sub_flat = (np.concatenate([flat[int(off[p]):int(off[p + 1])] for p in positions])
                    if positions else np.zeros(0, np.int8))
        return orc.score_db(q, sub_flat, sub_off, table, env.args.gapopen, env.args.gapextend)

    pos = [pos_of(g) for _, g in local]
    want = oracle_scores(pos)
    ok_a = all(int(w) == s for w, (s, _) in zip(want, local))
    rng = np.random.default_rng(12345 + env.rank)
    sample = [int(p) for p in rng.choice(n_local, size=min(64, n_local), replace=False)] if n_local else []
    ss = oracle_scores(sample)
    kth = int(keys[-1]) if len(keys) >= min(K, n_local) and len(keys) else 0
    listed = set(g for _, g in local)
    ok_b = True
    for p, s in zip(sample, ss):
        g = int(index[p]) if index is not None else p
        if swg.hit_key(int(s), g) > kth and g not in listed:
            ok_b = False
    res = {"candidates_checked": len(local), "candidates_equal_oracle": bool(ok_a),
           "sampled": len(sample), "sample_consistent_with_kth": bool(ok_b)}
    ok = ok_a and ok_b
    if env.use_dist:
        torch, dist = env.torch, env.dist
        mine = np.zeros(K, dtype=np.uint64)
        mine[:len(keys)] = keys
        merged = merger.merge_keys(mine)                          # the path the timed steps take
        gathered = [torch.zeros(K, dtype=torch.int64, device=env.coll_device) for _ in range(env.world)]
        dist.all_gather(gathered, torch.from_numpy(mine.view(np.int64)).to(env.coll_device))
        allk = np.concatenate([g.cpu().numpy().view(np.uint64) for g in gathered])
        allk = np.sort(allk[allk != 0])[::-1][:K]
        plain = [swg.key_hit(int(k)) for k in allk]
        ok_c = merged == plain and (timed_hits is None or list(timed_hits) == plain)
        flag = torch.tensor([1 if (ok and ok_c) else 0], dtype=torch.int64, device=env.coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res.update({"merge_equals_gather_and_sort": bool(ok_c), "all_ranks_ok": bool(int(flag.item()) == 1),
                    "best": plain[:3]})
        ok = ok and ok_c and int(flag.item()) == 1
    else:
        res["best"] = local[:3]
        if timed_hits is not None:
            res["timed_steps_returned_the_same"] = list(timed_hits) == local
            ok = ok and res["timed_steps_returned_the_same"]
    res["ok"] = bool(ok)
    if not ok:
        raise SystemExit("bench.py: top-K verification FAILED on rank %d: %s" % (env.rank, json.dumps(res)))
    return res


def host_inclusive(env, ctx, flat, off, k, cells):
    """The same search once from HOST buffers, outside the timed region and never `value`: pack the
    sequences (host work), copy the residue bytes over PCIe, build the per-database device tables on
    the device, fill, and copy every score back -- what a caller pays who searches a database exactly once."""
    swg = env.swg
    t0 = time.perf_counter()
    db = swg.Database(flat, off)
    t1 = time.perf_counter()
    db.upload(ctx)
    env.torch.cuda.synchronize()
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
            "note": "upload = residue bytes + 16 B per sequence; the first search builds the pair tokens on the device "
                    "from them and plans the geometry; all scores are copied back"}


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def available_cpus():
    """CPUs this process may use, whatever OMP_NUM_THREADS says (a launcher sets it to 1 for its ranks): the affinity
    mask, capped by a cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(swg, q, flat, off, sc, lq):
    """The reference's own fill (oracle/_ref: its alignment.c compiled from its sources,
    dispatched as its driver does) on this host's cores, over a bounded sample of the
    same database: whole 16-record groups, evenly spaced, about 3e10 cells; and on ONE thread over an
    eighth of that sample (BASELINE.md section 3 asks for both, with the CPU model)."""
    orc = swg_loader.oracle()
    n = len(off) - 1
    groups = n // 16
    budget_cells = 3.0e10
    total_cells = float(lq) * float(off[-1])
    take = max(1, min(groups, int(groups * budget_cells / max(total_cells, 1.0))))
    sel = np.unique(np.linspace(0, groups - 1, take).astype(np.int64))
    lens = np.diff(off.astype(np.int64))
    table = sc.table()
    model = cpu_model()
    if orc.have_ref():
        batches, bcells = [], []
        for g in sel:
            seqs = [flat[int(off[i]):int(off[i + 1])] for i in range(g * 16, g * 16 + 16)]
            batches.append(orc.make_batch16(seqs))
            bcells.append(lq * int(lens[g * 16:g * 16 + 16].sum()))
        cells = int(sum(bcells))
        # The reference takes omp_get_max_threads() threads (src/alignment_cmdline.c:341-347): all
        # hardware threads of the host.  This process may own fewer CPUs (cgroup quota, cpuset), so it is
        # timed with that many threads too and the better rate is the baseline.
        # (under a launcher OMP_NUM_THREADS is 1 for every rank; the baseline runs on rank 0 alone, outside the timed
        # region, while the other ranks wait: it takes the CPUs the process may use, not the launcher's setting)
        share = max(int(swg.lib.swg_host_threads()), available_cpus())
        hw = max(int(orc.rlib().swref_max_threads()), share)
        orc.ref_batches(q, batches[:min(len(batches), 64)], table, -2, -1)      # warm the pages
        runs = {}
        for t in sorted({hw, min(hw, share)}):
            _, secs = orc.ref_batches(q, batches, table, -2, -1, threads=t)
            runs[t] = cells / secs / 1e9
        one = batches[::8]
        _, secs1 = orc.ref_batches(q, one, table, -2, -1, threads=1)
        one_thread = sum(bcells[::8]) / secs1 / 1e9
        best = max(runs, key=runs.get)
        return {"value": round(runs[best], 3), "unit": "GCUPS", "cores": int(best), "kind": "reference",
                "cpu_model": model, "one_thread_gcups": round(one_thread, 3),
                "sample": "%d of %d 16-record batches of the same DB (%.3g real cells; one thread: every 8th of them), "
                          "reference alignment_fill_matrices under its OpenMP dynamic dispatch, fill region only, built by "
                          "oracle/Makefile with -O3 -march=x86-64-v3 -mavx2 -fopenmp (the reference's Makefile says -march=native: "
                          "the library is built off the GPU box, and the kernel is explicit AVX2 intrinsics either way); "
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
            "cores": os.cpu_count(), "kind": "port", "cpu_model": model,
            "sample": "%d sequences of the same DB, scalar int32 oracle with OpenMP" % len(idx)}


def line_from(block, env, scaling):
    """The driver's one-line contract from a configuration's block."""
    out = {"metric": METRIC, "value": block["value"], "unit": "GCUPS", "n_gpus": env.world,
           "steps": block["steps"], "warmup": block["warmup"], "ms_per_step": block["ms_per_step"],
           "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": block["dtype"],
           "data": "synthetic", "config": block["config"], "roofline": block["roofline"],
           "kernel_ms": block["kernel_ms"]}
    for k in ("verify", "host_inclusive", "cpu_baseline", "per_rank"):
        if k in block:
            out[k] = block[k]
    return out


_REAL_STDOUT = None


def emit(line):
    """The ONE line of the contract, on the process's real stdout."""
    data = (json.dumps(line) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def spawn_command(args, argv, port):
    """The launch `--gpus N` makes for itself when no launcher has set RANK / WORLD_SIZE: one rank per GPU of this
    node under torch.distributed.run (its children are fresh processes: nothing here has touched torch or HIP yet)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [
                x for x in argv if x != "--spawn-dry-run"]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(args, argv):
    """Runs the ranks as child processes, passes their one JSON line through and returns their exit code."""
    import subprocess
    cmd = spawn_command(args, argv, free_port())
    if args.spawn_dry_run:
        print(json.dumps({"spawn": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (the host driver supports dmabuf IPC only: RCCL needs this)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)    # stderr passes straight through
    out, _ = proc.communicate()
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode == 0 and len(lines) != 1:
        sys.stderr.write("bench.py: the ranks printed %d JSON lines instead of one\n" % len(lines))
        return 1
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return proc.returncode


def main():
    global _REAL_STDOUT
    args = parse_args()
    # (SWG_BENCH_FORCE_SPAWN=1: also for one GPU -- with SWG_BENCH_FORCE_DIST=1 that is the whole N > 1 path,
    # launcher, RCCL communicator and relay included, rehearsed on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("SWG_BENCH_FORCE_SPAWN") == "1") and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    # Libraries chat on stdout (RCCL prints a version banner at communicator creation): everything but the
    # result line goes to stderr, at file-descriptor level, so that stdout carries exactly one JSON line.
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    env = Env(args)
    K, W = args.steps, args.warmup
    if env.use_dist:
        # one global database dealt by bins; total work fixed as the number of GPUs grows
        cnum = args.config or SHARDED
        n_full = args.nseq or CONFIGS[cnum].get("n_full", CONFIGS[cnum]["n"])
        block = run_config(env, cnum, K, W, sharded=True, n_override=n_full, cpu_leg=True)
        if env.rank == 0:
            emit(line_from(block, env, "strong"))
    elif args.config and args.whole and CONFIGS[args.config].get("n_full"):
        block = run_config(env, args.config, K, W, sharded=True, n_override=args.nseq or CONFIGS[args.config]["n_full"], cpu_leg=True)
        out = line_from(block, env, "strong")
        out["configs"] = {"%d_whole" % args.config: {k: v for k, v in block.items() if k != "cpu_baseline"}}
        emit(out)
    elif args.config:
        block = run_config(env, args.config, K, W, host_inclusive_leg=True, cpu_leg=True,
                           legs=CONFIG_LEGS.get(args.config, ()))
        out = line_from(block, env, "strong")
        out["configs"] = {str(args.config): {k: v for k, v in block.items() if k not in ("cpu_baseline", "host_inclusive")}}
        emit(out)
    else:
        # The headline first (exactly K timed steps after W warm-up steps): config 4's ONE 10M-sequence database,
        # whole, on this GPU -- through the sharded path with one shard, i.e. the very workload `--gpus N` deals over
        # N ranks, so that value(N) / (N * value(1)) compares one workload with itself.  Then the other shapes with
        # step counts scaled to their step time so the whole run stays within minutes.
        blocks = {}
        head = run_config(env, SHARDED, K, W, sharded=True, n_override=CONFIGS[SHARDED]["n_full"], cpu_leg=True)
        out = line_from(head, env, "strong")
        if not args.only_headline:
            blocks["3"] = run_config(env, 3, K, W)
            # (a step of config 2 is 2 ms: W such steps are over before the clocks have settled after the host-side
            # set-up of the block, so this block warms up for 30 steps)
            blocks["2"] = run_config(env, 2, K, max(W, 30), host_inclusive_leg=True)
            blocks["4"] = run_config(env, 4, max(2, K // 5), min(W, 2))
            blocks["4_relatives"] = run_config(env, 7, max(2, K // 5), min(W, 2), first_search_leg=True)
            blocks["5"] = run_config(env, 5, 2, 1, legs=CONFIG_LEGS[5])
            blocks["5_stress"] = run_config(env, 6, 2, 1, legs=CONFIG_LEGS[6])
            blocks["peptides"] = run_config(env, 8, K, max(W, 30))
            if "host_inclusive" in blocks["2"]:
                out["host_inclusive"] = blocks["2"]["host_inclusive"]      # quoted on config 2, as in round 1
        blocks["4_whole"] = {k: v for k, v in head.items() if k != "cpu_baseline"}
        out["configs"] = blocks
        emit(out)
    env.close()


if __name__ == "__main__":
    main()
