"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py without GPUs.  Each rank packs
its shard of one database (round-robin bins), scores it with the oracle (standing in for the
GPU), and the ranks merge their top-K lists with the same single max-all-reduce bench.py uses."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import swg_loader
    import bench
    swg, orc = swg_loader.load(), swg_loader.oracle()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = swg.load_scoring("BLOSUM62")
        q = swg.synth_query(11, 64)
        flat, off = swg.synth_db(11, 700, max_len=200)
        want = orc.score_db(q, flat, off, sc.table(), -2, -1)
        shard = swg.Database(flat, off, rank, world)
        mine = shard.order()
        local = sorted(((-int(want[i]), int(i)) for i in mine))[:50]
        hits = [(-s, i) for s, i in local]                         # what swg_search would return
        merged = bench.TopKMerger(swg, 50, rank, world, "cpu").merge(hits)
        assert merged == orc.topk(want, 50), rank
        # the shards partition the database
        import torch
        seen = torch.zeros(700, dtype=torch.int64)
        seen[torch.from_numpy(mine.astype(np.int64))] = 1
        dist.all_reduce(seen)
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_topk_merge(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
