"""CPU, world sizes 2 and 4 over gloo: the multi-GPU path of bench.py without GPUs.  Each rank
generates and packs only its own bins of ONE global database (the sharded generator and
swg_db_pack_shard: the partition code bench.py --gpus N runs), scores them with the oracle (standing
in for the GPU), and the ranks merge their top-K lists with the same single max-all-reduce bench.py uses."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, tmp, n_seqs):
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
        flat, off = swg.synth_db(11, n_seqs, max_len=200)          # the whole database: for the oracle only
        want = orc.score_db(q, flat, off, sc.table(), -2, -1)
        # what bench.py --gpus N does on every rank: generate only this rank's bins of the one global
        # database and pack them as a shard that was cut elsewhere
        sh = swg.synth_db_shard(11, n_seqs, rank, world, max_len=200)
        assert sh["residues_total"] == int(off[-1])
        shard = swg.Database(sh["flat"], sh["offsets"], index=sh["index"], n_total=n_seqs)
        cut = swg.Database(flat, off, rank, world)                 # the same shard cut from the whole
        assert np.array_equal(shard.order(), cut.order()) and shard.residues == cut.residues
        for i in range(0, len(sh["index"]), 97):                   # the residues are the global database's
            g = int(sh["index"][i])
            assert np.array_equal(sh["flat"][int(sh["offsets"][i]):int(sh["offsets"][i + 1])],
                                  flat[int(off[g]):int(off[g + 1])])
        mine = shard.order()
        mine = mine[mine != 0xFFFFFFFF]                            # empty slots of the last bin
        assert len(mine) == shard.count
        local = sorted(((-int(want[i]), int(i)) for i in mine))[:50]
        hits = [(-s, i) for s, i in local]                         # what swg_search would return
        merged = bench.TopKMerger(swg, 50, rank, world, "cpu").merge(hits)
        assert merged == orc.topk(want, 50), rank
        # the shards partition the database
        import torch
        seen = torch.zeros(n_seqs, dtype=torch.int64)
        mine = mine[mine != 0xFFFFFFFF]
        seen[torch.from_numpy(mine.astype(np.int64))] = 1
        dist.all_reduce(seen)
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_seqs", [(2, 700), (4, 1100), (4, 300)])
def test_shards_and_topk_merge_over_gloo(tmp_path, world, n_seqs):
    """(2, 700): three bins per rank; (4, 1100): uneven shards (3, 2, 2, 2 bins, the last one partly
    empty); (4, 300): three bins for four ranks -- one rank holds nothing and contributes no hit."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() * 7 + world * 31 + n_seqs) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), n_seqs), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))
