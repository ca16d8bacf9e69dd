"""CPU, world_size 2, gloo: the N>1 path — weight broadcast and clip/file sharding (no data-path collective)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiotoken_amd.distributed import broadcast_weights, gather_scalars, shard_indices


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audiotoken_amd import weights as W
        w = None
        if rank == 0:
            w = {k: v for k, v in W.synth_encodec_weights(seed=0, with_decoder=False, n_codebooks=2).items()}
        got = broadcast_weights(w, torch.device("cpu"), dist)
        ref = W.synth_encodec_weights(seed=0, with_decoder=False, n_codebooks=2)
        assert list(got) == list(ref)
        assert all(np.array_equal(got[k], ref[k]) and got[k].shape == ref[k].shape for k in ref)
        mine = shard_indices(11, rank, world)
        allv = gather_scalars([float(len(mine)), float(sum(mine))], torch.device("cpu"), dist)
        assert sum(v[0] for v in allv) == 11 and sum(v[1] for v in allv) == sum(range(11))
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array(mine))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ok0.npy"), np.load(tmp_path / "ok1.npy")
    assert sorted(list(a) + list(b)) == list(range(11)) and len(a) - len(b) in (0, 1)


@pytest.mark.parametrize("n,world", [(0, 4), (3, 8), (512, 8), (10, 3)])
def test_shard_indices_partition(n, world):
    parts = [shard_indices(n, r, world) for r in range(world)]
    flat = [i for p in parts for i in p]
    assert flat == list(range(n))                      # contiguous blocks in rank order
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
