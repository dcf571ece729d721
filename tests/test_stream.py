"""Multi-GPU plumbing on CPU: world_size-2 gloo processes shard a stream by
sequence, run a stand-in compute (the oracle -- test-only injection) and hand
the keypoints back to rank 0.  Gathered result must equal the single-process
result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hand_pose_sl_amd.stream import ShardedStream, gather_to_root, shard_bounds, shard_sizes
from conftest import load_golden


def test_shard_bounds_partition():
    for n in (0, 1, 2, 7, 8, 9, 250, 2000, 2001):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sz = shard_sizes(n, w)
            assert sum(sz) == n and max(sz) - min(sz) <= 1
    assert shard_sizes(2000, 8) == [250] * 8          # BASELINE config 4
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_seq, out_path, chunk=2):
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rec = load_golden("cfg2_b64_t200_u55")
        x = torch.from_numpy(rec["x"][:n_seq, :40])           # short sequences keep the CPU test quick
        model = lambda t: torch.from_numpy(oracle.forward_from_state(t.numpy(), rec["state"]))  # noqa: E731
        stream = ShardedStream(model, max_batch=3)
        lo, hi = stream.local_slice(n_seq)
        y = stream.run(x[lo:hi], n_seq, gather=True)
        if rank == 0:
            assert y.shape == (n_seq, 40, 21, 2)
            np.save(out_path, y.numpy())
        else:
            assert y is None
        y_local = stream.run(x[lo:hi], n_seq, gather=False)
        assert y_local.shape[0] == hi - lo
        y_pipe = stream.run_pipelined(x[lo:hi], n_seq, chunk=chunk)  # piecewise hand-back, same result
        if rank == 0:
            assert torch.equal(y_pipe, y)
        else:
            assert y_pipe is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_seq", [7, 1, 8])
def test_gloo_world2_gather_equals_single_process(tmp_path, n_seq):
    import oracle
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_seq, out), nprocs=2, join=True)
    rec = load_golden("cfg2_b64_t200_u55")
    ref = oracle.forward_from_state(rec["x"][:n_seq, :40], rec["state"])
    assert np.array_equal(np.load(out), ref)


@pytest.mark.parametrize("world,n_seq,chunk", [(3, 10, 3), (8, 13, 1), (8, 5, 2), (8, 64, 3)])
def test_gloo_many_ranks_uneven_shards(tmp_path, world, n_seq, chunk):
    """The 8-rank shape of the node the bench targets, rehearsed on CPU: root posts up to 7 grouped
    receives per piece; shards are unequal (13 over 8 = 2,2,2,2,2,1,1,1), some ranks own nothing
    (5 over 8), piece counts differ between ranks (chunk does not divide the shards), and a 3-rank
    world with a piece length that divides nothing.  Gathered and pipelined results == one process."""
    import oracle
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_seq, out, chunk), nprocs=world, join=True)
    rec = load_golden("cfg2_b64_t200_u55")
    ref = oracle.forward_from_state(rec["x"][:n_seq, :40], rec["state"])
    assert np.array_equal(np.load(out), ref)


def test_single_process_gather_is_identity():
    y = torch.arange(12.).reshape(2, 3, 2)
    assert gather_to_root(y, 2) is y
