"""N>1 path on CPU: two gloo ranks shard the windows, each runs its block through a stand-in
compute, rank 0 gathers; the result equals the single-process result.  (The compute here is the
oracle -- test infrastructure -- because this container has no GPU; the sharding/gather code is the
product code that bench.py uses on N GPUs.)"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import radio_mapper_amd as rm
    from radio_mapper_amd.shard import gather_lags, window_shard
    from oracle import xcorr_ref as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    iq, _ = rm.synth.make_windows(5, 3, 256, 2.048e6, seed=3)
    s, c = window_shard(iq.shape[0], rank, world)
    li, lf, pk = orc.xcorr_batch_fast(iq[s:s + c])
    out = gather_lags(li, lf.astype(np.float32), pk)
    dist.barrier()
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    import radio_mapper_amd as rm
    from oracle import xcorr_ref as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    iq, _ = rm.synth.make_windows(5, 3, 256, 2.048e6, seed=3)
    li, lf, pk = orc.xcorr_batch_fast(iq)
    assert np.array_equal(got[0], li)
    assert np.array_equal(got[1], lf.astype(np.float32))
    assert np.array_equal(got[2], pk)
