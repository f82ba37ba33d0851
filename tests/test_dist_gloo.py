"""The N>1 path on CPU: world_size-2/3 gloo process groups exercise the stripe
geometry (rb_shard_layout / rb_shard_global_row), the gather and the
assembly; each rank fills its local stripe buffer with the ORACLE's rows (the
HIP kernels need a GPU), so the assembled frame must equal the whole-frame
oracle render bit for bit -- the property that makes the multi-GPU frame
identical to the single-GPU one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from renderbaby_amd import dist as rdist
from renderbaby_amd import scenes
from tests import _oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, stripe_rows, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene = scenes.cornell(w, h, 2, 3)
        owned, padded = rdist.shard_layout(h, rank, world, stripe_rows)
        rows = rdist.global_rows(h, rank, world, stripe_rows)
        local = np.zeros((padded, w, 4), dtype=np.uint8)
        n_owned = 0
        for lr, gr in enumerate(rows):
            if gr < h:
                _, _, rgba, _ = _oracle.render(scene, rows=(int(gr), int(gr) + 1), threads=1)
                local[lr] = rgba[gr]
                n_owned += 1
        assert n_owned == owned
        t = torch.from_numpy(local)
        gl = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gl, dst=0)
        if rank == 0:
            frame = rdist.assemble(gl, h, stripe_rows).numpy()
            np.save(out_path, frame)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,stripe_rows,h", [(2, 16, 37), (2, 1, 9), (3, 4, 22)])
def test_sharded_gather_equals_whole_frame(tmp_path, world, stripe_rows, h):
    w = 12
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), stripe_rows, w, h, out), nprocs=world, join=True)
    frame = np.load(out)
    _, _, rgba, _ = _oracle.render(scenes.cornell(w, h, 2, 3))
    assert frame.shape == rgba.shape and np.array_equal(frame, rgba)


def test_layout_partitions_every_row_exactly_once():
    for h, world, sr in [(1080, 8, 16), (1080, 8, 1), (37, 2, 16), (5, 4, 2), (4096, 8, 16), (1, 3, 16)]:
        seen = np.zeros(h, dtype=np.int32)
        pads = set()
        for r in range(world):
            owned, padded = rdist.shard_layout(h, r, world, sr)
            rows = rdist.global_rows(h, r, world, sr)
            assert len(rows) == padded
            seen[rows[rows < h]] += 1
            assert (rows < h).sum() == owned
            pads.add(padded)
        assert np.all(seen == 1) and len(pads) == 1
    assert rdist.shard_layout(100, 0, 1) == (100, 100)
    with pytest.raises(ValueError):
        rdist.shard_layout(100, 2, 2)
