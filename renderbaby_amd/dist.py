"""Multi-GPU: row-stripe sharding of one frame + one gather of the RGBA8 rows.

A pixel depends only on (x, y, width, pass, scene) because the seed uses the
global pixel index (shader.wgsl:682,693-694), so stripes render independently
and bit-identically to the single-GPU frame; the only exchange step is the
gather of the finished RGBA8 stripes to rank 0 (RCCL over xGMI on GPUs, gloo in
the CPU tests).  Sharding by samples is deliberately not offered: it would need
a float reduction and change the summation order.

One process per GPU; ``torch.distributed`` is plumbing (process group + the
gather), the rendering is librenderbaby_hip.so.
"""
import ctypes as C

import numpy as np

from ._lib import load

DEFAULT_STRIPE_ROWS = 8   # = rb_internal.hpp kDefaultStripeRows = bench.py --stripe-rows: profiles/r04_shard_rehearsal.txt


def shard_layout(height, rank, world, stripe_rows=DEFAULT_STRIPE_ROWS):
    """-> (owned_rows, padded_rows) via rb_shard_layout (pure host function)."""
    owned, padded = C.c_uint32(), C.c_uint32()
    rc = load().rb_shard_layout(height, rank, world, stripe_rows, C.byref(owned), C.byref(padded))
    if rc:
        raise ValueError(f"rb_shard_layout({height}, {rank}, {world}, {stripe_rows}) -> {rc}")
    return owned.value, padded.value


def global_rows(height, rank, world, stripe_rows=DEFAULT_STRIPE_ROWS):
    """Global row index of every local (padded) row of `rank`; entries >= height are padding."""
    _, padded = shard_layout(height, rank, world, stripe_rows)
    lib = load()
    return np.array([lib.rb_shard_global_row(rank, world, stripe_rows, r) for r in range(padded)], dtype=np.int64)


def assemble(gathered, height, stripe_rows=DEFAULT_STRIPE_ROWS):
    """gathered: list (one per rank, rank order) of [padded_rows, width, C] tensors in
    local stripe order -> [height, width, C] frame.  Pure tensor ops (CPU or GPU)."""
    import torch
    world = len(gathered)
    if world == 1:
        return gathered[0][:height]
    g = torch.stack(list(gathered), dim=0)                      # [G, P, W, C]
    G, P, W, Cn = g.shape
    per = P // stripe_rows
    g = g.reshape(G, per, stripe_rows, W, Cn).permute(1, 0, 2, 3, 4)  # stripe s = per_idx*G + rank
    return g.reshape(per * G * stripe_rows, W, Cn)[:height]


class _DevView:
    """Zero-copy view of library-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class ShardedRenderer:
    """One rank's share of a frame.  ``step()`` renders this rank's stripes for all
    samples and gathers the RGBA8 rows to rank 0 (returns the assembled
    [H, W, 4] uint8 tensor there, None elsewhere)."""

    def __init__(self, scene, rank=0, world=1, device=0, stripe_rows=DEFAULT_STRIPE_ROWS, kernel=0,
                 passes_per_launch=0, stats=False, host_gather=False, library_gather=False, **engine_kw):
        """``library_gather``: the exchange runs inside librenderbaby_hip.so (rb_comm_init_rank: one RCCL
        gather per frame, de-interleave and read-back on rank 0); torch.distributed only hands the communicator
        id round.  Otherwise the stripes are gathered with torch.distributed (``host_gather``: as CPU tensors,
        the gloo rehearsal)."""
        import torch
        from .engine import Engine, RenderConfig
        self.torch = torch
        self.scene, self.rank, self.world, self.device = scene, rank, world, device
        self.stripe_rows = stripe_rows
        self.host_gather = host_gather  # gloo rehearsal: gather CPU copies instead of device memory
        self.library_gather = library_gather and world > 1
        rc = RenderConfig.from_scene(scene)
        self.engine = Engine.new(rc, device=device, shard_rank=rank, shard_count=world, stripe_rows=stripe_rows,
                                 kernel=kernel, passes_per_launch=passes_per_launch, stats=stats, **engine_kw)
        self.engine.update(rc)
        self.width, self.height = scene.width, scene.height
        self.owned, self.padded = shard_layout(self.height, rank, world, stripe_rows)
        self.gather_note = "torch.distributed gather"
        if self.library_gather:
            # the communicator id travels by torch.distributed; every rank must agree on whether the library's
            # RCCL path came up, otherwise all fall back to the torch.distributed gather below
            import torch.distributed as dist
            # ncclCommInitRank is collective: a rank whose librccl does not load must not leave the others waiting
            # inside it.  Every rank first probes RCCL on its own (rb_comm_available: dlopen + dlsym, no id -- an id made
            # on a rank that then discards it would leave a bootstrap socket and a waiting thread behind), the ranks
            # agree on the outcome, rank 0 alone makes the id, and only then the communicator is made.
            ok, why, my_id = 1, "", None
            try:
                Engine.comm_available()
                if rank == 0:
                    my_id = Engine.comm_unique_id()
            except Exception as ex:   # librccl not loadable on this rank
                ok, why = 0, str(ex)
            flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{device}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                box = [my_id if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                try:
                    self.engine.comm_init_rank(box[0], rank, world)
                except Exception as ex:
                    ok, why = 0, str(ex)
            else:
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{device}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.gather_note = "one RCCL gather per frame inside librenderbaby_hip.so (rb_comm_init_rank)"
                self.local = self.gather_list = None
                return
            # a rank that did create its communicator keeps rendering its stripes only: start over without it
            self.library_gather = False
            self.gather_note = "torch.distributed gather (library RCCL path unavailable: %s)" % (why or "another rank failed")
            self.engine.close()
            self.engine = Engine.new(rc, device=device, shard_rank=rank, shard_count=world, stripe_rows=stripe_rows,
                                     kernel=kernel, passes_per_launch=passes_per_launch, stats=stats, **engine_kw)
            self.engine.update(rc)
        ptr, nbytes = self.engine.device_rgba()
        rows = self.padded if world > 1 else self.height
        assert nbytes >= rows * self.width * 4
        self.local = torch.as_tensor(_DevView(ptr, (rows, self.width, 4)), device=f"cuda:{device}")
        self.gather_list = None
        if world > 1 and rank == 0:
            like = self.local.cpu() if host_gather else self.local
            self.gather_list = [torch.empty_like(like) for _ in range(world)]

    def render_local(self):
        e = self.engine
        e.clear()
        e.dispatch(0, self.scene.total_samples)
        e.sync()  # the engine has its own stream; the gather below runs on torch's

    def step(self):
        if self.library_gather:   # rb_render: all passes, the RCCL gather, the frame on rank 0 (numpy) / None
            f = self.engine.render_current()
            return None if f is None else f.pixels
        if self.world == 1:   # rb_render: all passes and the frame read back, as on the root of a sharded run
            return self.engine.render_current().pixels
        self.render_local()
        import torch.distributed as dist
        # the rows live in the library's own allocation; hand the collective a torch-owned copy (1 MB per
        # rank) so nothing depends on how RCCL treats memory it did not see allocated
        src = self.local.cpu() if self.host_gather else self.local.clone()
        dist.gather(src, self.gather_list, dst=0)
        if self.rank == 0:
            return assemble(self.gather_list, self.height, self.stripe_rows)
        return None

    def close(self):
        self.local = None
        self.engine.close()
