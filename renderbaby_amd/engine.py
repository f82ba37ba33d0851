"""Host-side mirror of the reference's renderer interface over the C ABI.

Names and semantics follow the reference so parity tests read like its own code:

* ``Change`` / ``RenderConfig`` / ``RenderConfigBuilder`` --
  crates/engine-config/src/render_config.rs:37-57,99-109,312-606
* ``Engine.new(rc)`` / ``Engine.render(rc)`` / ``Engine.frame_iterator(rc)`` --
  ``impl Renderer for Engine``, crates/engine-pathtracer/src/lib.rs:58-119
* ``Frame`` / ``FrameIterator.has_next/next/destroy`` --
  crates/frame-buffer/src/frame_iterator.rs:3-51

Everything that computes is in librenderbaby_hip.so; this file only marshals.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np

from . import abi
from ._lib import load


class RenderError(RuntimeError):
    """anyhow::Error of the reference; ``code`` is the rb_abi.h status."""

    def __init__(self, code, message):
        super().__init__(f"[{abi.ERR.get(code, code)}] {message}")
        self.code = code
        self.message = message


@dataclass
class Change:
    """``enum Change<T> { Keep, Create(T), Update(T), Delete }``"""
    tag: int
    value: Any = None

    @staticmethod
    def keep():
        return Change(abi.KEEP)

    @staticmethod
    def create(v):
        return Change(abi.CREATE, v)

    @staticmethod
    def update(v):
        return Change(abi.UPDATE, v)

    @staticmethod
    def delete():
        return Change(abi.DELETE)


_FIELDS = ("uniforms", "spheres", "uvs", "meshes", "lights", "bvh_nodes", "bvh_indices", "bvh_triangles",
           "textures")
_DTYPES = {"uniforms": abi.UNIFORMS, "spheres": abi.SPHERE, "uvs": np.float32, "meshes": abi.MESH,
           "lights": abi.POINT_LIGHT, "bvh_nodes": abi.BVH_NODE, "bvh_indices": np.uint32,
           "bvh_triangles": abi.GPU_TRIANGLE}


class RenderConfig:
    """``struct RenderConfig`` -- nine ``Change`` fields, default Keep."""

    def __init__(self, **kw):
        for f in _FIELDS:
            setattr(self, f, kw.get(f, Change.keep()))

    @staticmethod
    def builder():
        return RenderConfigBuilder()

    @staticmethod
    def from_scene(scene, create=True):
        """What generate_full_render_command_builder emits
        (scene_engine_adapter.rs:463-490): all ``*_create`` on the first render;
        afterwards Update for everything except the three BVH fields, which stay Create."""
        mk = Change.create if create else Change.update
        return RenderConfig(
            uniforms=mk(scene.uniforms), spheres=mk(scene.spheres), uvs=mk(scene.uvs), meshes=mk(scene.meshes),
            lights=mk(scene.lights), bvh_nodes=Change.create(scene.bvh_nodes),
            bvh_indices=Change.create(scene.bvh_indices), bvh_triangles=Change.create(scene.bvh_triangles),
            textures=mk(scene.textures))

    # ---- marshalling
    def to_c(self):
        """Returns (abi.Config, keepalive list)."""
        cfg = abi.Config()
        keep: List[Any] = []
        for f in _FIELDS:
            ch: Change = getattr(self, f)
            fld = abi.Field()
            fld.change = ch.tag
            fld.ptr = None
            fld.count = 0
            if ch.tag in (abi.CREATE, abi.UPDATE):
                if f == "textures":
                    texs = ch.value or []
                    arr = (abi.Texture * max(len(texs), 1))()
                    for i, (w, h, data) in enumerate(texs):
                        d = np.ascontiguousarray(data, dtype=np.uint32)
                        keep.append(d)
                        arr[i].width, arr[i].height, arr[i].rgba_data = int(w), int(h), d.ctypes.data
                    keep.append(arr)
                    fld.ptr = C.cast(arr, C.c_void_p).value if texs else None
                    fld.count = len(texs)
                else:
                    a = np.ascontiguousarray(np.atleast_1d(ch.value), dtype=_DTYPES[f])
                    keep.append(a)
                    fld.ptr = a.ctypes.data if a.size else None
                    fld.count = a.size
            setattr(cfg, f, fld)
        return cfg, keep


class RenderConfigBuilder:
    """``RenderConfigBuilder``: ``x(v)`` = Update, ``x_create(v)``, ``x_no_change()``, ``x_delete()``."""

    def __init__(self):
        self._rc = RenderConfig()

    def build(self):
        return self._rc


def _add_builder_methods():
    for f in _FIELDS:
        def upd(self, v, _f=f):
            setattr(self._rc, _f, Change.update(v))
            return self

        def cre(self, v, _f=f):
            setattr(self._rc, _f, Change.create(v))
            return self

        def keep(self, _f=f):
            setattr(self._rc, _f, Change.keep())
            return self

        def dele(self, _f=f):
            setattr(self._rc, _f, Change.delete())
            return self
        setattr(RenderConfigBuilder, f, upd)
        setattr(RenderConfigBuilder, f + "_create", cre)
        setattr(RenderConfigBuilder, f + "_no_change", keep)
        setattr(RenderConfigBuilder, f + "_delete", dele)


_add_builder_methods()


@dataclass
class Frame:
    """``struct Frame { width, height, pixels: Vec<u8> }`` (RGBA8, x mirrored, A = 255)."""
    width: int
    height: int
    pixels: np.ndarray  # uint8, (height, width, 4)

    def expected_size(self):
        return self.width * self.height * 4

    def validate(self):
        if self.pixels.size != self.expected_size():
            raise RenderError(0, f"Frame pixel size mismatch: expected {self.expected_size()} bytes, got {self.pixels.size}")


class Engine:
    """``engine_pathtracer::Engine`` for the HIP backend."""

    def __init__(self, rc: RenderConfig, device=-1, shard_rank=0, shard_count=1, stripe_rows=0,
                 passes_per_launch=0, kernel=abi.KERNEL_DEFAULT, stats=False, blocks_per_cu=0, color_budget_mib=0,
                 no_sphere_bvh=False, fast_bvh=False, lds_mode=0, device_bvh=False, no_leaf_stepping=False,
                 device_lbvh=False, reference_walk=False, host_bvh=False, devices=None, gather_peer_copy=False,
                 no_run_ahead=False, own_tree=False, skip_near_degenerate=False, queue_batch=0, chunk_walk=False,
                 sphere_tree=None, chunk_tree=None):
        """``devices`` (list of HIP ordinals): one handle over several devices of this process
        (rb_create_multi): rows sharded in stripes, one RCCL gather per delivered frame."""
        self._lib = load()
        cfg, keep = rc.to_c()
        opt = abi.Options()
        opt.device = device
        opt.shard_rank, opt.shard_count, opt.stripe_rows = shard_rank, shard_count, stripe_rows
        opt.passes_per_launch = passes_per_launch
        opt.kernel = kernel
        # fast_bvh / host_bvh: the library's own tree, built on the host; device_bvh / device_lbvh: built on the
        # device; own_tree: the library's tree with the builder left to the library (by triangle count)
        host_bvh = host_bvh or fast_bvh
        own_tree = own_tree or host_bvh or device_bvh or device_lbvh
        opt.flags = (abi.FLAG_STATS if stats else 0) | (abi.FLAG_NO_SPHERE_BVH if no_sphere_bvh else 0) \
            | (abi.FLAG_FAST_BVH if own_tree else 0) | (abi.FLAG_DEVICE_BVH if (device_bvh or device_lbvh) else 0) \
            | (abi.FLAG_DEVICE_LBVH if device_lbvh else 0) | (abi.FLAG_REFERENCE_WALK if reference_walk else 0) \
            | (abi.FLAG_HOST_BVH if host_bvh else 0) | (abi.FLAG_GATHER_PEER_COPY if gather_peer_copy else 0) \
            | (abi.FLAG_NO_RUN_AHEAD if no_run_ahead else 0) | (abi.FLAG_SKIP_NEAR_DEGENERATE if skip_near_degenerate else 0) \
            | (abi.FLAG_CHUNK_WALK if chunk_walk else 0) \
            | {None: 0, "host": abi.FLAG_SPHERE_TREE_HOST, "device": abi.FLAG_SPHERE_TREE_DEVICE}[sphere_tree] \
            | {None: 0, "host": abi.FLAG_CHUNK_TREE_HOST, "device": abi.FLAG_CHUNK_TREE_DEVICE}[chunk_tree]   # who builds the chunked walk's tree
        opt._reserved[0] = blocks_per_cu
        opt._reserved[1] = color_budget_mib
        opt._reserved[2] = queue_batch   # items a wave reserves per queue atomic (0 = the launcher's choice)
        opt._reserved[3] = 1 if no_leaf_stepping else 0   # ablation: per-segment traversal for multi-node trees
        opt._reserved[4] = int(lds_mode)   # LDS staging of small meshes: 0 = when it fits, 1 = never
        if devices is not None:
            devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            self._h = self._lib.rb_create_multi(C.byref(cfg), C.byref(opt), devs, len(devices))
        else:
            self._h = self._lib.rb_create_ex(C.byref(cfg), C.byref(opt))
        del keep
        if not self._h:
            msg = self._lib.rb_last_error(None)
            raise RenderError(abi.ERR and 16, (msg or b"rb_create failed").decode())
        self.shard_count = max(shard_count, 1)
        self.comm_rank = None   # set by comm_init_rank: rank 0 then receives whole frames

    @classmethod
    def new(cls, rc, **kw):
        return cls(rc, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != abi.RB_OK:
            raise RenderError(rc, (self._lib.rb_last_error(self._h) or b"").decode())

    # ---- Renderer
    def update(self, rc: RenderConfig):
        cfg, keep = rc.to_c()
        self._check(self._lib.rb_update(self._h, C.byref(cfg)))
        del keep

    def size(self):
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rb_get_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def _frame_shape(self):
        w, h = self.size()
        if self.shard_count > 1 and self.comm_rank is None:
            h = self.local_rows()[1]
        return w, h

    # ---- one process per device: the stripes are gathered to rank 0 inside rb_render / rb_iter_next
    @staticmethod
    def comm_available():
        """Raises unless this process can load RCCL (no communicator id is made: rb_comm_available)."""
        rc = load().rb_comm_available()
        if rc:
            raise RenderError(rc, (load().rb_last_error(None) or b"").decode())

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
        rc = load().rb_comm_unique_id(buf)
        if rc:
            raise RenderError(rc, (load().rb_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init_rank(self, comm_id: bytes, rank: int, nranks: int):
        buf = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._check(self._lib.rb_comm_init_rank(self._h, buf, rank, nranks))
        self.comm_rank = rank

    def comm_info(self):
        """{"rccl_ranks", "rccl_rank", "gather_ms"}: the communicator as RCCL itself reports it (0 ranks when the
        exchange does not go through RCCL) and this rank's share of the last gather."""
        n, r, ms = C.c_uint32(), C.c_uint32(), C.c_float()
        self._check(self._lib.rb_comm_info(self._h, C.byref(n), C.byref(r), C.byref(ms)))
        return {"rccl_ranks": n.value, "rccl_rank": r.value, "gather_ms": ms.value}

    def render(self, rc: RenderConfig) -> Frame:
        cfg, keep = rc.to_c()
        self._check(self._lib.rb_update(self._h, C.byref(cfg)))
        del keep
        w, h = self._frame_shape()
        if self.comm_rank not in (None, 0):   # a non-root rank of a process group: its stripes go to rank 0
            self._check(self._lib.rb_render(self._h, None))
            return None
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.rb_render(self._h, out.ctypes.data))
        return Frame(w, h, out)

    def render_current(self) -> Frame:
        """rb_render without an update: all passes of the scene the engine already holds."""
        w, h = self._frame_shape()
        if self.comm_rank not in (None, 0):
            self._check(self._lib.rb_render(self._h, None))
            return None
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.rb_render(self._h, out.ctypes.data))
        return Frame(w, h, out)

    def frame_iterator(self, rc: RenderConfig, passes_per_frame: int = 1) -> "FrameIterator":
        """``passes_per_frame`` > 1 (extension): one frame per that many samples instead of per sample."""
        cfg, keep = rc.to_c()
        self._check(self._lib.rb_iter_begin(self._h, C.byref(cfg)))
        del keep
        self._check(self._lib.rb_iter_set_passes_per_frame(self._h, passes_per_frame))
        return FrameIterator(self)

    # ---- lower-level control (bench, tests, multi-GPU)
    def clear(self):
        self._check(self._lib.rb_clear(self._h))

    def dispatch(self, first_pass, n_passes):
        self._check(self._lib.rb_dispatch(self._h, first_pass, n_passes))

    def reserve(self, n_passes):
        """rb_reserve: what a dispatch of n_passes passes would allocate lazily (no tracing)."""
        self._check(self._lib.rb_reserve(self._h, n_passes))

    def sync(self):
        self._check(self._lib.rb_sync(self._h))

    def read_rgba(self):
        w, h = self._frame_shape()
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.rb_read_rgba(self._h, out.ctypes.data))
        return out

    def read_accumulation(self):
        w, h = self._frame_shape()
        out = np.empty((h, w, 4), dtype=np.float32)
        self._check(self._lib.rb_read_accumulation(self._h, out.ctypes.data))
        return out

    def device_rgba(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.rb_device_rgba(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def local_rows(self):
        r, pr = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rb_local_rows(self._h, C.byref(r), C.byref(pr)))
        return r.value, pr.value

    def global_row(self, local_row):
        g = C.c_uint32()
        self._check(self._lib.rb_global_row(self._h, local_row, C.byref(g)))
        return g.value

    def stats(self):
        s = abi.Stats()
        self._check(self._lib.rb_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self._lib.rb_reset_stats(self._h))

    def last_kernel_name(self):
        return (self._lib.rb_last_kernel_name(self._h) or b"").decode()

    def fast_bvh_builder(self):
        """("host-sah" | "device-lbvh" | "", build milliseconds) of the tree RB_FLAG_FAST_BVH walks."""
        ms = C.c_float()
        name = (self._lib.rb_fast_bvh_builder(self._h, C.byref(ms)) or b"").decode()
        return name, ms.value

    def sphere_tree_builder(self):
        """("device-median" | "host-median" | "", build milliseconds) of the library's sphere tree (> 64 spheres)."""
        ms = C.c_float()
        name = (self._lib.rb_sphere_tree_builder(self._h, C.byref(ms)) or b"").decode()
        return name, ms.value

    def chunk_tree_builder(self):
        """("device" | "host" | "", build milliseconds) of the chunked walk's tree (multi-node meshes, the default walk)."""
        ms = C.c_float()
        name = (self._lib.rb_chunk_tree_builder(self._h, C.byref(ms)) or b"").decode()
        return name, ms.value

    def debug_chunk_tree(self):
        """The tree the engine walks, read back and checked (rb_debug_engine_chunk_tree): dict of its census, or None."""
        out = (C.c_uint64 * 6)()
        self._check(self._lib.rb_debug_engine_chunk_tree(self._h, out))
        if not out[0]:
            return None
        return dict(nodes=out[1], positions=out[2], depth=out[3], chunks=out[4], unbounded=out[5])

    def last_dispatch_ms(self):
        ms = C.c_float()
        self._check(self._lib.rb_last_dispatch_ms(self._h, C.byref(ms)))
        return ms.value


class FrameIterator:
    """``RaytracerFrameIterator`` (lib.rs:127-234)."""

    def __init__(self, engine: Engine):
        self._e = engine

    def has_next(self) -> bool:
        return bool(self._e._lib.rb_iter_has_next(self._e._h))

    def next(self) -> Frame:
        w, h = self._e._frame_shape()
        if self._e.comm_rank not in (None, 0):
            self._e._check(self._e._lib.rb_iter_next(self._e._h, None))
            return None
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._e._check(self._e._lib.rb_iter_next(self._e._h, out.ctypes.data))
        return Frame(w, h, out)

    def destroy(self):
        self._e._lib.rb_iter_destroy(self._e._h)

    def __iter__(self):
        while self.has_next():
            yield self.next()


class PinnedFrame:
    """A page-locked RGBA8 frame buffer (rb_host_alloc) as a numpy array: read-backs into it are DMA copies."""

    def __init__(self, width, height):
        self._lib = load()
        self._p = self._lib.rb_host_alloc(width * height * 4)
        if not self._p:
            raise MemoryError("rb_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * (width * height * 4)).from_address(self._p)).reshape(height, width, 4)

    def free(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.rb_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_name(device=-1):
    buf = C.create_string_buffer(256)
    rc = load().rb_device_name(device, buf, 256)
    return buf.value.decode() if rc == 0 else "unknown"
