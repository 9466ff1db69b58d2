"""Host-side mirror of libmil.so's multi-GPU layer (include/mil.h, mil_mg_*): strips of one frame / one frame per GPU,
gathered to GPU 0 by grouped RCCL send / recv issued inside the library.

The reference renders on one device (/root/reference/renderer/src/render/vulkan.rs:106-108; lighting draw
render/mod.rs:189-219); these calls replace that draw for N devices.  Two process models:

  MultiGpu.single_process(devices)   one process drives every GPU (ncclCommInitAll)
  MultiGpu.per_process(world, rank, device, id)   one process per GPU (torchrun): rank 0 makes the id with
                                     MultiGpu.unique_id(), the host distributes it (bench.py: one torch.distributed
                                     broadcast), every rank joins with ncclCommInitRank

torch only owns device memory and, in the second model, carries the 128-byte id between the processes.
"""
import ctypes

import numpy as np

from . import lighting, ubo
from .lighting import Extras, GBuffer, MilError, Shadow, _np_ptr

TRANSPORT_RCCL, TRANSPORT_PEER_COPY = 0, 1
FLAG_LOOPBACK = 1
UNIQUE_ID_BYTES = 128

EXPORTS = ['mil_mg_create', 'mil_mg_unique_id', 'mil_mg_create_rank', 'mil_mg_destroy', 'mil_mg_world', 'mil_mg_local_ranks',
           'mil_mg_rccl_ranks', 'mil_mg_last_rccl_error', 'mil_mg_last_hip_error', 'mil_mg_rows_of_rank',
           'mil_mg_frame_rows_of_rank', 'mil_mg_render_strips', 'mil_mg_render_frames', 'mil_mg_sync', 'mil_mg_reserve', 'mil_mg_debug_delay_gather', 'mil_mg_debug_delay_senders']


class Config(ctypes.Structure):            # mil_mg_config
    _fields_ = [('max_point_lights', ctypes.c_uint32), ('band_rows', ctypes.c_uint32), ('chunks', ctypes.c_uint32),
                ('transport', ctypes.c_uint32), ('flags', ctypes.c_uint32)]


class RankIO(ctypes.Structure):            # mil_mg_rank_io
    _fields_ = [('gb', GBuffer), ('extras', ctypes.POINTER(Extras)), ('out', ctypes.c_void_p), ('out_pitch_px', ctypes.c_uint32)]


_bound = False


def _lib():
    global _bound
    lib = lighting.load_library()
    if not _bound:
        vp, u32, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        lib.mil_mg_create.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(ci), ctypes.POINTER(Config)]
        lib.mil_mg_unique_id.argtypes = [vp]
        lib.mil_mg_create_rank.argtypes = [ctypes.POINTER(vp), ci, ci, ci, vp, ctypes.POINTER(Config)]
        lib.mil_mg_destroy.argtypes = [vp]
        lib.mil_mg_destroy.restype = None
        for f in ('mil_mg_world', 'mil_mg_local_ranks', 'mil_mg_rccl_ranks', 'mil_mg_last_rccl_error', 'mil_mg_last_hip_error', 'mil_mg_sync'):
            getattr(lib, f).argtypes = [vp]
        lib.mil_mg_rows_of_rank.argtypes = [ctypes.POINTER(Config), ci, ci, u32]
        lib.mil_mg_rows_of_rank.restype = u32
        lib.mil_mg_frame_rows_of_rank.argtypes = [ctypes.POINTER(Config), ci, ci, u32, vp]
        lib.mil_mg_reserve.argtypes = [vp, u32, u32, ci]
        lib.mil_mg_debug_delay_gather.argtypes = [vp, u32]
        lib.mil_mg_debug_delay_senders.argtypes = [vp, u32]
        lib.mil_mg_render_strips.argtypes = [vp, ctypes.POINTER(RankIO), vp, vp, vp, ci, u32, vp, u32]
        lib.mil_mg_render_frames.argtypes = [vp, ctypes.POINTER(RankIO), vp, vp, vp, ci, u32, ctypes.POINTER(vp), u32]
        _bound = True
    return lib


def frame_rows_of_rank(height, world, rank, band_rows=0, chunks=0):
    """Which frame row each local row of `rank` is (host arithmetic in the library; works without a GPU)."""
    lib = _lib()
    cfg = Config(0, band_rows, chunks, 0, 0)
    n = lib.mil_mg_rows_of_rank(ctypes.byref(cfg), world, rank, height)
    rows = np.zeros(n, np.uint32)
    if n:
        rc = lib.mil_mg_frame_rows_of_rank(ctypes.byref(cfg), world, rank, height, _np_ptr(rows))
        if rc:
            raise MilError(f'mil error {rc}')
    return rows.astype(np.int64)


class MultiGpu:
    def __init__(self, handle, cfg):
        self.lib, self.h, self.cfg = _lib(), handle, cfg
        self.world = self.lib.mil_mg_world(handle)
        self.local = self.lib.mil_mg_local_ranks(handle)

    @staticmethod
    def _cfg(max_point_lights, band_rows, chunks, transport, flags):
        return Config(max_point_lights, band_rows, chunks, transport, flags)

    @staticmethod
    def _raise(lib, rc, h=None):
        extra = f' (rccl {lib.mil_mg_last_rccl_error(h)}, hip {lib.mil_mg_last_hip_error(h)})' if h else ''
        raise MilError(f'mil error {rc}: {lib.mil_strerror(rc).decode()}{extra}')

    @classmethod
    def single_process(cls, devices, max_point_lights=4096, band_rows=32, chunks=4, transport=TRANSPORT_RCCL, flags=0):
        lib = _lib()
        cfg = cls._cfg(max_point_lights, band_rows, chunks, transport, flags)
        h = ctypes.c_void_p()
        dev = (ctypes.c_int * len(devices))(*devices)
        rc = lib.mil_mg_create(ctypes.byref(h), len(devices), dev, ctypes.byref(cfg))
        if rc:
            cls._raise(lib, rc)
        return cls(h, cfg)

    @staticmethod
    def unique_id():
        lib = _lib()
        buf = (ctypes.c_ubyte * UNIQUE_ID_BYTES)()
        rc = lib.mil_mg_unique_id(buf)
        if rc:
            MultiGpu._raise(lib, rc)
        return bytes(buf)

    @classmethod
    def per_process(cls, world, rank, device, unique_id, max_point_lights=4096, band_rows=32, chunks=4, flags=0):
        lib = _lib()
        cfg = cls._cfg(max_point_lights, band_rows, chunks, TRANSPORT_RCCL, flags)
        h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        rc = lib.mil_mg_create_rank(ctypes.byref(h), world, rank, device, buf, ctypes.byref(cfg))
        if rc:
            cls._raise(lib, rc)
        return cls(h, cfg)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mil_mg_destroy(self.h)
            self.h = None

    __del__ = close

    @property
    def rccl_ranks(self):
        return self.lib.mil_mg_rccl_ranks(self.h)

    def rows_of_rank(self, rank, height, strips=True):
        if not strips:
            return height
        return int(self.lib.mil_mg_rows_of_rank(ctypes.byref(self.cfg), self.world, rank, height))

    def _io(self, parts, fmt, keep):
        """parts: one dict per local rank: gb (plane dict as LightingPass takes, the rank's rows), point_lights (host array or
        None), ambient, shadow (dict with the map on that rank's device), out (tensor or None for rank 0)."""
        io = (RankIO * len(parts))()
        for i, p in enumerate(parts):
            gb = p['gb']
            d = gb['depth']
            h, w = d.shape
            io[i].gb = GBuffer(gb['normal'].data_ptr(), gb['albedo_occ'].data_ptr(), gb['rough_metal'].data_ptr(), d.data_ptr(),
                               w, h, d.stride(0), 0)
            ex = Extras()
            pl = p.get('point_lights')
            if pl is not None and len(pl):
                pl = np.ascontiguousarray(pl, np.float32).reshape(-1, 8)
                keep.append(pl)
                ex.point_lights, ex.n_point = pl.ctypes.data, pl.shape[0]
            if p.get('ambient') is not None:
                ex.ambient = (ctypes.c_float * 3)(*[float(a) for a in p['ambient']])
            sh = p.get('shadow')
            if sh is not None:
                s = Shadow((ctypes.c_float * 16)(*[float(v) for v in sh['light_view_proj']]), sh['map'].data_ptr(), sh['map'].shape[0], float(sh['bias']))
                keep.append(s)
                ex.shadow = ctypes.pointer(s)
            keep.append(ex)
            io[i].extras = ctypes.pointer(ex)
            out = p.get('out')
            if out is not None:
                io[i].out = out.data_ptr()
                io[i].out_pitch_px = out.stride(0) // 4 if fmt == lighting.OUT_HDR else out.stride(0)
        return io

    def render_strips(self, parts, frame, lights, width, height, fmt=lighting.OUT_SRGB8, flags=0, full=None):
        """mil_mg_render_strips.  full: rank 0's (height, width[, 4]) tensor (None in processes that do not drive rank 0)."""
        keep = []
        io = self._io(parts, fmt, keep)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 8)
        push = ubo.push_constants(width, height, lights.shape[0])
        fp = full.data_ptr() if full is not None else None
        pitch = 0 if full is None else (full.stride(0) // 4 if fmt == lighting.OUT_HDR else full.stride(0))
        rc = self.lib.mil_mg_render_strips(self.h, io, _np_ptr(frame), _np_ptr(lights), _np_ptr(push), fmt, flags, fp, pitch or width)
        if rc:
            self._raise(self.lib, rc, self.h)

    def render_frames(self, parts, frames, lights, width, height, fmt=lighting.OUT_SRGB8, flags=0, slots=None):
        """mil_mg_render_frames.  frames: one 68-float FrameMatrixData per local rank; slots: on rank 0's device, one
        (height, width[, 4]) tensor per rank of the world (None in processes that do not drive rank 0)."""
        keep = []
        io = self._io(parts, fmt, keep)
        fr = np.ascontiguousarray(np.stack([np.asarray(f, np.float32).reshape(68) for f in frames]))
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 8)
        push = ubo.push_constants(width, height, lights.shape[0])
        sl = (ctypes.c_void_p * self.world)(*([s.data_ptr() for s in slots] if slots is not None else [None] * self.world))
        pitch = width if slots is None else (slots[0].stride(0) // 4 if fmt == lighting.OUT_HDR else slots[0].stride(0))
        rc = self.lib.mil_mg_render_frames(self.h, io, _np_ptr(fr), _np_ptr(lights), _np_ptr(push), fmt, flags, sl, pitch)
        if rc:
            self._raise(self.lib, rc, self.h)

    def reserve(self, width, height, fmt=lighting.OUT_SRGB8):
        """mil_mg_reserve: staging area + every local context's workspace for this frame size, ahead of the frame loop."""
        rc = self.lib.mil_mg_reserve(self.h, width, height, fmt)
        if rc:
            self._raise(self.lib, rc, self.h)

    def debug_delay_senders(self, microseconds):
        """mil_mg_debug_delay_senders (test hook): idles the gather streams of the local ranks other than 0 at this point."""
        rc = self.lib.mil_mg_debug_delay_senders(self.h, microseconds)
        if rc:
            self._raise(self.lib, rc, self.h)

    def debug_delay_gather(self, microseconds):
        """mil_mg_debug_delay_gather (test hook): idles rank 0's gather stream at this point."""
        rc = self.lib.mil_mg_debug_delay_gather(self.h, microseconds)
        if rc:
            self._raise(self.lib, rc, self.h)

    def sync(self):
        rc = self.lib.mil_mg_sync(self.h)
        if rc:
            self._raise(self.lib, rc, self.h)
