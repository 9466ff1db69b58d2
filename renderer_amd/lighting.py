"""Host-side mirror of the reference's lighting + tonemap subpasses over libmil.so.

The reference records these two draws inline in ``Frame::build``
(/root/reference/renderer/src/render/mod.rs:189-219 and :296-309); the bindings of
those draws (set 0 FrameMatrixData, set 1 the four G-buffer input attachments, set 2
Lights, push constants; render/mod.rs:26-29, pbr/mod.rs:207-229) are the arguments
here.  All work runs in hand-written HIP behind the C ABI of include/mil.h; torch is
only the owner of device memory and streams.  There is no CPU fallback: importing
works without a GPU, calling does not.
"""
import ctypes
import os

import numpy as np

from . import ubo

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MIL_LIB', os.path.join(_HERE, 'libmil.so'))

OUT_HDR, OUT_LDR, OUT_SRGB8, OUT_SRGB8_VIA_LDR = 0, 1, 2, 3
FLAG_STRICT, FLAG_NO_CULL, FLAG_SHADOW_LDS = 1, 2, 4
TILE = 16
NOALLOC_MAX_POINT_LIGHTS = 4096      # MIL_NOALLOC_MAX_POINT_LIGHTS

EXPORTS = ['mil_version', 'mil_strerror', 'mil_last_hip_error', 'mil_create', 'mil_destroy',
           'mil_set_tile_count_buffer', 'mil_set_tile_stamp_buffer', 'mil_sync', 'mil_light_directional', 'mil_tonemap',
           'mil_light_ex', 'mil_count_pairs', 'mil_fxaa', 'mil_sky_hosek', 'mil_transparency_resolve', 'mil_shade_frame',
           'mil_transparency_accumulate', 'mil_import_external_memory', 'mil_alloc_exportable_memory',
           'mil_release_external_memory', 'mil_shade_bands', 'mil_band_local_rows', 'mil_row_weights',
           'mil_light_ex_noalloc', 'mil_set_two_kernel_threshold', 'mil_reserve', 'mil_set_list_pool', 'mil_workspace_info']


class MilError(RuntimeError):
    pass


class GBuffer(ctypes.Structure):           # mil_gbuffer
    _fields_ = [('normal_l_model', ctypes.c_void_p), ('albedo_occlusion', ctypes.c_void_p),
                ('roughness_metallic', ctypes.c_void_p), ('depth', ctypes.c_void_p),
                ('width', ctypes.c_uint32), ('height', ctypes.c_uint32),
                ('pitch_px', ctypes.c_uint32), ('y_offset', ctypes.c_uint32)]


class Shadow(ctypes.Structure):            # mil_shadow
    _fields_ = [('light_view_proj', ctypes.c_float * 16), ('map', ctypes.c_void_p),
                ('size', ctypes.c_uint32), ('bias', ctypes.c_float)]


class Extras(ctypes.Structure):            # mil_extras
    _fields_ = [('point_lights', ctypes.c_void_p), ('n_point', ctypes.c_uint32),
                ('ambient', ctypes.c_float * 3), ('shadow', ctypes.POINTER(Shadow)),
                ('ldr_out', ctypes.c_void_p), ('ldr_out_pitch_px', ctypes.c_uint32)]


class FramePasses(ctypes.Structure):      # mil_frame_passes
    _fields_ = [('sky', ctypes.c_void_p), ('accum', ctypes.c_void_p), ('reveal', ctypes.c_void_p),
                ('accum_pitch_px', ctypes.c_uint32), ('reveal_pitch_px', ctypes.c_uint32)]


class WorkspaceStats(ctypes.Structure):    # mil_workspace_stats
    _fields_ = [('bytes', ctypes.c_uint64), ('half_tiles', ctypes.c_uint32), ('pool_entries', ctypes.c_uint32),
                ('list_entries_max', ctypes.c_uint32), ('last_items', ctypes.c_uint32), ('last_pool_entries', ctypes.c_uint32),
                ('last_brute_macro_tiles', ctypes.c_uint32), ('frame_call_allocations', ctypes.c_uint32)]


class BandLayout(ctypes.Structure):        # mil_band_layout
    _fields_ = [('band_rows', ctypes.c_uint32), ('n_ranks', ctypes.c_uint32), ('rank', ctypes.c_uint32),
                ('first_band', ctypes.c_uint32)]


class TransparentFragments(ctypes.Structure):   # mil_transparent_fragments
    _fields_ = [('position_depth', ctypes.c_void_p), ('normal_opacity', ctypes.c_void_p),
                ('albedo_roughness', ctypes.c_void_p), ('metallic', ctypes.c_void_p), ('count', ctypes.c_uint32)]


_lib = None


def load_library(path=LIB_PATH):
    """dlopen libmil.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise MilError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(hipcc --offload-arch=gfx950); there is no CPU fallback')
        # torch ships its own libamdhip64.so.7; load it FIRST so libmil.so (NEEDED
        # libamdhip64.so.7) binds to that same runtime instance.  Streams and device pointers
        # handed across the C ABI are only meaningful inside one HIP runtime.
        import torch  # noqa: F401
        lib = ctypes.CDLL(path)
        lib.mil_strerror.restype = ctypes.c_char_p
        lib.mil_strerror.argtypes = [ctypes.c_int]
        lib.mil_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_uint32]
        lib.mil_destroy.argtypes = [ctypes.c_void_p]
        lib.mil_destroy.restype = None
        lib.mil_last_hip_error.argtypes = [ctypes.c_void_p]
        lib.mil_set_tile_count_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        lib.mil_set_tile_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        lib.mil_sync.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        lib.mil_light_directional.argtypes = [vp, ctypes.POINTER(GBuffer), vp, vp, vp, vp, u32, u32, vp]
        lib.mil_tonemap.argtypes = [vp, vp, u32, u32, u32, vp, u32, ctypes.c_int, u32, vp]
        lib.mil_light_ex.argtypes = [vp, ctypes.POINTER(GBuffer), vp, vp, vp, ctypes.POINTER(Extras), vp, u32,
                                     ctypes.c_int, u32, vp]
        lib.mil_shade_frame.argtypes = [vp, ctypes.POINTER(GBuffer), vp, vp, vp, ctypes.POINTER(Extras),
                                        ctypes.POINTER(FramePasses), vp, u32, ctypes.c_int, u32, vp]
        lib.mil_fxaa.argtypes = [vp, vp, u32, u32, u32, u32, u32, vp, u32, u32, vp]
        lib.mil_sky_hosek.argtypes = [vp, vp, u32, u32, u32, u32, vp, vp, vp, vp, u32, u32, vp]
        lib.mil_transparency_resolve.argtypes = [vp, vp, u32, vp, u32, u32, u32, vp, u32, u32, vp]
        lib.mil_transparency_accumulate.argtypes = [vp, ctypes.POINTER(TransparentFragments), vp, vp, vp, vp, vp, u32, vp]
        lib.mil_import_external_memory.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(vp), ctypes.POINTER(vp)]
        lib.mil_alloc_exportable_memory.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
        lib.mil_release_external_memory.argtypes = [vp, vp]
        lib.mil_count_pairs.argtypes = [vp, ctypes.POINTER(GBuffer), vp, vp, vp, u32, vp, vp]
        lib.mil_light_ex_noalloc.argtypes = lib.mil_light_ex.argtypes
        lib.mil_set_two_kernel_threshold.argtypes = [vp, u32]
        lib.mil_reserve.argtypes = [vp, u32, u32]
        lib.mil_set_list_pool.argtypes = [vp, u32]
        lib.mil_workspace_info.argtypes = [vp, ctypes.POINTER(WorkspaceStats)]
        lib.mil_shade_bands.argtypes = [vp, ctypes.POINTER(GBuffer), vp, vp, vp, ctypes.POINTER(Extras),
                                        ctypes.POINTER(FramePasses), ctypes.POINTER(BandLayout), vp, u32, ctypes.c_int,
                                        ctypes.c_int, u32, vp]
        lib.mil_band_local_rows.argtypes = [ctypes.POINTER(BandLayout), u32]
        lib.mil_band_local_rows.restype = u32
        lib.mil_row_weights.argtypes = [vp, vp, u32, u32, ctypes.POINTER(BandLayout), vp, u32, vp]
        _lib = lib
    return _lib


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def band_local_rows(height, band_rows, n_ranks, rank):
    """Rows rank `rank` holds when the frame's bands of band_rows rows are dealt cyclically over n_ranks (host arithmetic)."""
    return int(load_library().mil_band_local_rows(ctypes.byref(BandLayout(band_rows, n_ranks, rank)), height))


def band_rows_of_rank(height, band_rows, n_ranks, rank):
    """Frame rows of that rank in local order (numpy int64): local row k is frame row result[k]."""
    nb = -(-height // band_rows)
    rows = [np.arange(b * band_rows, min(height, (b + 1) * band_rows)) for b in range(rank, nb, n_ranks)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def row_weights(frame, width, height, point_lights, gb_height=None, y_offset=0, bands=None):
    """mil_row_weights: the K2 launch-order weight of every local 16-row tile row (host arithmetic; works without a GPU)."""
    lib = load_library()
    gb_height = height if gb_height is None else gb_height
    frame = np.ascontiguousarray(frame, np.float32).reshape(68)
    pl = np.ascontiguousarray(point_lights, np.float32).reshape(-1, 8)
    push = ubo.push_constants(width, height, 0)
    w = np.zeros((gb_height + TILE - 1) // TILE, np.uint32)
    b = ctypes.byref(BandLayout(*bands)) if bands is not None else None
    rc = lib.mil_row_weights(_np_ptr(frame), _np_ptr(push), gb_height, y_offset, b, _np_ptr(pl), pl.shape[0], _np_ptr(w))
    if rc != 0:
        raise MilError(f'mil error {rc}: {lib.mil_strerror(rc).decode()}')
    return w


class LightingPass:
    """One mil_ctx on one GPU.  ``gb`` arguments are the plane dicts of
    ``renderer_amd.synth`` (int32/float32 CUDA tensors + width/height/y_offset)."""

    def __init__(self, device=0, max_point_lights=4096):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise MilError('no HIP device visible: the lighting pass has no CPU fallback')
        self.device = torch.device('cuda', device)
        h = ctypes.c_void_p()
        self._check(self.lib.mil_create(ctypes.byref(h), device, max_point_lights), None)
        self.ctx = h
        self._tile_counts = self._tile_stamps = None     # diagnostic buffers the ctx points at: kept alive here

    def set_two_kernel_threshold(self, min_point_lights):
        """mil_set_two_kernel_threshold: 1 = every culled fast-mode launch goes k_bin -> k_shade, 0xFFFFFFFF = none does."""
        self._check(self.lib.mil_set_two_kernel_threshold(self.ctx, min_point_lights))

    def reserve(self, width, height):
        """mil_reserve: no per-frame call allocates for frames up to this size."""
        self._check(self.lib.mil_reserve(self.ctx, width, height))

    def set_list_pool(self, entries):
        """mil_set_list_pool: light-list entries the two-kernel form's pool holds."""
        self._check(self.lib.mil_set_list_pool(self.ctx, entries))

    def workspace_info(self):
        """mil_workspace_info as a dict (waits for the last two-kernel launch)."""
        st = WorkspaceStats()
        self._check(self.lib.mil_workspace_info(self.ctx, ctypes.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in WorkspaceStats._fields_}

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.mil_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _check(self, rc, ctx='self'):
        if rc != 0:
            hip = self.lib.mil_last_hip_error(self.ctx) if ctx == 'self' and getattr(self, 'ctx', None) else 0
            raise MilError(f'mil error {rc}: {self.lib.mil_strerror(rc).decode()} (hip {hip})')

    def _stream(self, stream):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def _gb(self, gb):
        d = gb['depth']
        assert d.is_cuda and d.dtype == self.torch.float32 and d.stride(1) == 1
        for k in ('normal', 'albedo_occ', 'rough_metal'):
            assert gb[k].is_cuda and gb[k].dtype == self.torch.int32 and gb[k].stride() == d.stride(), k
        h, w = d.shape
        return GBuffer(gb['normal'].data_ptr(), gb['albedo_occ'].data_ptr(), gb['rough_metal'].data_ptr(),
                       d.data_ptr(), w, h, d.stride(0), gb.get('y_offset', 0))

    @staticmethod
    def _push(gb, n_dir):
        h, w = gb['depth'].shape
        return ubo.push_constants(gb.get('width', w), gb.get('height', h), n_dir)

    def light_directional(self, gb, frame, lights, out=None, strict=False, stream=None):
        """The lighting draw (render/mod.rs:201-216): returns hdr (h, w, 4) float32."""
        t = self.torch
        h, w = gb['depth'].shape
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 8)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        push = self._push(gb, lights.shape[0])
        if out is None:
            out = t.empty((h, w, 4), dtype=t.float32, device=self.device)
        g = self._gb(gb)
        self._check(self.lib.mil_light_directional(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(lights),
                                                   _np_ptr(push), out.data_ptr(), out.stride(0) // 4,
                                                   FLAG_STRICT if strict else 0, self._stream(stream)))
        return out

    def tonemap(self, hdr, fmt=OUT_LDR, out=None, strict=False, stream=None):
        """The tonemap draw (render/mod.rs:297-309): returns (h, w) int32 holding the packed u32."""
        t = self.torch
        h, w = hdr.shape[:2]
        if out is None:
            out = t.empty((h, w), dtype=t.int32, device=self.device)
        self._check(self.lib.mil_tonemap(self.ctx, hdr.data_ptr(), hdr.stride(0) // 4, w, h, out.data_ptr(),
                                         out.stride(0), fmt, FLAG_STRICT if strict else 0, self._stream(stream)))
        return out

    def fxaa(self, ldr, row_begin=0, row_count=None, out=None, strict=False, stream=None):
        """The FXAA draw (render/mod.rs:312-329): ldr (h, w) int32 R11G11B10F of the whole frame ->
        (row_count, w) int32 B8G8R8A8 sRGB of rows [row_begin, row_begin + row_count)."""
        t = self.torch
        h, w = ldr.shape
        row_count = h - row_begin if row_count is None else row_count
        if out is None:
            out = t.empty((row_count, w), dtype=t.int32, device=self.device)
        self._check(self.lib.mil_fxaa(self.ctx, ldr.data_ptr(), ldr.stride(0), w, h, row_begin, row_count, out.data_ptr(),
                                      out.stride(0), FLAG_STRICT if strict else 0, self._stream(stream)))
        return out

    def sky_hosek(self, gb, frame, params, hdr, strict=False, stream=None):
        """The sky draw (render/mod.rs:221-226) as a screen-space pass over hdr (in place).
        params: 44 floats = HosekWilkieParams (render/hosek/mod.rs:222-256)."""
        d = gb['depth']
        h, w = d.shape
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        prm = np.ascontiguousarray(params, np.float32).reshape(44)
        push = self._push(gb, 0)
        self._check(self.lib.mil_sky_hosek(self.ctx, d.data_ptr(), d.stride(0), w, h, gb.get('y_offset', 0), _np_ptr(frame),
                                           _np_ptr(push), _np_ptr(prm), hdr.data_ptr(), hdr.stride(0) // 4,
                                           FLAG_STRICT if strict else 0, self._stream(stream)))
        return hdr

    def transparency_resolve(self, accum, reveal, hdr, strict=False, stream=None):
        """The resolve draw (render/mod.rs:282-294): accum (h, w, 4) float16, reveal (h, w) float16, hdr in place."""
        t = self.torch
        assert accum.dtype == t.float16 and reveal.dtype == t.float16 and hdr.dtype == t.float32
        h, w = reveal.shape
        self._check(self.lib.mil_transparency_resolve(self.ctx, accum.data_ptr(), accum.stride(0) // 4, reveal.data_ptr(),
                                                      reveal.stride(0), w, h, hdr.data_ptr(), hdr.stride(0) // 4,
                                                      FLAG_STRICT if strict else 0, self._stream(stream)))
        return hdr

    def transparency_accumulate(self, fragments, frame, lights, accum=None, reveal=None, strict=False, stream=None):
        """The fragment shader of the accumulate draw (render/mod.rs:229-281, fs_mcguire13_accumulation.glsl:100-126) over a
        fragment list: fragments = dict of float32 CUDA tensors position_depth / normal_opacity / albedo_roughness (n, 4),
        metallic (n,).  Returns accum (n, 4), reveal (n,): the values the blend stage consumes."""
        t = self.torch
        planes = [fragments[k] for k in ('position_depth', 'normal_opacity', 'albedo_roughness', 'metallic')]
        n = planes[3].shape[0]
        for a, shape in zip(planes, ((n, 4), (n, 4), (n, 4), (n,))):
            assert a.is_cuda and a.dtype == t.float32 and a.is_contiguous() and tuple(a.shape) == shape
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 8)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        push = ubo.push_constants(1, 1, lights.shape[0])
        if accum is None:
            accum = t.empty((n, 4), dtype=t.float32, device=self.device)
        if reveal is None:
            reveal = t.empty((n,), dtype=t.float32, device=self.device)
        fr = TransparentFragments(*(a.data_ptr() for a in planes), n)
        self._check(self.lib.mil_transparency_accumulate(self.ctx, ctypes.byref(fr), _np_ptr(frame), _np_ptr(lights), _np_ptr(push),
                                                         accum.data_ptr(), reveal.data_ptr(), FLAG_STRICT if strict else 0,
                                                         self._stream(stream)))
        return accum, reveal

    # ---- SURVEY.md 8 f-3, HIP half: device memory shared with another API through a POSIX fd (dma-buf)
    def alloc_exportable_memory(self, size):
        """-> (mem handle, device pointer, fd, allocated size); the fd is the caller's to hand over and close."""
        mem, ptr, fd, real = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(-1), ctypes.c_uint64()
        self._check(self.lib.mil_alloc_exportable_memory(self.ctx, size, ctypes.byref(mem), ctypes.byref(ptr), ctypes.byref(fd), ctypes.byref(real)))
        return mem, ptr.value, fd.value, real.value

    def import_external_memory(self, fd, size):
        """-> (mem handle, device pointer) of an allocation another API exported as fd (vkGetMemoryFdKHR)."""
        mem, ptr = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self.lib.mil_import_external_memory(self.ctx, fd, size, ctypes.byref(mem), ctypes.byref(ptr)))
        return mem, ptr.value

    def release_external_memory(self, mem):
        self._check(self.lib.mil_release_external_memory(self.ctx, mem))

    def tensor_at(self, ptr, shape, dtype):
        """A torch tensor over raw device memory (no copy, no ownership): how tests and bench address imported memory."""
        t = self.torch
        typestr = {t.float32: '<f4', t.int32: '<i4', t.float16: '<f2', t.uint8: '|u1'}[dtype]

        class _Raw:
            __cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)
        return t.as_tensor(_Raw(), device=self.device)

    def light_ex(self, gb, frame, lights, point_lights=None, ambient=None, shadow=None, fmt=OUT_HDR,
                 flags=0, out=None, stream=None, ldr_out=None):
        """Lighting with the build-defined extensions, optionally fused with the tonemap.  ldr_out (fmt == OUT_HDR only): an
        int32 (h, w) CUDA tensor that receives the tonemapped sRGB8 image of the same pixels in the same launch
        (mil_extras.ldr_out)."""
        return self._shade(gb, frame, lights, point_lights, ambient, shadow, None, fmt, flags, out, stream, ldr_out=ldr_out)

    def shade_frame(self, gb, frame, lights, point_lights=None, ambient=None, shadow=None, sky=None, accum=None,
                    reveal=None, fmt=OUT_LDR, flags=0, out=None, stream=None, ldr_out=None):
        """Lighting -> sky -> transparency resolve -> tonemap (render/mod.rs:201-309) as ONE launch.
        sky: 44 floats HosekWilkieParams or None; accum (h, w, 4) / reveal (h, w) float16 or None."""
        t = self.torch
        fp = FramePasses()
        keep = []
        if sky is not None:
            prm = np.ascontiguousarray(sky, np.float32).reshape(44)
            keep.append(prm)
            fp.sky = prm.ctypes.data
        if accum is not None:
            assert accum.dtype == t.float16 and reveal.dtype == t.float16
            fp.accum, fp.reveal = accum.data_ptr(), reveal.data_ptr()
            fp.accum_pitch_px, fp.reveal_pitch_px = accum.stride(0) // 4, reveal.stride(0)
        return self._shade(gb, frame, lights, point_lights, ambient, shadow, fp, fmt, flags, out, stream, ldr_out=ldr_out)

    def light_ex_noalloc(self, gb, frame, lights, point_lights_dev=None, ambient=None, shadow=None, fmt=OUT_HDR, flags=0,
                         out=None, stream=None):
        """mil_light_ex_noalloc: mil_light_ex without any allocation inside the library (static device scratch).  The point
        lights are a (n, 8) float32 CUDA tensor (they already live on the device); `out` should be given, so that nothing
        allocates on this side of the ABI either."""
        t = self.torch
        h, w = gb['depth'].shape
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros((0, 8)), np.float32).reshape(-1, 8)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        push = self._push(gb, lights.shape[0])
        ex = Extras()
        if point_lights_dev is not None and point_lights_dev.shape[0]:
            assert point_lights_dev.is_cuda and point_lights_dev.dtype == t.float32 and point_lights_dev.is_contiguous()
            ex.point_lights, ex.n_point = point_lights_dev.data_ptr(), point_lights_dev.shape[0]
        if ambient is not None:
            ex.ambient = (ctypes.c_float * 3)(*[float(a) for a in ambient])
        if shadow is not None:
            sm = shadow['map']
            sh = Shadow((ctypes.c_float * 16)(*[float(v) for v in shadow['light_view_proj']]), sm.data_ptr(), sm.shape[0], float(shadow['bias']))
            ex.shadow = ctypes.pointer(sh)
        if out is None:
            out = (t.empty((h, w, 4), dtype=t.float32, device=self.device) if fmt == OUT_HDR
                   else t.empty((h, w), dtype=t.int32, device=self.device))
        pitch = out.stride(0) // 4 if fmt == OUT_HDR else out.stride(0)
        g = self._gb(gb)
        self._check(self.lib.mil_light_ex_noalloc(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(lights), _np_ptr(push),
                                                  ctypes.byref(ex), out.data_ptr(), pitch, fmt, flags, self._stream(stream)))
        return out

    def shade_bands(self, gb, frame, lights, bands, point_lights=None, ambient=None, shadow=None, sky=None, accum=None,
                    reveal=None, fmt=OUT_LDR, flags=0, out=None, out_is_frame=False, stream=None, ldr_out=None):
        """mil_shade_bands: mil_shade_frame on a rank's dealt bands.  gb: the rank's compacted planes with width / height
        of the WHOLE frame in gb['width'] / gb['height']; bands = (band_rows, n_ranks, rank[, first_band]).  out: the rank's compacted
        output, or (out_is_frame) the assembled frame."""
        t = self.torch
        fp = FramePasses()
        keep = []
        if sky is not None:
            prm = np.ascontiguousarray(sky, np.float32).reshape(44)
            keep.append(prm)
            fp.sky = prm.ctypes.data
        if accum is not None:
            assert accum.dtype == t.float16 and reveal.dtype == t.float16
            fp.accum, fp.reveal = accum.data_ptr(), reveal.data_ptr()
            fp.accum_pitch_px, fp.reveal_pitch_px = accum.stride(0) // 4, reveal.stride(0)
        # no later subpass: NULL passes = the lighting launch alone (mil_light_ex on dealt bands)
        return self._shade(gb, frame, lights, point_lights, ambient, shadow, fp if (sky is not None or accum is not None) else None,
                           fmt, flags, out, stream, bands=BandLayout(*(tuple(bands) + (0,) * (4 - len(bands)))), out_is_frame=out_is_frame,
                           ldr_out=ldr_out)

    def _shade(self, gb, frame, lights, point_lights, ambient, shadow, passes, fmt, flags, out, stream, bands=None,
               out_is_frame=False, ldr_out=None):
        t = self.torch
        h, w = gb['depth'].shape
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros((0, 8)), np.float32).reshape(-1, 8)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        push = self._push(gb, lights.shape[0])
        ex = Extras()
        keep = []
        if point_lights is not None and len(point_lights):
            pl = np.ascontiguousarray(point_lights, np.float32).reshape(-1, 8)
            keep.append(pl)
            ex.point_lights, ex.n_point = pl.ctypes.data, pl.shape[0]
        if ambient is not None:
            ex.ambient = (ctypes.c_float * 3)(*[float(a) for a in ambient])
        if shadow is not None:
            sm = shadow['map']
            assert sm.is_cuda and sm.dtype == t.float32 and sm.is_contiguous() and sm.shape[0] == sm.shape[1]
            sh = Shadow((ctypes.c_float * 16)(*[float(v) for v in shadow['light_view_proj']]), sm.data_ptr(),
                        sm.shape[0], float(shadow['bias']))
            keep.append(sh)
            ex.shadow = ctypes.pointer(sh)
        if ldr_out is not None:
            assert ldr_out.is_cuda and ldr_out.dtype == t.int32 and ldr_out.stride(1) == 1
            ex.ldr_out, ex.ldr_out_pitch_px = ldr_out.data_ptr(), ldr_out.stride(0)
        if out is None:
            oh = gb['height'] if out_is_frame else h
            out = (t.empty((oh, w, 4), dtype=t.float32, device=self.device) if fmt == OUT_HDR
                   else t.empty((oh, w), dtype=t.int32, device=self.device))
        pitch = out.stride(0) // 4 if fmt == OUT_HDR else out.stride(0)
        g = self._gb(gb)
        if bands is not None:
            self._check(self.lib.mil_shade_bands(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(lights), _np_ptr(push),
                                                 ctypes.byref(ex), ctypes.byref(passes) if passes is not None else None, ctypes.byref(bands), out.data_ptr(),
                                                 pitch, 1 if out_is_frame else 0, fmt, flags, self._stream(stream)))
        elif passes is None:
            self._check(self.lib.mil_light_ex(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(lights), _np_ptr(push),
                                              ctypes.byref(ex), out.data_ptr(), pitch, fmt, flags, self._stream(stream)))
        else:
            self._check(self.lib.mil_shade_frame(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(lights), _np_ptr(push),
                                                 ctypes.byref(ex), ctypes.byref(passes), out.data_ptr(), pitch, fmt, flags,
                                                 self._stream(stream)))
        return out

    def count_pairs(self, gb, frame, point_lights, stream=None):
        t = self.torch
        pl = np.ascontiguousarray(point_lights, np.float32).reshape(-1, 8)
        frame = np.ascontiguousarray(frame, np.float32).reshape(68)
        push = self._push(gb, 0)
        cnt = t.zeros(1, dtype=t.int64, device=self.device)
        g = self._gb(gb)
        self._check(self.lib.mil_count_pairs(self.ctx, ctypes.byref(g), _np_ptr(frame), _np_ptr(push), _np_ptr(pl),
                                             pl.shape[0], cnt.data_ptr(), self._stream(stream)))
        return int(cnt.item())

    def tile_counts(self, gb, enable=True):
        """Diagnostics: returns a (tiles_y, tiles_x) int32 tensor the next culled call fills."""
        t = self.torch
        h, w = gb['depth'].shape
        buf = t.zeros(((h + TILE - 1) // TILE, (w + TILE - 1) // TILE), dtype=t.int32, device=self.device)
        # the ctx keeps the raw pointer: the tensor lives on self until it is replaced or disabled, and the ctx knows
        # how many tiles it holds (a larger launch leaves it untouched)
        self._tile_counts = buf if enable else None
        self._check(self.lib.mil_set_tile_count_buffer(self.ctx, buf.data_ptr() if enable else None, buf.numel() if enable else 0))
        return buf

    def tile_stamps(self, gb, enable=True):
        """Diagnostics: (tiles_y, tiles_x, 4) int32 the next culled call fills with start, end (100 MHz), count, HW_ID."""
        t = self.torch
        h, w = gb['depth'].shape
        buf = t.zeros(((h + TILE - 1) // TILE, (w + TILE - 1) // TILE, 4), dtype=t.int32, device=self.device)
        self._tile_stamps = buf if enable else None
        self._check(self.lib.mil_set_tile_stamp_buffer(self.ctx, buf.data_ptr() if enable else None, buf.numel() // 4 if enable else 0))
        return buf

    def sync(self, stream=None):
        self._check(self.lib.mil_sync(self.ctx, self._stream(stream)))
