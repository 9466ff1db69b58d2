"""CPU-side checks of the drop-in boundary: libmil.so loads, exports every symbol that
include/mil.h declares, and the uniform structs have the reference's std140 sizes.  No
compute call is made here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from renderer_amd import lighting, multigpu, ubo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mil.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mil_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported():
    lib = lighting.load_library()
    syms = declared_symbols()
    assert sorted(syms) == sorted(lighting.EXPORTS + multigpu.EXPORTS)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.mil_version() == 3
    assert lib.mil_strerror(0) == b'ok' and b'bad argument' in lib.mil_strerror(-1)


def test_struct_layouts_match_reference():
    # mil_gbuffer / mil_shadow / mil_extras as the ctypes mirror sees them
    assert ctypes.sizeof(lighting.GBuffer) == 48
    assert ctypes.sizeof(lighting.Shadow) == 80
    assert ctypes.sizeof(lighting.Extras) == 48       # v3: + ldr_out, ldr_out_pitch_px
    # reference blocks (render/ubo.rs:28-63, fs_deferred_lighting.glsl:16-27)
    assert ubo.frame_matrix_data(np.eye(4), np.eye(4), (1, 2, 3)).nbytes == ubo.FRAME_UBO_BYTES == 272
    f = ubo.frame_matrix_data(np.eye(4), np.eye(4), (1, 2, 3))
    assert list(f[64:68]) == [1, 2, 3, 0]          # cameraPosition @256, 4 B pad
    assert ubo.default_directional_lights().itemsize * 8 == ubo.LIGHT_STRIDE_BYTES == 32
    assert ubo.push_constants(3840, 2160, 1).tobytes() == np.array([3840, 2160], np.float32).tobytes() + b'\x01\0\0\0'


def test_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(lighting.MilError):
        lighting.LightingPass(0)
    # the C ABI itself reports the missing device instead of aborting
    lib = lighting.load_library()
    h = ctypes.c_void_p()
    assert lib.mil_create(ctypes.byref(h), 0, 16) == -4 and not h.value
    assert lib.mil_create(None, 0, 16) == -1
    with pytest.raises(lighting.MilError, match='-4'):                # the multi-GPU layer too
        multigpu.MultiGpu.single_process([0])


def test_multigpu_layout_arithmetic():
    """mil_mg_frame_rows_of_rank (host arithmetic of the library): contiguous strips are sharding.strip_bounds (16-row cuts,
    4320 -> 7 x 544 + 512), dealt bands are lighting.band_rows_of_rank; every frame row belongs to one rank."""
    from renderer_amd import sharding
    for height, world in [(4320, 8), (2160, 8), (2160, 3), (100, 8), (16, 4)]:
        bounds = sharding.strip_bounds(height, world)
        got = [multigpu.frame_rows_of_rank(height, world, r, band_rows=0) for r in range(world)]
        for r, (y0, y1) in enumerate(bounds):
            assert np.array_equal(got[r], np.arange(y0, y1)), (height, world, r)
        for band in (16, 32, 128):
            rows = [multigpu.frame_rows_of_rank(height, world, r, band_rows=band) for r in range(world)]
            for r in range(world):
                assert np.array_equal(rows[r], lighting.band_rows_of_rank(height, band, world, r))
            assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(height))
    assert [len(multigpu.frame_rows_of_rank(4320, 8, r)) for r in range(8)] == [544] * 7 + [512]


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under renderer_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, 'renderer_amd')
    pat = re.compile(r'^\s*(from|import)\s+oracle\b|libmil_oracle|coracle|#include\s+".*oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.cpp', '.hpp', '.hip', '.h')) or fn == 'Makefile':
                assert not pat.search(open(os.path.join(dirpath, fn)).read()), fn


def test_hand_issued_scalar_loads_are_not_touched_before_their_wait():
    """The fast kernels fetch point-light records with s_load_dwordx8 issued by inline asm (mil_device.hpp srec_load) and wait
    for them by hand; the compiler does not know the destination SGPRs are in flight.  tools/check_srec.py compiles
    mil_fast.hip to assembly with the library's flags and proves, by data-flow over each kernel's control-flow graph, that no
    instruction names such a register between the load and the wait (a copy or a spill there would read stale data)."""
    import shutil
    import sys
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not present')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_srec
    path = check_srec.listing()
    try:
        total, bad = check_srec.main(path)
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    assert total >= 16 * 3 and bad == 0
