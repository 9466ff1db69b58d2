"""The oracles against fixtures made by EXECUTING THE REFERENCE'S OWN SHADER TEXT (tests/golden/make_glsl_golden.py:
oracle/glsl_exec.py interprets /root/reference/renderer/shaders/*.glsl in the development container; only its outputs
travel).  Bit for bit, NaN == NaN: the C oracle (the strict kernels' target) is the reference's arithmetic, operator by
operator; what the shader text does not fix (format decode / encode, built-in precision, blending) is listed in
glsl_exec.py and stays unpinned."""
import os

import numpy as np
import pytest

from oracle import mil_oracle as mo

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SHADERS = '/root/reference/renderer/shaders/'


def bits_equal(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope='module')
def ref():
    return np.load(os.path.join(GOLDEN, 'glsl_reference.npz'))


def planes(f):
    return f['normal'], f['albedo_occ'], f['rough_metal'], f['depth']


def test_lighting_shader(ref, coracle):
    sw, no = np.load(os.path.join(GOLDEN, 'sweep_96x54.npz')), np.load(os.path.join(GOLDEN, 'noise_64x36.npz'))
    assert bits_equal(ref['hdr_sweep'], coracle.shade(*planes(sw), sw['frame'], sw['dir_lights']))
    assert bits_equal(ref['hdr_sweep'], sw['hdr_dir_only_c32'])
    assert bits_equal(ref['hdr_noise'], coracle.shade(*planes(no), no['frame'], no['dir_lights']))      # worst-case texels, NaN included
    assert bits_equal(ref['hdr_sweep_1light'], coracle.shade(*planes(sw), sw['frame'], sw['dir_lights'][:1]))
    assert bits_equal(ref['hdr_sweep_0lights'], coracle.shade(*planes(sw), sw['frame'], sw['dir_lights'][:0]))
    assert np.all(ref['hdr_sweep_0lights'][..., :3] == 0) and np.all(ref['hdr_sweep'][..., 3] == 1)
    # a strip of the frame: the shader sees global gl_FragCoord.y and the full resolution = the oracle's y_offset / full_height
    strip = coracle.shade(*[p[16:32] for p in planes(sw)], sw['frame'], sw['dir_lights'], y_offset=16, full_height=54)
    assert bits_equal(ref['hdr_sweep_rows16_32'], strip) and bits_equal(ref['hdr_sweep_rows16_32'], ref['hdr_sweep'][16:32])
    # the float64 restatement agrees with the executed shader to FP32 rounding
    a = mo.shade(*planes(sw), sw['frame'], sw['dir_lights'])
    fin = np.isfinite(a).all(-1)
    assert (np.abs(a - ref['hdr_sweep'])[fin] > 1e-4 + 1e-4 * np.abs(a)[fin]).mean() < 2e-3


def test_tonemap_shader(ref, coracle):
    ti, to = ref['tonemap_in'], ref['tonemap_out']
    packed = coracle.tonemap(ti.reshape(-1, 1, 4), 1).reshape(-1)
    mine = np.array([coracle.pack_r11g11b10f(*t[:3]) for t in to], np.uint32)
    np.testing.assert_array_equal(packed, mine)                      # ACESFilm of the shader -> the same R11G11B10F codes
    assert np.abs(mo.aces_film(ti[:, :3].astype(np.float64)) - to[:, :3]).max() < 1e-6
    assert np.all(to[:, 3] == 1) and to[:, :3].min() >= 0 and to[:, :3].max() <= 1


def test_resolve_shader_and_blend(ref, coracle):
    got = coracle.transparency_resolve(ref['resolve_accum'], ref['resolve_reveal'], ref['hdr_sweep'].copy())
    assert bits_equal(ref['resolve_hdr'], got)
    r64 = mo.transparency_resolve(ref['resolve_accum'], ref['resolve_reveal'], ref['hdr_sweep'].astype(np.float64))
    assert np.all(np.abs(r64 - ref['resolve_hdr']) <= 1e-5 + 1e-5 * np.abs(r64))


def test_sky_shader(ref, coracle):
    sw = np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    got = coracle.sky_hosek(sw['depth'], sw['frame'], ref['sky_params'], ref['hdr_sweep'].copy())
    assert bits_equal(ref['sky_hdr'], got)
    sky = sw['depth'] >= 1
    assert sky.mean() > 0.2 and np.all(ref['sky_hdr'][sky][:, :3] > 0) and bits_equal(ref['sky_hdr'][~sky], ref['hdr_sweep'][~sky])


def test_fxaa_shader(ref, coracle):
    f = np.load(os.path.join(GOLDEN, 'fxaa_101x70.npz'))
    enc = mo.encode_b8g8r8a8_srgb(ref['fxaa_color'][..., :3].astype(np.float64))
    np.testing.assert_array_equal(enc, coracle.fxaa(f['ldr']))
    np.testing.assert_array_equal(enc, f['srgb8_c32'])
    assert np.all(ref['fxaa_color'][..., 3] == 1)


def accumulate_inputs(ref):
    return [ref['accumulate_' + k] for k in ('position_depth', 'normal_opacity', 'albedo_roughness', 'metallic', 'camera', 'dir_lights')]


def test_accumulate_shader(ref, coracle):
    acc, rev = coracle.transparency_accumulate(*accumulate_inputs(ref))
    assert bits_equal(ref['accumulate_accum'], acc) and bits_equal(ref['accumulate_reveal'], rev)
    assert np.isfinite(acc).all() and np.all(acc[:, 3] > 0) and np.array_equal(rev, ref['accumulate_normal_opacity'][:, 3])
    a64, r64 = mo.transparency_accumulate(*accumulate_inputs(ref))
    assert np.all(np.abs(a64 - acc) <= 1e-5 + 2e-4 * np.abs(a64)) and np.array_equal(r64, rev)
    # no lights: accum = (0, 0, 0, opacity) * w10, the weight alone
    a0, _ = coracle.transparency_accumulate(*accumulate_inputs(ref)[:5], np.zeros((0, 8), np.float32))
    assert np.all(a0[:, :3] == 0) and bits_equal(a0[:, 3], acc[:, 3])


@pytest.mark.skipif(not os.path.isdir(SHADERS), reason='the reference tree exists only in the development container')
def test_fixture_regenerates_from_the_reference_source(ref, tmp_path, monkeypatch):
    """Runs the interpreter on the reference's shader files again: the committed fixture is what they produce."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_glsl_golden', os.path.join(GOLDEN, 'make_glsl_golden.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sw = np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    hdr, _ = m.lighting(sw)
    assert bits_equal(hdr, ref['hdr_sweep'])


# ---- HosekWilkieParams blocks the reference itself would produce (tests/golden/make_hosek_golden.py)
@pytest.fixture(scope='module')
def hosek():
    return np.load(os.path.join(GOLDEN, 'hosek_params.npz'))


@pytest.mark.parametrize('name', ['default', 'basic'])
def test_sky_shader_on_reference_coefficient_blocks(hosek, coracle, name):
    """The sky of HosekSky::new (sun at the zenith, turbidity 1) and of scenes/basic.rs (sun = light 0, turbidity 8, red
    ground): make_hosek_wilkie_params evaluated on the reference's dataset.rs, then the reference's sky_hosek_frag.glsl
    executed on it - the C oracle reproduces that image bit for bit, the float64 oracle to 1e-5."""
    sw = np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    prm, img = hosek['params_' + name], hosek['sky_hdr_' + name]
    assert prm.shape == (11, 4) and prm.dtype == np.float32 and np.all(prm[:, 3] == 0)
    assert np.all(np.abs(prm[:10, :3] - hosek['params_' + name + '_f64'][:10, :3]) <= 2e-6 * np.abs(hosek['params_' + name + '_f64'][:10, :3]) + 1e-7)
    assert abs(np.linalg.norm(prm[10, :3]) - 1) < 1e-6
    hdr0 = sw['hdr_dir_only_c32']
    got = coracle.sky_hosek(sw['depth'], sw['frame'], prm, hdr0.copy())
    assert bits_equal(img, got)
    sky = sw['depth'] >= 1
    assert np.isfinite(img).all() and np.all(img[sky][:, :3] > 0) and bits_equal(img[~sky], hdr0[~sky])
    r64 = mo.sky_hosek(sw['depth'], sw['frame'], prm, hdr0.astype(np.float64))
    assert np.all(np.abs(r64 - img) <= 1e-5 + 1e-5 * np.abs(r64))
    if name == 'default':       # zenith sun: elevation^(1/3) = 1, every spline is its last control point; Z = the radiance table's
        assert np.allclose(prm[9, :3], [4.324573, 8.090833, 13.90238], rtol=1e-6)


@pytest.mark.skipif(not os.path.isdir(SHADERS), reason='the reference tree exists only in the development container')
def test_hosek_fixture_regenerates_from_the_reference_dataset(hosek):
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_hosek_golden', os.path.join(GOLDEN, 'make_hosek_golden.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for name, (sun, turb, alb) in m.parameter_sets().items():
        assert bits_equal(m.make_params(sun, turb, alb), hosek['params_' + name]), name
