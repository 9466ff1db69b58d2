"""Host-side logic of libmil.so that needs no GPU: the K2 launch-order weights (mil_row_weights), the band dealing
arithmetic (mil_band_local_rows) and their Python mirrors."""
import numpy as np
import pytest

from renderer_amd import lighting, synth, ubo


def test_row_weights_follow_the_projected_lights():
    """the row-cost heuristic once computed clip w with the wrong sign, so every light fell back to the
    whole frame and every row weighed n_point.  With w = P[11] z + P[15] w the weights of the headline scene vary,
    lights behind the camera weigh nothing, and each weight equals a brute-force projection of the sphere."""
    w, h = 3840, 2160
    cam = synth.sweep_camera(w, h)
    frame = cam.frame_matrix_data()
    pl = synth.point_light_field(256, 0x5EED0003)
    wt = lighting.row_weights(frame, w, h, pl)
    assert wt.shape == (135,) and wt.max() <= 256
    assert len(set(wt.tolist())) > 20 and wt.min() < wt.max()          # not the plain bottom-up order of equal weights
    # brute force: sample every light's sphere surface, project, mark the rows it reaches (a subset of the conservative range)
    view = cam.view_matrix().astype(np.float64)
    proj = cam.projection_matrix().astype(np.float64)
    rng = np.random.RandomState(1)
    d = rng.randn(400, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lower = np.zeros(135, np.int64)
    for l in pl:
        pts = l[0:3][None, :] + l[3] * d
        q = (np.c_[pts, np.ones(len(pts))] @ view.T) @ proj.T
        ok = q[:, 3] > 1e-3
        if not ok.any():
            continue
        y = (q[ok, 1] / q[ok, 3] * 0.5 + 0.5) * h
        y = y[(y >= 0) & (y < h)]
        if len(y):
            lower[int(y.min()) // 16: int(y.max()) // 16 + 1] += 1
    assert np.all(wt.astype(np.int64) >= lower)                        # conservative ...
    assert (wt.astype(np.int64) - lower).mean() < 64                   # ... and far from the old code's 256 everywhere
    assert np.corrcoef(wt, lower)[0, 1] > 0.9                          # the order it induces is the brute-force order
    # a light behind the camera reaches no row
    behind = ubo.point_lights(np.array([[30.0, 7.5, 4.5]]), np.array([2.0]), np.ones((1, 3)), 4.0)    # camera at x = 9 looking down -x
    assert lighting.row_weights(frame, w, h, behind).sum() == 0
    # a strip sees the rows of the whole frame it covers
    strip = lighting.row_weights(frame, w, h, pl, gb_height=272, y_offset=544)
    assert np.array_equal(strip, wt[34:51])


@pytest.mark.parametrize('height,band,n', [(2160, 32, 8), (4320, 32, 8), (4320, 64, 3), (100, 16, 4), (17, 16, 8), (2160, 16, 1)])
def test_band_dealing_arithmetic(height, band, n):
    """mil_band_local_rows == the Python mirror; the ranks' rows partition the frame; banded row weights are the
    monolithic weights of the rows a rank holds."""
    rows = [lighting.band_rows_of_rank(height, band, n, r) for r in range(n)]
    for r in range(n):
        assert lighting.band_local_rows(height, band, n, r) == len(rows[r])
    allrows = np.sort(np.concatenate(rows))
    assert np.array_equal(allrows, np.arange(height))
    if height >= 1000:
        w = height * 16 // 9
        cam = synth.sweep_camera(w, height)
        frame = cam.frame_matrix_data()
        pl = synth.point_light_field(64, 5)
        mono = lighting.row_weights(frame, w, height, pl)
        for r in (0, n - 1):
            got = lighting.row_weights(frame, w, height, pl, gb_height=len(rows[r]), bands=(band, n, r))
            want = mono[rows[r][::16] // 16]
            assert np.array_equal(got, want)


def test_band_layout_rejects_bad_values():
    assert lighting.band_local_rows(100, 24, 2, 0) == 0        # band_rows must be a power of two >= 16
    assert lighting.band_local_rows(100, 8, 2, 0) == 0
    assert lighting.band_local_rows(100, 16, 2, 2) == 0        # rank < n_ranks


def test_bench_gpus_n_starts_its_own_ranks_and_fails_cleanly_without_a_gpu():
    """the driver runs `python bench.py --gpus N` as is.  Outside a launcher bench.py must start the N ranks
    itself (child processes, before any GPU call) and relay rank 0's line; on a machine without a GPU every rank ends with
    MIL_ERR_NO_DEVICE, the exit code is non-zero and stdout carries no line (there is no CPU path to fall back to)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a machine with fewer than two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MIL_BENCH_REHEARSE')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    err = r.stderr.decode(errors='replace')
    assert r.returncode != 0
    assert r.stdout.strip() == b''
    assert 'starting 2 ranks as child processes' in err and 'torch.distributed.run' in err
    assert 'MIL_ERR_NO_DEVICE' in err


def test_bench_self_launch_falls_back_to_the_python_gather_once():
    """A mil_mg run whose ranks END with an error (forced here: MIL_BENCH_FAIL_MG) is followed by ONE run over --gather torch;
    without a GPU that one ends with MIL_ERR_NO_DEVICE, which is final (no third run, no line)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a machine with fewer than two GPUs')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MIL_BENCH_REHEARSE')}
    env['MIL_BENCH_FAIL_MG'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    err = r.stderr.decode(errors='replace')
    assert r.returncode != 0 and r.stdout.strip() == b''
    assert err.count('starting 2 ranks as child processes') == 2
    assert 'once more over the Python gather' in err and '--gather torch' in err
    assert 'MIL_ERR_NO_DEVICE' in err
