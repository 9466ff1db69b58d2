"""mil_mg with TWO PROCESSES: mil_mg_unique_id / mil_mg_create_rank had only ever run with a world of one.
Two rank processes (fresh children of a launcher that never touched the GPU, tests/mg_two_proc.py) create their handles
through the C ABI with the id carried over a pipe.  The reference renders on one device (render/vulkan.rs:106-108): there is
nothing to be at parity with but the single-GPU frame.
  * one GPU in the box: RCCL refuses two ranks on one device - both ranks must get a clean MIL_ERR_RCCL (-6), no hang, no
    abort (a timeout is a failure to diagnose, not to repeat);
  * two or more GPUs: both handles come up with 2 RCCL ranks, three frames in dealt bands are gathered to rank 0 and the
    assembled frame equals rank 0's own single-GPU frame bit for bit."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_rank_processes_through_the_c_abi(mg_launcher):
    assert mg_launcher is not None and mg_launcher.poll() is None, 'the launcher of tests/conftest.py is not running'
    n_dev = torch.cuda.device_count()
    devs = (0, 1) if n_dev >= 2 else (0, 0)
    mg_launcher.stdin.write(f'run {devs[0]} {devs[1]}\n')
    mg_launcher.stdin.flush()
    res = json.loads(mg_launcher.stdout.readline())
    assert not res['timed_out'] and 'error' not in res, res
    for r in res['ranks']:
        assert r is not None and r['rc'] == 0 and r['result'] is not None, res      # no abort, no crash: a status came back
    if n_dev >= 2:
        for r in res['ranks']:
            assert r['result']['created'] and r['result']['rccl_ranks'] == 2, res
        assert res['ranks'][0]['result']['assembled_equals_single_gpu_bitwise'] is True, res
    else:
        for r in res['ranks']:
            assert r['result']['created'] is False and r['result']['status'] == -6, res       # MIL_ERR_RCCL, cleanly
