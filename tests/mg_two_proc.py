"""Helper of tests/test_zy_two_process_mg.py (not collected by pytest): the two-process path of libmil's multi-GPU layer,
mil_mg_unique_id + mil_mg_create_rank through the C ABI, one process per rank.

    python tests/mg_two_proc.py launcher          started by tests/conftest.py BEFORE any test touches the GPU; makes no
                                                  GPU call itself; on 'run <devices>' it starts the two rank processes
                                                  (so they are fresh children of a process without GPU state), carries
                                                  rank 0's 128-byte id to rank 1 over pipes and prints one JSON line
    python tests/mg_two_proc.py rank <r> <dev>    one rank: creates its mil_mg; with two distinct devices it renders a
                                                  small frame in dealt bands and rank 0 compares the assembled frame with
                                                  its own single-GPU result bit for bit
"""
import json
import os
import subprocess
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIMEOUT = 150


def launcher():
    for line in sys.stdin:
        cmd = line.split()
        if not cmd:
            continue
        if cmd[0] == 'quit':
            return
        devs = cmd[1:3]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'rank', str(r), devs[r]], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for r in range(2)]
        res = {'ranks': [None, None], 'timed_out': False}
        killer = threading.Timer(TIMEOUT, lambda: [p.kill() for p in procs])
        killer.start()
        try:
            uid = procs[0].stdout.readline().strip()            # rank 0 prints the id first
            procs[0].stdin.write(uid + '\n'); procs[0].stdin.flush()
            procs[1].stdin.write(uid + '\n'); procs[1].stdin.flush()
            for r, p in enumerate(procs):
                out, err = p.communicate()
                last = [l for l in out.splitlines() if l.startswith('{')]
                res['ranks'][r] = dict(rc=p.returncode, result=json.loads(last[-1]) if last else None, stderr=err[-600:])
        except Exception as e:                                   # a broken pipe when a child died early: reported, not raised
            res['error'] = repr(e)
        res['timed_out'] = not killer.is_alive() and any(p.returncode in (-9, 137) for p in procs)
        killer.cancel()
        print(json.dumps(res), flush=True)


def rank_main(rank, device):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from renderer_amd import lighting, multigpu, synth, ubo
    if rank == 0:
        print(multigpu.MultiGpu.unique_id().hex(), flush=True)
    uid = bytes.fromhex(sys.stdin.readline().strip())
    try:
        mg = multigpu.MultiGpu.per_process(2, rank, device, uid, max_point_lights=64, band_rows=16, chunks=1)
    except lighting.MilError as e:
        print(json.dumps(dict(created=False, status=int(str(e).split('mil error ')[1].split(':')[0]), message=str(e))), flush=True)
        return
    # two ranks on two devices: a 320 x 176 frame in dealt 16-row bands, gathered to rank 0 over RCCL
    w, h = 320, 176
    dev = f'cuda:{device}'
    torch.cuda.set_device(device)
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=dev)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    pl = synth.point_light_field(48, 5)
    rows = torch.from_numpy(multigpu.frame_rows_of_rank(h, 2, rank, band_rows=16).astype(np.int64)).to(dev)
    mine = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
    mine['height'] = int(rows.numel())
    out = None if rank == 0 else torch.empty((rows.numel(), w), dtype=torch.int32, device=dev)
    full = torch.zeros((h, w), dtype=torch.int32, device=dev) if rank == 0 else None
    mg.reserve(w, h, lighting.OUT_SRGB8)
    for _ in range(3):                                           # frames back to back: the staging slices are reused
        mg.render_strips([dict(gb=mine, point_lights=pl, ambient=(0.03, 0.03, 0.03), shadow=None, out=out)], frame, dl, w, h,
                         fmt=lighting.OUT_SRGB8, full=full)
    mg.sync()
    res = dict(created=True, rccl_ranks=mg.rccl_ranks)
    if rank == 0:
        lp = lighting.LightingPass(device, max_point_lights=64)
        ref = lp.light_ex(gb, frame, dl, pl, (0.03, 0.03, 0.03), None, fmt=lighting.OUT_SRGB8)
        res['assembled_equals_single_gpu_bitwise'] = bool(torch.equal(full, ref))
        lp.close()
    mg.close()
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'launcher':
        launcher()
    else:
        rank_main(int(sys.argv[2]), int(sys.argv[3]))
