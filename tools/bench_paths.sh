#!/bin/bash
# Runs on the GPU box (via gpurun): the bench's N > 1 code paths as far as ONE GPU allows.
#   1. the mil_mg path with a world of one (torchrun, nccl process group, ncclCommInitRank inside libmil), strips and frames
#   2. the gloo rehearsal of two ranks on device 0 through the Python gather (--gather torch)
set -e
export TMPDIR=/tmp
O=gpurun_out
MIL_BENCH_FORCE_MG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --steps 10 --warmup 3 > $O/mg_world1_strips.json 2> $O/mg_world1_strips.err
MIL_BENCH_FORCE_MG=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 \
    bench.py --gpus 1 --steps 10 --warmup 3 --mode frames > $O/mg_world1_frames.json 2> $O/mg_world1_frames.err
MIL_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
    bench.py --gpus 2 --steps 5 --warmup 2 --gather torch --workload headline > $O/rehearse2_strips.json 2> $O/rehearse2_strips.err
MIL_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 \
    bench.py --gpus 2 --steps 5 --warmup 2 --gather torch --mode frames > $O/rehearse2_frames.json 2> $O/rehearse2_frames.err
