"""Runs the multi-rank benchmark driver in a group of ONE rank with backend nccl, so that its
transport selection, self-check and timing code run on a single-GPU box:
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29621 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/rehearse_dist_one_rank.py [--graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--mesh", "2048", "--gmres-m", "30"] + sys.argv[1:]
import bench
from krylov_amd import dist_bench
args = bench.parse_args()
dist_bench.run(args, 1, 0, 0, bench.claim_stdout())
