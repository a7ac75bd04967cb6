"""TEST HARNESS (run under torch.distributed.run by tests/test_gpu_bench.py): bench.py's multi-rank
flow -- fixture broadcast, barriers, max over ranks, rank 0 prints -- on a box with ONE GPU, where
RCCL refuses two ranks on one device.  bench.py itself knows only RCCL (and exits non-zero without
it); this wrapper swaps its collectives for gloo on the host and puts every rank on GPU 0, and says
so in the line's config.collectives."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LOCAL_RANK"] = "0"  # every rank on this box's single GPU

import bench  # noqa: E402


def _gloo_collectives(world, rank, local, args):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    return "gloo (tests/bench_two_ranks_gloo.py)", torch.device("cpu")


bench.init_collectives = _gloo_collectives
bench.main()
