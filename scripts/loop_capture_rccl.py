#!/usr/bin/env python3
"""Rebuilds and replays the hipGraph-captured training step WITH its RCCL all-reduces N times (default 30) in one child process
(the body of tests/test_gpu_model.py::test_reducer_on_a_one_rank_rccl_group; faulthandler on, stderr uncaptured).  r03 recorded
one abort inside capture_end of this path in a full-suite run; the log of this loop goes to profiles/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import torch.multiprocessing as mp
    from tests import test_gpu_model as t
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    mp.spawn(t._rccl_one_rank_worker, args=(t._free_port(), n), nprocs=1, join=True)
    print(f"loop_capture_rccl: {n} captures of the data-parallel step with RCCL all-reduces, no abort")


if __name__ == "__main__":
    main()
