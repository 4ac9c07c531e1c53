"""The reference call sequence (bench.Frame.step_reference_protocol) in a loop, for a kernel trace: which kernels one eager step
launches and what they cost on the GPU.  Run under rocprofv3 --kernel-trace --stats.    python tools/prof_dropin_kernels.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frame = bench.Frame(bench.make_workload("water", 0), torch.device("cuda"))
for _ in range(n):
    frame.step_reference_protocol("helper")
torch.cuda.synchronize()
print("steps", n)
