"""Dev tool: the 1-way k-shot eval episode (bench.py secondary.fs_1shot / fs_5shot), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.secondary_few_shot(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3, train_leg=False)
print({k: v for k, v in r.items() if k != "fs_train_episode_b4"})
