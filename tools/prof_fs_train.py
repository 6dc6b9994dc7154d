"""Dev tool: the training-mode few-shot episode (bench.py secondary.fs_train_episode_b4) a few times, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.secondary_few_shot_train(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
print(r)
