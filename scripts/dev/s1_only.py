"""Sweep 1 alone, back to back on one tensor (X stays in the Infinity Cache) and over three tensors in turn (cold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd import synth, vidcom2 as V
xs = [synth.make(128, 196, 3584, torch.bfloat16, sd, "drift").cuda() for sd in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else "warm"
for i in range(60):
    V._channel_variance(xs[0] if mode == "warm" else xs[i % 3])
torch.cuda.synchronize()
