import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vidcom2_amd.fused import keep_positions
dev = torch.device("cuda:0")
nvid, ntext = 64 * 324, 96
vm = torch.zeros(nvid + ntext, dtype=torch.bool, device=dev); vm[32:32 + nvid] = True
kept = torch.arange(0, nvid, 8, device=dev, dtype=torch.int64)
for _ in range(200): keep_positions(vm, kept, nvid)
torch.cuda.synchronize()
