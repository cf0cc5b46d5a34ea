import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import vidcom2_amd as vc
from conftest import load_core_cases, make_input
dev = torch.device("cuda:0")
bad = 0
for c in load_core_cases():
    x = make_input(c["F"], c["N"], c["D"], c["dtype"], c["seed"], c["dist"]).to(dev)
    o = vc.vidcom2.low_var_channel_order(x).cpu().tolist()
    if o != c["chan_idx"]:
        bad += 1
        diff = [i for i, (a, b) in enumerate(zip(o, c["chan_idx"])) if a != b]
        print("ORDER MISMATCH", c["name"], c["dtype"], c["dist"], c["seed"], "positions", len(diff), diff[:10], "same set", sorted(o) == sorted(c["chan_idx"]), flush=True)
print("order mismatches:", bad)
