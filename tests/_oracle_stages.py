"""CPU stage backend for vidcom2_amd.sharded.ShardedCompressor built on the oracle -- TEST ONLY.
Lets the collective logic (3 all-gathers + fixed-order reductions) run under gloo without a GPU."""
import torch

import oracle as O
from vidcom2_amd.sharded import ShardResult


class OracleStages:
    def __init__(self, F, N, D, dtype, base):
        self.F, self.N, self.D, self.dtype, self.base = F, N, D, dtype, base

    def chan_stats(self, x):
        xd = x.double()
        mean = xd.mean(0)
        return torch.stack([mean, ((xd - mean) ** 2).sum(0)])

    def select_channels(self, stats_all, R_total):
        P = stats_all.shape[0]
        n = R_total / P
        mean = stats_all[:, 0].mean(0)                                   # equal shard sizes
        m2 = (stats_all[:, 1] + n * (stats_all[:, 0] - mean) ** 2).sum(0)
        var = (m2 / R_total).float().to(self.dtype)
        self.chan_idx = O.topk_smallest(var, int(self.D * 0.5), True)

    def phase1(self, x):
        _, _, csum = O.gaussian_scores_sharded(x, self.chan_idx, self.N)
        return csum

    def phase2(self, x, csum_all, R_total):
        v, f, _ = O.gaussian_scores_sharded(x, self.chan_idx, self.N, csum_all, R_total)
        s, self.total = O.fuse(v, f)
        return s.float()

    def select(self, x, s_all, f0):
        scales = O.compute_scales(s_all.to(self.dtype), self.base)[f0: f0 + self.F].contiguous()
        idx = O.select_outlier_indices(self.total, scales, self.N)
        self.ks = torch.tensor([i.numel() for i in idx], dtype=torch.int64)
        self.local = O.map_linear_offset(idx, self.N)
        self.rows = x[self.local]

    def result(self, f0):
        return ShardResult(self.rows, self.local, self.local + f0 * self.N, self.ks, int(self.local.numel()))
