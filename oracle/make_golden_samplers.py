"""Generate tests/golden/samplers.npz from the REFERENCE's own samplers
(projects/mmdet3d_plugin/datasets/samplers/{group_sampler,distributed_sampler}.py), build container only.
Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_samplers.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as MG  # noqa: E402

CASES = [(23, 2, 4), (10, 1, 8), (7, 3, 2), (64, 1, 8)]       # (dataset size, samples_per_gpu, world size)


class _DS:
    def __init__(self, n, two_groups=False):
        self.n = n
        self.flag = np.zeros(n, dtype=np.uint8)
        if two_groups:
            self.flag[::3] = 1

    def __len__(self):
        return self.n


def main():
    MG.install_shims()
    MG._pkg("mmcv.utils")
    MG._mod("mmcv.utils.registry", Registry=MG._Registry, build_from_cfg=None)
    MG._mod("IPython", embed=None)
    base = os.path.join(MG.REF, "projects", "mmdet3d_plugin", "datasets")
    MG._pkg("projects.mmdet3d_plugin.datasets", base)
    MG._pkg("projects.mmdet3d_plugin.datasets.samplers", os.path.join(base, "samplers"))
    GS = importlib.import_module("projects.mmdet3d_plugin.datasets.samplers.group_sampler")
    DS = importlib.import_module("projects.mmdet3d_plugin.datasets.samplers.distributed_sampler")
    out = {}
    for (n, spg, world) in CASES:
        for two in (False, True):
            for epoch in (0, 3):
                for rank in range(world):
                    s = GS.DistributedGroupSampler(_DS(n, two), samples_per_gpu=spg, num_replicas=world, rank=rank, seed=0)
                    s.set_epoch(epoch)
                    out[f"group:{n}:{spg}:{world}:{int(two)}:{epoch}:{rank}"] = np.asarray(list(iter(s)), dtype=np.int64)
        for rank in range(world):
            s = DS.DistributedSampler(_DS(n), num_replicas=world, rank=rank, shuffle=False)
            out[f"dist:{n}:{world}:{rank}"] = np.asarray(list(iter(s)), dtype=np.int64)
    path = os.path.join(MG.OUT, "samplers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "index lists", os.path.getsize(path) / 1e3, "kB")


if __name__ == "__main__":
    main()
