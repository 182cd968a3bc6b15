"""Gradient gate at kitti_d192: GPU path vs oracle next to the oracle's own response to a one-ulp input perturbation
(the d112 statistic of tests/test_gpu_fullsize.py::test_gradient_gate_vs_oracle_noise_floor at the BASELINE metric's config)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fullsize as T

cfg_name, ac = (sys.argv[1] if len(sys.argv) > 1 else "kitti_d192"), True
model, smp, sd0, trainable, _l, _lg, grads = T._gpu_step(cfg_name, ac)
D = model.img_view_transformer.D
_, _, g0 = T._oracle_step(cfg_name, ac, sd0, trainable, smp, D)
_, _, g1 = T._oracle_step(cfg_name, ac, sd0, trainable, smp, D, perturb=1)
floor = T._l2_table(g1, g0)
ours = T._l2_table(grads, g0)
names = sorted(ours, key=lambda k: -ours[k])
print(f"{cfg_name}: GPU-vs-oracle L2 | oracle-vs-(one-ulp-perturbed oracle) L2, worst 16 of {len(names)}")
for k in names[:16]:
    print(f"{ours[k]:.3e}  floor {floor.get(k, float('nan')):.3e}  {k}")
