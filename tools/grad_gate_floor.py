"""Gradient gate at kitti_d192 (and the other step-test configurations): GPU path vs oracle next to the oracle's own response to
input perturbations at rounding level -- x * (1 + k 2^-23) for k = +1, -1, +2, -2 ("measure the oracle floor with
>= 3 perturbations and commit the table").  The gate of tests/test_gpu_fullsize.py::test_full_size_step_fwd_bwd_vs_oracle uses the MEAN
of the four.  usage: python tools/grad_gate_floor.py [cfg ...] > profiles/r6_grad_gate_floor.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fullsize as T

cases = [(c, True) for c in (sys.argv[1:] or ["kitti_d192"])]
for cfg_name, ac in cases:
    model, smp, sd0, trainable, _l, _lg, grads = T._gpu_step(cfg_name, ac)
    D = model.img_view_transformer.D
    _, _, g0 = T._oracle_step(cfg_name, ac, sd0, trainable, smp, D)
    per = {}
    floor = T._floor_table(cfg_name, ac, sd0, trainable, smp, D, g0, per_sample=per)
    ours = T._l2_table(grads, g0)
    names = sorted(ours, key=lambda k: -ours[k])
    print(f"{cfg_name}: GPU-vs-oracle L2 | oracle floor = mean response to k = {T.FLOOR_PERTURBATIONS} ulp | the four samples | ours / (3 floor + 2e-3); worst 20 of {len(names)}")
    for k in names[:20]:
        samples = " ".join(f"{per[p].get(k, float('nan')):.3e}" for p in T.FLOOR_PERTURBATIONS)
        print(f"{ours[k]:.3e}  floor {floor.get(k, float('nan')):.3e}  [{samples}]  ratio {ours[k] / (3 * floor[k] + 2e-3):.2f}  {k}")
    worst_ratio = max(ours[k] / (3 * floor[k] + 2e-3) for k in ours)
    print(f"{cfg_name}: worst ratio {worst_ratio:.2f}; tensors above 2e-2: {[k for k in names if ours[k] >= 2e-2]}; max {ours[names[0]]:.3e}")
    T._ORACLE_STEP.clear()
