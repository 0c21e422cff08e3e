#!/usr/bin/env python3
"""tests/golden/dit_g7b_unipc12.npz: a 12-step FlowUniPC trajectory (shift 5, the inference.py default) captured from
the reference scheduler -- exercises the steady second-order predictor/corrector steps and lower_order_final, which the
4-step fixture (orders 1,2,2,1) barely touches; and tests/golden/dit_g7c_unipc_orders.npz: solver_order 3 and solver_type bh1.
Build container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_import import load_reference                 # noqa: E402
from videocof_amd.weights import det_uniform                  # noqa: E402


@torch.no_grad()
def main():
    ns = load_reference()
    sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=2,
                                               prediction_type="flow_prediction")
    n = 12
    sch.set_timesteps(n, device="cpu", shift=5.0)
    x = det_uniform("g7b.x", (1, 16, 3, 4, 6), 1.0)
    vs = [det_uniform(f"g7b.v{i}", (1, 16, 3, 4, 6), 1.0) for i in range(n)]
    cur, traj, orders = x, [], []
    for i, t in enumerate(sch.timesteps):
        cur = sch.step(vs[i], t, cur, return_dict=False)[0]
        traj.append(cur)
        orders.append(sch.this_order)
    path = os.path.join(ROOT, "tests", "golden", "dit_g7b_unipc12.npz")
    np.savez_compressed(path, x=x.numpy(), v=torch.stack(vs).numpy(), traj=torch.stack(traj).numpy(),
                        timesteps=sch.timesteps.numpy(), sigmas=sch.sigmas.numpy(), orders=np.array(orders))
    print(os.path.getsize(path) // 1024, "KiB; orders", orders, "timesteps", sch.timesteps.tolist())

    # tests/golden/dit_g7c_unipc_orders.npz (round 5): the configurations beyond the CLIs' order-2 / bh2 -- solver_order 3 with bh2
    # (the five-term corrector, :590-600, and the solved predictor coefficients, :443-445) and solver_order 2 with bh1 (:402-403),
    # 9 steps at shift 5 each
    out = {}
    for tag, order, st in (("o3_bh2", 3, "bh2"), ("o2_bh1", 2, "bh1"), ("o3_bh1", 3, "bh1")):
        sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, solver_order=order, solver_type=st,
                                                   prediction_type="flow_prediction")
        n = 9
        sch.set_timesteps(n, device="cpu", shift=5.0)
        x = det_uniform("g7c.x", (1, 16, 3, 4, 6), 1.0)
        vs = [det_uniform(f"g7c.v{i}", (1, 16, 3, 4, 6), 1.0) for i in range(n)]
        cur, traj, orders = x, [], []
        for i, t in enumerate(sch.timesteps):
            cur = sch.step(vs[i], t, cur, return_dict=False)[0]
            traj.append(cur)
            orders.append(sch.this_order)
        out.update({"x": x.numpy(), "v": torch.stack(vs).numpy(), f"traj_{tag}": torch.stack(traj).numpy(),
                    f"orders_{tag}": np.array(orders), "timesteps": sch.timesteps.numpy(), "sigmas": sch.sigmas.numpy()})
        print(tag, "orders", orders)
    path = os.path.join(ROOT, "tests", "golden", "dit_g7c_unipc_orders.npz")
    np.savez_compressed(path, **out)
    print(os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
