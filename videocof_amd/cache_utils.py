"""TeaCache for the HIP-backed DiT (``videox_fun/models/cache_utils.py:4-76`` and the hook in
``WanTransformer3DModel.forward``, ``videox_fun/models/wan_transformer3d.py:956-1031, 1101-1104``).

Timestep-Embedding-Aware Cache (Liu et al., arXiv 2411.19108): the relative L1 change of the time-projection input
``e0`` between consecutive steps, passed through a model-specific polynomial, is accumulated; while the sum stays below
``rel_l1_thresh`` the transformer blocks are SKIPPED and the residual they produced the last time they ran is added to the
patch-embedded input instead.  It is a lossy speed-up that the caller opts into (``enable_teacache``); with it off -- the
default, and what every parity statement and ``bench.py`` uses -- nothing here runs.

Same names, arguments, state fields and error behaviour as the reference class.  ``offload`` is accepted and ignored: the
cached residual (1.4 GB at 14B / 67k tokens) stays in HBM.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

__all__ = ["TeaCache", "get_teacache_coefficients"]

# The published rescaling polynomials (constants of the TeaCache project, cache_utils.py:4-19), by model family.
_COEFFICIENTS = (
    (("wan2.1-t2v-1.3b", "wan2.1-fun-1.3b", "wan2.1-fun-v1.1-1.3b", "wan2.1-vace-1.3b"),
     [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02]),
    (("wan2.1-t2v-14b",),
     [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01]),
    (("wan2.1-i2v-14b-480p",),
     [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01]),
    (("wan2.1-i2v-14b-720p", "wan2.1-fun-14b", "wan2.2-fun", "wan2.2-i2v-a14b", "wan2.2-t2v-a14b", "wan2.2-ti2v-5b",
      "wan2.2-s2v", "wan2.1-vace-14b", "wan2.2-vace-fun"),
     [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]),
)


def get_teacache_coefficients(model_name: str) -> Optional[List[float]]:
    name = model_name.lower()
    for keys, coeff in _COEFFICIENTS:
        if any(k in name for k in keys):
            return list(coeff)
    print(f"The model {model_name} is not supported by TeaCache.")
    return None


class TeaCache:
    def __init__(self, coefficients: List[float], num_steps: int, rel_l1_thresh: float = 0.0,
                 num_skip_start_steps: int = 0, offload: bool = True):
        # the same three argument checks as the reference's constructor (cache_utils.py:37-46: ValueError each), own wording
        if num_steps < 1:
            raise ValueError(f"TeaCache: num_steps={num_steps}; a denoise loop has at least one step")
        if rel_l1_thresh < 0:
            raise ValueError(f"TeaCache: rel_l1_thresh={rel_l1_thresh}; the accumulated-distance threshold cannot be negative")
        if not 0 <= num_skip_start_steps <= num_steps:
            raise ValueError(f"TeaCache: num_skip_start_steps={num_skip_start_steps} lies outside [0, num_steps={num_steps}]")
        self.coefficients = list(coefficients)
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.num_skip_start_steps = num_skip_start_steps
        self.offload = offload
        self.rescale_func = np.poly1d(self.coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev: torch.Tensor, cur: torch.Tensor) -> float:
        """mean|cur - prev| / mean|prev| as a host float (one device sync per step, like the reference's ``.cpu().item()``)."""
        if prev is None:
            # the reference fails here with a TypeError (`cur - None`) when num_skip_start_steps == 0; same condition, clearer text
            raise TypeError("TeaCache has no previous modulated input: use num_skip_start_steps >= 1 (the reference fails "
                            "the same way, cache_utils.py:65)")
        return float((torch.abs(cur - prev).mean() / torch.abs(prev).mean()).item())

    def reset(self) -> None:
        self.cnt = 0
        self.should_calc = True
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_residual_cond = None
        self.previous_residual_uncond = None

    def decide(self, modulated_inp: torch.Tensor, cond_flag: bool = True) -> bool:
        """The decision block of the reference's forward (wan_transformer3d.py:956-978): run the blocks this step?"""
        if cond_flag:
            if self.cnt < self.num_skip_start_steps:
                should_calc = True
                self.accumulated_rel_l1_distance = 0
            else:
                d = self.compute_rel_l1_distance(self.previous_modulated_input, modulated_inp)
                self.accumulated_rel_l1_distance += self.rescale_func(d)
                if self.accumulated_rel_l1_distance < self.rel_l1_thresh:
                    should_calc = False
                else:
                    should_calc = True
                    self.accumulated_rel_l1_distance = 0
            self.previous_modulated_input = modulated_inp
            self.should_calc = should_calc
        return self.should_calc

    def step_done(self, cond_flag: bool = True) -> None:
        """wan_transformer3d.py:1101-1104."""
        if cond_flag:
            self.cnt += 1
            if self.cnt == self.num_steps:
                self.reset()
