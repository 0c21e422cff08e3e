"""Flow-matching UniPC multistep sampler (host PyTorch, as north_star prescribes).

Same contract as ``FlowUniPCMultistepScheduler`` of
``videox_fun/utils/fm_solvers_unipc.py`` (diffusers ``SchedulerMixin`` style):
``set_timesteps(n, device=, shift=)``, ``.timesteps`` (int64, truncated,
:208-211), ``.sigmas``, ``.order``, ``step(model_output, timestep, sample,
return_dict=False)[0]``, ``scale_model_input``.  Supported configuration: ``flow_prediction``,
``predict_x0`` (what every VideoCoF entry point builds, fast_infer.py:328-337), ``solver_type`` ``'bh1'`` or ``'bh2'``
(:402-407), ANY ``solver_order`` >= 1 (the reference's CLIs use 2; round 5 added the general form of :430-445 /
:590-600 -- the rho coefficients of order p come from the p - 1 (predictor) / p (corrector) Vandermonde system),
``lower_order_final``.  One deliberate difference: with ``bh1`` the reference's LAST step (sigma_t = 0) evaluates
``alpha_t * B_h * pred_res`` = 1 * (-inf) * 0 = NaN (:473); here the vanishing term is dropped (the limit), so the step
returns the x0 prediction as ``bh2`` does.

Own formulation: every UniP / UniC update is a linear combination of at most p + 2
tensors, so the per-step scalar algebra (:378-470, :520-612) is done once in
float64 on the host and each update is applied in one fused fp32 pass
(``torch.add``-chains on the device, no host sync), instead of the reference's
~20 small bf16 elementwise launches.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

__all__ = ["FlowUniPCMultistepScheduler"]


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


def _lam(sigma: float) -> float:
    """lambda = log(alpha) - log(sigma) with alpha = 1 - sigma (:272-273, 386-387)."""
    if sigma <= 0.0:
        return math.inf
    return math.log1p(-sigma) - math.log(sigma) if sigma < 1.0 else -math.inf


class _Config(dict):
    """The constructor arguments, readable as attributes and as keys (diffusers' FrozenDict, without the freezing machinery)."""
    __getattr__ = dict.__getitem__


class FlowUniPCMultistepScheduler:
    order = 1

    @classmethod
    def from_config(cls, config, **kwargs):
        """diffusers' ``SchedulerMixin.from_config``: a scheduler from another one's ``.config`` (or a plain dict), keyword overrides on
        top; entries the constructor does not take are ignored."""
        import inspect
        names = set(inspect.signature(cls.__init__).parameters) - {"self"}
        cfg = {k: v for k, v in dict(config).items() if k in names}
        cfg.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**cfg)

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: Optional[float] = 1.0,
                 use_dynamic_shifting: bool = False, thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0,
                 predict_x0: bool = True, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: List[int] = [], solver_p=None, timestep_spacing: str = "linspace",
                 steps_offset: int = 0, final_sigmas_type: Optional[str] = "zero"):
        if solver_type in ("midpoint", "heun", "logrho"):
            solver_type = "bh2"                                                  # :97-99
        if solver_type not in ("bh1", "bh2"):
            raise NotImplementedError(f"{solver_type} is not implemented for {self.__class__}")   # :42-43
        if prediction_type != "flow_prediction" or not predict_x0:
            raise NotImplementedError("only flow_prediction / predict_x0 is built (fast_infer.py:328-337)")
        if int(solver_order) < 1:
            raise ValueError(f"solver_order={solver_order} must be >= 1")
        if use_dynamic_shifting or thresholding or solver_p is not None or final_sigmas_type != "zero":
            raise NotImplementedError("dynamic shifting / thresholding / solver_p / sigma_min are not on the VideoCoF path")
        # what diffusers' @register_to_config would hold: EVERY constructor argument, by attribute and by key (fm_solvers_unipc.py:73-92)
        self.config = _Config(
            num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type, shift=shift,
            use_dynamic_shifting=False, thresholding=False, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
            sample_max_value=sample_max_value, predict_x0=True, solver_type=solver_type, lower_order_final=lower_order_final,
            disable_corrector=list(disable_corrector), solver_p=None, timestep_spacing=timestep_spacing, steps_offset=steps_offset,
            final_sigmas_type="zero")
        self.predict_x0 = True
        self.disable_corrector = list(disable_corrector)
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(torch.float32)
        sig = shift * sig / (1 + (shift - 1) * sig)                              # :113-116
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self.num_inference_steps = None
        self._reset()

    def _reset(self):
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.config.solver_order
        self.timestep_list = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None
        self._step_index = None
        self._begin_index = None

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[List[float]] = None, mu=None, shift: Optional[float] = None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]   # :185-188
        sigmas = np.asarray(sigmas, dtype=np.float64)
        if shift is None:
            shift = self.config.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                     # :195-196
        timesteps = sigmas * self.config.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0.0]]).astype(np.float32))   # stays on the host
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)    # truncation (:210-211)
        self.num_inference_steps = len(timesteps)
        self._reset()

    def scale_model_input(self, sample: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        return sample

    # ------------------------------------------------------------------ scalar algebra (float64, host)
    def _coeffs(self, i_t: int, i_s0: int, i_prevs: List[int], order: int, corrector: bool):
        """Return (a, b0, b_prevs, b_new): x_t = a * x + b0 * m0 + sum_k b_prevs[k] * m_k + b_new * m_new for a step from
        sigma[i_s0] to sigma[i_t]; m_k is the model output at sigma[i_prevs[k]] (order - 1 of them, nearest first).
        multistep_uni_p_bh_update (:406-470) and multistep_uni_c_bh_update (:560-612) in one place."""
        s = self.sigmas.double()
        sigma_t, sigma_s0 = s[i_t].item(), s[i_s0].item()
        alpha_t = 1.0 - sigma_t
        h = _lam(sigma_t) - _lam(sigma_s0)
        hh = -h
        h_phi_1 = math.expm1(hh) if hh > -math.inf else -1.0
        B_h = h_phi_1 if self.config.solver_type == "bh2" else hh                 # :402-407
        a = sigma_t / sigma_s0
        b0 = -alpha_t * h_phi_1
        assert len(i_prevs) == order - 1
        rks = [(_lam(s[i].item()) - _lam(sigma_s0)) / h for i in i_prevs]
        rhos: List[float] = []
        if (not corrector and order >= 3) or (corrector and order >= 2):
            rk_all = np.array(rks + [1.0])
            R, bvec = [], []
            h_phi_k = h_phi_1 / hh - 1.0
            fact = 1
            for i in range(1, order + 1):
                R.append(rk_all ** (i - 1))
                bvec.append(h_phi_k * fact / B_h)
                fact *= i + 1
                h_phi_k = h_phi_k / hh - 1.0 / fact
            R, bvec = np.stack(R), np.array(bvec)
            rhos = list(np.linalg.solve(R, bvec) if corrector else np.linalg.solve(R[:-1, :-1], bvec[:-1]))     # :600 / :443-445
        elif not corrector and order == 2:
            rhos = [0.5]                                                          # :437-438
        elif corrector and order == 1:
            rhos = [0.5]                                                          # :597-598
        k = -alpha_t * B_h
        b_prevs = [k * rhos[j] / rks[j] for j in range(order - 1)]                # D1s[j] = (m_j - m0) / rk_j
        b_new = k * rhos[-1] if corrector else 0.0                                # D1_t = model_t - m0
        b0 -= sum(b_prevs) + b_new
        return a, b0, b_prevs, b_new

    @staticmethod
    def _combine(dtype, terms: List[Tuple[float, Optional[torch.Tensor]]]) -> torch.Tensor:
        live = [(c, t) for c, t in terms if t is not None and c != 0.0]
        # (the fused kernel takes up to four terms -- everything orders 1 and 2 produce; higher orders take the torch chain below)
        if live and len(live) <= 4 and live[0][1].is_cuda and dtype in (torch.float32, torch.bfloat16) and math.isfinite(sum(c for c, _ in live)):
            from . import ops                      # fused single-pass HIP kernel (wan_lincomb)
            return ops.lincomb(live, dtype)
        acc = None
        for c, ten in terms:
            if ten is None or c == 0.0:
                continue
            acc = ten.float() * c if acc is None else acc.add_(ten.float(), alpha=c)
        return acc.to(dtype)

    # ------------------------------------------------------------------ step
    def index_for_timestep(self, timestep, schedule_timesteps=None):
        """fm_solvers_unipc.py:628-640: a timestep that occurs twice resolves to its SECOND position."""
        idx = ((self.timesteps if schedule_timesteps is None else schedule_timesteps) == timestep).nonzero()
        return idx[1 if len(idx) > 1 else 0].item()

    def __len__(self):
        return self.config.num_train_timesteps          # fm_solvers_unipc.py:798-799

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """fm_solvers_unipc.py:758-797: (1 - sigma) x0 + sigma noise with sigma looked up per sample -- by timestep before a loop has a
        begin index, else at the current (or the begin) step."""
        sigmas = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)
        ts = timesteps.to(original_samples.device)
        if self._begin_index is None:
            sched = self.timesteps.to(original_samples.device)
            idx = [self.index_for_timestep(t, sched) for t in ts]
        else:
            idx = [self._step_index if self._step_index is not None else self._begin_index] * ts.shape[0]
        sigma = sigmas[idx].flatten()
        sigma = sigma.view(-1, *([1] * (original_samples.dim() - 1)))
        return (1 - sigma) * original_samples + sigma * noise

    def step(self, model_output: torch.Tensor, timestep: Union[int, torch.Tensor], sample: torch.Tensor,
             return_dict: bool = True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = self._begin_index if self._begin_index is not None else self.index_for_timestep(
                timestep.to(self.timesteps.device) if torch.is_tensor(timestep) else timestep)
        i = self._step_index
        sig_i = float(self.sigmas[i])
        x0 = self._combine(sample.dtype, [(1.0, sample), (-sig_i, model_output)])      # convert_model_output :318-320
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        if use_corrector:
            p = self.this_order
            a, b0, bps, bn = self._coeffs(i, i - 1, [i - 1 - k for k in range(1, p)], p, True)
            sample = self._combine(sample.dtype, [(a, self.last_sample), (b0, self.model_outputs[-1])] +
                                   [(bps[k - 1], self.model_outputs[-(k + 1)]) for k in range(1, p)] + [(bn, x0)])
        self.model_outputs = self.model_outputs[1:] + [x0]
        self.timestep_list = self.timestep_list[1:] + [timestep]
        if self.config.lower_order_final:
            this_order = min(self.config.solver_order, len(self.timesteps) - i)                    # :713-716
        else:
            this_order = self.config.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)                               # :720
        assert self.this_order > 0
        self.last_sample = sample
        p = self.this_order
        a, b0, bps, _ = self._coeffs(i + 1, i, [i - k for k in range(1, p)], p, False)
        prev_sample = self._combine(sample.dtype, [(a, sample), (b0, x0)] +
                                    [(bps[k - 1], self.model_outputs[-(k + 1)]) for k in range(1, p)])
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev_sample,)
        return SchedulerOutput(prev_sample=prev_sample)
