"""Import shim for the *reference* Python sources (TEST INFRASTRUCTURE ONLY).

Runs only in the build container where ``/root/reference`` is mounted; nothing in
``-m gpu`` tests, ``bench.py`` or ``smoke()`` may import this module (the reference
does not exist on the GPU box).  It is used by ``oracle/gen_golden.py`` to
produce the committed fixtures under ``tests/golden/`` and by
``tests/test_oracle_vs_reference.py`` (live oracle-vs-reference checks on fresh inputs; skipped when the reference is
absent).

The reference needs ``diffusers`` (absent here) for a few base classes only.  We
install minimal stand-in modules in ``sys.modules`` -- behavioural stubs of
*diffusers*, not of the reference -- and then load the reference files by path
(SURVEY.md section 8c).  No reference source is copied.
"""
from __future__ import annotations

import enum
import importlib.util
import inspect
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("VIDEOCOF_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(
        REFERENCE_ROOT, "videox_fun", "models", "wan_transformer3d.py"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so sub-imports resolve
    sys.modules[name] = m
    return m


def _install_diffusers_stubs() -> None:
    if "diffusers" in sys.modules and not getattr(sys.modules["diffusers"], "_vcof_stub", False):
        return  # a real diffusers is present; use it

    class ConfigMixin:
        config_name = "config.json"

        @classmethod
        def from_config(cls, config, **kwargs):
            sig = inspect.signature(cls.__init__).parameters
            kw = {k: v for k, v in dict(config).items() if k in sig}
            kw.update({k: v for k, v in kwargs.items() if k in sig})
            return cls(**kw)

        def register_to_config(self, **kwargs):
            cfg = getattr(self, "config", None)
            if cfg is None:
                self.config = SimpleNamespace(**kwargs)
            else:
                cfg.__dict__.update(kwargs)

    def register_to_config(init):
        sig = inspect.signature(init)

        def wrapped(self, *args, **kwargs):
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
            self.config = SimpleNamespace(**cfg)
            init(self, *args, **kwargs)

        wrapped.__wrapped__ = init
        wrapped.__signature__ = sig
        return wrapped

    class ModelMixin(nn.Module):
        @property
        def dtype(self):
            return next(self.parameters()).dtype

        @property
        def device(self):
            return next(self.parameters()).device

    class FromOriginalModelMixin:
        pass

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    logging = SimpleNamespace(get_logger=lambda *a, **k: _Logger())

    def is_torch_version(op, ver):
        from packaging import version
        cur = version.parse(torch.__version__.split("+")[0])
        ref = version.parse(ver)
        return {"<": cur < ref, "<=": cur <= ref, ">": cur > ref,
                ">=": cur >= ref, "==": cur == ref}[op]

    class DecoderOutput:
        def __init__(self, sample):
            self.sample = sample

    class DiagonalGaussianDistribution:
        def __init__(self, parameters):
            self.parameters = parameters
            self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)

        def mode(self):
            return self.mean

    class AutoencoderKLOutput:
        def __init__(self, latent_dist):
            self.latent_dist = latent_dist

        def __getitem__(self, i):
            return (self.latent_dist,)[i]

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    class KarrasDiffusionSchedulers(enum.Enum):
        UniPCMultistepScheduler = 1

    root = _mod("diffusers", _vcof_stub=True)
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin,
         register_to_config=register_to_config)
    _mod("diffusers.loaders")
    _mod("diffusers.loaders.single_file_model", FromOriginalModelMixin=FromOriginalModelMixin)
    _mod("diffusers.models")
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.lora", LoRACompatibleConv=type("LoRACompatibleConv", (nn.Conv2d,), {}),
         LoRACompatibleLinear=type("LoRACompatibleLinear", (nn.Linear,), {}))
    _mod("diffusers.models.autoencoders")
    _mod("diffusers.models.autoencoders.vae", DecoderOutput=DecoderOutput,
         DiagonalGaussianDistribution=DiagonalGaussianDistribution)
    _mod("diffusers.models.modeling_outputs", AutoencoderKLOutput=AutoencoderKLOutput)
    _mod("diffusers.utils", is_torch_version=is_torch_version, logging=logging,
         deprecate=lambda *a, **k: None, is_scipy_available=lambda: True)
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         SchedulerOutput=SchedulerOutput,
         KarrasDiffusionSchedulers=KarrasDiffusionSchedulers)
    del root


def _load_by_path(modname: str, relpath: str):
    path = os.path.join(REFERENCE_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_CACHE = {}


def load_reference(sp_rank: int | None = None, sp_size: int | None = None) -> SimpleNamespace:
    """Return a namespace with the reference modules of the hot path.

    ``sp_rank``/``sp_size`` stub the two xfuser accessors that
    ``videox_fun/dist/wan_xfuser.py`` reads, so that its rank-sliced RoPE can be
    evaluated without xfuser (golden (11) of SURVEY.md section 8c).
    """
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    key = (sp_rank, sp_size)
    if key in _CACHE:
        return _CACHE[key]
    _install_diffusers_stubs()
    # warnings from torch.cuda.amp deprecations are noise here
    import warnings
    warnings.filterwarnings("ignore", category=FutureWarning)

    _mod("videox_fun")
    _mod("videox_fun.models")
    utils = _mod("videox_fun.utils")
    cfg = _load_by_path("videox_fun.utils.cfg_optimization",
                        "videox_fun/utils/cfg_optimization.py")
    utils.cfg_skip = cfg.cfg_skip
    # videox_fun/dist/fuser.py:27-33 sets these six names to None without xfuser
    dist = _mod("videox_fun.dist",
                get_sequence_parallel_rank=None,
                get_sequence_parallel_world_size=None,
                get_sp_group=None, usp_attn_forward=None,
                xFuserLongContextAttention=None, get_world_group=None)
    fuser = _mod("videox_fun.dist.fuser",
                 get_sequence_parallel_rank=(lambda: sp_rank) if sp_rank is not None else None,
                 get_sequence_parallel_world_size=(lambda: sp_size) if sp_size is not None else None,
                 get_sp_group=None, init_distributed_environment=None,
                 initialize_model_parallel=None, xFuserLongContextAttention=None)
    del dist, fuser

    ns = SimpleNamespace()
    ns.attention_utils = _load_by_path("videox_fun.models.attention_utils",
                                       "videox_fun/models/attention_utils.py")
    ns.cache_utils = _load_by_path("videox_fun.models.cache_utils",
                                   "videox_fun/models/cache_utils.py")
    ns.camera = _load_by_path("videox_fun.models.wan_camera_adapter",
                              "videox_fun/models/wan_camera_adapter.py")
    ns.transformer = _load_by_path("videox_fun.models.wan_transformer3d",
                                   "videox_fun/models/wan_transformer3d.py")
    ns.vae = _load_by_path("videox_fun.models.wan_vae", "videox_fun/models/wan_vae.py")
    ns.unipc = _load_by_path("videox_fun.utils.fm_solvers_unipc",
                             "videox_fun/utils/fm_solvers_unipc.py")
    if sp_rank is not None:
        ns.wan_xfuser = _load_by_path("videox_fun.dist.wan_xfuser",
                                      "videox_fun/dist/wan_xfuser.py")

    def lora_utils():
        return _load_by_path("videox_fun.utils.lora_utils", "videox_fun/utils/lora_utils.py")
    ns.load_lora_utils = lora_utils

    def text_encoder():
        return _load_by_path("videox_fun.models.wan_text_encoder", "videox_fun/models/wan_text_encoder.py")
    ns.load_text_encoder = text_encoder
    _CACHE[key] = ns
    return ns
