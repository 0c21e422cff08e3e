"""videocof_amd -- MI355X-native Wan2.1-DiT video denoising path (VideoCoF-compatible).

Host side mirrors the reference's interface for this path only:

    videocof_amd.WanTransformer3DModel      <- videox_fun/models/wan_transformer3d.py
    videocof_amd.attention                  <- videox_fun/models/attention_utils.py
    videocof_amd.FlowUniPCMultistepScheduler<- videox_fun/utils/fm_solvers_unipc.py
    videocof_amd.WanPipeline                <- videox_fun/pipeline/pipeline_wan.py
    videocof_amd.AutoencoderKLWan           <- videox_fun/models/wan_vae.py
    videocof_amd.WanT5EncoderModel          <- videox_fun/models/wan_text_encoder.py (umT5 encoder)
    videocof_amd.lora_utils                 <- videox_fun/utils/lora_utils.py (merge_lora on state dicts)
    videocof_amd.cache_utils                <- videox_fun/models/cache_utils.py (TeaCache, opt-in)
    videocof_amd.dist                       <- videox_fun/dist/{fuser,wan_xfuser}.py (Ulysses on RCCL)

Device arithmetic lives in ``libwan_hip.so`` (csrc/, C ABI in include/wan_hip.h).
"""
from .fm_solvers_unipc import FlowUniPCMultistepScheduler  # noqa: F401
from .pipeline_wan import WanPipeline, WanPipelineOutput  # noqa: F401
from .wan_transformer3d import WanTransformer3DModel  # noqa: F401
from .graph import GraphedForward, GraphedLoop  # noqa: F401
from .cache_utils import TeaCache, get_teacache_coefficients  # noqa: F401
from .attention_utils import attention, flash_attention  # noqa: F401
from .wan_vae import AutoencoderKLWan  # noqa: F401
from .wan_text_encoder import WanT5EncoderModel  # noqa: F401

__version__ = "0.1.0"
