"""VideoCoF denoising pipeline on the HIP-backed DiT.

Mirror of ``WanPipeline`` (``videox_fun/pipeline/pipeline_wan.py:125-138, 516-799``):
same constructor ``(tokenizer, text_encoder, vae, transformer, scheduler)``, same
``__call__`` argument names and ``WanPipelineOutput(videos, ground_videos,
edit_videos)``.  The denoise loop reproduces the reference's glue exactly:

    condition_count = (source_frames-1)//4 + 1                              (:630-631)
    G = 1 if reasoning_frames <= 1 else (reasoning_frames-1)//4 + 1          (:637)
    latents = cat([vae.encode(video).mode(), randn(Fs+G)], dim=2)            (:406-417)
    per step: v = transformer(x, t, ctx, seq_len, fsi=[cc]*B, gfi=[(cc,cc+G)]*B)
              CFG: v = v_uncond + s*(v_text - v_uncond), batch order [uncond, cond] (:700, 731-733)
              v[:, :, :cc] = 0                                              (:736)
              latents = scheduler.step(v, t, latents)[0]                    (:740)
    decode only the grounding and edit segments                             (:760-777)

Prompts: with a ``tokenizer`` (a HuggingFace tokenizer object, e.g. ``AutoTokenizer`` of
``google/umt5-xxl``) and a ``videocof_amd.WanT5EncoderModel`` the strings are encoded on the GPU as in
``_get_t5_prompt_embeds`` (:140-181); alternatively pass ``prompt_embeds`` /
``negative_prompt_embeds`` (lists of ``[len<=512, 4096]`` tensors), or a plain callable
``text_encoder(list[str]) -> list[Tensor]`` without a tokenizer.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Union

import numpy as np
import torch

from .fm_solvers_unipc import FlowUniPCMultistepScheduler

__all__ = ["WanPipeline", "WanPipelineOutput"]


@dataclass
class WanPipelineOutput:
    videos: Optional[Union[torch.Tensor, np.ndarray]]
    ground_videos: Optional[Union[torch.Tensor, np.ndarray]] = None
    edit_videos: Optional[Union[torch.Tensor, np.ndarray]] = None
    latents: Optional[torch.Tensor] = None          # extension: final latents (output_type="latent")


class WanPipeline:
    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None):
        if transformer is None or scheduler is None:
            raise ValueError("WanPipeline needs at least a transformer and a scheduler")
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer, self.scheduler = transformer, scheduler
        self._guidance_scale = 1.0
        self._num_timesteps = 0
        self._interrupt = False
        self._graphed = None
        self._graphed_loop = None
        self.release_workspaces_after_denoise = False     # see __call__: hand the DiT's activation buffers back before the VAE decode
        # Set to a dict to get synchronised wall seconds per stage of the next __call__ ("text_encoder", "vae_encode",
        # "denoise_loop", "vae_decode"): a measuring aid (bench.py's `e2e` object); None = no synchronisation anywhere.
        self.stage_seconds = None

    def _stage(self, name, t0=None):
        """Stage clock for ``stage_seconds``: call without t0 to start (returns the start time), with t0 to stop."""
        if self.stage_seconds is None:
            return None
        import time
        torch.cuda.synchronize()
        if t0 is None:
            return time.perf_counter()
        self.stage_seconds[name] = self.stage_seconds.get(name, 0.0) + time.perf_counter() - t0
        return None

    guidance_scale = property(lambda self: self._guidance_scale)
    num_timesteps = property(lambda self: self._num_timesteps)
    interrupt = property(lambda self: self._interrupt)

    # -------------------------------------------------------------- placement (fast_infer.py:347-362)
    def to(self, *args, device=None, **kwargs):
        """The models are packed on their HIP device by ``load_state_dict``; ``pipeline.to(device)`` only checks that
        the request matches (no parameters to move), so the reference's ``model_full_load`` branch works unchanged."""
        want = device if device is not None else next((a for a in args if isinstance(a, (str, torch.device))), None)
        if want is not None:
            want = torch.device(want)
            have = self.transformer.device
            if want.type != "cuda" or (want.index is not None and have is not None and have.index is not None
                                       and want.index != have.index):
                raise RuntimeError(f"WanPipeline.to({want}): the transformer was loaded on {have}; load it there instead")
        return self

    def enable_model_cpu_offload(self, *args, **kwargs):
        """No-op: 14B bf16 weights (28 GB) + umT5 (11 GB) + VAE stay resident in the 288 GB of HBM; there is nothing to
        offload (the reference's default ``sequential_cpu_offload`` exists for 24-80 GB cards)."""
        return None

    enable_sequential_cpu_offload = enable_model_cpu_offload

    # -------------------------------------------------------------- prompt handling (:140-256, :449-495) -- the reference's argument lists
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]              # :112-116

    def _get_t5_prompt_embeds(self, prompt=None, num_videos_per_prompt: int = 1, max_sequence_length: int = 512, device=None, dtype=None):
        """tokenizer(padding="max_length", max_length=512, truncation) -> text_encoder(ids, mask)[0], each
        sample trimmed to its own token count (pipeline_wan.py:140-181); ``num_videos_per_prompt`` repeats every sample in place."""
        device = device if device is not None else self.transformer.device
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        enc = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                             add_special_tokens=True, return_tensors="pt")
        ids = enc.input_ids if hasattr(enc, "input_ids") else enc["input_ids"]
        mask = enc.attention_mask if hasattr(enc, "attention_mask") else enc["attention_mask"]
        seq_lens = mask.gt(0).sum(dim=1).long().tolist()
        hidden = self.text_encoder(ids.to(device), attention_mask=mask.to(device))[0]
        if dtype is not None:
            hidden = hidden.to(dtype)
        out = [u[:v] for u, v in zip(hidden, seq_lens)]
        # (:176-179 repeats along the sequence axis and views as [B * n, seq, C]: sample b's n copies are consecutive)
        return [e for e in out for _ in range(int(num_videos_per_prompt))]

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance: bool = True, num_videos_per_prompt: int = 1,
                      prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length: int = 512, device=None, dtype=None):
        """:183-256.  Embeddings are LISTS of [len_i, 4096] tensors (what the transformer's ``context`` takes); a [len, 4096] or
        [B, len, 4096] tensor passed in is split into one.  Without a tokenizer the text encoder is called on the strings themselves."""
        device = device if device is not None else self.transformer.device

        def as_list(e):
            if torch.is_tensor(e):
                return [e] if e.dim() == 2 else list(e)
            return list(e)

        def enc(p):
            if self.text_encoder is None:
                raise ValueError("no text_encoder: provide `prompt_embeds`, or build the pipeline with a tokenizer "
                                 "and a WanT5EncoderModel")
            if self.tokenizer is not None:
                return self._get_t5_prompt_embeds(p, num_videos_per_prompt, max_sequence_length, device, dtype)
            return [e.to(device) for e in self.text_encoder(p) for _ in range(int(num_videos_per_prompt))]
        prompt = [prompt] if isinstance(prompt, str) else prompt
        if prompt_embeds is None:
            prompt_embeds = enc(prompt)
        prompt_embeds = as_list(prompt_embeds)
        batch_size = len(prompt) if prompt is not None else len(prompt_embeds)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = batch_size * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            if batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            negative_prompt_embeds = enc(negative_prompt)
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = as_list(negative_prompt_embeds)
            if do_classifier_free_guidance and len(negative_prompt_embeds) != len(prompt_embeds):
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same batch size")
        return prompt_embeds, negative_prompt_embeds

    def check_inputs(self, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:                                                   # :458-459
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(k in self._callback_tensor_inputs
                                                                       for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found "
                             f"{[k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt is not None and negative_prompt_embeds is not None:                           # :481-485
            raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`.")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if torch.is_tensor(prompt_embeds) and torch.is_tensor(negative_prompt_embeds) and \
                prompt_embeds.shape != negative_prompt_embeds.shape:                            # :492-498
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                             f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")

    # -------------------------------------------------------------- latents (:258-419), with the reference's names and argument lists
    @staticmethod
    def _randn(shape, generator, device, dtype):
        """diffusers' randn_tensor for one generator: drawn on the generator's device, then moved."""
        return torch.randn(tuple(shape), generator=generator, device=generator.device if generator is not None else device,
                           dtype=dtype).to(device)

    def _encode_modes(self, video, device, dtype):
        if self.vae is None:
            raise ValueError("no VAE: provide `source_latents` (or full `latents`)")
        video = video.to(device=device, dtype=dtype)
        return torch.cat([self.vae.encode(video[i:i + 1])[0].mode() for i in range(video.shape[0])])      # mode, no mean / std (:404-409)

    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, dtype, device, generator, latents=None):
        """:258-290 -- plain T2V noise [B, C, (num_frames - 1) / 4 + 1, H / 8, W / 8] (x ``scheduler.init_noise_sigma`` if it has one)."""
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        tr, sr = getattr(self.vae, "temporal_compression_ratio", 4), getattr(self.vae, "spatial_compression_ratio", 8)
        shape = (batch_size, num_channels_latents, (num_frames - 1) // tr + 1, height // sr, width // sr)
        latents = self._randn(shape, generator, device, dtype) if latents is None else latents.to(device)
        if hasattr(self.scheduler, "init_noise_sigma"):
            latents = latents * self.scheduler.init_noise_sigma
        return latents

    def prepare_video_latents(self, video, batch_size=1, num_channels_latents=16, height=480, width=832, dtype=torch.float32,
                              device=None, generator=None, condition_count=None, latents=None, timestep=None):
        """:292-341 -- the clip's latents with every frame from ``condition_count`` on replaced by noise."""
        if latents is not None:
            return latents.to(device=device, dtype=dtype)
        init = self._encode_modes(video, device, dtype)
        tr, sr = getattr(self.vae, "temporal_compression_ratio", 4), getattr(self.vae, "spatial_compression_ratio", 8)
        shape = (batch_size, num_channels_latents, (video.shape[2] - 1) // tr + 1, height // sr, width // sr)
        noise = self._randn(shape, generator, device, dtype)
        init[:, :, condition_count:] = noise[:, :, condition_count:]
        return init

    def prepare_video_latents_new(self, video, batch_size=1, num_channels_latents=16, height=480, width=832, dtype=torch.float32,
                                  device=None, generator=None, condition_count=None, latents=None, timestep=None, source_latents=None):
        """:343-378 -- [source latents | noise of the same shape] (the repeat / org layouts)."""
        return self.prepare_cot_video_latents(video, 0, batch_size, num_channels_latents, height, width, dtype, device, generator,
                                              condition_count, latents, timestep, source_latents)

    def prepare_cot_video_latents(self, video, reasoning_latent_count=1, batch_size=1, num_channels_latents=16, height=480, width=832,
                                  dtype=torch.float32, device=None, generator=None, condition_count=None, latents=None, timestep=None,
                                  source_latents=None):
        """:381-419 -- [source latents | noise over reasoning_latent_count + Fs frames].  The reference's argument list (its unused
        entries -- batch_size, num_channels_latents, height, width, condition_count, timestep -- are accepted and unused here too) plus
        ``source_latents``: the clip already encoded (no VAE in the pipeline, or the encode done elsewhere)."""
        if latents is not None:
            return latents.to(device=device, dtype=dtype)
        if source_latents is None:
            source_latents = self._encode_modes(video, device, dtype)
        org = source_latents.to(device=device, dtype=dtype)
        B, Cl, Fs, h, w = org.shape
        return torch.cat([org, self._randn((B, Cl, Fs + reasoning_latent_count, h, w), generator, device, dtype)], dim=2)

    def prepare_extra_step_kwargs(self, generator, eta):
        """:431-446 -- ``eta`` / ``generator`` only for schedulers whose ``step`` takes them (FlowUniPC takes ``generator``)."""
        import inspect
        names = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in names:
            kw["eta"] = eta
        if "generator" in names:
            kw["generator"] = generator
        return kw

    @property
    def attention_kwargs(self):
        return getattr(self, "_attention_kwargs", None)

    def decode_latents(self, latents: torch.Tensor, out: Optional[torch.Tensor] = None) -> np.ndarray:
        """:423-428 -- decode, (x / 2 + 0.5).clamp(0, 1), float32 numpy.  The reference converts on the host
        (``frames.cpu().float().numpy()``: a pageable copy, a CPU cast and, in ``__call__``, a concatenate: ~0.1 s for 81 frames at
        480 x 832); here the cast runs on the device and the frames go ONCE, through page-locked memory, to where they stay:
        ``out`` = a pinned float32 host tensor [B, 3, T, H, W] (or a frame slice of one) that the result is a view of."""
        frames = self.vae.decode(latents.to(self.vae.dtype)).sample
        frames = (frames / 2 + 0.5).clamp(0, 1).float()        # the arithmetic in the VAE's dtype, as the reference does it; the cast on the device
        if not frames.is_cuda:
            return frames.numpy()
        if out is None:
            out = torch.empty(frames.shape, dtype=torch.float32, pin_memory=True)
        if tuple(out.shape) != tuple(frames.shape):
            raise ValueError(f"decode_latents: out {tuple(out.shape)} for frames {tuple(frames.shape)}")
        for b in range(frames.shape[0]):
            for c in range(frames.shape[1]):
                out[b, c].copy_(frames[b, c], non_blocking=True)       # (a frame slice of a longer clip is contiguous per channel)
        torch.cuda.current_stream(frames.device).synchronize()
        return out.numpy()

    # -------------------------------------------------------------- __call__ (:516-799)
    @torch.no_grad()
    def __call__(self, video: Optional[torch.Tensor] = None, prompt=None, negative_prompt=None,
                 height: int = 480, width: int = 720, num_frames: int = 49, source_frames: int = 33,
                 reasoning_frames: int = 4, num_inference_steps: int = 50, timesteps=None,
                 guidance_scale: float = 6.0, num_videos_per_prompt: int = 1, eta: float = 0.0,
                 generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type: str = "numpy",
                 return_dict: bool = False, callback_on_step_end: Optional[Callable] = None,
                 attention_kwargs=None, callback_on_step_end_tensor_inputs=("latents",),
                 max_sequence_length: int = 512, comfyui_progressbar: bool = False,
                 shift: float = 5.0, repeat_rope: bool = True, cot: bool = False,
                 # extensions of this package (keyword-only in spirit; the names above are the reference's, :516-548)
                 source_latents: Optional[torch.Tensor] = None, device=None,
                 weight_dtype: torch.dtype = torch.bfloat16, cache_context: bool = True,
                 skip_source_prediction: bool = True, capture_graph=False):
        # `timesteps`: with a FlowUniPCMultistepScheduler the reference never looks at it (pipeline_wan.py:613-615 calls
        # set_timesteps(num_inference_steps, device=, shift=) and takes scheduler.timesteps); accepted and ignored here too.
        del timesteps
        num_videos_per_prompt = 1                  # :561 -- the reference overrides the argument at the top of __call__, whatever was passed
        self.check_inputs(prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, prompt_embeds, negative_prompt_embeds)
        self._guidance_scale = guidance_scale
        self._attention_kwargs = attention_kwargs                                               # :575 (stored, read by nothing, as there)
        self._interrupt = False
        device = torch.device(device) if device is not None else self.transformer.device
        do_cfg = guidance_scale > 1.0                                                           # :592
        t_stage = self._stage("text_encoder")
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, negative_prompt, do_cfg, num_videos_per_prompt=num_videos_per_prompt, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, max_sequence_length=max_sequence_length, device=device)
        self._stage("text_encoder", t_stage)
        in_prompt_embeds = (negative_prompt_embeds + prompt_embeds) if do_cfg else prompt_embeds   # :605-608

        if not isinstance(self.scheduler, FlowUniPCMultistepScheduler):
            raise NotImplementedError("only FlowUniPCMultistepScheduler is built (fast_infer.py:152)")
        self.scheduler.set_timesteps(num_inference_steps, device=device, shift=shift)           # :613-615
        timesteps = self.scheduler.timesteps
        self._num_timesteps = len(timesteps)

        ratio = getattr(self.vae, "temporal_compression_ratio", 4)
        condition_count = 1 if source_frames == 1 else (source_frames - 1) // ratio + 1         # :630-631
        ground_latent_count = 0
        t_stage = self._stage("vae_encode")
        if cot:
            ground_latent_count = 1 if reasoning_frames <= 1 else (reasoning_frames - 1) // ratio + 1   # :637
            latents = self.prepare_cot_video_latents(video, ground_latent_count, dtype=weight_dtype, device=device, generator=generator,
                                                     condition_count=condition_count, latents=latents, source_latents=source_latents)
        else:
            # repeat / org layouts: noise block has the source's frame count (:653-677 -> :343-378)
            latents = self.prepare_video_latents_new(video, dtype=weight_dtype, device=device, generator=generator,
                                                     condition_count=condition_count, latents=latents, source_latents=source_latents)
        self._stage("vae_encode", t_stage)
        B, _, Ftot, hl, wl = latents.shape
        ps = self.transformer.config.patch_size
        seq_len = math.ceil((hl * wl) / (ps[1] * ps[2]) * Ftot)                                 # :686-689
        self.transformer.num_inference_steps = num_inference_steps
        prev_cache = getattr(self.transformer, "cache_context", False)
        if hasattr(self.transformer, "cache_context"):
            self.transformer.cache_context = cache_context    # the prompt is fixed across steps
        if hasattr(self.transformer, "clear_context_cache"):
            self.transformer.clear_context_cache()            # never reuse another call's text K/V
        prev_skip = getattr(self.transformer, "skip_source_frames", 0)
        if hasattr(self.transformer, "skip_source_frames"):
            # noise_pred[:, :, :condition_count] is zeroed below (:736): the last block need not produce it
            self.transformer.skip_source_frames = condition_count if skip_source_prediction else 0

        # the CoF mask of :736 on the device: the unpatchify kernel writes zeros for the source frames
        prev_mask = getattr(self.transformer, "mask_source_frames", None)
        if prev_mask is not None:
            self.transformer.mask_source_frames = condition_count
        forward = self.transformer
        if capture_graph not in (False, True, "step", "loop"):
            raise ValueError(f"capture_graph={capture_graph!r}: False, True / 'step' (one hipGraph per forward) or 'loop' (the whole loop)")
        if capture_graph in (True, "step"):
            # one hipGraph per call shape, kept across calls (videocof_amd/graph.py): step 0 of the first call runs
            # eagerly, step 1 is captured, every later step (and call) is a replay -- bit-identical latents
            if self._graphed is None or self._graphed.model is not self.transformer:
                from .graph import GraphedForward
                self._graphed = GraphedForward(self.transformer)
            forward = self._graphed

        use_rope_map = repeat_rope and (video is not None or source_latents is not None or latents is not None)

        def denoise(latents, embeds):
            """The loop of :694-740 on `latents` with the prompt embeddings `embeds`; eager, or recorded into a hipGraph."""
            for i, t in enumerate(timesteps):                                                   # :694
                self.transformer.current_steps = i
                if self._interrupt:
                    continue
                latent_model_input = torch.cat([latents] * 2) if do_cfg else latents           # :700
                latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
                timestep = t.expand(latent_model_input.shape[0])                                # :705
                nb = latent_model_input.shape[0]
                fsi = gfi = None
                # :713 is `if repeat_rope and video is not None`, and the reference cannot run without `video` at all
                # (`video.to(...)` at :397 precedes the `latents` short-cut), so on every call the reference accepts
                # the condition is just `repeat_rope`.  Here `video` may be omitted when `latents` (reference argument)
                # or `source_latents` (extension) carry the encoded source: such calls are treated like the reference
                # treats the same call WITH its video; a call with none of the three has no source segment and keeps
                # plain T2V positions.  Documented in INTEGRATION.md section B.
                if use_rope_map:
                    fsi = [condition_count] * nb                                                # :713
                    if cot:
                        gfi = [(condition_count, condition_count + ground_latent_count)] * nb  # :716-718
                noise_pred = forward(x=latent_model_input, context=embeds, t=timestep,
                                     seq_len=seq_len, frame_split_indices=fsi,
                                     ground_frame_indices=gfi)                                  # :721-728
                if do_cfg:
                    nu, nt = noise_pred.chunk(2)
                    noise_pred = nu + self.guidance_scale * (nt - nu)                           # :731-733
                if prev_mask is None:
                    noise_pred[:, :, :condition_count] = 0                                      # :736
                # else: already zero -- written by wan_unpatchify(zero_frames); CFG keeps it (0 + s * (0 - 0))
                latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]     # :740
                if callback_on_step_end is not None:
                    # :742-750 -- the tensors named in callback_on_step_end_tensor_inputs; only `latents` coming back has an effect
                    # (there as here: the embeddings of the loop were concatenated before it started, :605-608)
                    vals = {"latents": latents, "prompt_embeds": prompt_embeds, "negative_prompt_embeds": negative_prompt_embeds}
                    out = callback_on_step_end(self, i, t, {k: vals[k] for k in (callback_on_step_end_tensor_inputs or ())})
                    if out:
                        latents = out.pop("latents", latents)
            return latents

        t_stage = self._stage("denoise_loop")
        try:
            if capture_graph == "loop":
                # the whole loop as ONE hipGraph (videocof_amd.GraphedLoop): first call of a signature eager, second captures
                if callback_on_step_end is not None:
                    raise NotImplementedError("capture_graph='loop' replays all steps in one launch: no per-step callback")
                if not getattr(self.transformer, "cache_context", False):
                    raise NotImplementedError("capture_graph='loop' needs cache_context (the text K/V are part of the graph)")
                if self._graphed_loop is None or self._graphed_loop.model is not self.transformer:
                    from .graph import GraphedLoop
                    self._graphed_loop = GraphedLoop(self.transformer)
                self.scheduler.set_begin_index(0)       # no device round trip (index_for_timestep) inside a capture
                key = (int(num_inference_steps), float(shift), bool(do_cfg), float(guidance_scale) if do_cfg else 0.0,
                       int(condition_count), int(ground_latent_count), bool(cot), bool(use_rope_map), int(seq_len),
                       int(self.scheduler.config.solver_order), bool(self.scheduler.config.lower_order_final))

                def loop_fn(lat, embeds):
                    self.scheduler._reset()
                    self.scheduler.set_begin_index(0)
                    return denoise(lat, embeds)
                latents = self._graphed_loop(key, latents, in_prompt_embeds, loop_fn, keep=(timesteps, self.scheduler.sigmas))
            else:
                latents = denoise(latents, in_prompt_embeds)
        finally:
            if hasattr(self.transformer, "cache_context"):
                self.transformer.cache_context = prev_cache
            if hasattr(self.transformer, "clear_context_cache"):
                self.transformer.clear_context_cache()        # releases the hoisted K/V^T and the prompt embeddings
            if hasattr(self.transformer, "skip_source_frames"):
                self.transformer.skip_source_frames = prev_skip
            if prev_mask is not None:
                self.transformer.mask_source_frames = prev_mask
            if self.release_workspaces_after_denoise and hasattr(self.transformer, "release_workspaces"):
                # opt-in: 7 GB of activation buffers at 14B / 67k tokens handed back before the VAE decode (for hosts that share
                # the device); the sets a captured graph replays from stay.  Off by default: 288 GB of HBM do not need it, every call
                # would re-allocate and zero-fill the set, and in capture_graph='loop' mode the capture call would then allocate
                # (and record the zero-fill of) its workspaces inside the graph.
                self.transformer.release_workspaces(keep_pinned=True)
        self._stage("denoise_loop", t_stage)

        # -- decode (:757-790)
        t_stage = self._stage("vae_decode")
        ground_video = edit_video = video_out = None
        if output_type == "numpy":
            if self.vae is None:
                raise ValueError("output_type='numpy' needs a VAE; use output_type='latent'")
            if cot:
                g0, g1 = condition_count, condition_count + ground_latent_count
                # grounding | edit frames side by side in ONE page-locked clip (what the reference builds with np.concatenate, :786):
                # each segment is decoded straight into its frame slice, `ground_videos` / `edit_videos` are views of `videos`
                tcr = self.vae.config.temporal_compression_ratio
                nfr = lambda n: 1 + tcr * (n - 1) if n > 0 else 0
                ng = nfr(g1 - g0) if (g1 > g0 and g0 < Ftot) else 0
                ne = nfr(Ftot - g1) if g1 < Ftot else 0
                scr = self.vae.config.spatial_compression_ratio
                clip = None
                if latents.is_cuda and ng + ne > 0:
                    clip = torch.empty((latents.shape[0], 3, ng + ne, latents.shape[3] * scr, latents.shape[4] * scr),
                                       dtype=torch.float32, pin_memory=True)
                parts = []
                if ng:
                    ground_video = self.decode_latents(latents[:, :, g0:g1], None if clip is None else clip[:, :, :ng])
                    parts.append(ground_video)
                if ne:
                    edit_video = self.decode_latents(latents[:, :, g1:], None if clip is None else clip[:, :, ng:])
                    parts.append(edit_video)
                video_out = clip.numpy() if clip is not None else np.concatenate(parts, axis=2)
            else:
                if condition_count < Ftot:
                    edit_video = self.decode_latents(latents[:, :, condition_count:])
                video_out = edit_video
            if not return_dict:
                video_out = torch.from_numpy(video_out) if isinstance(video_out, np.ndarray) else video_out
                ground_video = torch.from_numpy(ground_video) if isinstance(ground_video, np.ndarray) else ground_video
                edit_video = torch.from_numpy(edit_video) if isinstance(edit_video, np.ndarray) else edit_video
        elif output_type != "latent":
            raise ValueError(f"output_type {output_type!r} not supported ('numpy' or 'latent')")
        self._stage("vae_decode", t_stage)
        return WanPipelineOutput(videos=video_out, ground_videos=ground_video, edit_videos=edit_video, latents=latents)
