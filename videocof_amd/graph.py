"""hipGraph replay of the DiT forward (the loop of pipeline_wan.py:694-740).

One denoise step launches ~15 kernels per layer (600 at 40 layers) plus torch glue; at the 14B / 67k-token shape that
is 0.1 % of a 4.4 s step, at small shapes (BASELINE configs[0], the 1.3B model) it is most of the step.
``GraphedForward`` captures ``WanTransformer3DModel.forward`` -- every HIP kernel of libwan_hip.so is enqueued on
torch's current stream, so stream capture records them like torch's own kernels -- for one call shape and replays it
with new latents / timestep / prompt:

* latents and timestep live in static buffers that are overwritten before each replay;
* the step-invariant text K/V are hoisted (``cache_context``) and recomputed EAGERLY, in place, when the prompt changes,
  so the graph never contains them and stays valid across prompts of the same batch size;
* activation workspaces and the attention scratch are the model's cached buffers; every entry PINS the set it was captured
  with (the model keeps one workspace set per call shape and never evicts a pinned one; the entry also holds the attention
  scratch tensors), so a second call shape, a larger scratch request or an eager call in between can never free memory a
  captured graph still writes to;
* anything that replaces device memory the graph has baked in -- ``load_state_dict``, ``enable / disable_fp8_linear``,
  ``release_workspaces()`` -- bumps the model's ``_graph_epoch``; entries of an older epoch are dropped, not replayed
  (``merge_lora`` edits the weights in place: same addresses, the next replay reads the merged values);
* the CoF mask is part of the captured unpatchify kernel (``mask_source_frames``).

A replay executes the same kernels with the same arguments in the same order as the eager call: the result is
bit-identical (``tests/test_gpu_dit.py::test_graph_replay_is_bit_identical``).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

__all__ = ["GraphedForward", "GraphedLoop"]


def _check_capturable(model) -> None:
    """Single device, or a sequence-parallel model whose collectives run on the communicator the LIBRARY owns (``LibraryComm``).
    Round 4 recorded an SP forward with the RCCL calls on a side stream forked from and joined back into the capturing stream by events
    (the fork / join pattern of stream capture): the eager warm-up call was fine, `hipStreamEndCapture` then segfaulted inside the
    runtime (ROCm 7.2, one rank, both capture-error modes; profiles/r04/sp_graph_capture_rccl_segfault.log).  Round 5: while the
    caller's stream is being captured the library communicator enqueues every collective on THAT stream (csrc/sp_comm.cpp,
    `sp_runs_inline`: no side stream, no events -- at P = 8 an exchange is < 1 % of a layer, so the overlap it gives up inside a
    replayed graph is small), which is what this check admits.  torch.distributed's process group stays eager: its collectives run
    on a stream of its own.
    EVIDENCE: capture + bit-identical replay of that path has only ever run with ONE rank (tests/test_gpu_sp.py::
    test_graph_capture_of_a_sequence_parallel_forward; the development boxes have one GPU).  RCCL kernels of several ranks inside
    graphs that the ranks replay in lockstep are untested, so world_size > 1 is refused unless ``WAN_SP_GRAPH_MULTI_RANK=1`` opts in
    (for whoever has two GPUs to run the check on).  A communicator that was captured once is deliberately NOT destroyed by
    wan_sp_destroy (the graphs hold its RCCL kernels); it lives until the process exits."""
    if model.sp_world_size != 1 or getattr(model, "force_ulysses", False):
        from .dist import LibraryComm
        if not isinstance(getattr(model, "_sp", None), LibraryComm):
            raise NotImplementedError("graph capture of a sequence-parallel forward needs the library-owned communicator "
                                      "(init_sequence_parallel(backend=\"library\")); torch.distributed collectives stay eager")
        if model.sp_world_size != 1 and os.environ.get("WAN_SP_GRAPH_MULTI_RANK", "0") != "1":
            raise NotImplementedError("graph capture of a sequence-parallel forward is verified with one rank only; more ranks "
                                      "replaying RCCL collectives from graphs in lockstep are untested (set WAN_SP_GRAPH_MULTI_RANK=1 "
                                      "to try)")


class _Entry:
    __slots__ = ("x", "t", "graph", "out", "calls", "kv", "epoch", "bufs", "attn_bufs")


class GraphedForward:
    """Callable with the signature of ``WanTransformer3DModel.forward`` (T2V / CoF arguments)."""

    def __init__(self, model, warmup_calls: int = 1):
        _check_capturable(model)
        self.model = model
        self.warmup_calls = max(1, int(warmup_calls))
        self._entries: Dict[tuple, _Entry] = {}
        self.replays = 0

    def reset(self) -> None:
        """Drop every captured graph and un-pin its workspaces (the model's next eager call of another shape frees them)."""
        for ent in self._entries.values():
            if ent.bufs is not None:
                ent.bufs.pinned = False
        self._entries.clear()

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    @torch.no_grad()
    def __call__(self, x, t, context, seq_len, frame_split_indices=None, ground_frame_indices=None, **kw):
        m = self.model
        if any(v is not None for v in kw.values()):
            raise NotImplementedError("graph capture supports the T2V / CoF arguments only")
        if m.teacache is not None:
            raise NotImplementedError("TeaCache decides per step on the host whether the blocks run: not capturable")
        if isinstance(x, (list, tuple)):
            x = torch.stack(list(x))
        t = t.reshape(-1)
        if t.is_floating_point() and bool((t != t.round()).any()):
            # the eager forward takes fractional timesteps; a static integer buffer would truncate them silently
            key_t = t.dtype
        else:
            key_t = torch.int64
        key = (tuple(x.shape), x.dtype, key_t, int(seq_len), tuple(frame_split_indices or ()),
               tuple(tuple(g) for g in (ground_frame_indices or ())), int(m.skip_source_frames),
               int(m.mask_source_frames), len(context), torch.cuda.current_device(), tuple(m._fp8),
               bool(m.use_block_composite), bool(m.use_forward_composite), tuple(m.fp8_attn_exponents), bool(m.fp8_attn_smooth_k), bool(getattr(m, 'fp8_attn_calibrate', False)))
        epoch = m._graph_epoch
        for k in [k for k, e in self._entries.items() if e.epoch != epoch]:
            stale = self._entries.pop(k)                 # weights / fp8 copies / workspaces were replaced since the capture
            if stale.bufs is not None:
                stale.bufs.pinned = False
        ent = self._entries.get(key)
        if ent is None:
            ent = self._entries[key] = _Entry()
            ent.x, ent.t = torch.empty_like(x), torch.empty(x.shape[0], device=x.device, dtype=key_t)
            ent.graph, ent.out, ent.calls, ent.kv, ent.epoch, ent.bufs, ent.attn_bufs = None, None, 0, None, epoch, None, None
        ent.x.copy_(x)
        ent.t.copy_(t.to(key_t).expand(x.shape[0]))
        prev = m.cache_context
        m.cache_context = True
        try:
            # eager, outside the graph; in place (into the buffers the graph reads) when the prompt changed or the
            # model's cache was cleared in between
            ent.kv = m._hoisted_context(context, x.shape[0], into=ent.kv)
            args = (ent.x, ent.t, context, seq_len)
            kwargs = dict(frame_split_indices=frame_split_indices, ground_frame_indices=ground_frame_indices)
            if ent.graph is None:
                ent.calls += 1
                if ent.calls <= self.warmup_calls:           # eager: allocates workspaces, sets kernel attributes
                    return m(*args, **kwargs)
                events, m._attn_events = m._attn_events, None
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph):
                        ent.out = m(*args, **kwargs)
                finally:
                    m._attn_events = events
                ent.graph = graph
                # pin what the graph has baked in besides x / t / out / kv: the activation workspace set of this shape and the
                # attention scratch tensors of the four call sites (a later, larger request re-allocates the site's scratch;
                # this reference keeps the captured one alive)
                ent.bufs = m._bufs[m._bufs_last]
                ent.bufs.pinned = True
                ent.attn_bufs = [ws.buf for ws in (m._ws_self, m._ws_cross, m._ws_self_sfx, m._ws_cross_sfx)]
            ent.graph.replay()
            self.replays += 1
            return ent.out.clone()
        finally:
            m.cache_context = prev


class _LoopEntry:
    __slots__ = ("x", "ctx", "graph", "out", "calls", "epoch", "bufs", "attn_bufs", "keep")


class GraphedLoop:
    """The WHOLE denoise loop of ``WanPipeline.__call__`` (pipeline_wan.py:694-740: every step's forward, the CFG combine, the CoF
    mask and the UniPC update) as ONE hipGraph per call signature.  ``GraphedForward`` replays one forward and leaves the
    scheduler update and the CFG arithmetic eager between replays -- ~10 % of a step at launch-bound shapes; here a video is one
    graph launch.

    What makes it capturable: the sampler's coefficients are host float64 scalars that depend only on (steps, shift) -- they are
    baked into the captured ``wan_lincomb`` launches; the timesteps are device tensors kept alive by the entry; the latents and the
    prompt embeddings (zero-padded to ``text_len`` rows, which is what the text embedding does with them anyway) live in static
    buffers refreshed before each replay; the text K/V are computed INSIDE the graph by the first forward and reused by the later
    steps.  Call 1 of a signature runs eagerly (it allocates the workspaces and returns the result), call 2 captures and replays,
    later calls replay.  Entries pin their workspaces and follow the model's ``_graph_epoch`` exactly like ``GraphedForward``'s."""

    def __init__(self, model):
        _check_capturable(model)
        self.model = model
        self._entries: Dict[tuple, _LoopEntry] = {}
        self.replays = 0

    def reset(self) -> None:
        for ent in self._entries.values():
            if ent.bufs is not None:
                ent.bufs.pinned = False
        self._entries.clear()

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    @torch.no_grad()
    def __call__(self, key: tuple, latents: torch.Tensor, context, loop_fn, keep=()):
        """``loop_fn(latents, context) -> final latents`` runs the loop (eagerly or under capture); ``keep``: objects whose device
        memory the captured launches read (the scheduler's timestep tensor) -- held by the entry."""
        m = self.model
        if m.teacache is not None:
            raise NotImplementedError("TeaCache decides per step on the host whether the blocks run: not capturable")
        T, D = m.text_len, m.text_dim
        key = key + (tuple(latents.shape), latents.dtype, len(context), torch.cuda.current_device(), tuple(m._fp8),
                     bool(m.use_block_composite), bool(m.use_forward_composite), int(m.skip_source_frames), int(m.mask_source_frames),
                     tuple(m.fp8_attn_exponents), bool(m.fp8_attn_smooth_k), bool(getattr(m, 'fp8_attn_calibrate', False)))
        epoch = m._graph_epoch
        for k in [k for k, e in self._entries.items() if e.epoch != epoch]:
            stale = self._entries.pop(k)
            if stale.bufs is not None:
                stale.bufs.pinned = False
        ent = self._entries.get(key)
        if ent is None:
            ent = self._entries[key] = _LoopEntry()
            ent.x, ent.ctx, ent.graph, ent.out, ent.calls, ent.epoch = None, None, None, None, 0, epoch
            ent.bufs, ent.attn_bufs, ent.keep = None, None, None
        ent.calls += 1
        if ent.calls == 1:
            return loop_fn(latents, context)                 # eager: the result of this call, and the warm-up of the capture
        if ent.graph is None:
            ent.x = torch.empty_like(latents)
            ent.ctx = [torch.zeros(T, D, device=latents.device, dtype=u.dtype) for u in context]
            ent.keep = tuple(keep)
        ent.x.copy_(latents)
        for buf, u in zip(ent.ctx, context):
            if u.shape[0] > T or u.shape[1] != D:
                raise ValueError(f"context has shape {tuple(u.shape)}; expected [<= {T}, {D}]")
            buf[:u.shape[0]].copy_(u)
            buf[u.shape[0]:].zero_()
        if ent.graph is None:
            m._ctx_cache = None                              # the text K/V are recorded as part of the graph
            events, m._attn_events = m._attn_events, None
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    ent.out = loop_fn(ent.x, ent.ctx)
            finally:
                m._attn_events = events
            ent.graph = graph
            ent.bufs = m._bufs[m._bufs_last]
            ent.bufs.pinned = True
            ent.attn_bufs = [ws.buf for ws in (m._ws_self, m._ws_cross, m._ws_self_sfx, m._ws_cross_sfx)]
        ent.graph.replay()
        self.replays += 1
        return ent.out.clone()
