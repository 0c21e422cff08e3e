"""Sequence parallelism for the DiT: Ulysses head all-to-all on RCCL over xGMI.

Replaces ``videox_fun/dist/fuser.py`` (process-group bootstrap through xfuser) and
``videox_fun/dist/wan_xfuser.py`` (``usp_attn_forward`` through
``xFuserLongContextAttention``), whose third-party backend is absent from the
reference tree and whose attention forward does not accept VideoCoF's
``frame_split_indices`` ("Facts").  This layer is CoF-aware because RoPE
is applied on the local token shard with *global* token indices before the
exchange (``wan_rope_params.token_offset``).

Partition: the (f,h,w)-ordered token sequence, padded to a multiple of P, is cut
into P contiguous chunks (wan_transformer3d.py:904-905, 949-953).  Everything
except self-attention is token-local.  Per layer:

    q,k [Ll, H*128] --all-to-all--> [L, (H/P)*128]      (scatter heads / gather tokens)
    v^T [H*128, Ll] --all-to-all--> [(H/P)*128, L]
    flash attention on H/P heads over the full sequence
    o   [L, (H/P)*128] --all-to-all--> [Ll, H*128]      (inverse)

and one all-gather of the head output [Ll, 64] per forward (:1085-1086).
xGMI is a full mesh, so an all-to-all drives all 7 links of a GPU concurrently;
per-link traffic per tensor is local_bytes / P.  Collectives are issued with
``async_op=True`` (RCCL runs them on the process group's own HIP stream) and
waited for right before the attention launch, so the V projection GEMM overlaps
the q/k exchange.

Everything here is plain ``torch.distributed`` tensor plumbing and runs
identically on gloo/CPU (tests) and RCCL/GPU.

num_heads % P != 0 (12 heads on 8 GPUs; the reference's answer is ``ring_degree``, dist/fuser.py:46-49,
not built here): the model pads its heads to a multiple of P with zero-weight dummy heads dealt
round-robin (``WanTransformer3DModel._pad_heads_for_ulysses``), so this layer only ever sees equal splits.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

_SP: Optional["SequenceParallelGroup"] = None


class SequenceParallelGroup:
    """The Ulysses group on a torch.distributed process group.  NOTE on ``all_gather_tokens`` (both transports): the result is a
    view of a PERSISTENT per-shape buffer owned by the group -- the next call with the same shape overwrites it.  Consume (or
    clone) it before gathering again; the model's forward copies it out through the unpatchify kernel."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        # gloo cannot move device memory: stage through the host (used only by the 1-GPU
        # 2-rank validation test; RCCL exchanges device buffers directly over xGMI)
        self._host_staged = dist.get_backend(group) == "gloo"

    def _a2a(self, recv, send, async_op):
        if self._host_staged and send.is_cuda:
            s_cpu = send.cpu()
            r_cpu = torch.empty_like(s_cpu)
            dist.all_to_all_single(r_cpu, s_cpu, group=self.group)
            recv.copy_(r_cpu)
            return None
        return dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)

    # ------------------------------------------------------------------ wire-layout exchange (the DiT's per-layer path)
    # The model writes q / k / V^T straight into the send layouts of include/wan_hip.h (a21) from its kernels' epilogues
    # and reads the arrived buffers in place, so per layer the exchange layer adds nothing but the collective:
    # `exchange` is ONE all_to_all_single with equal splits between two persistent flat buffers.
    def exchange(self, recv: torch.Tensor, send: torch.Tensor, async_op: bool = False):
        """recv[s-th slab] <- send[my slab] of rank s (flat contiguous buffers of equal size, a multiple of P).
        Returns a callable that waits for completion (async_op) or None."""
        if send.numel() != recv.numel() or send.numel() % self.world_size or not (send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("exchange: send / recv must be contiguous, of equal size, divisible by the group size")
        work = self._a2a(recv.view(-1), send.view(-1), async_op)
        if not async_op:
            return None
        return (lambda: work.wait()) if work is not None else (lambda: None)

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """Element-wise maximum of a small fp32 tensor over the group, in place (what the ranks agree their fp8 attention
        exponents on: once per layer, on the first forward after ``enable_fp8_linear``)."""
        if self._host_staged and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    # torch statements of the wire layouts (CPU tests and documentation; the GPU path uses wan_sp_* / wan_rmsnorm_rope_sp)
    def pack_heads_ref(self, x: torch.Tensor) -> torch.Tensor:
        """[B, T, C] -> token-major wire [P, T, B, C/P]."""
        B, T, C = x.shape
        return x.reshape(B, T, self.world_size, C // self.world_size).permute(2, 1, 0, 3).contiguous()

    def unpack_heads_ref(self, wire: torch.Tensor) -> torch.Tensor:
        """arrived token-major wire [P, T, B, Cl] -> [B, T, P*Cl] (column s * Cl + c: the o projection's input)."""
        P, T, B, Cl = wire.shape
        return wire.permute(2, 1, 0, 3).reshape(B, T, P * Cl)

    @staticmethod
    def split_wire_ref(wire: torch.Tensor, split: int) -> torch.Tensor:
        """token-major wire [P, T, B, Cl] -> the two-head-group form of include/wan_hip.h's ``*_split`` kernels, flat: channels
        [0, split) of every slab as a complete wire buffer [P, T, B, split], then channels [split, Cl) as [P, T, B, Cl - split]."""
        return torch.cat([wire[..., :split].reshape(-1), wire[..., split:].reshape(-1)])

    @staticmethod
    def join_wire_ref(flat: torch.Tensor, P: int, T: int, B: int, Cl: int, split: int) -> torch.Tensor:
        """The inverse of ``split_wire_ref``: [P, T, B, Cl]."""
        n0 = P * T * B * split
        return torch.cat([flat[:n0].view(P, T, B, split), flat[n0:P * T * B * Cl].view(P, T, B, Cl - split)], dim=-1)

    def pack_vt_ref(self, vt: torch.Tensor) -> torch.Tensor:
        """[B, C, T] -> channel-major wire [P, C/P, B, T] (what the V projection writes per sample with ldo = B*T)."""
        B, C, T = vt.shape
        return vt.reshape(B, self.world_size, C // self.world_size, T).permute(1, 2, 0, 3).contiguous()

    def unpack_vt_ref(self, wire: torch.Tensor, ld: int) -> torch.Tensor:
        """arrived channel-major wire [P, Cl, B, T] -> [B, Cl, ld] with column s * T + t (pad columns zero)."""
        P, Cl, B, T = wire.shape
        out = torch.zeros(B, Cl, ld, dtype=wire.dtype, device=wire.device)
        out[:, :, :P * T].view(B, Cl, P, T).copy_(wire.permute(2, 1, 0, 3))
        return out

    # token-sharded [B, Ll, C] -> head-sharded [B, P*Ll, C/P]; `x` may be a strided view
    def scatter_heads(self, x: torch.Tensor, async_op: bool = False):
        P = self.world_size
        B, Ll, C = x.shape
        send = x.reshape(B, Ll, P, C // P).permute(2, 0, 1, 3).contiguous()
        recv = torch.empty_like(send)
        work = self._a2a(recv, send, async_op)

        def finish() -> torch.Tensor:
            if work is not None:
                work.wait()
            return recv.permute(1, 0, 2, 3).reshape(B, P * Ll, C // P)
        return finish if async_op else finish()

    # token-sharded transposed [B, C, Ll] -> head-sharded [B, C/P, ld] with columns [0, P*Ll) filled
    def scatter_heads_t(self, vt: torch.Tensor, ld: Optional[int] = None, async_op: bool = False):
        P = self.world_size
        B, C, Ll = vt.shape
        send = vt.reshape(B, P, C // P, Ll).permute(1, 0, 2, 3).contiguous()
        recv = torch.empty_like(send)
        work = self._a2a(recv, send, async_op)
        ld = ld or P * Ll

        def finish() -> torch.Tensor:
            if work is not None:
                work.wait()
            out = torch.zeros(B, C // P, ld, device=vt.device, dtype=vt.dtype)
            out[:, :, : P * Ll].view(B, C // P, P, Ll).copy_(recv.permute(1, 2, 0, 3))
            return out
        return finish if async_op else finish()

    # head-sharded [B, P*Ll, C/P] -> token-sharded [B, Ll, C]
    def gather_heads(self, o: torch.Tensor) -> torch.Tensor:
        P = self.world_size
        B, L, Cs = o.shape
        Ll = L // P
        send = o.reshape(B, P, Ll, Cs).permute(1, 0, 2, 3).contiguous()
        recv = torch.empty_like(send)
        self._a2a(recv, send, False)
        return recv.permute(1, 2, 0, 3).reshape(B, Ll, P * Cs)

    # [B, Ll, N] -> [B, P*Ll, N]
    def all_gather_tokens(self, y: torch.Tensor) -> torch.Tensor:
        """The one all-gather of a forward (the head output, wan_transformer3d.py:1085-1086), into a PERSISTENT receive buffer
        (one per shape, kept on the group): a forward allocates nothing for it.  The result is a view of that buffer (B = 1) or a
        second persistent buffer filled by one permuting copy (B > 1); it is overwritten by the next call of the same shape."""
        P = self.world_size
        y = y.contiguous()
        recv, out = _gather_buffers(self, y, P)
        if self._host_staged and y.is_cuda:
            yc = y.cpu()
            parts = [torch.empty_like(yc) for _ in range(P)]
            dist.all_gather(parts, yc, group=self.group)
            recv.copy_(torch.stack(parts))
        else:
            dist.all_gather_into_tensor(recv.view(-1), y.view(-1), group=self.group)
        return _gathered_view(recv, out)


def _gather_buffers(owner, y: torch.Tensor, P: int):
    """(receive buffer [P, B, T, N], output buffer [B, P, T, N] or None) of `owner` for inputs shaped like y -- allocated once."""
    cache = owner.__dict__.setdefault("_ag_bufs", {})
    key = (tuple(y.shape), y.dtype, str(y.device))
    if key not in cache:
        recv = torch.empty((P,) + tuple(y.shape), device=y.device, dtype=y.dtype)
        out = None if y.shape[0] == 1 else torch.empty((y.shape[0], P) + tuple(y.shape[1:]), device=y.device, dtype=y.dtype)
        cache[key] = (recv, out)
    return cache[key]


def _gathered_view(recv: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    P, B, T = recv.shape[0], recv.shape[1], recv.shape[2]
    if out is None:                                   # B == 1: [P, 1, T, N] is [1, P*T, N] as it stands
        return recv.view(1, P * T, *recv.shape[3:])
    out.copy_(recv.transpose(0, 1))
    return out.view(B, P * T, *recv.shape[3:])


class LibraryComm:
    """The same exchange interface on the communicator libwan_hip.so owns (``wan_sp_*``, include/wan_hip.h a21'): an RCCL comm, one
    side HIP stream and two events inside the library -- what a host without torch.distributed drives.  Here it is an alternative
    transport for the Python host (``init_sequence_parallel(backend="library")``): the rendezvous token travels through an existing
    torch.distributed group (any backend; gloo is enough) or is not needed at all for a single rank."""

    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None, group=None):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        lib = _lib.load()
        if (rank is None) != (world_size is None):
            raise ValueError("LibraryComm: give rank AND world_size, or neither (both are then taken from the process group)")
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0          # the rank INSIDE `group`
        uid = (ctypes.c_ubyte * 128)()
        if rank == 0:
            _lib.check(lib.wan_sp_unique_id(uid), "wan_sp_unique_id")
        if world_size > 1:
            box = [bytes(uid)]
            # broadcast_object_list addresses its source by GLOBAL rank: group rank 0 of a sub-group (an SP group inside a
            # data-parallel world) is generally not global rank 0
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            uid = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
        self._comm = ctypes.c_void_p()
        _lib.check(lib.wan_sp_init(ctypes.byref(self._comm), uid, int(rank), int(world_size)), "wan_sp_init")
        self.rank, self.world_size, self.group = int(rank), int(world_size), group
        self._host_staged = False

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def exchange(self, recv: torch.Tensor, send: torch.Tensor, async_op: bool = False):
        if send.numel() != recv.numel() or send.numel() % self.world_size or not (send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("exchange: send / recv must be contiguous, of equal size, divisible by the group size")
        lib, vp = self._lib.load(), self._ct.c_void_p
        self._lib.check(lib.wan_sp_a2a_scatter_heads(self._comm, vp(send.data_ptr()), vp(recv.data_ptr()),
                                                     send.numel() * send.element_size(), self._stream()), "wan_sp_a2a_scatter_heads")
        ticket = int(lib.wan_sp_ticket(self._comm))
        # the wait joins THIS exchange (and earlier ones) only: exchanges started after it stay in flight, as with the per-work
        # waits of the torch.distributed transport
        wait = lambda: self._lib.check(lib.wan_sp_wait_for(self._comm, ticket, self._stream()), "wan_sp_wait_for")
        if async_op:
            return wait
        wait()
        return None

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """Element-wise maximum over the ranks, in place: the library communicator has no reduction of its own, so the ranks'
        vectors are gathered (wan_sp_all_gather) and reduced locally -- a handful of floats, once per layer and model load."""
        if self.world_size > 1:
            flat = t.contiguous().view(1, 1, -1)
            t.copy_(self.all_gather_tokens(flat).view(self.world_size, -1).amax(dim=0).view(t.shape))
        return t

    def all_gather_tokens(self, y: torch.Tensor) -> torch.Tensor:
        P = self.world_size
        y = y.contiguous()
        recv, out = _gather_buffers(self, y, P)       # persistent: nothing is allocated per forward
        lib, vp = self._lib.load(), self._ct.c_void_p
        self._lib.check(lib.wan_sp_all_gather(self._comm, vp(y.data_ptr()), vp(recv.data_ptr()), y.numel() * y.element_size(),
                                              self._stream()), "wan_sp_all_gather")
        self._lib.check(lib.wan_sp_wait(self._comm, self._stream()), "wan_sp_wait")
        return _gathered_view(recv, out)

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.load().wan_sp_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EmulatedRank:
    """MEASUREMENT ONLY: the exchange interface of ONE rank of a P-way Ulysses group on one device, with no peers
    (``init_sequence_parallel(backend="emulated", rank=r, world_size=P)``; ``bench.py --emulate-sp P``).  The model then runs
    exactly what rank r of P would: the token-local Linears / norms on L/P rows, attention on H/P heads over all L keys, the wire
    layouts, pack / unpack passes and head-group pipelining -- and every exchange becomes a device-local copy of the send buffer
    into the receive buffer (the byte count a rank receives, the wrong contents: its own slabs where the peers' would arrive), the
    all-gather P copies of the local shard.  The OUTPUT IS THEREFORE NOT A DENOISED LATENT; what the mode gives is the compute side
    of one rank's step on the shard shapes, i.e. the bound a P-GPU run cannot beat and the number its exposed-communication time
    adds to.  Nothing in the product path or the tests' parity statements uses it."""

    def __init__(self, rank: int = 0, world_size: int = 1):
        if not (0 <= int(rank) < int(world_size)):
            raise ValueError(f"EmulatedRank: rank {rank} outside a group of {world_size}")
        self.rank, self.world_size, self.group = int(rank), int(world_size), None
        self._host_staged = False

    def exchange(self, recv: torch.Tensor, send: torch.Tensor, async_op: bool = False):
        if send.numel() != recv.numel() or send.numel() % self.world_size or not (send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("exchange: send / recv must be contiguous, of equal size, divisible by the group size")
        recv.view(-1).copy_(send.view(-1))
        return (lambda: None) if async_op else None

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_gather_tokens(self, y: torch.Tensor) -> torch.Tensor:
        P = self.world_size
        y = y.contiguous()
        recv, out = _gather_buffers(self, y, P)
        recv.copy_(y.unsqueeze(0).expand_as(recv))
        return _gathered_view(recv, out)


def get_sp_group() -> Optional[SequenceParallelGroup]:
    return _SP


def get_sequence_parallel_world_size() -> int:
    return _SP.world_size if _SP is not None else 1


def get_sequence_parallel_rank() -> int:
    return _SP.rank if _SP is not None else 0


def init_sequence_parallel(group=None, backend: str = "torch", rank: Optional[int] = None, world_size: Optional[int] = None):
    """Use an already initialised process group (or WORLD) as the Ulysses group.  ``backend="library"``: the collectives run on
    the communicator libwan_hip.so owns (``LibraryComm``); the torch group, if any, only carries the rendezvous token."""
    global _SP
    if backend == "library":
        _SP = LibraryComm(rank, world_size, group)
    elif backend == "torch":
        _SP = SequenceParallelGroup(group)
    elif backend == "emulated":         # measurement only: one rank's compute on the shard shapes, no peers (EmulatedRank)
        _SP = EmulatedRank(0 if rank is None else rank, 1 if world_size is None else world_size)
    else:
        raise ValueError(f"init_sequence_parallel: backend {backend!r} (\"torch\", \"library\" or \"emulated\")")
    return _SP


def destroy_sequence_parallel() -> None:
    global _SP
    if isinstance(_SP, LibraryComm):
        _SP.close()
    _SP = None


def set_multi_gpus_devices(ulysses_degree: int, ring_degree: int = 1, classifier_free_guidance_degree: int = 1):
    """Same contract as videox_fun/dist/fuser.py:35-54: returns the device for this rank and, when
    any degree > 1, boots the process group ("nccl" is RCCL on ROCm) and checks
    world_size == ring * ulysses * cfg.  Ring attention and CFG parallelism are not built
     and raise."""
    if ulysses_degree > 1 or ring_degree > 1 or classifier_free_guidance_degree > 1:
        if ring_degree > 1 or classifier_free_guidance_degree > 1:
            raise RuntimeError("only Ulysses sequence parallelism is implemented (ring_degree and "
                               "classifier_free_guidance_degree must be 1)")
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
        if dist.get_world_size() != ring_degree * ulysses_degree * classifier_free_guidance_degree:
            raise AssertionError(
                "number of GPUs(%d) should be equal to ring_degree * ulysses_degree * "
                "classifier_free_guidance_degree." % dist.get_world_size())
        init_sequence_parallel()
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", dist.get_rank() % max(torch.cuda.device_count(), 1)))
            torch.cuda.set_device(local)
            return torch.device(f"cuda:{local}")
        return torch.device("cpu")
    return torch.device("cuda")
