"""Multi-GPU use of the path: one process per GPU, ``torch.distributed`` (backend ``nccl`` == RCCL over xGMI).

The reference is a single-process CPU library; its only "parallelism" is a static split of one flat array
over pool threads (``src/piquant.cpp:132-176``).  Here the same split rule shards a tensor over ranks:

* ``quantize`` / ``dequantize`` are element-local -> every rank processes its own shard, **no collective**;
* ``compute_quant_params`` needs the global min/max -> each rank scans its shard on its GPU into two int32
  keys {key(min), key(-max)}; **one** 8-byte ``all_reduce(MIN)`` combines them; every rank then runs the same
  double-precision epilogue and obtains identical ``(scale, zero_point)``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import Context, DataType, decode_minmax_keys, quant_params_from_minmax
from .torch import _QUANT_TYPES, _ctx_for, torch_to_piquant_dtype


_P2P_MAX_TIMEOUT_S = 4294.0   # the C ABI carries microseconds in 32 bits


def _p2p_timeout_us(timeout: Optional[float]) -> int:
    """``timeout`` (seconds) of the peer-to-peer transports as the C ABI's microseconds.  None: ``PIQUANT_P2P_TIMEOUT_S`` from the environment,
    else the library's default (0 -> 10 minutes, what torch.distributed gives a collective before it calls a rank missing).  A rank may be
    minutes late for honest reasons -- a checkpoint write, an evaluation pass on rank 0, a first-call compile."""
    import os

    if timeout is None:
        env = os.environ.get('PIQUANT_P2P_TIMEOUT_S')
        if not env:
            return 0
        timeout = float(env)
    if not 0.0 < timeout <= _P2P_MAX_TIMEOUT_S:
        raise ValueError(f'timeout={timeout} s is outside (0, {_P2P_MAX_TIMEOUT_S:.0f}] s')
    return max(1, int(timeout * 1e6))


def _raise_peer_timeout(ctx: Context, what: str) -> None:
    """RuntimeError if a peer-to-peer wait of ``ctx`` ran out since the last look (the GPU queue is NOT faulted by such a wait: the kernel reports
    and the stream goes on, ``piquant_hip_peer_timeout``)."""
    t = ctx.peer_timeout()
    if t is None:
        return
    kind, rank, expected, seen = t
    # The buffers of the exchange that gave up are void from here on: the rank that was waited for may still store into them (its chunk into a receive
    # buffer of a LATER exchange's parity, its key pair into a mailbox slot nobody empties), and this rank signalled nothing behind the wait that ran out
    # (signal_flags_kernel), so its peers are about to give up on it in turn.  Every mesh of that kind on the device is poisoned until the group
    # rebuilds them (release_peer_meshes, a collective).
    for m in list((_PeerMesh if kind == 'flags' else _KeyMesh)._cache.values()):
        if m.device.index == ctx.device:
            m.poisoned = True
    if kind == 'flags':
        raise RuntimeError(f"{what}: rank {rank} did not arrive within the timeout (its flag read {seen}, exchange {expected} was waited for); the tensors of that "
                           f"all-reduce hold stale bytes.  Raise timeout= / PIQUANT_P2P_TIMEOUT_S if ranks may be that far apart.")
    raise RuntimeError(f"{what}: rank {rank} did not deliver its min/max keys within the timeout; the parameters of that call are void.  "
                       f"Raise timeout= / PIQUANT_P2P_TIMEOUT_S if ranks may be that far apart.")


class _StreamOrder:
    """The buffers and sequence numbers of a cached mesh are ONE resource: its two-parity argument ("a rank enters exchange s + 1 only behind its own
    decode of s") holds for exchanges issued in stream order.  Exchanges from different streams (a comm side stream, a DDP hook, two threads) are put
    in that order here: an event behind every exchange, waited for by the next one when it comes from another stream."""

    def __init__(self):
        import threading

        self._last_stream = None
        self._event = None
        self.lock = threading.Lock()      # the host side of an exchange (sequence number, parity, the launches) is one critical section per mesh

    def enter(self, device: torch.device) -> None:
        cur = torch.cuda.current_stream(device)
        if self._last_stream is not None and self._last_stream != cur.cuda_stream:
            cur.wait_event(self._event)

    def leave(self, device: torch.device) -> None:
        cur = torch.cuda.current_stream(device)
        if self._event is None:
            self._event = torch.cuda.Event()
        self._event.record(cur)
        self._last_stream = cur.cuda_stream


def shard_range(numel: int, rank: int, world_size: int, packed_bits: int = 8, align: int = 1) -> Tuple[int, int]:
    """[begin, end) of ``rank``'s shard: the reference's range split (``src/piquant.cpp:145-157``) -- boundaries are
    aligned down to a whole packed byte (2 elements for uint4, 4 for uint2); the last rank keeps the ragged end.
    ``align`` (1 = the reference rule only; otherwise a multiple of the pack factor, e.g. 4096) aligns interior boundaries down further, so that every shard of ONE
    shared allocation also starts on a 16-byte vector of both the float and the packed side."""
    world_size = max(1, world_size)
    pack = 8 // packed_bits if packed_bits < 8 else 1
    if align < 1 or (align != 1 and align % pack != 0):
        raise ValueError(f'align={align} must be 1 (whole packed bytes only) or a multiple of the pack factor {pack}')
    unit = max(pack, align)
    begin = numel * rank // world_size
    end = numel * (rank + 1) // world_size
    if unit > 1:
        begin -= begin % unit
        if rank + 1 != world_size:
            end -= end % unit
    return begin, max(begin, end)


def _rank_world(rank: Optional[int], world_size: Optional[int], group) -> Tuple[int, int]:
    if rank is None or world_size is None:
        if not (dist.is_available() and dist.is_initialized()):
            raise ValueError('rank= and world_size= are required when torch.distributed is not initialised')
        rank = dist.get_rank(group) if rank is None else rank
        world_size = dist.get_world_size(group) if world_size is None else world_size
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} is outside [0, {world_size})')
    return rank, world_size


def quantize_shard(
    tensor: torch.Tensor,
    *,
    scale: float,
    zero_point: int,
    dtype: torch.dtype,
    round_mode: str = 'nearest',
    out: Optional[torch.Tensor] = None,
    rank: Optional[int] = None,
    world_size: Optional[int] = None,
    group: Optional[dist.ProcessGroup] = None,
    align: int = 1,
    ctx: Optional[Context] = None,
    _quantize=None,
) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """This rank's share of ``quantize(tensor)`` for ONE logical tensor that every rank holds (or addresses): elements
    ``shard_range(numel, rank, world)`` -- the reference's pool split (``src/piquant.cpp:145-157``) with ranks in place of
    threads -- are quantized with the caller's (global) ``scale`` / ``zero_point``; no collective.

    Returns ``(packed bytes of the shard, (begin, end))``.  With ``out=`` (a contiguous quantized tensor or raw uint8 buffer
    for the WHOLE tensor) the bytes are written in place at ``begin * bits / 8`` and the returned tensor is that slice, so
    the concatenation over ranks is byte for byte what a single ``quantize`` call produces (the kernels are
    position-independent; tested)."""
    from .torch import packed_bytes, quantize

    if dtype not in _QUANT_TYPES:
        raise ValueError(f'{dtype} is not a quantized dtype')
    if not tensor.is_contiguous():
        raise ValueError('quantize_shard needs a contiguous tensor: a shard is a range of the flat element order')
    qdt = torch_to_piquant_dtype(dtype)
    rank, world_size = _rank_world(rank, world_size, group)
    flat = tensor.view(-1)
    begin, end = shard_range(flat.numel(), rank, world_size, qdt.bit_size, align)
    n_bytes = qdt.packed_nbytes(end - begin)
    first = begin * qdt.bit_size // 8
    if out is None:
        dst = torch.empty(n_bytes, dtype=torch.uint8, device=tensor.device)
    else:
        whole = out if out.dtype == torch.uint8 else packed_bytes(out)
        if whole.device != tensor.device or not whole.is_contiguous() or whole.numel() < qdt.packed_nbytes(flat.numel()):
            raise ValueError(f'out= must be a contiguous buffer on {tensor.device} holding the whole quantized tensor '
                             f'({qdt.packed_nbytes(flat.numel())} bytes)')
        dst = whole.view(-1)[first: first + n_bytes]
    if end > begin:
        (_quantize or quantize)(flat[begin:end], scale=scale, zero_point=zero_point, dtype=dtype,
                                round_mode=round_mode, ctx=ctx, out=dst, uniform=True)
    return dst, (begin, end)


def dequantize_shard(
    packed: torch.Tensor,
    *,
    numel: int,
    scale: float,
    zero_point: int,
    quant_dtype: torch.dtype,
    out: torch.Tensor,
    reduce_op: str = 'set',
    rank: Optional[int] = None,
    world_size: Optional[int] = None,
    group: Optional[dist.ProcessGroup] = None,
    align: int = 1,
    ctx: Optional[Context] = None,
    _dequantize=None,
) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """This rank's share of ``dequantize``: ``packed`` holds the WHOLE quantized tensor of ``numel`` elements (a quantized
    torch tensor or its raw uint8 bytes), ``out`` the whole float tensor; elements ``shard_range(numel, rank, world)`` of
    ``out`` are set (or accumulated into, ``reduce_op='add'``).  Returns ``(out[begin:end] view, (begin, end))``."""
    from .torch import dequantize, packed_bytes

    if quant_dtype not in _QUANT_TYPES:
        raise ValueError(f'{quant_dtype} is not a quantized dtype')
    qdt = torch_to_piquant_dtype(quant_dtype)
    raw = packed if packed.dtype == torch.uint8 else packed_bytes(packed)
    if not raw.is_contiguous() or raw.numel() < qdt.packed_nbytes(numel):
        raise ValueError(f'packed must be contiguous and hold {qdt.packed_nbytes(numel)} bytes for {numel} elements')
    if not out.is_contiguous() or out.numel() != numel or out.device != raw.device:
        raise ValueError('out= must be the contiguous float tensor of the whole logical tensor, on the same device')
    rank, world_size = _rank_world(rank, world_size, group)
    begin, end = shard_range(numel, rank, world_size, qdt.bit_size, align)
    dst = out.view(-1)[begin:end]
    if end > begin:
        first = begin * qdt.bit_size // 8
        src = raw.view(-1)[first: first + qdt.packed_nbytes(end - begin)]
        (_dequantize or dequantize)(src, scale=scale, zero_point=zero_point, dtype=out.dtype, reduce_op=reduce_op, ctx=ctx, out=dst,
                                    quant_dtype=quant_dtype, shape=(end - begin,), uniform=True)
    return dst, (begin, end)


def local_minmax_keys(tensor: torch.Tensor, ctx: Optional[Context] = None) -> torch.Tensor:
    """int32[2] tensor on ``tensor.device`` holding {key(min), key(-max)} of the local shard (HIP scan, async)."""
    if not tensor.is_cuda:
        raise RuntimeError('local_minmax_keys needs a ROCm device tensor: the min/max scan is a HIP kernel, there is no CPU path')
    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    ctx = _ctx_for(tensor, ctx)
    keys = torch.empty(2, dtype=torch.int32, device=tensor.device)
    ctx.minmax_keys_ptr(tensor.data_ptr(), torch_to_piquant_dtype(tensor.dtype), tensor.numel(), keys.data_ptr(), init=True, _device_ptrs=True)
    return keys


def compute_quant_params(
    local_shard: torch.Tensor,
    *,
    dtype: torch.dtype,
    group: Optional[dist.ProcessGroup] = None,
    ctx: Optional[Context] = None,
    transport: str = 'collective',
    timeout: Optional[float] = None,
    _scan=local_minmax_keys,
) -> Tuple[float, int]:
    """Quantization parameters of the tensor whose shards are spread over ``group`` (identical on every rank).

    ``transport='collective'``: the local keys, ONE 8-byte ``all_reduce(MIN)`` (RCCL over xGMI), the epilogue.  ``transport='p2p'`` (GPUs of
    one node): the same MIN over peer-mapped mailboxes by one one-wave kernel behind the scan -- a store to and a poll for every peer instead
    of a collective's launch and protocol (``piquant_hip_exchange_minmax_keys``); same keys, hence the same parameters.  At most 64 ranks.
    ``timeout`` (seconds, p2p only; default ``PIQUANT_P2P_TIMEOUT_S`` or 10 minutes): how long a rank waits for its peers before the call raises
    RuntimeError naming the missing rank."""
    if dtype not in _QUANT_TYPES:
        raise ValueError(f'{dtype} is not a quantized dtype')
    if transport not in ('collective', 'p2p'):
        raise ValueError(f"transport must be 'collective' or 'p2p', got {transport!r}")
    keys = _scan(local_shard, ctx)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1 and transport == 'p2p':
        if world > _KEY_MESH_MAX_RANKS:
            raise ValueError(f"transport='p2p' exchanges the keys in one wave: at most {_KEY_MESH_MAX_RANKS} ranks, the group has {world}")
        if not keys.is_cuda:
            raise RuntimeError("transport='p2p' exchanges device memory between GPUs: the shard must live on one")
        cx = _ctx_for(local_shard, ctx)
        mesh = _KeyMesh.get(group, keys.device, world, dist.get_rank(group))
        keys = mesh.exchange(keys, cx, _p2p_timeout_us(timeout))
        k = keys.cpu()                    # synchronises: the exchange is over, one way or the other
        try:
            _raise_peer_timeout(cx, "compute_quant_params(transport='p2p')")
        except RuntimeError:
            mesh.poisoned = True          # the late rank's word will land in a slot nobody empties any more
            raise
    elif world > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)   # the path's only collective: 8 bytes
    k = keys.cpu()
    r_min, r_max = decode_minmax_keys(int(k[0]), int(k[1]))
    return quant_params_from_minmax(r_min, r_max, torch_to_piquant_dtype(dtype))


def _device_identity(index: int) -> str:
    props = torch.cuda.get_device_properties(index)
    uuid = getattr(props, 'uuid', None)
    return str(uuid) if uuid is not None else f'{props.name}#{index}'


def _check_peers_reachable(group, device: torch.device, world: int, rank: int) -> None:
    """Collective.  Before anything is mapped: every rank must sit on the same host, and every GPU of the group must be able to address every other
    one (``hipDeviceCanAccessPeer``).  If a pair cannot, EVERY rank raises the same RuntimeError naming the pairs -- a refusal before the first
    IPC handle is opened instead of a fault inside a kernel.  (``PIQUANT_P2P_PRETEND_UNREACHABLE="i-j"`` declares a pair of ranks unreachable:
    the refusal can be tested on a box where everything is reachable.)  A peer's GPU that this process cannot see at all (per-rank
    HIP_VISIBLE_DEVICES) cannot be asked about; the mapping itself then decides."""
    import os
    import socket

    me = (socket.gethostname(), _device_identity(device.index))
    everyone = [None] * world
    dist.all_gather_object(everyone, me, group=group)
    hosts = {h for h, _ in everyone}
    if len(hosts) > 1:
        raise RuntimeError(f"transport='p2p' maps device memory between the GPUs of ONE node; the group spans {sorted(hosts)}")
    visible = {_device_identity(i): i for i in range(torch.cuda.device_count())}
    pretend = set()
    for pair in filter(None, os.environ.get('PIQUANT_P2P_PRETEND_UNREACHABLE', '').split(',')):
        a, b = (int(v) for v in pair.split('-'))
        pretend.add((min(a, b), max(a, b)))
    mine = []
    for j, (_, ident) in enumerate(everyone):
        if j == rank:
            continue
        if (min(rank, j), max(rank, j)) in pretend:
            mine.append((rank, j))
        elif ident != me[1] and ident in visible and not torch.cuda.can_device_access_peer(device.index, visible[ident]):
            mine.append((rank, j))
    unreachable = [None] * world
    dist.all_gather_object(unreachable, mine, group=group)
    pairs = sorted({(min(a, b), max(a, b)) for per_rank in unreachable for a, b in per_rank})
    if pairs:
        raise RuntimeError(f"transport='p2p' refused: the GPUs of ranks {pairs} cannot address each other's memory (hipDeviceCanAccessPeer); "
                           f"use transport='collective'")


class _PeerMapped:
    """One allocation per rank that every other rank of the group maps into its own address space (HIP IPC through the library's
    ``piquant_hip_peer_*`` calls; the 64-byte handles travel over the process group).  ``ptrs[j]`` is the local address of rank j's
    allocation (``ptrs[rank]`` the own one).  On one node every GPU reaches every other over its own xGMI link; several processes on one GPU
    -- the tests -- share memory the same way."""

    def __init__(self, ctx: Context, group, nbytes: int, world: int, rank: int, fine_grained: bool, fill_word: int = 0):
        self.ctx, self.rank = ctx, rank
        self.own, handle = ctx.peer_alloc(nbytes, fine_grained, fill_word)
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
        self.ptrs = [self.own if j == rank else ctx.peer_open(h) for j, h in enumerate(handles)]

    def release(self, group) -> None:
        """Collective: everybody unmaps before anybody frees."""
        dist.barrier(group=group)
        for j, p in enumerate(self.ptrs):
            if j != self.rank:
                self.ctx.peer_close(p)
        self.ptrs = []
        dist.barrier(group=group)
        self.ctx.peer_free(self.own)
        self.own = 0

    def discard(self) -> None:
        """NOT collective: the peers are gone (their process group was destroyed); unmap what was mapped, free what was allocated."""
        torch.cuda.synchronize()
        for j, p in enumerate(self.ptrs):
            if j != self.rank:
                self.ctx.peer_close(p)
        self.ptrs = []
        if self.own:
            self.ctx.peer_free(self.own)
            self.own = 0


_KEY_MESH_MAX_RANKS = 64   # include/piquant_hip.h, piquant_hip_exchange_minmax_keys: one lane per rank


class _KeyMesh:
    """Mailboxes of ``compute_quant_params(transport='p2p')``: per rank two arrays (parities) of ``world`` 8-byte words in FINE-GRAINED device
    memory (another GPU writes them while a kernel of this one polls).  Word j of rank r's mailbox is where rank j stores its key pair for r;
    a word is emptied by its reader; exchanges alternate between the parities (``include/piquant_hip.h``)."""

    _cache = {}

    def __init__(self, group, device: torch.device, world: int, rank: int):
        self.world, self.rank, self.device = world, rank, device
        self.ctx = Context.get(device.index)
        _check_peers_reachable(group, device, world, rank)
        self.mem = _PeerMapped(self.ctx, group, 16 * world, world, rank, fine_grained=True, fill_word=0x7fffffff)   # every word = two keys of NaN patterns: empty
        self.out = torch.empty(2, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        dist.barrier(group=group)
        self.seq = 0
        self.order = _StreamOrder()
        self.poisoned = False            # an exchange of this mesh gave up: a late word may sit in a mailbox slot for ever

    @classmethod
    def get(cls, group, device, world, rank):
        owner = group if group is not None else dist.group.WORLD
        key = (id(owner), device.index, world)
        m = cls._cache.get(key)
        if m is not None and m.owner is not owner:      # an id recycled by a later process group: those peers are gone
            cls._cache.pop(key).mem.discard()
            m = None
        if m is None:
            m = cls._cache[key] = cls(group, device, world, rank)
            m.owner = owner
        return m

    def exchange(self, keys: torch.Tensor, ctx: Context, timeout_us: int = 0) -> torch.Tensor:
        if self.poisoned:
            raise RuntimeError("compute_quant_params(transport='p2p'): an earlier exchange of this group gave up on a late rank, whose keys may still sit in a "
                               "mailbox; every rank must call piquant.distributed.release_peer_meshes(group) before the group uses transport='p2p' again")
        _raise_peer_timeout(ctx, "an earlier compute_quant_params(transport='p2p')")
        with self.order.lock:
            self.order.enter(self.device)
            self.seq += 1
            par = self.seq & 1
            slots = [self.mem.ptrs[j] + 8 * (par * self.world + self.rank) for j in range(self.world)]
            ctx.exchange_minmax_keys_ptr(keys.data_ptr(), slots, self.mem.own + 8 * par * self.world, self.out.data_ptr(), timeout_us)
            self.order.leave(self.device)
        return self.out

    def release(self, group) -> None:
        torch.cuda.synchronize(self.device)
        self.mem.release(group)


# -----------------------------------------------------------------------------------------------------------------
# Quantized ring all-reduce: the caller pattern the reference's SET/ADD store operators were
# designed for ("useful for ring-reduction operations", reference README.md:29), built from the path's primitives.
# -----------------------------------------------------------------------------------------------------------------
_HEADER_BYTES = 16   # wire header per hop == the device parameter record {float scale, float 1/scale, int64 zero_point}


class _DeviceOps:
    """Wire encode / decode on ROCm device tensors (HIP kernels through libpiquant.so).  The parameters are derived on
    the device straight into the buffer's header and read back from it by the receiver's dequantize kernel: a hop
    needs no host synchronisation at all.

    The calls go through the context's raw-pointer entry points (round 5): every buffer here is one this module laid out itself (16-byte header +
    packed bytes, slots on 16-byte boundaries), and the tensor-level wrappers of ``piquant.torch`` -- two slices and a dozen attribute checks per
    buffer -- cost 17 us of host time for a one-term ``reduce_encode`` and 42 us for a seven-term one, more than the kernels they launch
    (profiles/EXPERIMENTS.md): an all-reduce of 109 MB was bound by its host."""

    def __init__(self, ctx: Optional[Context]):
        self.ctx = ctx

    def _cx(self, t: torch.Tensor) -> Context:
        if not t.is_cuda:
            raise RuntimeError('the wire kernels are HIP kernels: quantized_all_reduce needs ROCm device tensors (there is no CPU path)')
        return _ctx_for(t, self.ctx)    # the tensor's device, PyTorch's current stream, stream-ordered

    @staticmethod
    def _mode(round_mode: str):
        from . import RoundMode

        return RoundMode.NEAREST if round_mode == 'nearest' else RoundMode.STOCHASTIC

    def encode(self, x: torch.Tensor, buf: torch.Tensor, qdtype: torch.dtype, round_mode: str) -> None:
        p = buf.data_ptr()
        self._cx(x).quantize_dynamic_ptr(x.data_ptr(), torch_to_piquant_dtype(x.dtype), p + _HEADER_BYTES, torch_to_piquant_dtype(qdtype), x.numel(), p,
                                         self._mode(round_mode), _device_ptrs=True)

    def decode(self, buf: torch.Tensor, out: torch.Tensor, qdtype: torch.dtype, reduce_op: str) -> None:
        from . import ReduceOp

        p = buf.data_ptr()
        self._cx(out).dequantize_dp_ptr(p + _HEADER_BYTES, torch_to_piquant_dtype(qdtype), out.data_ptr(), torch_to_piquant_dtype(out.dtype), out.numel(), p,
                                        ReduceOp.ADD if reduce_op == 'add' else ReduceOp.SET, _device_ptrs=True)

    def encode_batch(self, xs, bufs, qdtype: torch.dtype, round_mode: str) -> None:
        """encode(xs[i], bufs[i]) for all i with one kernel launch per 16 chunks (each chunk its own parameters)."""
        if xs:
            ps = [b.data_ptr() for b in bufs]
            self._cx(xs[0]).quantize_dynamic_batch_ptr([x.data_ptr() for x in xs], torch_to_piquant_dtype(xs[0].dtype), [p + _HEADER_BYTES for p in ps],
                                                       torch_to_piquant_dtype(qdtype), [x.numel() for x in xs], ps, self._mode(round_mode), _device_ptrs=True)

    def decode_batch(self, bufs, outs, qdtype: torch.dtype, reduce_op: str) -> None:
        """decode(bufs[i], outs[i]) for all i with one kernel launch per 16 chunks."""
        from . import ReduceOp

        if bufs:
            ps = [b.data_ptr() for b in bufs]
            self._cx(outs[0]).dequantize_dp_batch_ptr([p + _HEADER_BYTES for p in ps], torch_to_piquant_dtype(qdtype), [o.data_ptr() for o in outs],
                                                      torch_to_piquant_dtype(outs[0].dtype), [o.numel() for o in outs], ps,
                                                      ReduceOp.ADD if reduce_op == 'add' else ReduceOp.SET, _device_ptrs=True)

    def reduce_encode(self, bufs, acc: torch.Tensor, buf: torch.Tensor, qdtype: torch.dtype, round_mode: str) -> None:
        """encode(acc + sum of the wire buffers) into ``buf`` as one call (``acc`` is scratch afterwards)."""
        ps = [b.data_ptr() for b in bufs]
        p = buf.data_ptr()
        self._cx(acc).reduce_quantize_dynamic_ptr(acc.data_ptr(), torch_to_piquant_dtype(acc.dtype), [q + _HEADER_BYTES for q in ps], ps, p + _HEADER_BYTES,
                                                  torch_to_piquant_dtype(qdtype), acc.numel(), p, self._mode(round_mode), _device_ptrs=True)

    def decode_sum(self, bufs, out: torch.Tensor, qdtype: torch.dtype) -> None:
        """out += sum of the wire buffers, one pass over ``out`` (same result as decode(..., 'add') buffer by buffer)."""
        from . import ReduceOp

        ps = [b.data_ptr() for b in bufs]
        self._cx(out).dequantize_sum_ptr([p + _HEADER_BYTES for p in ps], ps, torch_to_piquant_dtype(qdtype), out.data_ptr(), torch_to_piquant_dtype(out.dtype),
                                         out.numel(), ReduceOp.ADD, _device_ptrs=True)


def _exchange(send: torch.Tensor, recv: torch.Tensor, nxt: int, prv: int, group) -> None:
    """Send `send` to the next rank of the ring while receiving `recv` from the previous one.  RCCL (backend nccl)
    moves device buffers directly over xGMI; other backends (gloo, used in tests) are staged through host memory."""
    if dist.get_backend(group) == 'nccl':
        ops = [dist.P2POp(dist.isend, send, nxt, group), dist.P2POp(dist.irecv, recv, prv, group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return
    s_host = send.cpu()
    r_host = torch.empty(recv.shape, dtype=recv.dtype)
    req_s = dist.isend(s_host, nxt, group=group)
    req_r = dist.irecv(r_host, prv, group=group)
    req_s.wait()
    req_r.wait()
    recv.copy_(r_host)


def _all_to_all(send: torch.Tensor, recv: torch.Tensor, group) -> None:
    """Equal-split all-to-all of byte buffers (slot j of `send` goes to rank j).  RCCL moves device buffers peer to peer -- on
    MI355X every pair of GPUs has its own xGMI link, so all G-1 transfers of a rank run at once; other backends (gloo, tests)
    are staged through host memory."""
    if dist.get_backend(group) == 'nccl':
        dist.all_to_all_single(recv, send, group=group)
        return
    r_host = torch.empty(recv.shape, dtype=recv.dtype)
    dist.all_to_all_single(r_host, send.cpu(), group=group)
    recv.copy_(r_host)


def _all_gather(mine: torch.Tensor, everyone: torch.Tensor, group) -> None:
    """`everyone` = concatenation over ranks of equally sized `mine` buffers."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == 'nccl':
        dist.all_gather_into_tensor(everyone, mine, group=group)
        return
    parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
    dist.all_gather(parts, mine.cpu(), group=group)
    everyone.copy_(torch.cat(parts))


# -----------------------------------------------------------------------------------------------------------------
# Peer-to-peer transport of the mesh schedule: the encode kernels store straight into the peers' receive buffers.
# -----------------------------------------------------------------------------------------------------------------
class _PeerMesh:
    """Buffers of ``quantized_all_reduce_direct(transport='p2p')``, per rank one payload allocation and one small flag allocation, both mapped
    by every other rank (``_PeerMapped``):

        recv[2][world][slot]   chunk j of peer i lands in recv[parity][i] of rank j -- written by PEER i's encode kernel, no copy
        mine[2][slot]          the owner's finished chunk -- READ by every peer's decode kernel, no all-gather
        arrived[world]         uint32 sequence numbers: arrived[i] = s  <=>  peer i's chunk of exchange s is in recv[s & 1][i]
        finished[world]        finished[i] = s  <=>  owner i's mine[s & 1] holds its finished chunk of exchange s

    The payload is ordinary device memory: it is produced by one launch and consumed by a LATER one on the other side (the flag wait sits
    between them), and kernel boundaries are where ordinary device memory becomes coherent between GPUs.  The flags are polled by a running
    kernel while another GPU writes them: fine-grained memory.  Two parities suffice: a rank enters exchange s + 1 only behind its own decode
    of exchange s, a peer passes its wait of exchange s + 1 only after that rank's encode of s + 1 -- so when anybody writes parity
    (s + 2) & 1 = s & 1 again, every reader of exchange s is done.
    """

    _cache = {}

    def __init__(self, group, device: torch.device, slot: int, world: int, rank: int):
        self.world, self.rank, self.slot, self.device = world, rank, slot, device
        self.ctx = Context.get(device.index)
        self.off_mine = 2 * world * slot
        payload = -(-(self.off_mine + 2 * slot) // 256) * 256
        _check_peers_reachable(group, device, world, rank)
        self.data = _PeerMapped(self.ctx, group, payload, world, rank, fine_grained=False)
        self.flags = _PeerMapped(self.ctx, group, 8 * world, world, rank, fine_grained=True)     # arrived[world] then finished[world], all 0
        self.off_arrived, self.off_finished = 0, 4 * world
        torch.cuda.synchronize(device)
        dist.barrier(group=group)        # nobody signals into memory somebody has not finished mapping
        self.seq = 0
        self.order = _StreamOrder()
        self.poisoned = False            # a wait of an exchange on this device gave up (_raise_peer_timeout): late stores may land in these buffers

    @classmethod
    def get(cls, group, device, slot, world, rank):
        """The group's mesh on this device, grown when a tensor needs larger slots than it has (every rank sees the same sizes in the same order,
        so every rank grows at the same call).  Growing is a collective behind a device synchronisation: this rank's last decode -- the last
        reader of the peers' old buffers -- has finished before the barriers inside release() let anybody unmap or free them."""
        owner = group if group is not None else dist.group.WORLD
        key = (id(owner), device.index, world)
        m = cls._cache.get(key)
        if m is not None and m.owner is not owner:      # an id recycled by a later process group: those peers are gone
            m = cls._cache.pop(key)
            m.data.discard()
            m.flags.discard()
            m = None
        if m is None or m.slot < slot:
            if m is not None:
                cls._cache.pop(key).release(group)
            m = cls._cache[key] = cls(group, device, slot, world, rank)
            m.owner = owner
        return m

    def release(self, group) -> None:
        torch.cuda.synchronize(self.device)
        self.data.release(group)
        self.flags.release(group)

    def flag_ptr(self, j: int, which: str, i: int) -> int:
        return self.flags.ptrs[j] + (self.off_arrived if which == 'arrived' else self.off_finished) + 4 * i


def release_peer_meshes(group: Optional[dist.ProcessGroup] = None) -> None:
    """Drops the peer-mapped buffers of ``transport='p2p'`` for ``group`` (a collective: every rank calls it, e.g. before
    ``destroy_process_group``).  The next p2p all-reduce builds them again."""
    owner = group if group is not None else dist.group.WORLD
    for cache in (_PeerMesh._cache, _KeyMesh._cache):
        for key in [k for k, m in cache.items() if m.owner is owner]:
            cache.pop(key).release(group)


def ring_chunks(numel: int, world_size: int, packed_bits: int = 8, align: int = 4096):
    """Chunk boundaries of the ring: `world_size` contiguous chunks; interior boundaries are multiples of `align` elements
    (whole packed bytes and 16-byte vectors on both sides of every kernel); the last chunk keeps the ragged end."""
    if align % (8 // packed_bits if packed_bits < 8 else 1) != 0:
        raise ValueError(f'align={align} must be a multiple of the pack factor')
    per = -(-numel // world_size)
    per = -(-per // align) * align
    bounds = [min(i * per, numel) for i in range(world_size)] + [numel]
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


def quantized_all_reduce(
    tensor: torch.Tensor,
    *,
    quant_dtype: torch.dtype = torch.uint8,
    round_mode: str = 'nearest',
    group: Optional[dist.ProcessGroup] = None,
    ctx: Optional[Context] = None,
    algorithm: str = 'direct',
    transport: str = 'collective',
    timeout: Optional[float] = None,
    _ops=None,
    _single_rank_collectives: bool = False,
) -> torch.Tensor:
    """In-place SUM all-reduce of a contiguous float32/bfloat16 tensor whose wire format is quantized.

    ``transport='p2p'`` (``algorithm='direct'`` only, one node; EXPERIMENTAL until it has run between two GPUs): no collective at all -- the
    encode kernels store into the peers' receive buffers over xGMI and flags order the steps (``quantized_all_reduce_direct``).  ``timeout``
    (seconds; default ``PIQUANT_P2P_TIMEOUT_S`` or 10 minutes) bounds every wait for a peer; a rank that is later than that does not fault the
    GPU: the NEXT p2p call on the device (or ``check_peer_timeouts``) raises RuntimeError naming it.  The all-reduce is asynchronous, so a caller of
    ``transport='p2p'`` MUST call ``check_peer_timeouts()`` (it synchronises) before it consumes the result -- e.g. at the step's synchronisation point,
    in front of the optimizer step: behind a wait that ran out the tensor holds sums of stale bytes.  After such a failure the rank signals nothing
    more (its peers give up on it in turn, so every rank learns of it) and the group's peer-mapped buffers stay refused until every rank has called
    ``release_peer_meshes(group)``.

    (``_single_rank_collectives`` is a test hook: with a one-rank group the function normally returns at once; with the hook it
    runs the whole schedule -- encode, the group's collectives with the rank as its own only peer, decode -- so that the RCCL
    branches execute on a box with one GPU.  The result is then ``dequantize(quantize(x))``.)

    ``algorithm='direct'`` (the default: MI355X's xGMI is a point-to-point mesh) is the schedule of
    ``quantized_all_reduce_direct`` -- one all-to-all + one all-gather, every value quantized exactly twice, 78 us of kernel
    time per rank for an 8-way all-reduce of 109 MB against 152 us for the ring.  ``algorithm='ring'`` is the schedule described
    here, for topologies where one neighbour link is all there is.

    Ring reduce-scatter: at every hop a rank receives ``header + packed bytes`` of a chunk's partial sum from its predecessor, adds
    them to its own values of that chunk and quantizes the new partial sum -- parameters from the sum itself, computed on the device
    into the 16-byte wire header -- for its successor: ONE kernel launch per hop (``reduce_quantize_dynamic``) and no host round trip.
    Ring all-gather: the owner of a
    finished chunk quantizes it once; the bytes travel round the ring unchanged and every rank (the owner included)
    stores ``dequantize(..., 'set')`` of the same bytes, so all ranks end bit-identical.  Wire traffic per element is
    1 byte (uint8) / 0.5 (uint4) instead of 4, over the same 2(G-1)/G ring schedule; each xGMI link carries one
    point-to-point stream, which is what the per-link (not NVSwitch-style) bandwidth of MI355X wants.
    """
    if not (tensor.is_contiguous() and tensor.dtype in (torch.float32, torch.bfloat16)):
        raise ValueError('quantized_all_reduce needs a contiguous float32 or bfloat16 tensor')
    if algorithm not in ('ring', 'direct'):
        raise ValueError(f"algorithm must be 'ring' or 'direct', got {algorithm!r}")
    if transport not in ('collective', 'p2p'):
        raise ValueError(f"transport must be 'collective' or 'p2p', got {transport!r}")
    if algorithm == 'direct':
        return quantized_all_reduce_direct(tensor, quant_dtype=quant_dtype, round_mode=round_mode, group=group, ctx=ctx, transport=transport, timeout=timeout,
                                           _ops=_ops, _single_rank_collectives=_single_rank_collectives)
    if transport != 'collective':
        raise ValueError("transport='p2p' is the mesh schedule's (algorithm='direct'); the ring forwards through its neighbours")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not _single_rank_collectives:
        return tensor
    ops = _ops or _DeviceOps(ctx)
    qdt = torch_to_piquant_dtype(quant_dtype)
    flat = tensor.view(-1)
    chunks = ring_chunks(flat.numel(), world, qdt.bit_size)
    nxt = dist.get_global_rank(group, (rank + 1) % world) if group is not None else (rank + 1) % world
    prv = dist.get_global_rank(group, (rank - 1) % world) if group is not None else (rank - 1) % world
    max_bytes = max(qdt.packed_nbytes(e - b) for b, e in chunks) + _HEADER_BYTES
    send = torch.empty(max_bytes, dtype=torch.uint8, device=tensor.device)
    recv = torch.empty(max_bytes, dtype=torch.uint8, device=tensor.device)

    def wire(idx):
        b, e = chunks[idx]
        return flat[b:e], _HEADER_BYTES + qdt.packed_nbytes(e - b)

    # ---- reduce-scatter: after G-1 hops rank r owns the complete sum of chunk (r+1) % G ----
    # What arrives at a hop is added to the local chunk and the sum is what the next hop forwards: that is ONE call (and one
    # launch), reduce_encode = quantize(local + dequantize(received)); the local chunk itself need not be updated, the forwarded
    # buffer carries the sum and every chunk is overwritten by the all-gather at the end.
    nxt_send = torch.empty(max_bytes, dtype=torch.uint8, device=tensor.device)
    x_first, n_first = wire(rank)
    if x_first.numel():
        ops.encode(x_first, send[:n_first], quant_dtype, round_mode)
    n_cur = n_first
    for step in range(world - 1):
        x_recv, n_recv = wire((rank - step - 1) % world)
        _exchange(send[:n_cur], recv[:n_recv], nxt, prv, group)
        if x_recv.numel():
            ops.reduce_encode([recv[:n_recv]], x_recv, nxt_send[:n_recv], quant_dtype, round_mode)
        send, nxt_send = nxt_send, send
        n_cur = n_recv
    if world == 1 and n_cur:   # test hook only: the encoded chunk makes one trip through the transport, to this rank itself
        _exchange(send[:n_cur], recv[:n_cur], nxt, prv, group)
        send, recv = recv, send

    # ---- all-gather: the finished chunk's bytes (now in `send`) circulate unchanged ----
    x_own, n_own = wire((rank + 1) % world)
    assert n_cur == n_own
    if x_own.numel():
        ops.decode(send[:n_own], x_own, quant_dtype, 'set')   # the owner keeps exactly what everyone else will see
    for step in range(world - 1):
        x_recv, n_recv = wire((rank - step) % world)
        _exchange(send[:n_cur], recv[:n_recv], nxt, prv, group)
        if x_recv.numel():
            ops.decode(recv[:n_recv], x_recv, quant_dtype, 'set')
        send, recv = recv, send          # forward the received bytes as they are
        n_cur = n_recv
    return tensor


def quantized_all_reduce_direct(
    tensor: torch.Tensor,
    *,
    quant_dtype: torch.dtype = torch.uint8,
    round_mode: str = 'nearest',
    group: Optional[dist.ProcessGroup] = None,
    ctx: Optional[Context] = None,
    transport: str = 'collective',
    timeout: Optional[float] = None,
    _ops=None,
    _single_rank_collectives: bool = False,
) -> torch.Tensor:
    """In-place quantized SUM all-reduce for a point-to-point mesh (MI355X: every GPU has its own xGMI link to each of its 7 peers).

    A ring keeps one link per direction busy and re-quantizes a partial sum at each of its G-1 hops.  Here rank r owns chunk r:

    1. every rank quantizes chunk j of its tensor for every peer j (parameters from that chunk, 16-byte header + packed bytes;
       all G-1 chunks in ONE kernel launch, ``quantize_dynamic_batch``),
    2. ONE all-to-all delivers them -- G-1 transfers per rank, each on its own link, all at once,
    3. the owner adds the G-1 received chunks to its own (unquantized) values and
    4. quantizes the finished chunk -- both in ONE launch that keeps the sum on chip (``reduce_quantize_dynamic``); ONE all-gather
       distributes it, and every rank (the owner included) stores
       ``dequantize(..., 'set')`` of the same bytes (all G chunks in one launch), so all ranks end bit-identical.

    Every value is quantized exactly twice whatever the world size (a ring: up to G times), the wire carries the same
    2(G-1)/G x packed bytes per element, and the two collectives are what RCCL implements natively over the mesh.

    ``transport='p2p'``: the same four steps and the same bytes with NO collective.  Step 1's kernel stores chunk j straight into rank j's
    receive buffer (an address of rank j's memory mapped here once: HIP IPC, ``_PeerMesh``) and a flag store behind it says so; step 3 waits for its
    G-1 flags on the stream; step 4 leaves the finished chunk in the owner's own buffer, and every rank's decode kernel READS the G finished
    chunks from their owners.  Against the collective transport that is one HBM write and one read of the wire bytes less per phase on every
    rank, no staging copy inside RCCL and no RCCL launch: 4 kernel launches + 2 flag stores + 2 flag waits per all-reduce.  One node only
    (IPC-mapped device memory); results are bit-identical to the collective transport (tests/test_gpu_distributed.py).
    """
    if not (tensor.is_contiguous() and tensor.dtype in (torch.float32, torch.bfloat16)):
        raise ValueError('quantized_all_reduce needs a contiguous float32 or bfloat16 tensor')
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not _single_rank_collectives:
        return tensor
    ops = _ops or _DeviceOps(ctx)
    qdt = torch_to_piquant_dtype(quant_dtype)
    flat = tensor.view(-1)
    chunks = ring_chunks(flat.numel(), world, qdt.bit_size)
    slot = _HEADER_BYTES + max(qdt.packed_nbytes(e - b) for b, e in chunks)
    slot = -(-slot // 16) * 16                       # every slot starts on a 16-byte boundary (vector kernels on both sides)
    if transport == 'p2p':
        if world == 1:
            raise ValueError("transport='p2p' needs peers (a one-rank group has none)")
        return _all_reduce_direct_p2p(tensor, flat, chunks, slot, quant_dtype, qdt, round_mode, group, ctx, ops, world, rank, _p2p_timeout_us(timeout))
    if transport != 'collective':
        raise ValueError(f"transport must be 'collective' or 'p2p', got {transport!r}")
    send = torch.zeros(world * slot, dtype=torch.uint8, device=tensor.device)
    recv = torch.empty(world * slot, dtype=torch.uint8, device=tensor.device)

    def wire_len(idx):
        b, e = chunks[idx]
        return _HEADER_BYTES + qdt.packed_nbytes(e - b)

    # ---- reduce-scatter over the mesh ----
    peers = [j for j in range(world) if j != rank and chunks[j][1] > chunks[j][0]]
    ops.encode_batch([flat[chunks[j][0]:chunks[j][1]] for j in peers], [send[j * slot: j * slot + wire_len(j)] for j in peers], quant_dtype, round_mode)
    _all_to_all(send, recv, group)
    b_own, e_own = chunks[rank]
    x_own = flat[b_own:e_own]
    n_own = wire_len(rank)
    # ---- the owner's sum, quantized for the all-gather in the same launch (the sum itself is never stored: the final value of the
    # own chunk, too, comes from decoding the gathered bytes) ----
    mine = torch.zeros(slot, dtype=torch.uint8, device=tensor.device)
    if x_own.numel():
        ops.reduce_encode([recv[i * slot: i * slot + n_own] for i in range(world) if i != rank], x_own, mine[:n_own], quant_dtype, round_mode)
    gathered = torch.empty(world * slot, dtype=torch.uint8, device=tensor.device)   # not `recv`: the launch above is still reading it
    _all_gather(mine, gathered, group)
    recv = gathered
    full = [j for j in range(world) if chunks[j][1] > chunks[j][0]]
    ops.decode_batch([recv[j * slot: j * slot + wire_len(j)] for j in full], [flat[chunks[j][0]:chunks[j][1]] for j in full], quant_dtype, 'set')
    return tensor


def _all_reduce_direct_p2p(tensor, flat, chunks, slot, quant_dtype, qdt, round_mode, group, ctx, ops, world, rank, timeout_us=0):
    """The mesh schedule over peer-mapped buffers (``quantized_all_reduce_direct``, ``transport='p2p'``).  The calls go through the context's
    raw-pointer entry points: the peers' buffers are addresses of THEIR devices' memory, which the tensor-level wrappers (one device per
    call, by design) would refuse."""
    if not tensor.is_cuda:
        raise RuntimeError("transport='p2p' moves device memory between GPUs: the tensor must live on one")
    slot = -(-slot // 65536) * 65536    # coarse sizes: a mesh is grown (an IPC exchange and a barrier) only when a tensor needs more than any before it
    mesh = _PeerMesh.get(group, tensor.device, slot, world, rank)      # its own slot size lays the buffers out (>= what this tensor needs)
    cx = _ctx_for(tensor, ctx)          # the tensor's device, PyTorch's current stream, stream-ordered
    _raise_peer_timeout(cx, "an earlier quantized_all_reduce(transport='p2p')")
    if mesh.poisoned:
        raise RuntimeError("quantized_all_reduce(transport='p2p'): an earlier exchange of this group gave up on a late rank, whose stores may still land in the "
                           "peer-mapped buffers; every rank must call piquant.distributed.release_peer_meshes(group) before the group uses transport='p2p' again")
    with mesh.order.lock:               # one exchange of a mesh at a time on the host, whatever thread it comes from ...
        mesh.order.enter(tensor.device)     # ... and behind the mesh's previous exchange on the device, whatever stream that ran on
        _all_reduce_direct_p2p_locked(tensor, flat, chunks, mesh, cx, qdt, round_mode, world, rank, timeout_us)
        mesh.order.leave(tensor.device)
    return tensor


def _all_reduce_direct_p2p_locked(tensor, flat, chunks, mesh, cx, qdt, round_mode, world, rank, timeout_us):
    from . import ReduceOp, RoundMode

    slot = mesh.slot
    fdt = torch_to_piquant_dtype(tensor.dtype)
    rmode = RoundMode.NEAREST if round_mode == 'nearest' else RoundMode.STOCHASTIC
    mesh.seq += 1
    seq, par = mesh.seq, mesh.seq & 1
    esize = tensor.element_size()
    base = flat.data_ptr()

    def chunk_ptr(j):
        return base + chunks[j][0] * esize

    def chunk_len(j):
        return chunks[j][1] - chunks[j][0]

    def recv_ptr(j, i):                 # recv[par][i] of rank j: 16-byte header (the parameter record), then the packed bytes
        return mesh.data.ptrs[j] + (par * world + i) * slot

    def mine_ptr(j):
        return mesh.data.ptrs[j] + mesh.off_mine + par * slot

    def wait_for_everybody(offset):     # ONE wait over all `world` flags (the own one is signalled with the peers'): flag index == rank, which is what a timeout reports
        cx.wait_flags_ptr(mesh.flags.own + offset, world, seq, timeout_us)

    peers = [j for j in range(world) if j != rank]
    everybody = list(range(world))
    # ---- 1. every peer's chunk, quantized straight into that peer's recv[par][rank]; then the flags ----
    full = [j for j in peers if chunk_len(j) > 0]
    cx.quantize_dynamic_batch_ptr([chunk_ptr(j) for j in full], fdt, [recv_ptr(j, rank) + _HEADER_BYTES for j in full], qdt, [chunk_len(j) for j in full],
                                  [recv_ptr(j, rank) for j in full], rmode, _device_ptrs=True)
    cx.signal_flags_ptr([mesh.flag_ptr(j, 'arrived', rank) for j in everybody], seq)
    # ---- 2./3. wait for the G-1 chunks of MY range, add them to my own values, quantize the finished chunk into mine[par] ----
    wait_for_everybody(mesh.off_arrived)
    if chunk_len(rank) > 0:
        cx.reduce_quantize_dynamic_ptr(chunk_ptr(rank), fdt, [recv_ptr(rank, i) + _HEADER_BYTES for i in peers], [recv_ptr(rank, i) for i in peers],
                                       mine_ptr(rank) + _HEADER_BYTES, qdt, chunk_len(rank), mine_ptr(rank), rmode, _device_ptrs=True)
    cx.signal_flags_ptr([mesh.flag_ptr(j, 'finished', rank) for j in everybody], seq)
    # ---- 4. every finished chunk, read from its owner (the own one from this rank's buffer: all ranks decode the same bytes) ----
    wait_for_everybody(mesh.off_finished)
    everyone = [j for j in range(world) if chunk_len(j) > 0]
    cx.dequantize_dp_batch_ptr([mine_ptr(j) + _HEADER_BYTES for j in everyone], qdt, [chunk_ptr(j) for j in everyone], fdt, [chunk_len(j) for j in everyone],
                               [mine_ptr(j) for j in everyone], ReduceOp.SET, _device_ptrs=True)


def check_peer_timeouts(device: Optional[torch.device] = None, ctx: Optional[Context] = None) -> None:
    """Synchronises the device and raises RuntimeError if a wait of a ``transport='p2p'`` call issued on it ran out (the asynchronous all-reduce
    cannot raise by itself: a late rank is found at the next p2p call, or here)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    torch.cuda.synchronize(device)
    _raise_peer_timeout(ctx if ctx is not None else Context.get(device.index), "quantized_all_reduce / compute_quant_params (transport='p2p')")
