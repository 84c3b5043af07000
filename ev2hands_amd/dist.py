"""Multi-GPU: one process per GPU, contiguous batch shards, ONE all-gather of the packed predictions.

Event windows are independent in eval mode (BatchNorm uses running statistics, no cross-sample op in
/root/reference/src/Ev2Hands/model/TEHNet.py:168-197), so the path shards with no data-path collective;
the only exchange is gathering the per-window outputs.  torch.distributed's
"nccl" backend is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
The reference's analogue is nn.DataParallel's scatter/gather (train.py:68), training only.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import synth

PER_HAND = synth.N_MANO_OUT + synth.MANO_NV * 3 + 21 * 3      # 22 + 2334 + 63


def packed_width(N: int, n_pose: int = synth.MANO_CMPS) -> int:
    return 4 * N + 2 * (PER_HAND + n_pose - synth.MANO_CMPS)


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_fps_inits(inits, lo: int, hi: int):
    """The four FPS start vectors are drawn once for the GLOBAL batch (reference RNG order) and sliced,
    so that sharded == unsharded bit for bit."""
    return [t[lo:hi].contiguous() for t in inits]


def pack_outputs(out: dict) -> torch.Tensor:
    """{'class_logits','left','right'} -> [B, 4N + 2*(22+2334+63)] float32 (one row per window)."""
    B = out["class_logits"].shape[0]
    logits = out["class_logits"]
    parts = [logits.reshape(B, logits.shape[1] * logits.shape[2])]       # explicit widths: a shard may hold no window
    for side in ("left", "right"):
        d = out[side]
        parts += [d["global_orient"], d["hand_pose"], d["betas"], d["transl"], d["vertices"].reshape(B, synth.MANO_NV * 3),
                  d["j3d"].reshape(B, 63)]
    return torch.cat(parts, 1).contiguous()


def unpack_outputs(buf: torch.Tensor, N: int, n_pose: int = synth.MANO_CMPS) -> dict:
    B = buf.shape[0]
    out = {"class_logits": buf[:, :4 * N].reshape(B, 4, N)}
    o = 4 * N
    for side in ("left", "right"):
        d = {}
        for k, w in (("global_orient", 3), ("hand_pose", n_pose), ("betas", 10), ("transl", 3),
                     ("vertices", synth.MANO_NV * 3), ("j3d", 63)):
            d[k] = buf[:, o:o + w]
            o += w
        d["vertices"] = d["vertices"].reshape(B, synth.MANO_NV, 3)
        d["j3d"] = d["j3d"].reshape(B, 21, 3)
        out[side] = d
    return out


def all_gather_outputs(out: dict, N: int, group=None, global_batch: int | None = None) -> dict:
    """All ranks end up with the predictions of the whole global batch, in shard order.

    `global_batch` = the batch that shard_range() split over the ranks.  Equal shards take ONE
    all_gather_into_tensor; unequal shards (global_batch % world != 0) are padded to the largest shard for the
    collective and trimmed afterwards.  Without `global_batch` the shard sizes are exchanged first (one small
    collective) so that mismatched counts can never reach RCCL, where they hang instead of raising."""
    local = pack_outputs(out)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if global_batch is None:
        sizes_t = torch.zeros(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(sizes_t, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
        sizes = [int(v) for v in sizes_t.tolist()]
    else:
        sizes = [hi - lo for lo, hi in (shard_range(global_batch, r, world) for r in range(world))]
    if sizes[rank] != local.shape[0]:
        raise ValueError(f"rank {rank}: local batch {local.shape[0]} is not this rank's shard of {sizes} windows")
    big = max(sizes)
    n_pose = out["left"]["hand_pose"].shape[1]
    if min(sizes) == big:
        full = torch.empty(world * big, local.shape[1], device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(full, local, group=group)
        return unpack_outputs(full, N, n_pose)
    padded = local if local.shape[0] == big else torch.cat([local, local.new_zeros(big - local.shape[0], local.shape[1])], 0)
    full = torch.empty(world * big, local.shape[1], device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(full, padded.contiguous(), group=group)
    rows = torch.cat([full[r * big:r * big + sizes[r]] for r in range(world)], 0)
    return unpack_outputs(rows, N, n_pose)


def _group_moves_device_memory(group) -> bool:
    """Can this process group run a collective on CUDA tensors?  Asked of the group itself, not of its name: a group made by
    init_process_group() without a backend ("cpu:gloo,cuda:nccl") reports another string than "nccl" and still moves device memory."""
    try:
        be = group._get_backend(torch.device("cuda")) if group is not None else dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda"))
        name = type(be).__name__.lower()
        return "nccl" in name or "rccl" in name
    except Exception:  # noqa: BLE001 -- no backend registered for cuda devices
        pass
    try:
        return "nccl" in str(dist.get_backend_config(group)).lower()
    except Exception:  # noqa: BLE001
        return str(dist.get_backend(group)).lower() == "nccl"


_warned_host_staged = False


def _warn_host_staged(backend) -> None:
    global _warned_host_staged
    if not _warned_host_staged:
        _warned_host_staged = True
        import warnings
        warnings.warn(f"ev2hands_amd.dist: the process group (backend {backend!r}) cannot move device memory -- the gather is staged through "
                      "pinned host memory with a host synchronisation per step (the path of the CPU tests and of two ranks sharing one GPU). "
                      "A multi-GPU run wants backend 'nccl' (RCCL).", RuntimeWarning, stacklevel=3)


class GatherBuffer:
    """This rank's persistent all-gather buffer: float32 [world * big, packed_width(N)], big = the largest shard.

    `rows()` is the slice the local forward writes straight into (TEHNet.forward(..., rows=buf.rows()): ev2h_outputs' window
    strides), `gather()` runs ONE in-place all_gather_into_tensor (the send buffer is this rank's slice of the receive buffer,
    which RCCL recognises as its in-place form) and returns the global predictions as views of the buffer -- no packing copy, no
    concatenation, no per-step allocation.  Unequal shards: a short shard leaves the tail of its slice unused and `gather()`
    trims it (the only case that copies)."""

    def __init__(self, N: int, global_batch: int, device, group=None, n_pose: int = synth.MANO_CMPS):
        self.group, self.N, self.n_pose = group, N, n_pose
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # Transport of the one collective: RCCL ("nccl") moves device memory itself.  A process group without a device transport
        # (gloo: the CPU tests, and tests/test_gpu_dist.py's two ranks that SHARE one GPU, which RCCL refuses) is served through a
        # pinned host mirror of the buffer: D2H of this rank's slice, the same in-place all_gather_into_tensor on the mirror, H2D
        # of the whole.  Only bytes move; the predictions are the forward's, bit for bit.
        self.host_staged = torch.device(device).type == "cuda" and not _group_moves_device_memory(group)
        if self.host_staged:
            _warn_host_staged(dist.get_backend(group))
        if global_batch < self.world:
            # a rank without a window would skip its forward (ev2h_forward needs B > 0) while the others wait in the collective:
            # refused here, on EVERY rank (the sizes are a pure function of global_batch and the world size)
            raise ValueError(f"global batch {global_batch} < world size {self.world}: every rank needs at least one window")
        self.generation = 0          # bumped whenever rows() hands the buffer to a new forward (GatherPipeline.Pending checks it)
        self.sizes = [hi - lo for lo, hi in (shard_range(global_batch, r, self.world) for r in range(self.world))]
        self.big = max(self.sizes)
        self.full = torch.zeros(self.world * self.big, packed_width(N, n_pose), dtype=torch.float32, device=device)
        self.host = torch.zeros(self.full.shape, dtype=torch.float32).pin_memory() if self.host_staged else None

    def _gather(self, async_op: bool):
        """the collective on whichever copy of the buffer the process group can move; returns (work or None)"""
        lo = self.rank * self.big
        if not self.host_staged:
            return dist.all_gather_into_tensor(self.full, self.full[lo:lo + self.big], group=self.group, async_op=async_op)
        self.host[lo:lo + self.big].copy_(self.full[lo:lo + self.big], non_blocking=True)
        torch.cuda.current_stream(self.full.device).synchronize()          # the forward's rows have reached the mirror
        return dist.all_gather_into_tensor(self.host, self.host[lo:lo + self.big], group=self.group, async_op=async_op)

    def _landed(self):
        """host-staged transport: the gathered mirror back onto the device (in stream order)"""
        if self.host_staged:
            self.full.copy_(self.host, non_blocking=True)

    def rows(self) -> torch.Tensor:
        lo = self.rank * self.big
        self.generation += 1
        return self.full[lo:lo + self.sizes[self.rank]]

    def gather(self) -> dict:
        self._gather(async_op=False)
        self._landed()
        if min(self.sizes) == self.big:
            return unpack_outputs(self.full, self.N, self.n_pose)
        return unpack_outputs(torch.cat([self.full[r * self.big:r * self.big + self.sizes[r]] for r in range(self.world)], 0), self.N, self.n_pose)


class GatherPipeline:
    """Ping-pong of `depth` GatherBuffers for back-to-back steps: the all-gather of step i is issued with async_op=True and runs on
    RCCL's stream while the forward of step i+1 already executes on the caller's -- the xGMI transfer (7 x 13.4 MB arriving per
    rank at world 8; ring collectives are per-link bound) is hidden under compute instead of being added to every step; the
    forward of step i+depth waits for gather i before it overwrites that buffer.

        pipe = GatherPipeline(N, global_B, device)
        for batch in batches:
            net.net(batch, net.hands, rows=pipe.rows())     # waits only for the gather that last used this buffer
            pending = pipe.submit()                         # all-gather in flight
            ...
            outputs = pending.result()                      # the caller's stream now waits for the collective; views of the buffer
    A result (views of its buffer) stays valid until that buffer comes round again, `depth` steps later; results nobody asks for
    cost nothing."""

    class Pending:
        def __init__(self, buf, work):
            self.buf, self.work, self._out = buf, work, None
            self.generation = buf.generation                # the forward whose predictions this gather carries

        def wait(self):
            if self.work is not None:
                self.work.wait()                            # device-side: the current stream waits for the collective
                self.work = None
                self.buf._landed()

        def result(self) -> dict:
            if self.buf.generation != self.generation:
                raise RuntimeError("this gather's buffer was handed to a later forward (GatherPipeline.rows()): its predictions are "
                                   "overwritten -- ask for result() before the buffer comes round again, or raise `depth`")
            self.wait()
            if self._out is None:
                b = self.buf
                if min(b.sizes) == b.big:
                    self._out = unpack_outputs(b.full, b.N, b.n_pose)
                else:
                    self._out = unpack_outputs(torch.cat([b.full[r * b.big:r * b.big + b.sizes[r]] for r in range(b.world)], 0), b.N, b.n_pose)
            return self._out

    def __init__(self, N: int, global_batch: int, device, group=None, depth: int = 2, n_pose: int = synth.MANO_CMPS, inflight: int = 0, net=None):
        """inflight = K > 1 (with net = the TEHNetWrapper): `forward(xyz)` is then ONE call per step -- forward i runs on slot stream
        i mod K with its own workspace (ev2hands_amd/inflight.py: a rank's share can be too small to fill its GPU; 16 windows of 8192
        points: +35 % with two in flight, also inside the RCCL process -- the slot streams and their side streams are bound to hardware
        queues by measurement, _lib.concurrent_streams / ev2h_bind_stream), writes into gather buffer i mod depth and the asynchronous all-gather is issued from
        the SLOT'S stream right behind it, so that the caller's stream never waits for a forward.  The reference's analogue is the one
        `net(lnes)` call under nn.DataParallel (train.py:68,83).  depth is raised to K (two forwards in flight never share a buffer)."""
        if depth < 1:
            raise ValueError("depth >= 1")
        self.inflight = None
        if inflight and inflight > 1:
            if net is None:
                raise ValueError("GatherPipeline(inflight=K) needs net= (the TEHNetWrapper whose forwards it issues)")
            from .inflight import InflightForward
            self.inflight = InflightForward(net, inflight)
            depth = max(depth, inflight)
        self.net = net
        self.bufs = [GatherBuffer(N, global_batch, device, group, n_pose) for _ in range(depth)]
        self.pending = [None] * depth
        self.i = 0

    def forward(self, xyz: torch.Tensor, fps_init=None, post=None) -> "GatherPipeline.Pending":
        """One step: this rank's batch through the network into the next gather buffer, then the all-gather in flight.  Returns the
        Pending of that gather (result() = the GLOBAL predictions).  fps_init: this rank's slice of the globally drawn FPS start
        vectors (shard_fps_inits); post: optional callable(out) queued right behind the forward (before the gather)."""
        if self.net is None:
            raise ValueError("GatherPipeline.forward needs net= at construction")
        if fps_init is not None:
            self.net.net.fps_init = fps_init
        if self.inflight is None:
            with torch.no_grad():
                out = self.net.net(xyz, self.net.hands, rows=self.rows())
            if post is not None:
                post(out)
            return self.submit()
        box = []

        def after(out):
            if post is not None:
                post(out)
            box.append(self.submit())                       # issued on the slot's stream: RCCL orders the collective behind the forward

        # rows(): on the SLOT'S stream -- it waits for the gather that last read this buffer, the caller's stream waits for nothing
        self.inflight.submit(xyz, pre=self.rows, post=after)
        return box[0]

    def rows(self) -> torch.Tensor:
        p = self.pending[self.i]
        if p is not None:
            p.wait()                                        # the gather that read this buffer must be done before it is overwritten
        return self.bufs[self.i].rows()

    def submit(self) -> "GatherPipeline.Pending":
        b = self.bufs[self.i]
        work = b._gather(async_op=True)
        p = self.pending[self.i] = GatherPipeline.Pending(b, work)
        self.i = (self.i + 1) % len(self.bufs)
        return p

    def drain(self) -> None:
        if self.inflight is not None:
            self.inflight.drain()                           # the caller's stream waits for every slot's forwards (and gathers' issue)
        for p in self.pending:
            if p is not None:
                p.wait()
