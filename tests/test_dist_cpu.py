"""CPU, world_size 2, gloo: batch sharding + the one all-gather of packed predictions."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ev2hands_amd import dist as evdist, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_outputs(lo, hi, N):
    """Deterministic per-window 'predictions' standing in for the GPU forward (value = f(window index))."""
    B = hi - lo
    w = torch.arange(lo, hi, dtype=torch.float32).view(B, 1)
    out = {"class_logits": (w.view(B, 1, 1) + torch.arange(4 * N, dtype=torch.float32).view(1, 4, N) * 1e-3)}
    for s, side in enumerate(("left", "right")):
        out[side] = {"global_orient": w + torch.arange(3) + s, "hand_pose": w + torch.arange(6) * 2 + s,
                     "betas": w + torch.arange(10) * 3 + s, "transl": w + torch.arange(3) * 4 + s,
                     "vertices": (w.view(B, 1, 1) + torch.arange(778 * 3, dtype=torch.float32).view(1, 778, 3) * 1e-2 + s),
                     "j3d": (w.view(B, 1, 1) + torch.arange(63, dtype=torch.float32).view(1, 21, 3) + s)}
    return out


def _worker(rank, world, port, gB, N, q, pass_gb=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = evdist.shard_range(gB, rank, world)
    inits = synth.fps_inits(gB, N, 5)
    mine = evdist.shard_fps_inits(inits, lo, hi)
    assert all(m.shape[0] == hi - lo for m in mine) and torch.equal(mine[2], inits[2][lo:hi])
    full = evdist.all_gather_outputs(_fake_outputs(lo, hi, N), N, global_batch=gB if pass_gb else None)
    want = _fake_outputs(0, gB, N)
    ok = torch.equal(full["class_logits"], want["class_logits"])
    for side in ("left", "right"):
        for k in want[side]:
            ok = ok and torch.equal(full[side][k], want[side][k]) and full[side][k].shape == want[side][k].shape
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_inplace(rank, world, port, gB, N, q, steps=2):
    """GatherBuffer: the 'forward' writes its windows straight into this rank's slice of the gather buffer (as ev2h_forward does
    through ev2h_outputs' window strides), then ONE in-place all-gather; repeated, because the buffer is persistent."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = evdist.shard_range(gB, rank, world)
    if gB < world:                                         # a rank without a window would leave the others in the collective:
        try:                                               # refused on EVERY rank, before any collective
            evdist.GatherBuffer(N, gB, "cpu")
            q.put((rank, False))
        except ValueError as e:
            q.put((rank, "every rank needs at least one window" in str(e)))
        dist.destroy_process_group()
        return
    buf = evdist.GatherBuffer(N, gB, "cpu")
    ok = buf.rows().shape == (hi - lo, evdist.packed_width(N)) and (hi == lo or buf.rows().data_ptr() == buf.full[rank * buf.big:].data_ptr())
    for step in range(steps):
        buf.rows().copy_(evdist.pack_outputs(_fake_outputs(lo + 100 * step, hi + 100 * step, N)))
        full = buf.gather()
        want = _fake_outputs(100 * step, gB + 100 * step, N)
        ok = ok and torch.equal(full["class_logits"], want["class_logits"])
        for side in ("left", "right"):
            for k in want[side]:
                ok = ok and torch.equal(full[side][k], want[side][k]) and full[side][k].shape == want[side][k].shape
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _worker_pipeline(rank, world, port, gB, N, q, steps=5):
    """GatherPipeline: the gather of step i is in flight while step i+1's 'forward' already writes the other buffer; every step's
    result must still be that step's, and a buffer is only overwritten after its gather completed."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = evdist.shard_range(gB, rank, world)
    pipe = evdist.GatherPipeline(N, gB, "cpu", depth=2)
    ok, handles = True, []
    for step in range(steps):
        pipe.rows().copy_(evdist.pack_outputs(_fake_outputs(lo + 100 * step, hi + 100 * step, N)))
        handles.append((step, pipe.submit()))
        if len(handles) == 2:                              # consume one step late: the next forward has already been "enqueued"
            st, h = handles.pop(0)
            full, want = h.result(), _fake_outputs(100 * st, gB + 100 * st, N)
            ok = ok and torch.equal(full["class_logits"], want["class_logits"]) and torch.equal(full["right"]["vertices"], want["right"]["vertices"])
    pipe.drain()
    for st, h in handles:
        full, want = h.result(), _fake_outputs(100 * st, gB + 100 * st, N)
        ok = ok and torch.equal(full["class_logits"], want["class_logits"]) and torch.equal(full["left"]["j3d"], want["left"]["j3d"])
    # a result asked for AFTER its buffer was handed to a later forward is refused instead of returning overwritten rows
    stale = pipe.submit()
    pipe.rows(); pipe.submit(); pipe.rows()                # depth 2: the second rows() hands stale's buffer out again
    try:
        stale.result()
        ok = False
    except RuntimeError as e:
        ok = ok and "handed to a later forward" in str(e)
    pipe.drain()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("gB", [6, 7])
def test_gather_pipeline_world2(gB):
    world, N = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, world, port, gB, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("gB", [6, 7, 1])
def test_in_place_gather_buffer_world2(gB):
    world, N = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_inplace, args=(r, world, port, gB, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("gB,pass_gb", [(6, True), (7, True), (7, False), (1, True)])
def test_all_gather_of_sharded_predictions_world2(gB, pass_gb):
    """gB = 7 and 1: shards of unequal size (one rank may even own no window) are padded for the collective and trimmed."""
    world, N = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gB, N, q, pass_gb)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_ranges_cover_batch():
    for gB in (1, 7, 256, 2048):
        for world in (1, 2, 4, 8):
            r = [evdist.shard_range(gB, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == gB
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))


def test_pack_unpack_roundtrip():
    N = 128
    out = _fake_outputs(3, 8, N)
    back = evdist.unpack_outputs(evdist.pack_outputs(out), N)
    assert torch.equal(back["class_logits"], out["class_logits"])
    assert torch.equal(back["right"]["vertices"], out["right"]["vertices"])
    assert evdist.pack_outputs(out).shape[1] == evdist.packed_width(N)


class _FakeNet:
    """stands in for TEHNetWrapper on CPU: `.net(xyz, hands, rows=)` writes one packed row per window (value = f(window id)), as
    ev2h_forward does through ev2h_outputs' window strides; records the fps_init it was handed"""

    def __init__(self, N):
        self.N, self.hands, self.net, self.seen_inits = N, None, self, []
        self.fps_init = None

    def __call__(self, xyz, hands, rows=None, ws=None):
        ids = xyz[:, 0, 0].long().tolist()
        rows.copy_(torch.cat([evdist.pack_outputs(_fake_outputs(i, i + 1, self.N)) for i in ids], 0))
        self.seen_inits.append(self.fps_init)
        self.fps_init = None
        return evdist.unpack_outputs(rows, self.N)


def _worker_pipeline_forward(rank, world, port, gB, N, q):
    """[r6] GatherPipeline.forward: ONE call per step (forward into the next buffer + asynchronous gather), results one step late"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = evdist.shard_range(gB, rank, world)
    net = _FakeNet(N)
    pipe = evdist.GatherPipeline(N, gB, "cpu", depth=2, net=net)
    ok, prev, posted = True, None, []
    for step in range(4):
        xyz = (torch.arange(lo, hi, dtype=torch.float32) + 100 * step).view(-1, 1, 1).expand(hi - lo, 4, N).contiguous()
        cur = pipe.forward(xyz, fps_init=[step] * 4, post=lambda out: posted.append(int(out["class_logits"][0, 0, 0])))
        if prev is not None:
            st, h = prev
            full, want = h.result(), _fake_outputs(100 * st, gB + 100 * st, N)
            ok = ok and torch.equal(full["class_logits"], want["class_logits"]) and torch.equal(full["left"]["vertices"], want["left"]["vertices"])
        prev = (step, cur)
    pipe.drain()
    ok = ok and net.seen_inits == [[s] * 4 for s in range(4)] and posted == [lo + 100 * s for s in range(4)]
    try:                                                   # inflight mode needs the net (and a GPU: slot streams)
        evdist.GatherPipeline(N, gB, "cpu", inflight=2)
        ok = False
    except ValueError as e:
        ok = ok and "needs net=" in str(e)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_pipeline_forward_world2():
    world, N, gB = 2, 64, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline_forward, args=(r, world, port, gB, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_transport_is_chosen_by_capability_not_by_name():
    """a gloo group cannot move device memory (host-staged transport for CUDA buffers, with ONE warning); the decision is
    asked of the group (`_group_moves_device_memory`), so a group whose backend string is not literally "nccl" but which has a
    device backend is not silently sent down the host-staged path"""
    class _G:                                              # a group with a device backend under another name string
        def _get_backend(self, device):
            class ProcessGroupNCCL:
                pass
            return ProcessGroupNCCL()
    assert evdist._group_moves_device_memory(_G())

    class _H:
        def _get_backend(self, device):
            raise RuntimeError("no backend for cuda")
    import unittest.mock as mock
    with mock.patch.object(dist, "get_backend_config", lambda g: "cpu:gloo"), mock.patch.object(dist, "get_backend", lambda g: "gloo"):
        assert not evdist._group_moves_device_memory(_H())
    with mock.patch.object(dist, "get_backend_config", lambda g: "cpu:gloo,cuda:nccl"):
        assert evdist._group_moves_device_memory(_H())
