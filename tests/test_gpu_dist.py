"""GPU: the multi-rank path on the one GPU there is -- TWO ranks on device 0, `gloo` process group.

RCCL refuses two ranks on one device, so everything of the N > 1 path EXCEPT RCCL's own transport runs on hardware here
(the reference's analogue is nn.DataParallel, /root/reference/src/Ev2Hands/train.py:68,105): each rank shards
the global batch and the globally drawn FPS start vectors, runs the REAL HIP forward into its `GatherPipeline.rows()` slice of
the gather buffer while a process group is alive (torch's stream pool, gloo's threads and a second process on the same GPU
around the library's two-stream schedule), gathers in place (host-staged transport, ev2hands_amd/dist.py), and the gathered
predictions of BOTH ranks must equal the unsharded forward BIT FOR BIT.  Also exercised on hardware: the pipeline's
generation guard, unequal shards, and the per-rank two-stream self-check bench.py reports at world 8.
"""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, gB, N, C, precision, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", ERPC="1" if C == 5 else "0",
                          EV2H_PRECISION=precision)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import time

        import torch.distributed as dist

        from ev2hands_amd import _lib, dist as evdist, synth
        from ev2hands_amd.model import TEHNetWrapper
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        _lib.lib().ev2h_init()                                   # side stream first, then the process group
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assets = {s: synth.synth_mano_assets(s, 3) for s in ("left", "right")}
        net = TEHNetWrapper(dev, mano_assets=assets)
        net.load_state_dict(synth.synth_state_dict(C, 3), strict=True)
        net.eval()
        xyz = synth.synth_cloud("E", gB, C, N, 12).to(dev)       # the GLOBAL batch (every rank can build it: it is synthetic)
        inits = synth.fps_inits(gB, N, 12)
        lo, hi = evdist.shard_range(gB, rank, world)
        pipe = evdist.GatherPipeline(N, gB, dev, depth=2)
        assert pipe.bufs[0].host_staged
        res = {}
        with torch.no_grad():
            # reference: the unsharded forward on this rank's own device context
            net.net.fps_init = inits
            full = net(xyz)
            torch.cuda.synchronize()
            want = {"class_logits": full["class_logits"].clone()}
            for side in ("left", "right"):
                want[side] = {k: full[side][k].clone() for k in ("global_orient", "hand_pose", "betas", "transl", "vertices", "j3d")}
            # three pipelined steps: forward into the gather buffer, asynchronous gather, result one step later
            pend = []
            for step in range(3):
                net.net.fps_init = evdist.shard_fps_inits(inits, lo, hi)
                net.net(xyz[lo:hi], net.hands, rows=pipe.rows())
                pend.append(pipe.submit())
                if step >= 1:
                    got = pend[step - 1].result() if step == 1 else None
                    if got is not None:
                        ok = torch.equal(got["class_logits"], want["class_logits"])
                        for side in ("left", "right"):
                            for k in want[side]:
                                ok = ok and torch.equal(got[side][k], want[side][k])
                        res["gathered_equals_unsharded"] = bool(ok)
            # generation guard: pend[0]'s buffer was handed to step 2's forward -- its result must be refused
            try:
                pend[0].result()
                res["generation_guard"] = False
            except RuntimeError:
                res["generation_guard"] = True
            got = pend[2].result()
            ok = torch.equal(got["class_logits"], want["class_logits"])
            for side in ("left", "right"):
                for k in want[side]:
                    ok = ok and torch.equal(got[side][k], want[side][k])
            res["last_gather_equals_unsharded"] = bool(ok)
            pipe.drain()

            # the same with two forwards in flight per rank, as ONE call per step [r6]: GatherPipeline(inflight=2).forward -- forward i on
            # slot stream i mod 2 with its own workspace, the gather of step i issued from that stream right behind its forward, the
            # wait for the gather that last read the buffer on the slot's stream too (config 5 as 16-window shards)
            pipe2 = evdist.GatherPipeline(N, gB, dev, depth=2, inflight=2, net=net)
            my_inits = evdist.shard_fps_inits(inits, lo, hi)
            pend2 = [pipe2.forward(xyz[lo:hi], fps_init=my_inits) for _ in range(2)]
            oks = []
            for step in range(2, 6):                         # results consumed one step late, buffers and slots reused
                got = pend2[step - 1].result()
                ok = torch.equal(got["class_logits"], want["class_logits"])
                for side in ("left", "right"):
                    for k in want[side]:
                        ok = ok and torch.equal(got[side][k], want[side][k])
                oks.append(bool(ok))
                pend2.append(pipe2.forward(xyz[lo:hi], fps_init=my_inits))
            try:                                             # the generation guard holds in this mode too
                pend2[0].result()
                oks.append(False)
            except RuntimeError:
                oks.append(True)
            pipe2.drain()
            got = pend2[-1].result()
            ok = torch.equal(got["class_logits"], want["class_logits"])
            for side in ("left", "right"):
                for k in want[side]:
                    ok = ok and torch.equal(got[side][k], want[side][k])
            res["inflight_gather_equals_unsharded"] = bool(ok) and all(oks)
            del pipe2

            # the per-rank two-stream self-check of bench.py, with both ranks hammering the same GPU
            L = _lib.lib()

            def timed(k):
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(k):
                    net.net(xyz[lo:hi], net.hands, rows=pipe.bufs[0].rows())
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            timed(3)
            t_on = timed(10)
            prev = L.ev2h_set_side_stream(0)
            timed(3)
            t_off = timed(10)
            res["two_stream_gain"] = round(t_off / t_on, 4)
            # the single-stream schedule is the same function: bit-identical rows (fps_init is consumed by a forward: set it again)
            b0 = pipe.bufs[0]
            mine = b0.full[b0.rank * b0.big:][:hi - lo]
            net.net.fps_init = evdist.shard_fps_inits(inits, lo, hi)
            net.net(xyz[lo:hi], net.hands, rows=b0.rows())
            torch.cuda.synchronize()
            a = mine.clone()
            L.ev2h_set_side_stream(prev)
            net.net.fps_init = evdist.shard_fps_inits(inits, lo, hi)
            net.net(xyz[lo:hi], net.hands, rows=b0.rows())
            torch.cuda.synchronize()
            res["single_stream_rows_equal_two_stream_rows"] = bool(torch.equal(a, mine))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


@pytest.mark.parametrize("gB,N,C,precision", [(6, 2048, 4, "f16x2"), (5, 1024, 5, "bf16x3"), (32, 8192, 4, "f16x2"), (6, 2048, 4, "f16")])
def test_two_ranks_on_one_gpu_gather_equals_unsharded(gB, N, C, precision):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, gB, N, C, precision, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    try:
        for _ in range(2):
            rank, res, err = q.get(timeout=600)
            assert err is None, f"rank {rank}: {err}"
            out[rank] = res
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    print(f"two ranks on one GPU, global batch {gB} (shards {[(gB + 1 - r) // 2 for r in range(2)]}), N = {N}, {precision}: "
          + "; ".join(f"rank {r}: two_stream_gain {out[r]['two_stream_gain']}" for r in sorted(out)))
    for r in (0, 1):
        assert out[r]["gathered_equals_unsharded"], out
        assert out[r]["last_gather_equals_unsharded"], out
        assert out[r]["generation_guard"], out
        assert out[r]["inflight_gather_equals_unsharded"], out
        assert out[r]["single_stream_rows_equal_two_stream_rows"], out
        assert 0.3 < out[r]["two_stream_gain"] < 3.0, out          # a report (two processes share the GPU here), not a bar


def _rank_auto(rank, world, port, q):
    """precision = "auto" under a process group: rank 1's verification is made to fail (tolerance 0); the decision is collective"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", ERPC="0")
        os.environ.pop("EV2H_PRECISION", None)
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import torch.distributed as dist

        from ev2hands_amd import _lib, dist as evdist, synth
        from ev2hands_amd.model import TEHNetWrapper
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        _lib.lib().ev2h_init()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        net = TEHNetWrapper(dev, mano_assets={s: synth.synth_mano_assets(s, 3) for s in ("left", "right")})
        net.load_state_dict(synth.synth_state_dict(4, 3), strict=True)
        net.eval()
        assert net.net.precision == "auto"
        if rank == 1:
            net.net.AUTO_TOLERANCE = 0.0                       # this rank's shard "fails" the check
        gB, N = 4, 512
        xyz = synth.synth_cloud("E", gB, 4, N, 12).to(dev)
        inits = synth.fps_inits(gB, N, 12)
        lo, hi = evdist.shard_range(gB, rank, world)
        net.net.fps_init = evdist.shard_fps_inits(inits, lo, hi)
        with torch.no_grad():
            net(xyz[lo:hi])
        torch.cuda.synchronize()
        rep = net.net.auto_report
        res = {"chose": net.net.effective_precision(), "ok": rep["ok"], "ok_this_rank": rep["ok_this_rank"], "ranks": rep.get("ranks")}
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, res, None))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


def test_auto_precision_decision_is_collective():
    """under torch.distributed every rank verifies f16x2 on its own shard and the group takes the MINIMUM -- one rank whose
    check fails sends ALL ranks to bf16x3 (a lone bf16x3 rank would break sharded == unsharded and stall every all-gather)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_auto, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    try:
        for _ in range(2):
            rank, res, err = q.get(timeout=600)
            assert err is None, f"rank {rank}: {err}"
            out[rank] = res
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert out[0]["ok_this_rank"] is True and out[1]["ok_this_rank"] is False, out
    assert out[0]["chose"] == out[1]["chose"] == "bf16x3" and out[0]["ok"] is False and out[0]["ranks"] == 2, out
