"""Several forwards in flight: throughput for batches that are too small to fill the chip on their own.

A rank's share of BASELINE.json config 5 -- 128 dense windows over 8 GPUs -- is 16 windows of 8192 points.  One such forward is
latency-bound for a fifth of its time: farthest point sampling is 512 DEPENDENT steps on 48 of the 256 CUs (~0.49 ms however few
windows there are), followed by selection kernels that are chains of L2 round trips.  Consecutive batches are independent
(eval-mode windows share nothing, /root/reference/src/Ev2Hands/model/TEHNet.py:168-197), so the head of batch i + 1 can run
under the matrix-pipe-bound tail of batch i: `InflightForward` issues forward i on stream i mod depth with its own workspace.
16 windows of 8192 points: 7 500 -> 10 300-10 500 windows/s with depth 2 (round 6, profiles/r6_inflight.txt,
profiles/r6_side_streams_by_measurement.txt; round 5: 7 090 -> 8 560), 8 windows of 2048 points: 7 200 -> 11 160; depth 3
over-subscribes the four hardware queues a process gets (9 000); at 128 windows per GPU the chip is already full (+2.5 %), at the
headline shape (256 windows of 2048 points) +2 % -- the default is one in flight.

Numbers are unchanged by construction: the same ev2h_forward with the same arguments, only on another stream
(tests/test_gpu_forward.py::test_inflight_forwards_are_bit_identical).
"""
from __future__ import annotations

import threading

import torch

from . import _lib


class InflightForward:
    """
        pipe = InflightForward(net, depth=2)            # net: TEHNetWrapper (eval)
        for batch in batches:
            ticket = pipe.submit(batch)                 # returns at once; the forward runs on one of `depth` streams
            ...
            out = ticket.result()                       # the CURRENT stream waits (device-side) for that forward; dict as net(batch)
        pipe.drain()

    A slot (stream + workspace) is reused every `depth` submissions; submit() makes the slot's stream wait for the caller's
    current stream first (the input is ready) and for the slot's previous forward by stream order.  `rows=` as in TEHNet.forward
    (the multi-GPU gather buffer; `dist.GatherPipeline(..., inflight=K)` composes the two).

    One host thread at a time: every slot's forward forks onto the side stream the library keeps for that slot's stream (per host
    thread and device; up to four, beyond that slot 0's is shared) and joins it with that slot's events; the library's tables are
    per host thread, so submit() holds a lock for the duration of the enqueue."""

    class Ticket:
        def __init__(self, out, event, device):
            self.out, self.event, self.device = out, event, device

        def result(self) -> dict:
            cur = torch.cuda.current_stream(self.device)          # the stream of the OUTPUTS' device, whatever the current device is
            cur.wait_event(self.event)
            # the tensors were allocated on the slot's stream and are now used on the caller's: tell the caching allocator
            for v in self.out.values():
                for t in (v.values() if isinstance(v, dict) else (v,)):
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
            return self.out

    def __init__(self, net, depth: int = 2):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.net, self.depth, self.i = net, depth, 0
        self.streams, self.ws, self.device = None, [None] * depth, None
        self._lock = threading.Lock()

    def _slot(self, device, nbytes: int):
        if self.streams is None:
            with torch.cuda.device(device):             # the library's side stream first, ON THIS DEVICE (it wants a hardware queue of its own)
                _lib.check(_lib.lib().ev2h_init(), "ev2h_init")
            # [r6] slot streams that the device really runs side by side, each with a side stream that runs beside all of them: HIP
            # maps streams onto a few hardware queues and two that share one run in order -- measured, not assumed
            # (_lib.concurrent_streams, ev2h_bind_stream; what comes out depends on what else the process created: profiles/r6_side_slots_ab.txt)
            self.streams = _lib.concurrent_streams(device, self.depth)
            self.device = torch.device(device)
            with torch.cuda.device(device):
                self.binding = [_lib.bind_stream(s_.cuda_stream) for s_ in self.streams]
        elif torch.device(device) != self.device:
            raise RuntimeError(f"InflightForward was started on {self.device}, got a batch on {device}")
        k = self.i % self.depth
        self.i += 1
        if self.ws[k] is None or self.ws[k].numel() < nbytes:
            # allocated ON the slot's stream: a workspace that is outgrown while its last forward is still running goes back to
            # that stream's pool, where any reuse is ordered behind that forward (the caching allocator's own rule)
            with torch.cuda.stream(self.streams[k]):
                self.ws[k] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.streams[k], self.ws[k]

    def submit(self, xyz: torch.Tensor, rows=None, post=None, pre=None) -> "InflightForward.Ticket":
        """post: optional callable(out) run right behind the forward ON THE SLOT'S STREAM (e.g. the collision term of the same
        batch, or the all-gather of its predictions) -- work queued on the caller's stream instead would make the next submit() wait
        for it.  pre: optional callable() run on the slot's stream BEFORE the forward and returning its `rows` (dist.GatherPipeline:
        wait for the gather that last read the buffer, then hand its rows out)."""
        B, _, N = xyz.shape
        with self._lock:
            stream, ws = self._slot(xyz.device, _lib.lib().ev2h_workspace_bytes(B, N))
            stream.wait_stream(torch.cuda.current_stream(xyz.device))
            with torch.cuda.stream(stream), torch.no_grad():
                if pre is not None:
                    rows = pre()
                out = self.net.net(xyz, self.net.hands, rows=rows, ws=ws)
                if post is not None:
                    post(out)
                ev = torch.cuda.Event()
                ev.record(stream)
            xyz.record_stream(stream)
        return InflightForward.Ticket(out, ev, xyz.device)

    def drain(self) -> None:
        if self.streams:
            cur = torch.cuda.current_stream(self.device)
            for s in self.streams:
                cur.wait_stream(s)
