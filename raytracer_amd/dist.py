"""Multi-GPU row partition of a frame and the gather of its slices (one process per GPU,
torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

The reference's only parallel build splits the image pixel-cyclically over its workers and has the
host collect the pixels (ppu/ppu_raytrace.c:237-263, 312-338).  Here the unit is an 8-row block
(the height of a wave's pixel tile): block b belongs to rank b % world, which balances sky rows
against sphere rows, and each rank renders its blocks into a compact local buffer
(rt_hip_render_interleaved).  One all-gather of the equally sized local buffers plus one
de-interleave copy rebuilds the frame on every rank.  Rendering needs no communication at all.
"""
from dataclasses import dataclass

import torch

BLOCK_ROWS = 8


@dataclass(frozen=True)
class RowPlan:
    height: int
    world: int

    @property
    def nblocks(self):
        return (self.height + BLOCK_ROWS - 1) // BLOCK_ROWS

    @property
    def blocks_per_rank(self):
        """Block slots in every rank's local buffer (ranks short of one block leave the last unused)."""
        return (self.nblocks + self.world - 1) // self.world

    @property
    def local_rows(self):
        return self.blocks_per_rank * BLOCK_ROWS

    def blocks_of(self, rank):
        return list(range(rank, self.nblocks, self.world))

    def rows_of(self, rank):
        """[(first frame row, row count, first local row)] of the blocks rank renders."""
        out = []
        for k, b in enumerate(self.blocks_of(rank)):
            y = b * BLOCK_ROWS
            out.append((y, min(BLOCK_ROWS, self.height - y), k * BLOCK_ROWS))
        return out


def deinterleave(gathered, plan, out=None):
    """gathered: [world * local_rows, ...] = the ranks' local buffers stacked in rank order.
    Returns the frame [height, ...]: frame block k*world + r = rank r's local block k."""
    rest = tuple(gathered.shape[1:])
    g = gathered.view(plan.world, plan.blocks_per_rank, BLOCK_ROWS, *rest)
    perm = (1, 0, 2) + tuple(range(3, 3 + len(rest)))
    frame = g.permute(*perm).reshape(plan.world * plan.local_rows, *rest)[:plan.height]
    if out is None:
        return frame.contiguous()
    out.copy_(frame)
    return out


class FrameGather:
    """Pre-allocated buffers for `all-gather the local slices, de-interleave into the frame`."""

    def __init__(self, plan, width, dtype, device, trailing=(), rank=0, force=False):
        self.plan = plan
        self.force = force  # a world of one goes through the collective too (rehearsal of the multi-rank path on one GPU)
        self.width = width
        shape_tail = (width,) + tuple(trailing)
        self.gathered = torch.zeros((plan.world * plan.local_rows,) + shape_tail, dtype=dtype, device=device)
        # this rank's slice of the gathered buffer IS its render target: the all-gather is in place
        self.local = self.gathered[rank * plan.local_rows:(rank + 1) * plan.local_rows]
        self.frame = torch.zeros((plan.height,) + shape_tail, dtype=dtype, device=device)
        # 32-bit pixel frames on the GPU are de-interleaved by the library's streaming kernel
        self._native = (self.gathered.is_cuda and not trailing and self.gathered.element_size() == 4)

        # (views and addresses made once: a rank of eight submits a frame every 0.15 ms, and the host is what bounds it)
        self._flat = (self.gathered.view(-1), self.local.reshape(-1))
        self._ptrs = (self.gathered.data_ptr(), self.frame.data_ptr())
        if self._native:
            from . import _lib
            self._deinterleave = _lib.load().rt_hip_deinterleave

    def gather(self, group=None, collective=True, stream=None):
        """all-gather (on the current stream) + de-interleave (on `stream`, default the current one)"""
        if (self.plan.world > 1 or self.force) and collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self._flat[0], self._flat[1], group=group)
        return self.rebuild(stream)

    def rebuild(self, stream=None):
        """gathered -> frame on `stream` (a raw stream handle; default: the current stream)"""
        if self._native:
            if stream is None:
                stream = torch.cuda.current_stream(self.gathered.device).cuda_stream
            rc = self._deinterleave(self._ptrs[0], self._ptrs[1], self.width, self.plan.height, self.plan.world, stream)
            if rc:
                from . import _lib
                _lib.check(rc, "rt_hip_deinterleave")
            return self.frame
        return deinterleave(self.gathered, self.plan, out=self.frame)


class _NoStream:
    """Stand-in for a HIP stream / event on a device without streams (the CPU tests): everything runs in
    program order, so waiting and recording are nothing."""
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class FramePipeline:
    """`inflight` frames of one rank submitted ahead of their collection: what bench.py times and what
    tests/test_dist_cpu.py drives with gloo on the CPU - one implementation for both.

    Frame i uses slot k = i % inflight: a FrameGather (its `local` slice is the render target and, in
    place, this rank's contribution to the all-gather) and the events that order the two streams:
      render stream:  [wait gathered[k] of the frame that used the slot before]  render(i, k, local, stream)
      done stream:    where the frame is complete (the render stream, or the tail stream of a context)
      comm stream:    wait rendered[k]; all-gather in place; de-interleave into frame; record gathered[k]
    With gather_stream == "same" the comm stream IS the render stream (frames wait for their gathers).
    render_priority "high": the shared render stream is created with the high stream priority, so that what runs
    beside it on other streams (a context's tail stream) takes the issue slots it leaves, not its own.

    render(i, k, local, stream): enqueue the render of frame i into `local` ([local_rows, width, ...]) on
    `stream` (asynchronous).  finish(i, k): wait for the render, return its statistics (an object with
    overflow_retries, or None).  The gather is enqueued at submit time, behind the render on the device
    but BEFORE the host has checked the render for queue overflow: a batch that overflowed is rendered
    again by finish(), after the gather has copied its rows.  collect() therefore refuses such a frame
    (size the context so that it cannot happen, or gather after collect)."""

    def __init__(self, plan, width, dtype, device, rank, render, finish, inflight=2, gather_stream="side",
                 done_stream=None, collective=True, group=None, render_streams=None, trailing=(),
                 render_priority="normal", force_collective=False):
        self.plan, self.rank, self.inflight = plan, rank, max(1, inflight)
        self.render_fn, self.finish_fn = render, finish
        self.collective, self.group = collective, group
        # force_collective: one rank takes the path of many - local slice, all-gather (of one), de-interleave: every
        # RCCL call of the multi-rank path runs, on the one GPU there is (bench.py RT_BENCH_RCCL_WORLD1, tests)
        self.multi = plan.world > 1 or force_collective
        self.cuda = torch.device(device).type == "cuda"
        self.gathers = [FrameGather(plan, width, dtype, device, trailing=trailing, rank=rank, force=force_collective) for _ in range(self.inflight)]
        if self.cuda:
            shared = torch.cuda.Stream(device=device, priority=-1 if render_priority == "high" else 0)
            self.streams = list(render_streams) if render_streams else [shared] * self.inflight
            self.comm = (torch.cuda.Stream(device=device) if gather_stream == "side" else None) if self.multi else None
            self.rendered = [torch.cuda.Event() for _ in range(self.inflight)]
            self.gathered = [torch.cuda.Event() for _ in range(self.inflight)]
        else:
            self.streams = [_NoStream()] * self.inflight
            self.comm = _NoStream() if self.multi and gather_stream == "side" else None
            self.rendered = [_NoStream() for _ in range(self.inflight)]
            self.gathered = [_NoStream() for _ in range(self.inflight)]
        self.done_stream = done_stream  # where a frame is complete when it is not its render stream

    def _on(self, stream):
        return torch.cuda.stream(stream) if self.cuda else stream

    def submit(self, i):
        k = i % self.inflight
        st, fg = self.streams[k], self.gathers[k]
        # (render_fn is handed its stream; no stream context around it: entering and leaving one costs the host
        # 10 us a frame, and with more than one rank the host is what bounds the frame rate)
        if not self.multi:
            self.render_fn(i, k, fg.frame, st)  # one rank: straight into the frame, nothing to gather
            return
        if i >= self.inflight:
            st.wait_event(self.gathered[k])  # the previous gather out of this slot's buffers is done
        self.render_fn(i, k, fg.local, st)
        self.rendered[k].record(self.done_stream if self.done_stream is not None else st)
        comm = self.comm if self.comm is not None else st
        comm.wait_event(self.rendered[k])
        if self.collective and self.cuda:
            with torch.cuda.stream(comm):  # (the collective runs on the current stream)
                fg.gather(group=self.group, collective=True, stream=comm.cuda_stream)
        else:
            fg.gather(group=self.group, collective=self.collective, stream=comm.cuda_stream if self.cuda else None)
        self.gathered[k].record(comm)

    def collect(self, i, sizing=False):
        """sizing=True: a frame rendered only so that its context sizes its queues (a context that sizes itself
        renders its first frame of a shape in batches that may overflow and be rendered again): its gathered copy
        is not looked at, and the retries are no error"""
        k = i % self.inflight
        stats = self.finish_fn(i, k)
        if self.multi and not sizing and stats is not None and getattr(stats, "overflow_retries", 0):
            raise RuntimeError("a batch of frame %d overflowed its queues and was rendered again after its rows had "
                               "been gathered: raise the context's capacity" % i)
        return stats

    def frame(self, i):
        """the complete frame i (after collect(i)): waits for its gather"""
        k = i % self.inflight
        self.gathered[k].synchronize()
        return self.gathers[k].frame


def tone_slices(local_radiance, local_argb, local_pixels, total_pixels, op, ld_max, l_max, key_pixel_local=-1,
                use_key=False, group=None):
    """The tone-reproduction step (apply_tone, raytrace.c:792-871) on a frame whose pixels are spread
    over the ranks - the one exchange step of the path: every rank sums the luminance and
    log-luminance of its own `local_pixels` (the valid prefix of its compact buffer), the 3 doubles are
    all-reduced (RCCL), every rank applies the operator to its pixels.  key_pixel_local: index of
    Reinhard's key pixel inside THIS rank's buffer, or -1 on the ranks that do not hold it;
    use_key says whether a key pixel is used at all (the same on every rank)."""
    import torch.distributed as dist
    from . import api
    sums = torch.zeros(3, dtype=torch.float64, device=local_radiance.device)
    api.tone_sums(local_radiance.data_ptr(), local_pixels, l_max, key_pixel_local, sums.data_ptr(),
                  torch.cuda.current_stream(local_radiance.device).cuda_stream)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    api.tone_apply(local_radiance.data_ptr(), local_argb.data_ptr() if local_argb is not None else 0, local_pixels,
                   op, ld_max, l_max, sums.data_ptr(), total_pixels, use_key,
                   torch.cuda.current_stream(local_radiance.device).cuda_stream)
    return sums
