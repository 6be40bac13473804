"""rt_hip_tone_static: the tone step through an entry point that allocates nothing (its partial sums live
in a static device array of the library).  Called through the C-ABI and compared bit for bit with
rt_hip_tone, the existing entry point whose kernels it shares, on two calls in a row - the second with a
smaller frame, so that whatever the first call left in the shared array would show."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _lib

pytestmark = pytest.mark.gpu

MAX_PIXELS = 1 << 28  # RT_TONE_STATIC_MAX_PIXELS, include/rt_hip.h


def _frame(torch, rng, n):
    # radiance as the render writes it: [0, 1] with clamped highlights and black pixels
    rad = np.clip(rng.gamma(0.6, 0.5, (n, 3)), 0.0, 1.0).astype(np.float32)
    rad[rng.random(n) < 0.05] = 0.0
    return torch.from_numpy(rad).cuda()


def test_static_scratch_entry_point_equals_rt_hip_tone_twice_in_a_row():
    import torch
    assert torch.cuda.is_available(), "this test needs the MI355X"
    rng = np.random.default_rng(7)
    first = True
    # (pixels, operator, key pixel): a 1080p frame, then smaller ones; Reinhard with and without a key pixel
    for n, op, key in [(1920 * 1080, rt.TONE_WARD, -1), (333 * 217, rt.TONE_REINHARD, 1234),
                       (640 * 480, rt.TONE_REINHARD, -1), (77, rt.TONE_WARD, -1), (1, rt.TONE_WARD, -1)]:
        src = _frame(torch, rng, n)
        want_rad, want_argb = src.clone(), torch.zeros(n, dtype=torch.int32, device="cuda")
        rt.tone(want_rad.data_ptr(), want_argb.data_ptr(), n, op, 100.0, 1000.0, key)
        torch.cuda.synchronize()
        got_rad, got_argb = src.clone(), torch.full((n,), -1, dtype=torch.int32, device="cuda")
        free_before = torch.cuda.mem_get_info()[0]
        rt.tone_static(got_rad.data_ptr(), got_argb.data_ptr(), n, op, 100.0, 1000.0, key)
        torch.cuda.synchronize()
        if not first:  # (the first call of a process may still load the code object)
            assert torch.cuda.mem_get_info()[0] == free_before, "the entry point allocated device memory"
        first = False
        assert torch.equal(got_rad.view(torch.int32), want_rad.view(torch.int32)), (n, op)
        assert torch.equal(got_argb, want_argb), (n, op)
        assert not torch.equal(got_rad, src)  # it did something
    # on a side stream, without a pixel buffer
    src = _frame(torch, rng, 4096)
    want = src.clone()
    rt.tone(want.data_ptr(), 0, 4096, rt.TONE_WARD, 100.0, 1000.0, -1)
    side = torch.cuda.Stream()
    got = src.clone()
    torch.cuda.synchronize()
    rt.tone_static(got.data_ptr(), 0, 4096, rt.TONE_WARD, 100.0, 1000.0, -1, stream=side.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    # limits and argument errors: a clear error, nothing launched
    with pytest.raises(rt.RtHipError, match="RT_TONE_STATIC_MAX_PIXELS"):
        rt.tone_static(got.data_ptr(), 0, MAX_PIXELS + 1, rt.TONE_WARD, 100.0, 1000.0, -1)
    with pytest.raises(rt.RtHipError):
        rt.tone_static(got.data_ptr(), 0, 4096, 7, 100.0, 1000.0, -1)
    with pytest.raises(rt.RtHipError):
        rt.tone_static(got.data_ptr(), 0, 4096, rt.TONE_REINHARD, 100.0, 1000.0, 4096)
    rt.tone_static(got.data_ptr(), 0, 0, rt.TONE_WARD, 100.0, 1000.0, -1)  # an empty frame is a no-op
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
