"""rt_render_device_noalloc against rt_render_device, through the C ABI, on a real MI355X.

The no-alloc entry point keeps the launch's per-workgroup counter rows in a fixed device array inside the library
instead of a block grown with hipMalloc.  It must produce the same frame, accumulator and ray count, bit for bit, on
two calls in a row, the second with a smaller input (the array then still holds the first call's rows).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(pkg, ctx, torch, name, size, noalloc, events=False):
    s = pkg.Scene(name)
    w, h = size
    ctx.load(s, w, h)
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.render_device(frame_index=1, accumulate=False, seed=0x5EED, collect_events=events, d_rgba8=rgba.data_ptr(),
                      d_accum=accum.data_ptr(), noalloc=noalloc)
    st = ctx.stats()   # waits for the launch
    return rgba.cpu().numpy(), accum.cpu().numpy(), st


def test_noalloc_entry_point_equals_render_device_on_two_calls_in_a_row(pkg):
    import torch
    ctx = pkg.Context(0)
    try:
        # first call: 640x360 with the counting variant (40 counter words per workgroup), then a smaller frame of another
        # scene through the plain variant, then a ray-marching frame; every one against rt_render_device
        for name, size, events in (("scene1", (640, 360), True), ("scene3", (200, 120), False), ("scene4", (96, 64), False)):
            got = _frames(pkg, ctx, torch, name, size, noalloc=True, events=events)
            want = _frames(pkg, ctx, torch, name, size, noalloc=False, events=events)
            assert np.array_equal(got[0], want[0]), f"{name} {size}: RGBA8 bytes differ"
            assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), f"{name} {size}: accumulator differs"
            assert got[2]["rays_total"] == want[2]["rays_total"] > 0
            assert {k: got[2][k] for k in pkg.COUNTER_NAMES} == {k: want[2][k] for k in pkg.COUNTER_NAMES}
        # beyond the fixed array: 4K with 40 counter words per workgroup
        s = pkg.Scene("scene3")
        ctx.load(s, 3840, 2160)
        with pytest.raises(pkg.BackendError, match="2\\^20 counter words"):
            ctx.render_device(frame_index=1, collect_events=True, noalloc=True)
        ctx.render_device(frame_index=1, noalloc=True)          # the plain variant fits at 4K
        a = ctx.stats()["rays_total"]
        ctx.render_device(frame_index=1, noalloc=False)
        assert ctx.stats()["rays_total"] == a
        # no device memory is taken or given back by the entry point: the free memory hipMemGetInfo reports stays the same over
        # calls that a growing counter block would have made allocate (first frame of a camera, table frame, accumulating frame)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for f, acc in ((1, False), (1, False), (2, True), (3, True)):
            ctx.render_device(frame_index=f, accumulate=acc, noalloc=True, no_timing=True)
        ctx.stats()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "rt_render_device_noalloc changed the amount of free device memory"
        # with the words of an earlier rt_render_device launch of this shape in place, the entry point renders with the sky
        # classification too (before such a launch it renders without): same frame either way, still no allocation
        assert ctx.hierarchy_info()["classified_workgroups"] > 0
        # a large diffuse scene: the one-lane-per-pixel kernel before rt_render_device has allocated the path tracer's pools,
        # the re-grouping kernel after; both leave rt_render_device's frame, accumulator and ray count
        s = pkg.Scene("spheres1k"); s.desc.max_ray_bounces = 4
        fresh = pkg.Context(0)
        try:
            fresh.load(s, 320, 200)
            rgba = torch.zeros((200, 320, 4), dtype=torch.uint8, device="cuda")
            accum = torch.zeros((200, 320, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()   # torch fills on its own stream; the context's stream does not wait for it
            frames = []
            for noalloc in (True, False, True):
                accum.zero_()
                torch.cuda.synchronize()
                free_before = torch.cuda.mem_get_info()[0]
                rays = 0
                for f in (1, 2, 3, 4):   # (the fourth reads the camera rays' hits that the third -- the second of its kind -- wrote)
                    fresh.render_device(frame_index=f, accumulate=True, seed=9, noalloc=noalloc, d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr())
                    rays += fresh.stats()["rays_total"]
                torch.cuda.synchronize()
                if noalloc:
                    assert torch.cuda.mem_get_info()[0] == free_before
                frames.append((rgba.cpu().numpy(), accum.cpu().numpy(), rays))
            for got in (frames[0], frames[2]):
                assert np.array_equal(got[0], frames[1][0]) and np.array_equal(got[1].view(np.uint32), frames[1][1].view(np.uint32))
                assert got[2] == frames[1][2] > 0
        finally:
            fresh.close()
        # flag bits the library does not know are refused, by both entry points
        for noalloc in (True, False):
            with pytest.raises(pkg.BackendError, match="unknown bits"):
                ctx.render_device(frame_index=1, noalloc=noalloc, flags=1 << 9)
    finally:
        ctx.close()
