"""The multi-GPU split of a frame on the one GPU of the test box: the interleaved row shares of N ranks rendered one
after the other must compose the single-launch frame, and the sample-sharded exchange (reduce-scatter of accumulators,
resolve of the owned rows, all-gather of RGBA8 bands) runs through RCCL on device tensors with a world of one rank and
is checked against the CPU oracle."""
import os
import socket

import numpy as np
import pytest

from helpers import assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,size,world", [("scene1", (200, 150), 3), ("scene4", (96, 75), 2), ("spheres200", (136, 88), 8)])
def test_interleaved_tile_row_shares_compose_the_frame(pkg, ctx, name, size, world):
    import torch
    s = pkg.Scene(name)
    w, h = size
    ctx.load(s, w, h)
    acc = bool(s.desc.diffuse)
    full_rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    full_acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # torch fills on its own stream; the context's stream does not wait for it
    ctx.render_device(frame_index=1, accumulate=acc, seed=3, d_rgba8=full_rgba.data_ptr(), d_accum=full_acc.data_ptr())
    rays = ctx.stats()["rays_total"]
    rgba = torch.zeros_like(full_rgba)
    accum = torch.zeros_like(full_acc)
    torch.cuda.synchronize()   # torch fills on its own stream; the context's stream does not wait for it
    share_rays = 0
    for rank in range(world):
        ctx.render_device(frame_index=1, accumulate=acc, seed=3, rows=(rank * 8, h), tile_row_stride=world,
                          d_rgba8=rgba.data_ptr(), d_accum=accum.data_ptr())
        share_rays += ctx.stats()["rays_total"]
        mine = np.zeros(h, dtype=bool)
        for b, e in pkg.interleaved_tile_rows(h, rank, world):
            mine[b:e] = True
        got = rgba.cpu().numpy()
        assert np.array_equal(got[mine], full_rgba.cpu().numpy()[mine])
    assert share_rays == rays
    assert torch.equal(rgba, full_rgba) and torch.equal(accum.view(torch.int32), full_acc.view(torch.int32))
    with pytest.raises(pkg.BackendError, match="multiple of 8"):
        ctx.render_device(frame_index=1, rows=(4, h), tile_row_stride=2)


def test_sample_sharded_exchange_through_rccl_with_one_rank(pkg, orc):
    """samples -> reduce_scatter_tensor -> rt_resolve_device -> all_gather_into_tensor on device tensors, backend nccl
    (= RCCL), world size 1: every collective is the identity, so the frame must equal the oracle's bit for bit."""
    import torch
    import torch.distributed as dist
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ctx = pkg.Context(0)
    try:
        s = pkg.Scene("spheres200")
        w, h, spp = 160, 120, 4
        ctx.load(s, w, h)
        xch = pkg.FrameExchange(dist, h, w, 0, 1, device="cuda", backend="nccl", force_collectives=True)
        assert xch.collectives and xch.reduce_scatter
        sh = torch.cuda.current_stream().cuda_stream
        for idx in pkg.shard_samples(spp, 0, 1):
            ctx.render_device(frame_index=idx, accumulate=True, seed=11, d_rgba8=xch.frame.data_ptr(), d_accum=xch.accum.data_ptr(), stream=sh)
        rb, re, band = xch.sum_accumulators()
        assert (rb, re) == (0, h) and band.data_ptr() == xch.band_accum.data_ptr()
        ctx.resolve_device(spp, rows=(0, re - rb), d_accum=band.data_ptr(), d_rgba8=xch.band_rgba.data_ptr(), stream=sh)
        xch.gather_bands(rb, re)
        torch.cuda.synchronize()
        acc = np.zeros((h, w, 4), dtype=np.float32)
        for f in range(1, spp + 1):
            ref = orc.render(s.desc, s.camera(w, h), (), frame_index=f, accumulate=True, seed=11, accum=acc, n_threads=8)
        assert_parity(xch.frame[:h].cpu().numpy(), ref["rgba"], band.cpu().numpy(), acc, what="sample-sharded frame through RCCL")
        # the rows mode's collective as well: an all-gather of interleaved tile rows
        s3 = pkg.Scene("scene3")
        ctx.load(s3, w, h)
        rb, re, stride = xch.render_rows()
        ctx.render_device(frame_index=1, rows=(rb, re), tile_row_stride=stride, d_rgba8=xch.frame.data_ptr(), d_accum=xch.accum.data_ptr(), stream=sh)
        xch.gather_interleaved()
        torch.cuda.synchronize()
        ref3 = orc.render(s3.desc, s3.camera(w, h), s3.textures())
        assert np.array_equal(xch.frame[:h].cpu().numpy(), ref3["rgba"])
    finally:
        ctx.close()
        dist.destroy_process_group()
