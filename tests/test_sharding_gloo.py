"""The N > 1 path on the CPU: two processes under torch.distributed's gloo backend drive the package's sharding logic
with the oracle as the renderer.  Row bands must reassemble the single-process frame exactly; sample-index sharding
must give the same accumulated frame up to the summation order of the all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A rendezvous port below the range the kernel hands out for outgoing connections (32768...): a port that the kernel
    picked for this process and that was closed again can be taken as a source port by a process that is still winding down
    before the ranks get to bind it."""
    import random
    for _ in range(64):
        port = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
            return port
        except OSError:
            continue
        finally:
            s.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = g.load_package(), g.load_oracle()
    w, h = 96, 72
    if mode == "rows":
        s = pkg.Scene("scene1")
        cam = s.camera(w, h)
        rows = pkg.shard_rows(h, rank, world)
        r = orc.render(s.desc, cam, s.textures(), rows=rows)
        local = torch.from_numpy(r["rgba"])
        full = pkg.gather_row_bands(dist, local, h, rank, world)
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), full.numpy())
    elif mode in ("interleaved", "interleaved_ragged"):
        # the rows mode as bench.py drives it: every world-th 8-row tile row per rank, one all-gather of tile rows
        if mode == "interleaved_ragged":
            h = 75   # 10 tile rows, the last one of 3 rows
        s = pkg.Scene("scene1")
        cam = s.camera(w, h)
        xch = pkg.FrameExchange(dist, h, w, rank, world, device="cpu", backend="gloo")
        rb, re, stride = xch.render_rows()
        assert (rb, re, stride) == (rank * 8, h, world)
        frame = xch.frame.numpy()
        for b, e in pkg.interleaved_tile_rows(h, rank, world):
            frame[b:e] = orc.render(s.desc, cam, s.textures(), rows=(b, e))["rgba"][b:e]
        xch.gather_interleaved()
        np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), xch.frame[:h].numpy())
    elif mode in ("scatter", "scatter_ragged"):
        # sample sharding as bench.py drives it: sum of the accumulators, resolve of the owned rows, all-gather of RGBA8
        if mode == "scatter_ragged":
            h = 75   # not divisible by 2: 8-row-aligned bands of different sizes
        s = pkg.Scene("scene2")
        cam = s.camera(w, h)
        xch = pkg.FrameExchange(dist, h, w, rank, world, device="cpu", backend="gloo")
        assert not xch.reduce_scatter, "reduce-scatter is chosen for RCCL only; gloo sums with an all-reduce"
        acc = xch.accum.numpy()
        for idx in pkg.shard_samples(6, rank, world):
            orc.render(s.desc, cam, (), frame_index=idx, accumulate=True, seed=0x5EED, accum=acc)
        rb, re, band = xch.sum_accumulators()
        assert (re - rb) == band.shape[0] and 0 <= rb < re <= h
        # resolve as render_chunk's accumulation tail does (src/renderer.rs:62-66, :29-36)
        lin = np.clip(band.numpy() / np.float32(6.0), 0.0, 1.0)
        rgba_band = np.stack([(lin[..., 0] * 255.0).astype(np.uint8), (lin[..., 1] * 255.0).astype(np.uint8),
                              (lin[..., 2] * 255.0).astype(np.uint8), np.full(lin.shape[:2], 255, np.uint8)], axis=-1)
        xch.frame[rb:re] = torch.from_numpy(rgba_band)
        xch.gather_bands(rb, re)
        np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), xch.frame[:h].numpy())
        np.save(os.path.join(out_dir, f"{mode}_band_{rank}.npy"), np.array([rb, re]))
    else:
        s = pkg.Scene("scene2")
        cam = s.camera(w, h)
        acc = np.zeros((h, w, 4), dtype=np.float32)
        acc_t = torch.from_numpy(acc)

        def render(idx):
            orc.render(s.desc, cam, (), frame_index=idx, accumulate=True, seed=0x5EED, accum=acc)

        n = pkg.render_sample_sharded(dist, render, acc_t, 6, rank, world)
        assert n == 3
        np.save(os.path.join(out_dir, f"samples_{rank}.npy"), acc_t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rows", "samples"])
def test_two_ranks_gloo(pkg, orc, tmp_path, mode):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    w, h = 96, 72
    if mode == "rows":
        s = pkg.Scene("scene1")
        ref = orc.render(s.desc, s.camera(w, h), s.textures())
        for r in range(2):
            assert np.array_equal(np.load(tmp_path / f"rows_{r}.npy"), ref["rgba"]), f"rank {r} did not reassemble the frame"
    else:
        s = pkg.Scene("scene2")
        acc = np.zeros((h, w, 4), dtype=np.float32)
        for f in range(1, 7):
            orc.render(s.desc, s.camera(w, h), (), frame_index=f, accumulate=True, seed=0x5EED, accum=acc)
        a, b = np.load(tmp_path / "samples_0.npy"), np.load(tmp_path / "samples_1.npy")
        assert np.array_equal(a, b), "all-reduce left the ranks with different accumulators"
        # same six samples, summed in a different order: relative 1e-6, far inside the 1e-3 linear budget after /6
        assert np.allclose(a, acc, rtol=1e-6, atol=1e-6)
        assert a[..., 3].min() == 6.0 and a[..., 3].max() == 6.0


@pytest.mark.parametrize("mode", ["interleaved", "interleaved_ragged"])
def test_two_ranks_interleaved_tile_rows(pkg, orc, tmp_path, mode):
    """The rows mode end to end: interleaved 8-row tile rows, one all-gather, both ranks hold the single-process frame."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    w, h = 96, (72 if mode == "interleaved" else 75)
    s = pkg.Scene("scene1")
    ref = orc.render(s.desc, s.camera(w, h), s.textures())
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"{mode}_{r}.npy"), ref["rgba"]), f"rank {r} did not reassemble the frame"


def test_interleaved_tile_rows_cover_every_row_once(pkg):
    for h in (1, 8, 9, 72, 75, 2160):
        for world in (1, 2, 3, 8):
            rows = sorted(y for r in range(world) for b, e in pkg.interleaved_tile_rows(h, r, world) for y in range(b, e))
            assert rows == list(range(h))


@pytest.mark.parametrize("mode", ["scatter", "scatter_ragged"])
def test_two_ranks_reduce_scatter_resolve_all_gather(pkg, orc, tmp_path, mode):
    """Sample sharding end to end: both ranks end with the same full RGBA8 frame, within 1 LSB of the single-process one."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    w, h = 96, (72 if mode == "scatter" else 75)
    s = pkg.Scene("scene2")
    ref = None
    acc = np.zeros((h, w, 4), dtype=np.float32)
    for f in range(1, 7):
        ref = orc.render(s.desc, s.camera(w, h), (), frame_index=f, accumulate=True, seed=0x5EED, accum=acc)
    a, b = np.load(tmp_path / f"{mode}_0.npy"), np.load(tmp_path / f"{mode}_1.npy")
    assert np.array_equal(a, b), "the ranks hold different frames after the all-gather"
    bands = [tuple(np.load(tmp_path / f"{mode}_band_{r}.npy")) for r in range(2)]
    assert bands[0][0] == 0 and bands[0][1] == bands[1][0] and bands[1][1] == h
    # the summation order differs from the sequential +=, so bytes may differ by one step at rounding boundaries
    assert np.abs(a.astype(np.int16) - ref["rgba"].astype(np.int16)).max() <= 1
    assert (a != ref["rgba"]).mean() < 1e-3


def test_sample_shards_cover_every_index_once(pkg):
    for n in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in pkg.shard_samples(n, r, world))
            assert seen == list(range(1, n + 1))


def _pattern_worker(rank, world, port, out_dir):
    """Exchange logic alone, at the heights of the benchmark frames: rows of a known pattern instead of rendered ones."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = g.load_package()
    w = 8
    ok = True
    for h in (2160, 4320, 600, 75):
        want = (torch.arange(h, dtype=torch.int64) * 7 % 251).to(torch.uint8).view(h, 1, 1).expand(h, w, 4)
        # rows mode: this rank's interleaved tile rows, then one all-gather
        xch = pkg.FrameExchange(dist, h, w, rank, world, device="cpu", backend="gloo")
        for b, e in pkg.interleaved_tile_rows(h, rank, world):
            xch.frame[b:e] = want[b:e]
        xch.gather_interleaved()
        ok = ok and bool(torch.equal(xch.frame[:h], want))
        # samples mode: accumulators summed, owned rows "resolved" (here: the pattern where the sum is right), bands gathered
        xch = pkg.FrameExchange(dist, h, w, rank, world, device="cpu", backend="gloo")
        xch.accum.fill_(float(rank + 1))
        b, e, band = xch.sum_accumulators()
        ok = ok and bool((band == float(world * (world + 1) // 2)).all()) and band.shape[0] == e - b
        xch.frame[b:e] = want[b:e]
        xch.gather_bands(b, e)
        ok = ok and bool(torch.equal(xch.frame[:h], want))
    with open(os.path.join(out_dir, f"pattern_{rank}.txt"), "w") as fh:
        fh.write("ok" if ok else "bad")
    dist.barrier()   # no rank closes its connections while another is still inside the last collective
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 4])
def test_exchange_shapes_at_the_benchmark_heights_with_more_ranks(tmp_path, world):
    """FrameExchange with 3 and 4 ranks at 2160, 4320, 600 and 75 rows (270 and 540 tile rows do not divide by 4, 75 rows
    end in a short tile): the interleaved all-gather and the sum / band gather of the samples mode rebuild every row on
    every rank."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pattern_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"pattern_{r}.txt").read() == "ok", f"rank {r}"
