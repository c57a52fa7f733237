"""One-rank rehearsal of the RCCL calls the rows / samples sharding modes make (dtypes, shapes, in-place forms), for a
box with a single GPU: world size 1 cannot show scaling, but it does show that every collective is accepted by the
backend on device tensors of the shapes bench.py uses.  Usage: python tools/rccl_rehearsal.py [height width]"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29512")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
accum = torch.rand((h, w, 4), dtype=torch.float32, device="cuda")
want = accum.clone()
dist.barrier()
dist.all_reduce(accum)
band = torch.empty_like(accum)
dist.reduce_scatter_tensor(band, accum)
rgba = (band * 255.0).to(torch.uint8)
full = torch.empty_like(rgba)
dist.all_gather_into_tensor(full, rgba.clone())
recv = [torch.empty_like(rgba)]
dist.all_gather(recv, rgba)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(band, want) and torch.equal(full, rgba) and torch.equal(recv[0], rgba) and float(t.item()) == 1.5
dist.barrier()
dist.destroy_process_group()
print("rccl rehearsal ok", h, w)
