"""GPU box: the RCCL control-plane calls bench.py makes around the timed region (init with device_id, barrier, float64 MAX
all-reduce, destroy), with world size 1 -- the only size a one-GPU box allows; the 2/4/8-GPU runs are the driver's."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", float(t.item()), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.barrier(); dist.destroy_process_group()
