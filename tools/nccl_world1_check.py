import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
x = torch.arange(12, device=dev, dtype=torch.uint8).reshape(3, 4)
bufs = [torch.empty_like(x)]
dist.gather(x, bufs, dst=0)
torch.cuda.synchronize()
assert torch.equal(bufs[0], x) and t.item() == 1.5
dist.barrier(); dist.destroy_process_group()
print("nccl world-1 ok")
