import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29512", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, dtype=torch.bfloat16, device="cuda")
dist.broadcast(x, src=0)
meta = [[("a", (2, 3), 0, 6)], 123]
dist.broadcast_object_list(meta, src=0)
t = torch.tensor([1.5], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
out = [torch.empty(4, dtype=torch.uint8, device="cuda")]
dist.all_gather(out, torch.arange(4, dtype=torch.uint8, device="cuda"))
torch.cuda.synchronize()
print("rccl world-1 ok", dist.get_backend(), float(t), out[0].tolist())
dist.destroy_process_group()
