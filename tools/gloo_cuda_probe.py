"""Probe: does gloo's all_gather_into_tensor work on CUDA tensors when two ranks share cuda:0? (rehearsal helper)"""
import os, sys, time
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
x = torch.full((1024,), rank + 1, dtype=torch.uint8, device=dev)
full = torch.empty((world, 1024), dtype=torch.uint8, device=dev)
for form in ("flat", "list"):
    t0 = time.time()
    if form == "flat":
        dist.all_gather_into_tensor(full.view(-1), x.view(-1))
    else:
        dist.all_gather(list(full.unbind(0)), x)
    torch.cuda.synchronize()
    print(f"rank {rank} {form}: {full[:, 0].tolist()} in {time.time() - t0:.2f}s", flush=True)
dist.barrier()
dist.destroy_process_group()
