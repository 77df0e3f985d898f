"""1-rank RCCL group on one GPU: does init / all_reduce / barrier work on this box?  (run with a short timeout)"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
t0 = time.time()
def log(msg): print(f"[{time.time() - t0:6.2f}s] {msg}", flush=True)
torch.cuda.set_device(0)
log("init_process_group ...")
if os.environ.get("EAGER", "1") == "1":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
log("initialised")
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize(); log(f"all_reduce ok {float(x[0])}")
w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize(); log("async all_reduce ok")
dist.barrier(); log("barrier ok")
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); log(f"max ok {t.item()}")
dist.destroy_process_group(); log("done")
