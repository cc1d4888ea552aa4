import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as td
torch.cuda.set_device(0)
t0 = time.time()
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(9350, device="cuda")
td.all_reduce(x); torch.cuda.synchronize()
print("nccl init + first all_reduce: %.2f s" % (time.time() - t0), float(x.sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): td.all_reduce(x)
e1.record(); torch.cuda.synchronize()
print("all_reduce 37KB, world 1: %.1f us each" % (e0.elapsed_time(e1) * 1e3 / 200))
td.barrier(); 
from three_mlagents_amd import dist
print("dist helpers:", dist.world_size(), dist.rank(), dist.allreduce_max_float(1.5, device="cuda"))
td.destroy_process_group()
