"""What the barrier around a timed region of bench.py costs on this box: dist.barrier() against a one-element all-reduce
+ synchronize, RCCL process group of one rank (25 us / 19 us on an MI355X: not what separates a cold region from a warm one).
    python tools/rccl_barrier_cost.py"""
import os, time, torch, torch.distributed as dist, datetime
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
tok=torch.zeros(1,device=dev)
def A():
    torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
def B():
    torch.cuda.synchronize(dev); dist.all_reduce(tok); torch.cuda.synchronize(dev)
def C():
    torch.cuda.synchronize(dev); dist.barrier(device_ids=[0]); torch.cuda.synchronize(dev)
for name,f in (("barrier",A),("all_reduce+sync",B),("barrier(device_ids)",C)):
    for _ in range(5): f()
    ts=[]
    for _ in range(50):
        t=time.perf_counter(); f(); ts.append(time.perf_counter()-t)
    ts.sort(); print(name, "median %.1f us, min %.1f us"%(ts[25]*1e6, ts[0]*1e6), flush=True)
dist.destroy_process_group()
