import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
import libtsd_amd as t
import bench
class A: pass
args = A(); args.log2n = bench.LOG2N; args.force_dist = True
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev) if os.environ.get("DIAG_SIDE_STREAM") else None
if side is not None:
    torch.cuda.set_stream(side)
for name in sys.argv[1:]:
    w = bench.WORKLOADS[name](t, torch, dev, 0, 1, args)
    for _ in range(30): w.step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "enqueue ms/step", round((t1 - t0) / K * 1e3, 4), "total ms/step", round((t2 - t0) / K * 1e3, 4), flush=True)
    # pieces: exchange only
    if getattr(w, "pipe", None) is not None:
        t0 = time.perf_counter()
        for _ in range(K):
            ex = w.pipe.post(w.halo_out); ex.finish(); w.pipe.consumed()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(name, "exchange alone: enqueue", round((t1 - t0) / K * 1e3, 4), "total", round((t2 - t0) / K * 1e3, 4), flush=True)
dist.destroy_process_group()
