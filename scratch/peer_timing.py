"""us per all-reduce of the 24-pose reduced system: peer-read kernel between two processes on one GPU vs RCCL (world 1).
   python scratch/peer_timing.py"""
import os, socket, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dba-fusion_amd"))

def worker(rank, world, port):
    import torch.distributed as dist
    from dbaf_amd.peer import PeerAllReduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    def exchange(mine):
        hs = [None] * world; dist.all_gather_object(hs, mine); return hs
    peer = PeerAllReduce(rank, world, 1 << 18, exchange)
    dist.barrier()
    for n in (20880, 6 * 64 * (6 * 64 + 1)):
        t = torch.ones(n, dtype=torch.float64, device="cuda")
        for _ in range(20): peer.all_reduce(t)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(200): peer.all_reduce(t)
        torch.cuda.synchronize()
        if rank == 0: print("peer-read, %d processes on one GPU, %7d doubles: %.1f us per all-reduce" % (world, n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
        dist.barrier()
    peer.close(); dist.destroy_process_group()

if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])); sys.exit(0)
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [subprocess.Popen([sys.executable, __file__, str(r), str(world), str(port)]) for r in range(world)]
        [p.wait(timeout=300) for p in ps]
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    for n in (20880, 6 * 64 * (6 * 64 + 1)):
        t = torch.ones(n, dtype=torch.float64, device="cuda")
        for _ in range(20): dist.all_reduce(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): dist.all_reduce(t)
        torch.cuda.synchronize()
        print("RCCL, world 1, %7d doubles: %.1f us per all-reduce (no peer to wait for: launch + copy only)" % (n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
    dist.destroy_process_group()
