"""Can the flat-gradient all-reduce (RCCL through torch.distributed's nccl backend) sit INSIDE a captured HIP graph?
Single-rank nccl group on the one GPU of the box; the child is a fresh process of a parent that never touches the GPU.
Prints one JSON line: {"bare": ..., "step": ...} with "ok" or the error text."""
import json, os, socket, sys, traceback
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        x = torch.arange(68715, device=dev, dtype=torch.float32) / 7.0
        ref = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                dist.all_reduce(x, op=dist.ReduceOp.AVG)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                dist.all_reduce(x, op=dist.ReduceOp.AVG)
            same = []
            for _ in range(3):
                x.mul_(1.5); want = x.clone()
                g.replay(); torch.cuda.synchronize()
                same.append(bool(torch.equal(x, want)))
            ret["bare"] = "ok" if all(same) else f"replayed result differs: {same}"
        except Exception as e:
            ret["bare"] = "capture failed: " + repr(e)[:600]
            torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    with mp.Manager() as mgr:
        ret = mgr.dict()
        try:
            mp.spawn(worker, args=(1, port, ret), nprocs=1, join=True)
        except Exception as e:
            ret["spawn"] = repr(e)[:600]
        print(json.dumps(dict(ret)))
