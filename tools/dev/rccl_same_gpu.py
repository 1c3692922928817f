"""Dev probe: does RCCL accept two ranks on ONE device (it would let the gather path run under the real backend on a 1-GPU box)?"""
import os, sys, torch, torch.distributed as dist
import torch.multiprocessing as mp


def w(rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2)
        t = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "all_reduce ->", t.tolist(), flush=True)
    except Exception as e:
        print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)


if __name__ == "__main__":
    mp.spawn(w, nprocs=2)
