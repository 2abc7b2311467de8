import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    x = torch.full((4, 8), float(rank + 1), device=dev)
    for name, fn in (("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(8, 8, device=dev), x)),
                     ("all_gather", lambda: dist.all_gather([torch.empty_like(x) for _ in range(world)], x)),
                     ("all_reduce", lambda: dist.all_reduce(x.clone())),
                     ("broadcast", lambda: dist.broadcast(x.clone(), 0)),
                     ("all_gather_into_tensor_i64", lambda: dist.all_gather_into_tensor(torch.empty(8, 2, dtype=torch.int64, device=dev), torch.ones(4, 2, dtype=torch.int64, device=dev)))):
        try:
            fn(); torch.cuda.synchronize()
            if rank == 0: print(name, "ok", flush=True)
        except Exception as e:
            if rank == 0: print(name, "FAILED", type(e).__name__, str(e)[:150], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2)
