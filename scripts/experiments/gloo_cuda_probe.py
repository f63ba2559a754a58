"""Can two ranks that SHARE one GPU exchange device tensors over gloo (a rehearsal transport for the N = 2 bench path on a
one-GPU box; RCCL refuses two ranks on one device)?  torchrun --nproc-per-node 2 scripts/experiments/gloo_cuda_probe.py"""
import os
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
x = torch.arange(8, device="cuda", dtype=torch.int64) + 100 * rank
out = torch.empty_like(x)
for name, fn in (("all_to_all_single", lambda: dist.all_to_all_single(out, x)),
                 ("all_reduce", lambda: dist.all_reduce(x.clone())),
                 ("all_gather", lambda: dist.all_gather([torch.empty_like(x) for _ in range(world)], x)),
                 ("batch_isend_irecv", lambda: [r.wait() for r in dist.batch_isend_irecv(
                     [dist.P2POp(dist.isend, x, (rank + 1) % world), dist.P2POp(dist.irecv, out, (rank - 1) % world)])])):
    try:
        fn()
        torch.cuda.synchronize()
        print(rank, name, "ok", out.tolist()[:3], flush=True)
    except Exception as e:
        print(rank, name, "FAILED", repr(e)[:200], flush=True)
dist.destroy_process_group()
