"""diagnostic (torch ops only, none of this repository's kernels): two processes on one GPU, each with its own constant in buffers that
very likely sit at the same VIRTUAL addresses; each keeps re-reading its buffers with small kernels in lock-step with the other and
counts values that are not its own constant."""
import os, sys, time
import torch
import torch.multiprocessing as mp


def worker(rank, n, go, iters, big):
    torch.cuda.set_device(0)
    c = float(rank + 1)
    bufs = [torch.full((64, 48, 64), c, device="cuda") for _ in range(8)]
    scratch = torch.empty(big, device="cuda")                 # something large in flight, like the pyramid build
    torch.cuda.synchronize()
    print("rank %d: buffers at %s" % (rank, [hex(b.data_ptr()) for b in bufs[:3]]), flush=True)
    while time.time() < go:
        pass
    bad_total = 0
    for it in range(iters):
        scratch.fill_(float(it))                              # long kernel: 1-2 ms
        outs = [b * 1.0 for b in bufs]                        # eight small kernels reading the constants right behind it
        nb = sum(int((o != c).sum()) for o in outs)
        if nb:
            bad_total += nb
            o = torch.cat([o.flatten() for o in outs])
            vals = o[o != c]
            print("rank %d it %d: %d foreign values, e.g. %s" % (rank, it, nb, vals[:8].tolist()), flush=True)
    print("rank %d: %d foreign values in %d iterations" % (rank, bad_total, iters), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]); iters = int(sys.argv[2]); big = int(sys.argv[3])
    go = time.time() + 15
    mp.spawn(worker, args=(n, go, iters, big), nprocs=n, join=True)
