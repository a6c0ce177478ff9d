import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORLD, BATCH = 2, 32


def worker(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    from oracle import steps as osteps
    from disentangle_mlp_amd import trainer as T
    b = osteps.synthetic_batch(BATCH)
    lo, hi = rank * BATCH // WORLD, (rank + 1) * BATCH // WORLD
    tr = T.BetaVAEGANTrainer(beta=25.0, lr=0.0)
    orig_launch = T.FlatGrads._launch
    def _launch(self, bidx):
        print(f"rank {rank}: launch bucket {bidx} pending {self._pending}", flush=True)
        return orig_launch(self, bidx)
    T.FlatGrads._launch = _launch
    got = {}
    tr.step(*(b[k][lo:hi].cuda() for k in ("data", "noise", "eps2", "eps3")),
            grad_hook=lambda ph, net: got.__setitem__(ph, {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}))
    torch.cuda.synchronize()
    # single-process reference of the local gradient (no exchange)
    dist.barrier()
    if rank == 0:
        ref = {}
        for r in range(WORLD):
            eg, d, oeg, od = osteps.build_nets(dtype=torch.float64)
            for o in (oeg, od):
                o.param_groups[0]["lr"] = 0.0
            bb = osteps.synthetic_batch(BATCH, dtype=torch.float64)
            l, h = r * BATCH // WORLD, (r + 1) * BATCH // WORLD
            def hook(ph, net):
                for k, p in net.named_parameters():
                    ref.setdefault(ph, {})
                    ref[ph][k] = ref[ph].get(k, 0) + p.grad.detach().clone()
            osteps.betavaegan_step(eg, d, oeg, od, bb["data"][l:h], bb["noise"][l:h], bb["eps2"][l:h], bb["eps3"][l:h], beta=25.0, bce_divisor=BATCH, grad_hook=hook)
        for ph in ("D", "EG2", "EG3"):
            for k, r_ in ref[ph].items():
                if float(r_.norm()) == 0:
                    continue
                g = got[ph][k]
                ratio = float((g * r_).sum() / (r_ * r_).sum())
                e = float((g - r_).norm() / r_.norm())
                if e > 1e-2:
                    print(f"{ph}/{k}: err {e:.3f} ratio {ratio:.3f} |got| {float(g.norm()):.3e} |ref| {float(r_.norm()):.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, port)) for r in range(WORLD)]
    [p.start() for p in ps]
    [p.join() for p in ps]
