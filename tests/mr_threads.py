"""N logical ranks of the row-partitioned HIP path as THREADS of one process sharing GPU 0.

The GPU boxes of this pool admit few processes on the card, so 8 ranks cannot be 8 processes there.  The
host-staged transport of HipGCN (host/comm.cpp, HostComm) only needs two callbacks — an in-place all-gather of
float blocks and an all-reduce of doubles; here they are barriers over shared numpy buffers.  ctypes releases
the GIL around every library call and the callbacks re-take it, so the ranks really run concurrently on their
own HIP streams.  Everything of the N > 1 path except RCCL itself is exercised: partition, exchange plan,
table layout, the cut operators of HIPGCN_OVERLAP_EXCHANGE, the fused all-reduce.
"""
import ctypes as C
import threading

import numpy as np


class ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def callbacks(self, rank):
        world, slots, barrier = self.world, self.slots, self.barrier

        def allgather(user, ptr, block):
            arr = np.ctypeslib.as_array(ptr, (block * world,))
            slots[rank] = arr
            barrier.wait()
            for q in range(world):
                if q != rank:
                    arr[q * block:(q + 1) * block] = slots[q][q * block:(q + 1) * block]
            barrier.wait()

        def allreduce(user, ptr, n):
            arr = np.ctypeslib.as_array(ptr, (n,))
            slots[rank] = arr.copy()
            barrier.wait()
            total = slots[0].copy()
            for q in range(1, world):           # rank order: every rank forms the same sum
                total += slots[q]
            barrier.wait()
            arr[:] = total
        return allgather, allreduce


def run_ranks(ds, world, flags, epochs, hidden, dropout, seed=4, run_async=False, env_flags=None):
    """returns dict(trace [epochs x 4], test (loss, acc), w1, h1 [N x hidden] in global row order, exchange of rank 0)"""
    from cuda_gcn_amd.model import HipGCNModel
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def body(rank):
        try:
            ag, ar = tw.callbacks(rank)
            m = HipGCNModel(ds, seed=seed, device=0, flags=flags, rank=rank, world=world, host_allgather=ag, host_allreduce=ar,
                            hidden_dim=hidden, dropout=dropout, epochs=epochs)
            info = m.info()
            if run_async:
                tr = m.run_epochs(epochs)
            else:
                tr = np.array([m.train_epoch() + m.eval(2) for _ in range(epochs)], np.float32)
            test = m.eval(3)
            ids, renumbered = m.row_ids()
            results[rank] = dict(trace=tr, test=np.array(test, np.float32), w1=m.var(2), h1=m.var(3), row_start=info["row_start"],
                                 exchange=m.exchange(), ids=ids, renumbered=renumbered)
            m.close()
        except BaseException as e:      # a failed rank must not leave the others at a barrier forever
            errors.append((rank, e))
            tw.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
        raise RuntimeError(f"rank {real[0][0]} failed: {real[0][1]!r}")
    h1 = np.zeros((ds["num_nodes"], results[0]["h1"].shape[1]), np.float32)
    for r in range(world):                                  # rows back in the caller's node numbering
        h1[results[r]["ids"]] = results[r]["h1"]
    return dict(trace=results[0]["trace"], test=results[0]["test"], w1=results[0]["w1"], h1=h1, exchange=results[0]["exchange"],
                renumbered=results[0]["renumbered"], traces=[results[r]["trace"] for r in range(world)])
