"""Worker for the multi-rank tests (one process per rank, torch.distributed gloo
rendezvous on 127.0.0.1).  Modes:
  cpu  — no GPU: the row partition, the padded all-gather layout and the
         host-staged callbacks, with the aggregation done in numpy on every
         rank's block (checked against the single-process CPU oracle);
  gpu  — every rank drives the HIP path on GPU 0 through HipGCNModel with the
         host-staged transport (D2H -> gloo -> H2D), i.e. everything of the
         N > 1 path except RCCL itself; rank 0 writes its trace to a file;
  rccl — needs one GPU per rank: rank r on GPU r, RCCL over xGMI (the product's transport):
         gcnhost_rccl_selftest_world first, then the same model run as `gpu`.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_callbacks(dist, world):
    import torch

    def allgather(user, ptr, block):
        arr = np.ctypeslib.as_array(ptr, (block * world,))
        t = torch.from_numpy(arr)
        mine = t[dist.get_rank() * block:(dist.get_rank() + 1) * block].clone()
        dist.all_gather(list(t.view(world, block).unbind(0)), mine)

    def allreduce(user, ptr, n):
        arr = np.ctypeslib.as_array(ptr, (n,))
        dist.all_reduce(torch.from_numpy(arr))
    return allgather, allreduce


def main():
    mode, name, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuda_gcn_amd import datagen, model
    ds = datagen.make_dataset(name)
    gp, gi, N = ds["g_indptr"], ds["g_indices"], ds["num_nodes"]
    if mode == "cpu_halo":
        # the HALO plan with a real point-to-point exchange (gloo isend/irecv): pack the rows each peer needs, receive
        # the peers' pieces into the table segments, aggregate through the table, compare with the oracle's rows
        from oracle.pyoracle import Oracle
        p = model.exchange_plan(gp, gi, world, rank, 2)
        assert p["halo"]
        start, _ = model.partition(gp, world)
        r0, r1 = int(start[rank]), int(start[rank + 1])
        dim = 12
        x = np.random.default_rng(0).standard_normal((N, dim)).astype(np.float32)     # same on every rank
        table = np.full((p["table_rows"], dim), np.nan, np.float32)
        table[:r1 - r0] = x[r0:r1]
        reqs, bufs = [], {}
        for q in range(world):
            if q == rank:
                continue
            n_recv = int(p["recv_off"][q + 1] - p["recv_off"][q])
            if n_recv:
                bufs[q] = torch.empty((n_recv, dim), dtype=torch.float32)
                reqs.append(dist.irecv(bufs[q], src=q))
            rows = p["send_rows"][p["send_off"][q]:p["send_off"][q + 1]]
            if rows.size:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(table[rows])), dst=q))
        for r in reqs:
            r.wait()
        for q, b in bufs.items():
            table[p["n_local"] + p["recv_off"][q]:p["n_local"] + p["recv_off"][q + 1]] = b.numpy()
        assert not np.isnan(table).any()
        assert np.array_equal(table, x[p["table_global"]])                             # every segment holds the rows the plan names
        ip, ix, cd = p["indptr"], p["indices"], p["col_deg"]
        deg = np.diff(ip).astype(np.int64)
        src = np.repeat(np.arange(r1 - r0), deg)
        coef = (1.0 / np.sqrt((deg[src] * cd[ix].astype(np.int64)).astype(np.float32)).astype(np.float64)).astype(np.float32)
        local = np.zeros((r1 - r0, dim), np.float64)
        np.add.at(local, src, coef[:, None].astype(np.float64) * table[ix].astype(np.float64))
        want = Oracle().graphsum(gp, gi, x, dim)[r0:r1]
        assert np.allclose(local, want, rtol=1e-5, atol=1e-5), np.abs(local - want).max()
        if rank == 0:
            np.save(out, np.array([float(p["table_rows"])]))
    elif mode == "cpu":
        import ctypes as C
        import scipy.sparse as sp
        from oracle.pyoracle import Oracle
        start, rows_max = model.partition(gp, world)
        ip, ix, cd, n_cols = model.local_graph(gp, gi, world, rank)
        r0, r1 = int(start[rank]), int(start[rank + 1])
        dim = 12
        x = np.random.default_rng(0).standard_normal((N, dim)).astype(np.float32)     # same on every rank
        # in-place all-gather of this rank's block through the host callbacks
        full = np.zeros((world * rows_max, dim), np.float32)
        full[rank * rows_max:rank * rows_max + (r1 - r0)] = x[r0:r1]
        ag, ar = make_callbacks(dist, world)
        ag(None, full.ctypes.data_as(C.POINTER(C.c_float)), rows_max * dim)
        for q in range(world):                                                        # every block landed where padded() says
            a, b = int(start[q]), int(start[q + 1])
            assert np.array_equal(full[q * rows_max:q * rows_max + (b - a)], x[a:b])
        deg = np.diff(ip).astype(np.int64)
        src = np.repeat(np.arange(r1 - r0), deg)
        coef = (1.0 / np.sqrt((deg[src] * cd[ix].astype(np.int64)).astype(np.float32)).astype(np.float64)).astype(np.float32)
        A = sp.csr_matrix((coef, ix, ip), shape=(r1 - r0, n_cols))
        local = A @ full
        want = Oracle().graphsum(gp, gi, x, dim)[r0:r1]
        assert np.allclose(local, want, rtol=1e-5, atol=1e-5), np.abs(local - want).max()
        # the cut HIPGCN_OVERLAP_EXCHANGE makes: edges into this rank's own rows need nothing from the exchange, the rest does;
        # the two parts add up to the whole
        own = (ix >= rank * rows_max) & (ix < rank * rows_max + (r1 - r0))
        A_own = sp.csr_matrix((np.where(own, coef, 0).astype(np.float32), ix, ip), shape=(r1 - r0, n_cols))
        A_rest = sp.csr_matrix((np.where(own, 0, coef).astype(np.float32), ix, ip), shape=(r1 - r0, n_cols))
        before_exchange = np.zeros_like(full)
        before_exchange[rank * rows_max:rank * rows_max + (r1 - r0)] = x[r0:r1]
        assert np.array_equal(A_own @ before_exchange, A_own @ full)              # the first part reads own rows only
        assert np.allclose(A_own @ full + A_rest @ full, want, rtol=1e-5, atol=1e-5)
        assert 0 < own.sum() < own.size or world == 1
        v = np.array([rank + 1.0, 10.0 * (rank + 1)], np.float64)
        ar(None, v.ctypes.data_as(C.POINTER(C.c_double)), 2)
        tot = world * (world + 1) / 2
        assert np.array_equal(v, [tot, 10 * tot])
        if rank == 0:
            np.save(out, np.array([1.0]))
    else:
        ag, ar, nccl_id, device = None, None, None, 0
        if mode == "rccl":
            device = int(os.environ.get("LOCAL_RANK", rank))
            box = [model.nccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            nccl_id = box[0]
            from cuda_gcn_amd import _lib
            rc = _lib.gcnhost().gcnhost_rccl_selftest_world(device, rank, world, nccl_id)
            assert rc == 0, (rc, _lib.gcnhost().gcnhost_last_error())
            box = [model.nccl_unique_id() if rank == 0 else None]      # a fresh id for the model's communicator
            dist.broadcast_object_list(box, src=0)
            nccl_id = box[0]
        else:
            ag, ar = make_callbacks(dist, world)
        epochs = int(sys.argv[4])
        flags = int(sys.argv[5])
        dropout = float(sys.argv[6])
        m = model.HipGCNModel(ds, seed=4, device=device, flags=flags, rank=rank, world=world, nccl_id=nccl_id, host_allgather=ag, host_allreduce=ar,
                              hidden_dim=int(os.environ.get("MR_HIDDEN", "16")), dropout=dropout, epochs=epochs)
        info = m.info()
        assert info["world"] == world and info["rank"] == rank
        if os.environ.get("MR_ASYNC") == "1":
            tr = m.run_epochs(epochs)                 # training lane + overlapped validation lane
        else:
            tr = np.array([m.train_epoch() + m.eval(2) for _ in range(epochs)], np.float32)
        test = m.eval(3)
        w1 = m.var(2)
        h1 = m.var(3)                                   # this rank's rows of H1
        m.close()
        gathered = [None] * world
        dist.all_gather_object(gathered, (info["row_start"], h1))
        if rank == 0:
            H1 = np.concatenate([g[1] for g in sorted(gathered, key=lambda t: t[0])], axis=0)
            np.savez(out, trace=tr, test=np.array(test, np.float32), w1=w1, h1=H1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
