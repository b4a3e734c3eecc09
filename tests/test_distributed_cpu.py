"""N>1 path on CPU: world_size-2 gloo processes run the per-step gather (pgdrive_amd/dist.py) on oracle-produced shards
and must reproduce the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q, transport="collective"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import orc
    from pgdrive_amd import _abi, bank, mapdata, scenario
    from pgdrive_amd import dist as pdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    descs = bank.load_descriptions()[:4]
    mb = mapdata.MapBank(descs)
    sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_traffic=8)
    lo, hi = pdist.shard_range(n_total, rank, world)
    n = hi - lo
    cfg = _abi.make_config(n, num_traffic=8, num_lasers=24)
    o = orc.Oracle(cfg, mb, sb)  # the oracle stands in for the engine: this test is about sharding + the collective
    o.reset(pdist.scenario_ids_for(lo, hi, 4))
    D = _abi.obs_dim(cfg)
    g = pdist.StepGather(torch, dist, n, D, transport=transport)  # the class bench.py and the GPU world-size-2 test drive with the real engine
    rng = np.random.default_rng(0)
    outs = []
    for t in range(5):
        act = rng.uniform(-1, 1, size=(n_total, 1, 2)).astype(np.float32)[lo:hi]

        def produce(rows):  # what pgd_step_packed does on the device
            obs, rew, done, flags = o.step(act)
            pdist.pack(torch, torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(rew.astype(np.float32)),
                       torch.from_numpy(done), out=rows)
        b = g.step(produce)
        go, gr, gd = g.result(b)
        outs.append((go.numpy().copy(), gr.numpy().copy(), gd.numpy().copy()))
    # the exchange's self-check (bench.py prints it as `gather_ok`): clean rows pass, one damaged value among the rows that ARRIVED
    # from the last rank fails -- on every rank, since the verdict is all-reduced
    act = rng.uniform(-1, 1, size=(n_total, 1, 2)).astype(np.float32)[lo:hi]

    def produce(rows):
        obs, rew, done, flags = o.step(act)
        pdist.pack(torch, torch.from_numpy(obs.astype(np.float32)), torch.from_numpy(rew.astype(np.float32)),
                   torch.from_numpy(done), out=rows)
    ok_clean, detail = g.validate(produce)

    def damage(buf):
        if buf.shape[0] == n_total:  # a rank that holds the gathered rows
            buf[n_total - 1, 5] += 0.25
    ok_bad, detail_bad = g.validate(produce, corrupt=damage)

    def swap(buf):  # two rows of the last rank's slice exchanged: the plain sum does not see it, the weighted one does
        if buf.shape[0] == n_total:
            tmp = buf[n_total - 1].clone()
            buf[n_total - 1] = buf[n_total - 2]
            buf[n_total - 2] = tmp
    ok_swap, _ = g.validate(produce, corrupt=swap)
    if rank == 0:
        q.put((outs, (ok_clean, ok_bad, ok_swap, detail, detail_bad)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("transport", ["root", "collective"])
def test_two_rank_gather_matches_single_process(transport):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    n_total = 8
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = (29611 if transport == "collective" else 29631) + world
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q, transport)) for r in range(world)]
        for p in procs:
            p.start()
        res[world], checks = q.get(timeout=240)
        ok_clean, ok_bad, ok_swap, detail, detail_bad = checks
        assert ok_clean and not detail["mismatched_ranks"], detail
        assert not ok_bad and not ok_swap
        if world == 2:
            assert detail_bad["mismatched_ranks"] == [1] and detail_bad["ranks_checked"] == 2
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for (o1, r1, d1), (o2, r2, d2) in zip(res[1], res[2]):
        assert o1.shape == o2.shape == (n_total, 1, 8 + 10 + 16 + 24)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)


def _fake_world8(rank, q):
    """One rank of a world-8 'fake' process group (torch.testing._internal.distributed.fake_pg: collectives are no-ops, the
    Python argument checks of torch.distributed run): the StepGather of BASELINE C4's shape on that rank."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from torch.testing._internal.distributed.fake_pg import FakeStore
    from pgdrive_amd import dist as pdist
    world, n_local, D, A = 8, 4096, 274, 1
    dist.init_process_group(backend="fake", rank=rank, world_size=world, store=FakeStore())

    class AsNccl:  # the same group, reporting the backend name RCCL has: StepGather then lays its buffers out as for RCCL
        def __getattr__(self, k):
            return getattr(dist, k)

        def get_backend(self):
            return "nccl"

    out = {}
    for transport in ("root", "collective"):
        g = pdist.StepGather(torch, AsNccl(), n_local, D, A, transport=transport)
        W = pdist.pack_width(D, A)
        assert g.world == world and g.rank == rank and g.W == W == 276
        for b in range(g.nbuf):
            send, recv = g.send[b], g.recv[b]
            assert send.shape == (n_local, W) and send.is_contiguous() and send.dtype == torch.float32
            if transport == "collective":
                # RCCL / NCCL all-gather in place: the input must be exactly this rank's slice of the output
                assert g.inplace and recv.shape == (world * n_local, W) and recv.is_contiguous()
                assert send.data_ptr() == recv.data_ptr() + rank * n_local * W * 4
                assert send.untyped_storage().data_ptr() == recv.untyped_storage().data_ptr()
            else:
                if rank == 0:
                    # gather to the root: the receive list is world contiguous views of ONE buffer, in rank order, and the
                    # root's own rows come from a separate send buffer (send and receive memory must not alias)
                    assert recv.shape == (world * n_local, W) and len(g.parts[b]) == world
                    for q_, part in enumerate(g.parts[b]):
                        assert part.is_contiguous() and part.shape == (n_local, W)
                        assert part.data_ptr() == recv.data_ptr() + q_ * n_local * W * 4
                    assert send.untyped_storage().data_ptr() != recv.untyped_storage().data_ptr()
                else:
                    assert g.parts[b] is None and recv.shape == (n_local, W) and send.data_ptr() == recv.data_ptr()
        calls = []
        for t in range(5):  # double-buffered: the exchange of step t is waited for when its buffer comes round again
            b = g.step(lambda rows: (calls.append(rows.data_ptr()), rows.fill_(float(t)))[0])
            assert b == t % g.nbuf and calls[-1] == g.send[b].data_ptr()
        g.drain()
        obs, rew, done = g.result((5 - 1) % g.nbuf)
        n_rows = world * n_local if (transport == "collective" or rank == 0) else n_local
        assert obs.shape == (n_rows, A, D) and rew.shape == (n_rows, A) and done.shape == (n_rows, A)
        out[transport] = g.describe()
        g.close()
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_step_gather_world8_buffer_layout_under_a_fake_process_group():
    """RCCL readiness without hardware (the RCCL backend itself has never run: one GPU per box here).  For the shapes of BASELINE
    C4 (8 ranks x 4096 envs x 276 floats) on the root and on a middle rank: the buffers StepGather hands to
    `dist.gather(gather_list=...)` and to the in-place `dist.all_gather_into_tensor` obey the aliasing rules RCCL requires, the
    double buffering addresses the right slices, and torch.distributed's own argument checks accept every call."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fake_world8, args=(r, q)) for r in (0, 3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "RCCL gather" in got[0]["root"] and "RCCL all_gather_into_tensor" in got[3]["collective"] and "in place" in got[3]["collective"]
