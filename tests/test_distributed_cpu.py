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
    if rank == 0:
        q.put(outs)
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
        res[world] = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for (o1, r1, d1), (o2, r2, d2) in zip(res[1], res[2]):
        assert o1.shape == o2.shape == (n_total, 1, 8 + 10 + 16 + 24)
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
