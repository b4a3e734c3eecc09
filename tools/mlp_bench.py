"""pgd_mlp_policy alone: mean launch time over back-to-back launches (HIP events around 200 launches), against the torch ops of the same
network; rows = N envs.  usage: mlp_bench.py [N]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch  # noqa: E402
from pgdrive_amd import _abi, bank, mapdata, scenario  # noqa: E402
from pgdrive_amd.engine import Engine  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
descs = bank.get_descriptions(range(1000, 1008))
mb = mapdata.MapBank(descs)
sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=16)
eng = Engine(_abi.make_config(N, seed=1), mb, sb)
eng.reset(np.arange(N) % 8)
torch.manual_seed(0)
lin = [torch.nn.Linear(274, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 2)]
W = [l.weight.detach().t().contiguous().cuda() for l in lin]
B = [l.bias.detach().contiguous().cuda() for l in lin]
w = (W[0], B[0], W[1], B[1], W[2], B[2])
act = torch.zeros((N, 1, 2), device="cuda")
obs2d = eng.obs.view(N, -1)


def torch_policy():
    h = torch.tanh(torch.addmm(B[0], obs2d, W[0]))
    h = torch.tanh(torch.addmm(B[1], h, W[1]))
    torch.tanh(torch.addmm(B[2], h, W[2]), out=act.view(N, 2))


prep = eng.mlp_prepare(w)
eng.sync()
for name, fn in (("pgd_mlp_policy", lambda: eng.mlp_policy(w, act, final_tanh=True)),
                 ("pgd_mlp_policy_prepared (bf16 x 3)", lambda: eng.mlp_policy(None, act, final_tanh=True, prepared=prep)),
                 ("torch 3 x addmm + 3 x tanh", torch_policy)):
    with torch.no_grad(), torch.cuda.stream(eng.stream):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, stream=eng.stream):
        for _ in range(50):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 400
    flops = 2.0 * N * (274 * 256 + 256 * 256 + 256 * 2)
    print("%-36s %d rows: %.2f us per call back to back (%.1f TFLOP/s fp32; f32 MFMA peak 157)" % (name, N, us, flops / us / 1e6))
eng.close()
