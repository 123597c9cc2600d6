import os, sys, time, json
sys.path.insert(0, '/root/repo/pi-quant_amd')
import torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
import piquant.distributed as D
x = torch.empty(100_000, device='cuda').uniform_(-1, 1)
out = {}
for name in ('p2p', 'collective'):
    for _ in range(10):
        D.compute_quant_params(x, dtype=torch.quint8, transport=name)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(200):
        D.compute_quant_params(x, dtype=torch.quint8, transport=name)
    out[name + '_us_per_call'] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
D.release_peer_meshes()
if rank == 0: print(json.dumps(out))
dist.destroy_process_group()
