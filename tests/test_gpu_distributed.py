"""GPU: piquant.distributed on a real device -- the HIP scan feeding the RCCL all-reduce (single rank here:
the box has one GPU; world_size 2/3 is covered on CPU over gloo in test_distributed_cpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_sharded_params_through_rccl(pg, oracle_mod):
    import piquant.distributed as D

    O = oracle_mod
    x = np.random.default_rng(5).normal(size=3_000_001).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for tdt, odt in ((torch.quint8, O.UINT8), (torch.quint4x2, O.UINT4), (torch.quint2x4, O.UINT2)):
        assert D.compute_quant_params(xd, dtype=tdt) == O.compute_quant_params(x, O.F32, odt)
    xb = O.f32_to_bf16(x)
    xbd = torch.from_numpy(xb.view(np.int16)).cuda().view(torch.bfloat16)
    assert D.compute_quant_params(xbd, dtype=torch.quint8) == O.compute_quant_params(xb, O.BF16, O.UINT8)


def test_keys_fold_like_an_all_reduce(pg, oracle_mod):
    """MIN over the key pairs of several shards == keys of the whole tensor: what all_reduce(MIN) computes."""
    import piquant
    import piquant.distributed as D

    O = oracle_mod
    x = np.random.default_rng(6).uniform(-3, 5, 1_000_000).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    world = 8
    parts = []
    for r in range(world):
        b, e = D.shard_range(x.size, r, world, 4)
        parts.append(D.local_minmax_keys(xd[b:e]))
    folded = torch.stack(parts).min(dim=0).values.cpu()
    whole = D.local_minmax_keys(xd).cpu()
    assert torch.equal(folded, whole)
    assert piquant.decode_minmax_keys(int(folded[0]), int(folded[1])) == O.minmax(x, O.F32)
