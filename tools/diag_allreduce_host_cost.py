#!/usr/bin/env python3
"""HOST time of the three calls of the mesh all-reduce schedule (encode_batch for 7 peers, reduce_encode with 7 terms, decode_batch of 8 chunks) as
piquant.distributed issues them, on chunks so small that the GPU never is the bottleneck: what the schedule costs the launching thread.  Round 5: 220 us
through the tensor-level wrappers of piquant.torch (two slices and a dozen attribute checks per buffer), 67 us through the context's raw-pointer entry
points -- the kernels of an 8-way all-reduce of 109 MB take 72 us."""
import sys, time
sys.path.insert(0, "pi-quant_amd")
import torch, piquant, piquant.distributed as D
ops = D._DeviceOps(piquant.Context())
n, W = 40000, 8
x = torch.empty(n * W, device="cuda").uniform_(-1, 1)
slot = 16 + n
bufs = torch.zeros(W * slot + 64, dtype=torch.uint8, device="cuda")
mine = torch.zeros(slot, dtype=torch.uint8, device="cuda")
chunks = [(i * n, (i + 1) * n) for i in range(W)]
for j, (b, e) in enumerate(chunks):
    ops.encode(x[b:e], bufs[j * slot: j * slot + 16 + n], torch.uint8, "nearest")
def direct():
    peers = list(range(1, W))
    ops.encode_batch([x[chunks[j][0]:chunks[j][1]] for j in peers], [bufs[j * slot: j * slot + 16 + n] for j in peers], torch.uint8, "nearest")
    ops.reduce_encode([bufs[i * slot: i * slot + 16 + n] for i in range(1, W)], x[chunks[0][0]:chunks[0][1]], mine[: 16 + n], torch.uint8, "nearest")
    ops.decode_batch([bufs[j * slot: j * slot + 16 + n] for j in range(W)], [x[chunks[j][0]:chunks[j][1]] for j in range(W)], torch.uint8, "set")
for _ in range(50): direct()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): direct()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host us per direct schedule (3 calls, 8 ranks, tiny chunks): {(t1 - t0) / 500 * 1e6:.1f}; incl. drain {(t2 - t0) / 500 * 1e6:.1f}")
