import sys, time
sys.path.insert(0, 'pi-quant_amd')
import torch, piquant
x = torch.rand(1_000_000, device='cuda')
scale, zp = piquant.torch.compute_quant_params(x, dtype=torch.quint8)
def t(f, n=20000):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
out8 = torch.empty(x.shape, dtype=torch.uint8, device='cuda')
outq = torch.empty(x.shape, dtype=torch.quint8, device='cuda')
ctx = piquant.Context.get(0)
from piquant import DataType, RoundMode
print("native front end: %s" % (piquant.torch._native is not None))
print("full quantize(dtype=quint8)       %.2f us" % t(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8)))
print("full quantize(dtype=uint8)        %.2f us" % t(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.uint8)))
print("quantize(out=preallocated quint8) %.2f us" % t(lambda: piquant.torch.quantize(x, scale=scale, zero_point=zp, dtype=torch.quint8, out=outq)))
print("torch.empty quint8                %.2f us" % t(lambda: torch.empty(x.shape, dtype=torch.quint8, device=x.device)))
print("torch.empty uint8                 %.2f us" % t(lambda: torch.empty(x.shape, dtype=torch.uint8, device=x.device)))
print("_ctx_for                          %.2f us" % t(lambda: piquant.torch._ctx_for(x, None)))
print("current_stream.cuda_stream        %.2f us" % t(lambda: torch.cuda.current_stream(0).cuda_stream))
pi, po, n = x.data_ptr(), out8.data_ptr(), x.numel()
print("ctx.quantize_ptr (python+ctypes)  %.2f us" % t(lambda: ctx.quantize_ptr(pi, DataType.F32, po, DataType.UINT8, n, scale, zp, RoundMode.NEAREST, _device_ptrs=True)))
from piquant._bootstrap import C_LIB as C
print("raw ctypes C.piquant_quantize     %.2f us" % t(lambda: C.piquant_quantize(ctx._ctx, pi, 0, po, 4, n, scale, zp, 0)))
print("torch.quantize_per_tensor         %.2f us" % t(lambda: torch.quantize_per_tensor(x, scale, zp, torch.quint8)))
print("x.is_contiguous + data_ptr        %.2f us" % t(lambda: (x.is_contiguous(), x.data_ptr(), x.numel())))
