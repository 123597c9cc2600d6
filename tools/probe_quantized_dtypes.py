import torch
for dt in (torch.quint8, torch.quint4x2, torch.quint2x4):
    try:
        t = torch.empty((3,5), dtype=dt, device='cuda')
        print(dt, "empty OK", t.shape, t.is_quantized, t.untyped_storage().nbytes())
    except Exception as e:
        print(dt, "empty FAIL", repr(e)[:200])
    try:
        t = torch._empty_affine_quantized((3,5), scale=0.1, zero_point=3, dtype=dt, device='cuda')
        print(dt, "_empty_affine_quantized OK", t.shape, t.untyped_storage().nbytes(), t.data_ptr() % 256)
        try:
            print("  int_repr", t.int_repr().dtype, t.int_repr().shape)
        except Exception as e: print("  int_repr FAIL", repr(e)[:120])
        try:
            d = t.dequantize(); print("  dequantize ok", d.dtype, d.device)
        except Exception as e: print("  dequantize FAIL", repr(e)[:120])
    except Exception as e:
        print(dt, "_empty_affine_quantized FAIL", repr(e)[:200])
x = torch.randn(10, device='cuda')
try:
    q = torch.quantize_per_tensor(x, 0.1, 3, torch.quint8); print("quantize_per_tensor cuda OK", q.int_repr()[:4])
except Exception as e: print("quantize_per_tensor cuda FAIL", repr(e)[:200])
