"""does torch wrap a raw device pointer through __cuda_array_interface__ on this ROCm build? (parallel.witness_all_gather relies on it)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fawkes_crypto_amd as fk
from fawkes_crypto_amd import parallel
ctx = fk.Context(0)
dptr, st = ctx.witness_slot(0, 4096)
z = np.arange(1024, dtype=np.uint32)
ctx.witness_upload_part_async(0, z, 0)
ctx.witness_mark_ready(0)
ctx.witness_ptr(0); ctx.sync()
t = torch.as_tensor(parallel._DevBytes(dptr, 4096), device=torch.device('cuda', 0))
torch.cuda.synchronize()
print('as_tensor ok', t.dtype, t.shape, t.data_ptr() == dptr, bool((t.cpu().numpy().view(np.uint32) == z).all()))
s = torch.cuda.ExternalStream(st, device=torch.device('cuda', 0))
with torch.cuda.stream(s):
    t[:8] = 7
s.synchronize()
print('write through view ok', ctx.download(dptr, 16, np.uint8).tolist())
