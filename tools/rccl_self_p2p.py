"""One-GPU rehearsal of the torch.distributed (backend nccl = RCCL) exchange used by bench.py / distributed.FrameAssembler:
a world of ONE rank sends a device-resident band to itself and receives it into a row-slice of a device framebuffer with
batch_isend_irecv (a grouped ncclSend/ncclRecv).  Prints OK or the error."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
frame = torch.zeros((8, 16, 3), device=dev)
band = torch.arange(3 * 16 * 3, dtype=torch.float32, device=dev).reshape(3, 16, 3)
ops = [dist.P2POp(dist.irecv, frame[2:5], 0), dist.P2POp(dist.isend, band, 0)]
for req in dist.batch_isend_irecv(ops):
    req.wait()
torch.cuda.synchronize()
ok = bool(torch.equal(frame[2:5], band)) and float(frame[:2].abs().sum()) == 0.0 and float(frame[5:].abs().sum()) == 0.0
print("RCCL self P2P into a framebuffer slice:", "OK" if ok else "MISMATCH", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
