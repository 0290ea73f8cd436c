"""Multi-GPU glue: one process per GPU.

The data path (m-vector and CG-scalar all-reduces) lives in the C library and goes over RCCL
(`Context.comm_init_rccl`); torch.distributed is only the control plane that ships the RCCL
unique id and synchronises processes.  `torch_allreduce_callback` is the alternative transport
(the host supplies the all-reduce): it is what the CPU test-suite uses with gloo, and a fallback
on GPUs via torch's own RCCL process group."""
from __future__ import annotations

import ctypes as C

import numpy as np


def init_rccl_from_torch(ctx, dist):
    """Create the library's RCCL communicator using an initialised torch.distributed group
    (any backend) to broadcast the unique id."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init_rccl(rank, world, box[0])


def torch_allreduce_callback(device_index: int | None = None, group=None):
    """All-reduce callback for Context.comm_init_callback built on torch.distributed.
    device_index None => the buffer is HOST memory (emulator build, gloo); otherwise it is device
    memory on cuda:<device_index> and is reduced through torch's nccl(=RCCL) group ``group`` (default group if None)."""
    import torch
    import torch.distributed as dist

    def fn(ptr: int, count: int, op: int, stream: int) -> int:
        rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
        if device_index is None:
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=rop, group=group)
            return 0

        class _Holder:  # zero-copy view of the library's device buffer
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        t = torch.as_tensor(_Holder(), device=f"cuda:{device_index}")
        ext = torch.cuda.ExternalStream(stream, device=f"cuda:{device_index}") if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=rop, group=group)
        return 0

    return fn


def host_staged_allreduce_callback(device_index: int):
    """FUNCTIONAL-TEST transport: stage the device buffer through the host and reduce it with the
    gloo group.  Lets several ranks share ONE GPU (RCCL refuses duplicate devices), so the whole
    multi-rank flow of bench.py can be exercised on a 1-GPU box.  Not a performance path."""
    import torch
    import torch.distributed as dist

    def fn(ptr: int, count: int, op: int, stream: int) -> int:
        class _Holder:
            __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        ext = torch.cuda.ExternalStream(stream, device=f"cuda:{device_index}") if stream else torch.cuda.current_stream()
        with torch.cuda.stream(ext):
            t = torch.as_tensor(_Holder(), device=f"cuda:{device_index}")
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
            t.copy_(h)
            ext.synchronize()
        return 0

    return fn
