"""The sweep's one collective behind the C ABI (include/cald_hip.h: cald_comm_*, cald_allgather_scores): an RCCL communicator per
process (one process per GPU) and the all-gather of the per-image score rows, without torch.distributed in the data path.

    comm = RcclComm.from_store(rank, world, exchange)        # `exchange(id_bytes | None) -> id_bytes`: any way to ship 128 bytes
    comm = RcclComm.from_torch_group(group=None)             # ... e.g. through an existing torch.distributed group (any backend)
    full_cons, full_cls = comm.allgather_scores(local_pos, cons, cls, pool_size)

Replaces detection/utils.py:75-115 (`all_gather`) / :302-324 (`init_distributed_mode`).  Fails loudly without a GPU or without RCCL."""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .detector import get_ctx


class RcclComm:
    def __init__(self, handle, device, world, rank):
        self.handle, self.device, self.world, self.rank = handle, device, world, rank

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _ffi.check(_ffi.lib().cald_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, id_bytes, world, rank, device=None):
        device = torch.cuda.current_device() if device is None else device
        assert len(id_bytes) == 128
        h = C.c_void_p()
        _ffi.check(_ffi.lib().cald_comm_init_rank(get_ctx(device), C.create_string_buffer(bytes(id_bytes), 128), world, rank, C.byref(h)))
        return cls(h, device, world, rank)

    @classmethod
    def from_store(cls, rank, world, exchange, device=None):
        """exchange(x): rank 0 passes the id bytes in, every rank gets them back (a file, a socket, the launcher's own channel)."""
        return cls.init_rank(exchange(cls.unique_id() if rank == 0 else None), world, rank, device)

    @classmethod
    def from_torch_group(cls, group=None, device=None):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls.init_rank(box[0], world, rank, device)

    def allgather_rows(self, rows):
        """rows: float64 device tensor [n][row_len], the same n on every rank -> [world * n][row_len] on every rank."""
        assert rows.is_cuda and rows.dtype == torch.float64 and rows.is_contiguous() and rows.dim() == 2
        out = torch.empty((self.world * rows.shape[0], rows.shape[1]), dtype=torch.float64, device=rows.device)
        # the collective runs on the CONTEXT's stream (bound when the context was created), which need not be torch's current stream
        # here: finish whatever produced `rows` (and torch.empty's allocation) first -- once per sweep, microseconds
        torch.cuda.current_stream(rows.device).synchronize()
        _ffi.check(_ffi.lib().cald_allgather_scores(self.handle, C.c_void_p(rows.data_ptr()), C.c_void_p(out.data_ptr()), rows.shape[0], rows.shape[1]))
        _ffi.check(_ffi.lib().cald_ctx_sync(get_ctx(self.device)))
        return out

    def allgather_scores(self, local_pos, cons, cls, pool_size):
        """Same contract as sweep.allgather_scores (strided shard, rows padded to ceil(pool / world)), over the C ABI."""
        from .sweep import shard_positions
        world, rank = self.world, self.rank
        Cm1 = cls.shape[1]
        rows = (pool_size + world - 1) // world
        mine = shard_positions(pool_size, rank, world)
        if list(local_pos) != mine:
            raise ValueError("rank %d must hold exactly the strided shard p %% %d == %d of a pool of %d, in order" % (rank, world, rank, pool_size))
        buf = torch.zeros((rows, 1 + Cm1), dtype=torch.float64)
        k = len(mine)
        if k:
            buf[:k, 0] = torch.from_numpy(np.ascontiguousarray(cons, dtype=np.float64))
            buf[:k, 1:] = torch.from_numpy(np.ascontiguousarray(cls, dtype=np.float64))
        out = self.allgather_rows(buf.to(torch.device("cuda", self.device))).cpu().numpy().reshape(world, rows, 1 + Cm1)
        full_cons = np.zeros(pool_size, np.float64)
        full_cls = np.zeros((pool_size, Cm1), np.float64)
        for r in range(world):
            pos = shard_positions(pool_size, r, world)
            full_cons[pos] = out[r, :len(pos), 0]
            full_cls[pos] = out[r, :len(pos), 1:]
        return full_cons, full_cls

    def close(self):
        if self.handle is not None:
            _ffi.lib().cald_comm_destroy(self.handle)
            self.handle = None
