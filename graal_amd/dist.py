"""Multi-GPU glue: one process per GPU, the COO contact list sharded by rank, ONE exchange of the per-candidate
Q vector (13*K int64 = 520 bytes) per MCMC step -- either an RCCL all-reduce of a device buffer, or (ranks of one node,
the default there) through pinned host memory the ranks share: each rank's GPU publishes its sums to its slot, each host
adds up the slots (``Group.shared_host_segment``, ``graal_attach_exchange`` in include/graal_hip.h).  The reference has no distributed path; its only hint is
the comment "place where we want to spread the workload accross the network!" at ``cuda_lib_gl.py:1886``.

Because every log-likelihood contribution is an int64 fixed-point number, the all-reduced vector is bit-identical
for any number of ranks: every rank then draws the same move from its own (identically seeded) RandomState and no
broadcast of the decision is needed.
"""
import os

import numpy as np


def env_world():
    """(rank, world, local_rank) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(nnz, rank, world):
    """Contiguous, equal-nnz slice [lo, hi) of the (row, col)-sorted contact list owned by `rank`."""
    nnz, rank, world = int(nnz), int(rank), int(world)
    assert world >= 1 and 0 <= rank < world
    return (nnz * rank) // world, (nnz * (rank + 1)) // world


def shard_take(nnz, rank, world, block=4096):
    """Indices of the (row, col)-sorted contact list owned by `rank`: blocks of `block` consecutive contacts dealt round robin.
    (A contiguous slice per rank -- shard_range -- is balanced in bytes but not in work: the contacts a step has to PRICE are
    those of the one or two affected contigs, a contiguous row range, i.e. one rank's slice.)  The order inside a shard is
    still (row, col), which is all the streaming pass wants."""
    nnz, rank, world, block = int(nnz), int(rank), int(world), int(block)
    assert world >= 1 and 0 <= rank < world and block >= 1
    if world == 1:
        return slice(0, nnz)
    b = np.arange(rank, (nnz + block - 1) // block, world, dtype=np.int64)
    idx = (b[:, None] * block + np.arange(block, dtype=np.int64)[None, :]).ravel()
    return idx[idx < nnz]


class Group:
    """Thin wrapper over torch.distributed (RCCL for CUDA tensors, gloo for CPU tensors); world == 1 needs no
    process group at all."""

    def __init__(self, rank=0, world=1):
        self.rank, self.world = int(rank), int(world)
        if self.world > 1:
            import torch.distributed as td
            if not td.is_initialized():
                raise RuntimeError("torch.distributed must be initialised before building a multi-rank sampler")
            assert td.get_world_size() == self.world and td.get_rank() == self.rank

    def all_reduce_sum_(self, tensor):
        """In-place SUM all-reduce of an int64 tensor (device tensor -> RCCL over xGMI)."""
        if self.world > 1:
            import torch.distributed as td
            td.all_reduce(tensor, op=td.ReduceOp.SUM)
        return tensor

    def all_reduce_sum_int(self, value):
        """All-reduce one Python int (used for the contact part of the full likelihood)."""
        if self.world == 1:
            return int(value)
        import torch
        import torch.distributed as td
        dev = "cuda" if td.get_backend() == "nccl" else "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return int(t.cpu()[0])

    def all_reduce_max_int(self, value):
        if self.world == 1:
            return int(value)
        import torch
        import torch.distributed as td
        dev = "cuda" if td.get_backend() == "nccl" else "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return int(t.cpu()[0])

    def barrier(self):
        if self.world > 1:
            import torch.distributed as td
            td.barrier()

    def single_node(self):
        """True if every rank of the group runs on this host (then the ranks can share host memory)."""
        if self.world == 1:
            return True
        import socket
        import torch.distributed as td
        names = [None] * self.world
        td.all_gather_object(names, (socket.gethostname(), os.path.exists("/dev/shm")))
        return all(n == names[0] and n[1] for n in names)

    def shared_host_segment(self, nbytes):
        """A zero-filled, page-aligned host memory segment mapped by every rank of the (single-node) group: rank 0
        creates a /dev/shm file, the others open it, and the name is removed once all of them have it mapped.  Setup
        traffic (the file name, two barriers) goes through torch.distributed; the per-step exchange does not."""
        import mmap
        import secrets
        import torch.distributed as td
        nbytes = (int(nbytes) + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
        name = ["/dev/shm/graal_x_%d_%s" % (os.getpid(), secrets.token_hex(6))] if self.rank == 0 else [None]
        td.broadcast_object_list(name, src=0)
        path = name[0]
        if self.rank == 0:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, nbytes)
        td.barrier()
        try:
            if self.rank != 0:
                fd = os.open(path, os.O_RDWR)
            m = mmap.mmap(fd, nbytes, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            os.close(fd)
        finally:
            td.barrier()
            if self.rank == 0:
                os.unlink(path)
        return m
