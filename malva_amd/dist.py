"""Multi-GPU glue for the scan: one process per GPU (torch.distributed, backend
"nccl" = RCCL on ROCm; "gloo" on CPU for tests).

The KMC table shards by rows; every rank holds the whole (read-only) index and
accumulates into its own counters.  The counters are two commutative wrapping
sums (u16 cells kept as u32, and u32 map values -- SURVEY Appendix A.2), so one
sum all-reduce over the concatenated vector is the only exchange step.  Variant
blocks are independent, so the genotyping step shards by variants with no
collective.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n, rank, world):
    """contiguous [lo, hi) of n items for `rank`; sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counters_(t):
    """In-place sum all-reduce of a counter vector held as int32: two's-complement
    addition wraps exactly like the reference's u32 arithmetic (bf cells are masked
    to 16 bits when read)."""
    import torch
    import torch.distributed as dist
    assert t.dtype == torch.int32
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class _DevArray:
    """__cuda_array_interface__ carrier: lets torch alias device memory owned by the HIP library"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


def alias_int32(ptr, n, device):
    """torch int32 tensor over n device words at ptr (no copy; the owner must outlive the tensor)"""
    import torch
    if n == 0:
        return torch.zeros(0, dtype=torch.int32, device=device)
    t = torch.as_tensor(_DevArray(ptr, n, "<i4"), device=device)
    assert t.data_ptr() == ptr
    return t
