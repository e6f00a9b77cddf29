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


def allreduce_counters_packed_(t, limit_world=16):
    """Same result as allreduce_counters_ with half the bytes on the wire, when that is provably exact.

    Two counters travel in one int32 (low and high 16 bits).  That is exact iff no 16-bit lane can carry into
    its neighbour, i.e. iff every rank's every partial counter is <= 65535 // world.  The bound is checked with a
    MAX all-reduce of one scalar first; if it does not hold the plain 32-bit all-reduce runs instead (an odd
    last element travels unpacked at the end of the packed vector).  Partial counters are tiny in practice: KMC lists each distinct k-mer once, so a
    counter receives one count (<= 255) plus the counts of the few k-mers that collide with it.
    Returns True if the packed path was taken."""
    import torch
    import torch.distributed as dist
    assert t.dtype == torch.int32
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return False
    world = dist.get_world_size()
    n = t.numel()
    ok = n > 1 and world <= limit_world
    if ok:
        # counters are u32 bit patterns: a value >= 2^31 shows up negative here, so test both ends
        ext = torch.stack([t.max(), -t.min()]).to(torch.int64)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX)
        mx, neg = (int(x) for x in ext.tolist())
        ok = neg <= 0 and mx <= 65535 // world
    if not ok:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return False
    m = n // 2
    pairs = t[: 2 * m].view(-1, 2)
    packed = torch.empty(m + (n & 1), dtype=torch.int32, device=t.device)
    packed[:m] = pairs[:, 0] | (pairs[:, 1] << 16)
    if n & 1:
        packed[m] = t[n - 1]
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    pairs[:, 0] = packed[:m] & 0xFFFF
    pairs[:, 1] = (packed[:m] >> 16) & 0xFFFF
    if n & 1:
        t[n - 1] = packed[m]
    return True


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
