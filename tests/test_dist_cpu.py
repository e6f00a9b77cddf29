"""The N>1 path on CPU: two gloo ranks shard a k-mer table by rows, scan their shard into
private counters, exchange with ONE sum all-reduce (malva_amd.dist), and must end with exactly
the counters of a single whole-table scan.  The per-rank scan is done by the CPU oracle here (no
GPU in this test); what is under test is the sharding and the wrapping-int32 reduction glue that
bench.py uses unchanged with backend "nccl"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _counters(panel, hi, lo, cnt, k, ref_k, bits, pre=None):
    """scan rows with the oracle; return the [bf counters | map counters] vector as u32 (untruncated bf sums)"""
    from malva_amd import synth
    from oracle import capi as ocapi
    sig, valid = synth.snp_signature_rows(panel, k)
    rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
    rows[:, :k] = sig
    is_ref = np.zeros(rows.shape[0], dtype=np.uint8)
    is_ref[0::2] = 1
    obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
    ocapi.add_kmers(obf, omap, rows, is_ref)
    obf.switch_mode()
    ocapi.ref_scan(obf, octx, panel.genome.tobytes(), k, ref_k)
    octx.switch_mode()
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
    keys = sorted(k_ for k_, _ in omap.items())
    vals = dict(omap.items())
    return np.concatenate([obf.counts().astype(np.uint32), np.array([vals[k_] for k_ in keys], dtype=np.int64).astype(np.uint32)])


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from malva_amd import synth
    from malva_amd.dist import allreduce_counters_, rank_world, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert rank_world() == (rank, world)
    k, ref_k, bits = 35, 43, 1 << 18
    panel = synth.snp_panel(400, 5)
    hi, lo, cnt = synth.kmer_table(panel, 30001, k, ref_k, 6)
    cnt[:] = 0x7FFFFF00 + (cnt & 0xFF)                 # large counts: the u32 sums must wrap identically
    a, b = shard_range(len(hi), rank, world)
    mine = _counters(panel, hi[a:b], lo[a:b], cnt[a:b], k, ref_k, bits)
    t = torch.from_numpy(mine.view(np.int32).copy())
    allreduce_counters_(t)
    if rank == 0:
        np.save(out, t.numpy().view(np.uint32))
    dist.destroy_process_group()


def test_two_rank_shard_scan_allreduce_equals_whole(tmp_path):
    from malva_amd import synth
    from malva_amd.dist import shard_range
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    out = str(tmp_path / "reduced.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    reduced = np.load(out)
    k, ref_k, bits = 35, 43, 1 << 18
    panel = synth.snp_panel(400, 5)
    hi, lo, cnt = synth.kmer_table(panel, 30001, k, ref_k, 6)
    cnt[:] = 0x7FFFFF00 + (cnt & 0xFF)
    # the oracle keeps bf cells as u16: compare those mod 2^16, the map values mod 2^32
    whole = _counters(panel, hi, lo, cnt, k, ref_k, bits)
    n_map = 400                                           # one REF signature per SNP, all distinct
    n_bf = len(whole) - n_map
    assert np.array_equal(reduced[:n_bf] & 0xFFFF, whole[:n_bf] & 0xFFFF)
    assert np.array_equal(reduced[n_bf:], whole[n_bf:])
    assert whole[n_bf:].max() > 0 and (whole[:n_bf] > 0).any()


def _pack_worker(rank, world, port, out, big):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from malva_amd.dist import allreduce_counters_packed_
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    v = rng.integers(0, 65535 // world + 1, size=20001, dtype=np.int64)
    if big and rank == 1:
        v[777] = 0xFFFFFF00                       # one large u32 partial: the guard must fall back to 32-bit
    t = torch.from_numpy(v.astype(np.uint32).view(np.int32).copy())
    took = allreduce_counters_packed_(t)
    if rank == 0:
        np.save(out, np.concatenate([t.numpy().view(np.uint32), np.array([int(took)], dtype=np.uint32)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("big", [False, True])
def test_packed_allreduce_is_exact_or_falls_back(tmp_path, big):
    world = 3
    out = str(tmp_path / "p.npy")
    mp.spawn(_pack_worker, args=(world, _free_port(), out, big), nprocs=world, join=True)
    got = np.load(out)
    want = np.zeros(20001, dtype=np.uint64)
    for r in range(world):
        v = np.random.default_rng(100 + r).integers(0, 65535 // world + 1, size=20001, dtype=np.int64)
        if big and r == 1:
            v[777] = 0xFFFFFF00
        want += v.astype(np.uint64)
    assert np.array_equal(got[:-1], (want & 0xFFFFFFFF).astype(np.uint32))
    assert int(got[-1]) == (0 if big else 1)
