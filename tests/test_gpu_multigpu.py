"""SURVEY 8(e) through the C ABI: the k-mer table shards by rows over several contexts, each scans its shard into
its own counters, ONE exchange (mg_counters_allreduce*: RCCL's ncclAllReduce(sum, uint32) inside libmalva_hip.so, or
the kernel sum for contexts that share a device) must leave every context with exactly the counters of a
whole-table scan -- compared with the oracle, which scans the whole table on the CPU.

A one-GPU box can hold only one RCCL rank per device, so there:
  * the RCCL calls themselves run on hardware as a one-rank communicator (all-reduce = identity),
  * the N-way layout runs with N contexts on device 0 (MG_COMM_LOCAL),
  * the two-process RCCL test is skipped; it runs wherever two GPUs are visible."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import build_index_pair, map_values_by_key
from malva_amd import BF_ALT, Context, capi, synth
from malva_amd.dist import shard_range
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, R, BITS = 35, 43, 1 << 20          # a small filter: collisions, so bf counters really are sums over several k-mers


def _n_devices():
    import torch
    return torch.cuda.device_count()


def _case(seed, n_vars=3000, n_rows=200000):
    panel = synth.snp_panel(n_vars, seed)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, K, R, seed + 1)
    cnt[:] = 0x7FFFFF00 + (cnt & 0xFF)          # u32 sums wrap, u16 cells wrap many times
    return panel, hi, lo, cnt


def _assert_equals_oracle(ctx, obf, omap):
    _, _, _, counts = ctx.bf_export(BF_ALT)
    assert np.array_equal(counts, obf.counts())
    assert map_values_by_key(ctx) == dict(omap.items())


def test_rccl_one_rank_communicator_runs_on_hardware():
    panel, hi, lo, cnt = _case(31)
    with Context(K, R, BITS) as ctx:
        obf, octx, omap = build_index_pair(ctx, panel, K, R, BITS)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
        ctx.kmc_scan(hi, lo, cnt)
        uid = capi.comm_unique_id()
        assert len(uid) == capi.COMM_ID_BYTES and any(uid)
        ctx.comm_init(0, 1, uid)
        assert ctx.comm_info() == (0, 1, capi.COMM_RCCL)
        ctx.counters_allreduce()                 # ncclAllReduce over one rank: must leave every counter as it was
        ctx.synchronize()
        _assert_equals_oracle(ctx, obf, omap)
        ctx.comm_destroy()
        assert ctx.comm_info() == (0, 0, capi.COMM_NONE)
        with pytest.raises(capi.MalvaError):
            ctx.counters_allreduce()


@pytest.mark.parametrize("world", [2, 3])
def test_contexts_shard_scan_allreduce_equals_whole_table(world):
    panel, hi, lo, cnt = _case(40 + world)
    distinct = _n_devices() >= world
    ctxs = [Context(K, R, BITS, device=(r if distinct else 0)) for r in range(world)]
    try:
        obf = octx = omap = None
        for c in ctxs:                            # every rank builds (or loads) the same index
            obf, octx, omap = build_index_pair(c, panel, K, R, BITS)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)      # the oracle scans the WHOLE table
        capi.comm_init_all(ctxs)
        assert ctxs[1].comm_info() == (1, world, capi.COMM_RCCL if distinct else capi.COMM_LOCAL)
        for r, c in enumerate(ctxs):
            a, b = shard_range(len(hi), r, world)
            c.kmc_scan(hi[a:b], lo[a:b], cnt[a:b])
        partial = ctxs[0].bf_export(BF_ALT)[3]
        assert not np.array_equal(partial, obf.counts())              # one shard alone is not the answer
        capi.counters_allreduce_all(ctxs)
        for c in ctxs:
            c.synchronize()
            _assert_equals_oracle(c, obf, omap)
        # a second round on the same group: reset, scan other shards, exchange again
        for r, c in enumerate(ctxs):
            c.counters_reset()
            a, b = shard_range(len(hi), world - 1 - r, world)
            c.kmc_scan(hi[a:b], lo[a:b], cnt[a:b])
        capi.counters_allreduce_all(ctxs)
        for c in ctxs:
            c.synchronize()
            _assert_equals_oracle(c, obf, omap)
    finally:
        for c in ctxs:
            c.close()


def _small_counts_case(seed, n_vars=3000, n_rows=200000):
    """partial counters small enough for the 16-bit packed form: KMC counts as they come (2..63), two table rows per signature at most"""
    panel = synth.snp_panel(n_vars, seed)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, K, R, seed + 1)
    return panel, hi, lo, cnt


@pytest.mark.parametrize("world,pack", [(2, 2), (3, 2), (3, 0)])
def test_packed_exchange_equals_whole_table(world, pack):
    """the 16-bit packed form of the exchange (two counters per word on the wire, guarded by the ranks' largest partial counter):
    the same pack / sum / unpack kernels whether RCCL or, on a one-GPU box, the kernel sum moves the words.  Small counts take it,
    counts that could carry fall back to the 32-bit sum by themselves; either way every context ends with the whole table's counters."""
    distinct = _n_devices() >= world
    for big in (False, True):
        panel, hi, lo, cnt = (_case(60 + world) if big else _small_counts_case(60 + world))
        ctxs = [Context(K, R, 1 << 26, device=(r if distinct else 0)) for r in range(world)]
        try:
            obf = octx = omap = None
            for c in ctxs:
                c.set_option("exchange_pack", pack)
                obf, octx, omap = build_index_pair(c, panel, K, R, 1 << 26)
            ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
            capi.comm_init_all(ctxs)
            for r, c in enumerate(ctxs):
                a, b = shard_range(len(hi), r, world)
                c.kmc_scan(hi[a:b], lo[a:b], cnt[a:b])
            capi.counters_allreduce_all(ctxs)
            for c in ctxs:
                c.synchronize()
                assert c.get_option("exchange_packed") == (1 if pack and not big else 0)
                _assert_equals_oracle(c, obf, omap)
        finally:
            for c in ctxs:
                c.close()


@pytest.mark.parametrize("pack", [0, 2])
def test_exchange_on_its_own_stream_beside_the_block_cut(pack):
    """mg_counters_allreduce_begin / _end on hardware (a one-rank communicator: the sum is the identity, the streams, events and the
    packed form's guard, pack and unpack are real): scan -> begin -> block cut of the resident record loop -> end -> coverages and
    likelihoods.  Counters, cuts, coverages, GT and GQ equal the oracle's."""
    import torch
    from malva_amd.resident import ResidentPanel
    from test_gpu_resident import oracle_blocks
    k, ref_k, bits = 35, 43, 1 << 28
    panel = synth.clustered_snp_panel(40_000, seed=47, n_contigs=2)
    args = oracle_blocks(panel, k)
    hi, lo, cnt = synth.flat_kmer_table(panel, 400_000, k, ref_k, seed=9, max_records=6_000)
    obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
    ocapi.index_blocks(obf, omap, panel.genome, **args, haploid=False, k=k)
    obf.switch_mode()
    for b, l in zip(panel.contig_base, panel.contig_len):
        ocapi.ref_scan(obf, octx, panel.genome[int(b):int(b) + int(l)].tobytes(), k, ref_k)
    octx.switch_mode()
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
    dev = torch.device("cuda", 0)
    d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
    d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    with Context(k, ref_k, bits) as ctx:
        ctx.set_option("exchange_pack", pack)
        ctx.reference_upload(panel.genome)
        rp = ResidentPanel(panel, 0, haploid=False)
        assert rp.index(ctx).sum() == 0
        ctx.bf_finalize(BF_ALT)
        for b, l in zip(panel.contig_base, panel.contig_len):
            ctx.ref_scan_resident(int(b), int(l))
        ctx.bf_finalize(1)
        ctx.comm_init(0, 1, capi.comm_unique_id())
        with pytest.raises(capi.MalvaError):
            ctx.counters_allreduce_end()                       # nothing begun
        for _ in range(2):                                     # twice: the streams and events are reused
            ctx.counters_reset()
            ctx.kmc_scan_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), len(hi))
            ctx.counters_allreduce_begin()
            with pytest.raises(capi.MalvaError):
                ctx.counters_allreduce_begin()                 # one exchange at a time
            rp.cut(ctx)                                        # needs no counters: runs beside the collective
            ctx.counters_allreduce_end()
            rp.cover(ctx)
            rp.genotype(ctx)
            got = rp.results()
            ms, packed = ctx.exchange_stats()
            assert ms >= 0 and packed == (1 if pack else 0)
            want = ocapi.cover_blocks(obf, omap, panel.genome, **args, haploid=False, k=k)
            assert np.array_equal(got["blk_var_off"], args["blk_var_off"]) and np.array_equal(got["cov"], want) and got["overflow"].sum() == 0
            g1, g2, gq = ocapi.genotype_panel(want, panel.freq, panel.var_allele_off, 0.001, 200, False)
            assert np.array_equal(got["g1"], g1) and np.array_equal(got["g2"], g2) and np.array_equal(got["gq"], gq)
            assert (want > 0).sum() > 3_000
        _assert_equals_oracle(ctx, obf, omap)


def test_allreduce_refuses_contexts_with_different_indexes():
    a, b = Context(K, R, BITS), Context(K, R, BITS)
    try:
        build_index_pair(a, synth.snp_panel(500, 1), K, R, BITS)
        build_index_pair(b, synth.snp_panel(700, 2), K, R, BITS)
        capi.comm_init_all([a, b])               # both on device 0
        with pytest.raises(capi.MalvaError):
            capi.counters_allreduce_all([a, b])
    finally:
        a.close()
        b.close()


_RANK_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, uid_hex, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
from malva_amd import BF_ALT, BF_CTX, Context, synth
from malva_amd.dist import shard_range
K, R, BITS = 35, 43, 1 << 20
panel = synth.snp_panel(3000, 77)
hi, lo, cnt = synth.kmer_table(panel, 200000, K, R, 78)
cnt[:] = 0x7FFFFF00 + (cnt & 0xFF)
sig, valid = synth.signature_rows(panel, K)
rows = np.zeros((sig.shape[0], 40), dtype=np.uint8); rows[:, :K] = sig
is_ref = np.zeros(sig.shape[0], dtype=np.uint8); is_ref[panel.var_allele_off[:-1]] = 1
ctx = Context(K, R, BITS, device=rank)
ctx.map_insert(rows[valid][is_ref[valid] == 1]); ctx.bf_insert(BF_ALT, rows[valid][is_ref[valid] == 0])
ctx.bf_finalize(BF_ALT); ctx.ref_scan(panel.genome.tobytes()); ctx.bf_finalize(BF_CTX)
ctx.comm_init(rank, world, bytes.fromhex(uid_hex))
a, b = shard_range(len(hi), rank, world)
ctx.kmc_scan(hi[a:b], lo[a:b], cnt[a:b])
ctx.counters_allreduce(); ctx.synchronize()
keys, vals = ctx.map_export()
order = np.argsort(np.array(keys))
np.savez(out, counts=ctx.bf_export(BF_ALT)[3], keys=np.array(keys)[order], vals=vals[order])
ctx.close()
"""


def test_two_processes_rccl_allreduce(tmp_path):
    """one process per GPU, as bench.py and a multi-node driver would run it: needs two visible GPUs"""
    if _n_devices() < 2:
        pytest.skip("one GPU visible: RCCL rejects two ranks on one device (covered by the one-rank and shared-device tests)")
    uid = capi.comm_unique_id().hex()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", uid, str(tmp_path / ("r%d.npz" % r))]) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    panel = synth.snp_panel(3000, 77)
    hi, lo, cnt = synth.kmer_table(panel, 200000, K, R, 78)
    cnt[:] = 0x7FFFFF00 + (cnt & 0xFF)
    with Context(K, R, BITS) as ctx:
        obf, octx, omap = build_index_pair(ctx, panel, K, R, BITS)
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
    want = dict(omap.items())
    for r in range(2):
        got = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert np.array_equal(got["counts"], obf.counts())
        assert dict(zip((bytes(k) for k in got["keys"]), (int(v) for v in got["vals"]))) == want


_ORDER_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1])
assert "torch" not in sys.modules
import numpy as np
from malva_amd import BF_ALT, BF_CTX, Context, synth
K, R, BITS = 35, 43, 1 << 20
ctx = Context(K, R, BITS)                       # the library initialises HIP first ...
panel = synth.snp_panel(1000, 3)
sig, valid = synth.signature_rows(panel, K)
rows = np.zeros((sig.shape[0], 40), dtype=np.uint8); rows[:, :K] = sig
is_ref = np.zeros(sig.shape[0], dtype=np.uint8); is_ref[panel.var_allele_off[:-1]] = 1
ctx.map_insert(rows[valid][is_ref[valid] == 1]); ctx.bf_insert(BF_ALT, rows[valid][is_ref[valid] == 0])
ctx.bf_finalize(BF_ALT); ctx.ref_scan(panel.genome.tobytes()); ctx.bf_finalize(BF_CTX)
import torch                                     # ... and torch comes second
assert torch.cuda.is_available(), "torch sees no GPU after libmalva_hip.so initialised HIP"
from malva_amd.dist import alias_int32
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)               # share a stream with torch
hi, lo, cnt = synth.kmer_table(panel, 50000, K, R, 4)
with torch.cuda.stream(stream):
    d_hi = torch.from_numpy(hi.view(np.int64)).to(dev); d_lo = torch.from_numpy(lo.view(np.int64)).to(dev)
    d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
    ctx.kmc_scan_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), len(hi))
    ptr, n_bf, n_map = ctx.counters_view()
    seen_by_torch = alias_int32(ptr, n_bf + n_map, dev).clone()     # ordered behind the scan on the shared stream
stream.synchronize()
seen = seen_by_torch.cpu().numpy().view(np.uint32)
counts = ctx.bf_export(BF_ALT)[3]
assert np.array_equal((seen[:n_bf] & 0xFFFF).astype(np.uint16), counts) and counts.any()
keys, vals = ctx.map_export()
assert int(seen[n_bf:].astype(np.uint64).sum()) == int(vals.astype(np.uint32).astype(np.uint64).sum()) > 0
ctx.close()
print("ORDER_OK")
"""


def test_context_before_torch_import_then_shared_stream(tmp_path):
    """The library must not care who initialises HIP first: a fresh interpreter creates a Context,
    only then imports torch, and the two share a stream."""
    script = tmp_path / "order.py"
    script.write_text(_ORDER_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ORDER_OK" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("workload,extra", [("c3", ["--kmers", "2e6", "--variants", "2e4", "--strong-c4-kmers", "4e6", "--strong-c4-variants", "3e5"]),
                                            ("c5", ["--kmers", "1e6", "--clusters", "3e3"])])
def test_bench_self_launch_two_ranks_on_one_gpu(workload, extra):
    """plain `python bench.py --gpus 2` (no launcher): the ranks are started by bench.py itself; on a one-GPU box they share
    the device and reduce over gloo (--rehearse-on-one-gpu).  ONE JSON line comes out, for two ranks, with the table rows and
    the record loop split between them."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload", workload, "--steps", "2",
                        "--warmup", "1", "--cpu-sample", "0", "--sustained-s", "0.2"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["config"]["kmers_total"] == 2 * d["config"]["kmers_per_gpu"]
    assert d["config"]["variants_genotyped_per_gpu"] < d["config"]["panel_variants"]
    assert d["overflow_records"] == 0
    if workload == "c3":       # the whole-genome leg every N > 1 line of the default workload carries, here at a reduced size
        s4 = d["strong_c4"]
        assert "error" not in s4, s4
        assert s4["n_gpus"] == 2 and s4["scaling"] == "strong" and s4["overflow_records"] == 0 and s4["kernels_ms"]["general_records"] > 1000
        assert "speedup_vs_one_gpu" not in s4
