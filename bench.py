#!/usr/bin/env python3
"""bench.py -- the malva-geno `call` hot path on N MI355X GPUs of one node.

One step = one pass of the hot path over one batch of synthetic input that is
already resident in HBM (BASELINE.json config C3 per GPU):

    1. KMC scan (main.cpp:482-500) of this rank's shard of the k-mer table
    2. exchange: one sum all-reduce of the counter vector over RCCL (N > 1 only)
    3. per-variant path (main.cpp:556-559) on this rank's slice of the variants

Scaling with N GPUs: the KMC table is what shards (north_star: "the KMC k-mer table
shards naturally across the 8 GPUs ... with RCCL all-reduce of per-allele counts").
Every rank scans `--kmers` rows of a table N times as large (weak scaling of the
metric's unit); the panel (`--variants` SNPs) is the fixed database: its index
(both filters + exact map) is replicated on every GPU as SURVEY 8(e) prescribes, its
per-allele counters are all-reduced, and its genotyping is split N ways with no
collective.  `--grow-panel` makes the panel N times as large instead (every rank then
genotypes `--variants` SNPs and the replicated index, gate included, grows with N).

Launch:  python bench.py --gpus 1            (default, single process)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
                --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCAN_BYTES_PER_KMER = 44      # 20 B streamed + 3 probes x 8 B
GENO_BYTES_PER_SNP = 128
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kmers", type=float, default=1e8, help="k-mer table rows per GPU")
    ap.add_argument("--variants", type=float, default=1e6, help="isolated biallelic SNPs per GPU")
    ap.add_argument("--b", type=int, default=4, help="filter size in units of 2^33 bits (malva-geno -b)")
    ap.add_argument("--k", type=int, default=35, help="signature k-mer length (malva-geno -k)")
    ap.add_argument("--r", type=int, default=43, help="context k-mer length of the KMC table (malva-geno -r); 63 = config C5's")
    ap.add_argument("--table", choices=["auto", "host", "device"], default="auto",
                    help="where the synthetic table is drawn: host = numpy (malva_amd.synth.kmer_table, SURVEY 8(d)), device = random rows drawn on the "
                         "GPU with the windows around 20 %% / 7.5 of the rows' worth of variant sites planted (minutes -> seconds); auto: device above 2e8 rows")
    ap.add_argument("--layout", choices=["compact", "soa"], default="compact",
                    help="table layout in HBM: compact = 12-byte rows (count << 2r | r-mer; needs 33 <= r <= 44), soa = {hi[], lo[], cnt[]} 20 B/row")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --kmers is the WHOLE table, sharded over the ranks (north_star's 8-GPU claim: "
                         "--strong --kmers 3e9 --variants 8e7 --b 16); default is weak scaling, --kmers rows per GPU")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1: native = mg_counters_allreduce (RCCL inside libmalva_hip.so); torch = torch.distributed over the aliased vector")
    ap.add_argument("--cpu-sample", type=float, default=5e6, help="rows of the table the CPU oracle scans (0 = skip)")
    ap.add_argument("--cpu-variants", type=float, default=2e5)
    ap.add_argument("--no-summary", action="store_true", help="A/B: disable the cache-resident gate")
    ap.add_argument("--grow-panel", action="store_true", help="panel of N x --variants SNPs instead of a fixed one (see module docstring)")
    ap.add_argument("--pack16-min-mb", type=float, default=32.0,
                    help="counter vectors of at least this size try the 16-bit packed all-reduce (smaller ones: the guard costs more than it saves)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: mg_set_option before the index is built (use_partition=0, gate_log2=26, ...)")
    ap.add_argument("--scan-ablate", type=int, default=0,
                    help="profiling only (results invalid): filter-kernel ablation mask, see scan_filter_kernel")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and reduce over gloo: exercises the multi-rank code path on a 1-GPU box (numbers meaningless)")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON: whatever a library prints there (gloo's rank banner, RCCL warnings)
    # goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from malva_amd import BF_ALT, BF_CTX, Context, synth
    from malva_amd.dist import alias_int32, allreduce_counters_, allreduce_counters_packed_, rank_world

    rank, world = rank_world()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    K, R = args.k, args.r
    n_rows = int(args.kmers)
    if args.strong:                                          # fixed total work: this rank's share of the table
        from malva_amd.dist import shard_range as _sr
        a_, b_ = _sr(int(args.kmers), rank, world)
        n_rows = b_ - a_
    n_vars_total = int(args.variants) * (world if args.grow_panel else 1)
    from malva_amd.dist import shard_range
    v0, v1 = shard_range(n_vars_total, rank, world)
    n_vars = v1 - v0                                    # variants this rank genotypes
    bf_bits = args.b << 33

    # ---- setup (untimed): synthetic panel, index build on the device ------------------
    t0 = time.time()
    panel = synth.snp_panel(n_vars_total, seed=20261003)
    log(rank, "panel: %d SNPs on a %.3g-base genome (%.1fs)" % (n_vars_total, panel.genome.size, time.time() - t0))
    ctx = Context(K, R, bf_bits, device=local)
    # one explicit stream for the library's kernels AND torch's work on the aliased counters (a NULL handle would
    # mean the library's private stream: torch's default stream has handle 0)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    ctx.set_stream(work_stream.cuda_stream)
    if args.no_summary:
        ctx.set_option("use_summary", 0)
    if args.scan_ablate:
        ctx.set_option("scan_ablate", args.scan_ablate)
    for kv in args.opt:
        name, value = kv.split("=")
        ctx.set_option(name, int(value))
    t0 = time.time()
    sig, _ = synth.snp_signature_rows(panel, K)
    stride = (K + 1 + 7) // 8 * 8
    batch = 1 << 20
    for a in range(0, sig.shape[0], 2 * batch):
        chunk = sig[a:a + 2 * batch]
        rows = np.zeros((chunk.shape[0], stride), dtype=np.uint8)
        rows[:, :K] = chunk
        ctx.map_insert(rows[0::2])          # allele 0 -> ref_bf (main.cpp:137)
        ctx.bf_insert(BF_ALT, rows[1::2])   # others   -> bf     (main.cpp:139)
    del sig
    ctx.bf_finalize(BF_ALT)
    ctx.ref_scan(panel.genome)
    ctx.bf_finalize(BF_CTX)
    _, n_alt, _ = ctx.bf_info(BF_ALT)
    _, n_ctx, _ = ctx.bf_info(BF_CTX)
    log(rank, "index: %d bf bits set, %d context bits set, %d map keys (%.1fs)" % (n_alt, n_ctx, ctx.map_size(), time.time() - t0))

    # this rank's shard of the table and of the variants
    t0 = time.time()
    sub = synth.Panel(genome=panel.genome, pos=panel.pos[v0:v1], var_allele_off=(panel.var_allele_off[v0:v1 + 1] - panel.var_allele_off[v0]),
                      allele_off=panel.allele_off[2 * v0:2 * v1 + 1] - panel.allele_off[2 * v0], pool=panel.pool[2 * v0:2 * v1],
                      freq=panel.freq[2 * v0:2 * v1], present_mask=panel.present_mask[v0:v1], flags=panel.flags[v0:v1],
                      donor_gt=panel.donor_gt[v0:v1])
    def dev_i64(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)

    device_table = args.table == "device" or (args.table == "auto" and n_rows > 200_000_000)
    if device_table:
        plant = max(1, min(n_vars, int(n_rows * 0.2 / 7.5)))      # 5 windows x 1.5 haplotypes per planted variant -> 20 % of the rows
        tb = synth.device_table(sub, n_rows, K, R, 777 + rank, dev, plant_variants=plant)
        d_hi, d_lo, d_cnt = tb["d_hi"], tb["d_lo"], tb["d_cnt"]
        ns_cpu = int(min(args.cpu_sample, n_rows))
        hi, lo, cnt = (d_hi[:ns_cpu].cpu().numpy().view(np.uint64), d_lo[:ns_cpu].cpu().numpy().view(np.uint64),
                       d_cnt[:ns_cpu].cpu().numpy().view(np.uint32))             # what the CPU leg scans
    else:
        hi, lo, cnt = synth.kmer_table(sub, n_rows, K, R, seed=777 + rank)
        d_hi, d_lo = dev_i64(hi), dev_i64(lo)
        d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
    log(rank, "table: %d rows per GPU, drawn on the %s (%.1fs)" % (n_rows, "device" if device_table else "host", time.time() - t0))
    compact = args.layout == "compact" and 33 <= R <= 44
    d_rows = None
    if compact:                     # the table as it stays resident: packed once, outside the timed region
        d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())

    def scan(n):
        if compact:
            ctx.kmc_scan_rows_device(d_rows.data_ptr(), n)
        else:
            ctx.kmc_scan_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n)
    ctx.reference_upload(panel.genome)
    d_pos = dev_i64(sub.pos.astype(np.uint64))
    d_vo = torch.from_numpy(sub.var_allele_off.astype(np.uint32).view(np.int32)).to(dev)
    d_ao = torch.from_numpy(sub.allele_off.astype(np.uint32).view(np.int32)).to(dev)
    d_pool = torch.from_numpy(np.ascontiguousarray(sub.pool)).to(dev)
    d_freq = torch.from_numpy(np.ascontiguousarray(sub.freq)).to(dev)
    d_pm = dev_i64(sub.present_mask)
    d_fl = torch.from_numpy(np.ascontiguousarray(sub.flags)).to(dev)
    na = int(sub.var_allele_off[-1])
    d_cov = torch.zeros(na, dtype=torch.int32, device=dev)
    d_g1 = torch.zeros(n_vars, dtype=torch.int32, device=dev)
    d_g2 = torch.zeros(n_vars, dtype=torch.int32, device=dev)
    d_gq = torch.zeros(n_vars, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n_vars, dtype=torch.uint8, device=dev)
    d_goff = dev_i64((3 * np.arange(n_vars + 1)).astype(np.uint64))       # biallelic diploid: 3 genotypes per variant
    d_probs = torch.zeros(3 * n_vars, dtype=torch.float64, device=dev)   # normalised likelihoods (GTS) + workspace
    cptr, n_bf, n_map = ctx.counters_view()                 # [bf counters | map counters], one allocation inside the context
    d_counters = alias_int32(cptr, n_bf + n_map, dev)         # reduced in place: no export/import copies
    exchange = "none"
    if world > 1:
        exchange = "torch.distributed all_reduce(sum,int32), in place"
        if args.exchange == "native" and not args.rehearse_on_one_gpu:
            # RCCL inside the library: rank 0's ncclUniqueId travels over the process group, then every step's
            # exchange is mg_counters_allreduce on the library's stream.  If the library cannot bring RCCL up on
            # this node the run still measures (torch's RCCL, same collective) and the JSON line says so.
            from malva_amd import capi
            uid = [None]
            if rank == 0:                   # (a failure here must not leave the other ranks waiting in the broadcast)
                try:
                    uid = [capi.comm_unique_id()]
                except Exception as e:      # noqa: BLE001 -- reported, never silent
                    log(0, "rank 0: native exchange unavailable (%s)" % e)
            dist.broadcast_object_list(uid, src=0)
            ok = torch.zeros(1, dtype=torch.int32, device=dev)
            if uid[0] is not None:
                try:
                    ctx.comm_init(rank, world, uid[0])
                    ok = torch.ones(1, dtype=torch.int32, device=dev)
                except Exception as e:      # noqa: BLE001
                    print("[bench] rank %d: native exchange unavailable (%s)" % (rank, e), file=sys.stderr, flush=True)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                exchange = "mg_counters_allreduce: ncclAllReduce(sum,uint32) inside libmalva_hip.so, in place"
            else:
                exchange += " (native RCCL init failed on some rank: see stderr)"
    native = exchange.startswith("mg_counters_allreduce")

    scan_ms = []
    packed_steps = []

    def step(record=False):
        ctx.counters_reset()
        scan(n_rows)
        if world > 1 and native:
            ctx.counters_allreduce()
        elif world > 1:
            # a large vector is worth halving on the wire; for a small one the guard's extra round trip costs more
            if 4.0 * d_counters.numel() >= args.pack16_min_mb * (1 << 20):
                packed_steps.append(allreduce_counters_packed_(d_counters))
            else:
                allreduce_counters_(d_counters)
        ctx.call_isolated_device(n_vars, d_pos.data_ptr(), d_vo.data_ptr(), d_ao.data_ptr(), d_pool.data_ptr(), d_freq.data_ptr(),
                                 d_pm.data_ptr(), d_fl.data_ptr(), 0.001, 200, False, d_cov.data_ptr(), d_g1.data_ptr(),
                                 d_g2.data_ptr(), d_gq.data_ptr(), d_st.data_ptr(), d_probs.data_ptr(), d_goff.data_ptr())
        if record:
            scan_ms.append(ctx.scan_stats())     # waits on the scan's own events only

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    t_start = time.perf_counter()
    for s in range(args.steps):
        step(record=False)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel timing for the roofline (outside the timed region, same launches) --
    geno_ms = []
    for _ in range(max(3, args.steps)):
        ctx.counters_reset()
        scan(n_rows)
        scan_ms.append(ctx.scan_stats())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.call_isolated_device(n_vars, d_pos.data_ptr(), d_vo.data_ptr(), d_ao.data_ptr(), d_pool.data_ptr(), d_freq.data_ptr(),
                                 d_pm.data_ptr(), d_fl.data_ptr(), 0.001, 200, False, d_cov.data_ptr(), d_g1.data_ptr(),
                                 d_g2.data_ptr(), d_gq.data_ptr(), d_st.data_ptr(), d_probs.data_ptr(), d_goff.data_ptr())
        e1.record()
        e1.synchronize()
        geno_ms.append(e0.elapsed_time(e1))
    filt_ms = float(np.mean([m[0] for m in scan_ms]))
    probe_ms = float(np.mean([m[1] for m in scan_ms]))
    hits_ms = float(np.mean([m[2] for m in scan_ms]))
    n_open, n_hits = int(scan_ms[-1][3]), int(scan_ms[-1][4])
    geno_ms_avg = float(np.mean(geno_ms))

    # ---- size-independent check of the full-size run: GT histogram is sane and the counters are consistent --
    g1 = d_g1.cpu().numpy(); g2 = d_g2.cpu().numpy(); gq = d_gq.cpu().numpy()
    called = {"0/0": int(np.sum((g1 == 0) & (g2 == 0))), "0/1": int(np.sum((g1 == 0) & (g2 == 1))), "1/1": int(np.sum((g1 == 1) & (g2 == 1)))}

    cpu_baseline = None
    parity_sample = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:       # the CPU leg runs at N=1 only
        from oracle import capi as ocapi
        ns = int(min(args.cpu_sample, n_rows))
        log(rank, "cpu baseline: importing the device-built filters into the oracle ...")
        obf, octx, omap = ocapi.BF(bf_bits), ocapi.BF(bf_bits), ocapi.KMAP()
        _, _, words, _ = ctx.bf_export(BF_ALT)
        obf.load_words(words)
        _, _, words, _ = ctx.bf_export(BF_CTX)
        octx.load_words(words)
        del words
        obf.switch_mode(); octx.switch_mode()
        sigs, _ = synth.snp_signature_rows(panel, K)
        refrows = np.zeros((n_vars_total, stride), dtype=np.uint8)
        refrows[:, :K] = sigs[0::2]
        ocapi.add_kmers(obf, omap, refrows, np.ones(n_vars_total, dtype=np.uint8))
        del sigs, refrows
        t0 = time.perf_counter()
        ocapi.kmc_scan_packed(octx, obf, omap, hi[:ns], lo[:ns], cnt[:ns], K, R)
        cpu_scan_s = time.perf_counter() - t0
        # parity of the same sample through the device path
        ctx.counters_reset()
        scan(ns)
        ctx.synchronize()
        _, _, _, counts = ctx.bf_export(BF_ALT)
        keys, vals = ctx.map_export()
        ok_bf = bool(np.array_equal(counts, obf.counts()))
        ok_map = dict(zip(keys, (int(v) for v in vals))) == dict(omap.items())
        nv = int(min(args.cpu_variants, n_vars))
        t0 = time.perf_counter()
        ocov, og1, og2, ogq = ocapi.call_isolated(obf, omap, panel.genome, sub.pos[:nv], sub.allele_off[:2 * nv + 1], sub.var_allele_off[:nv + 1],
                                                  sub.pool[:2 * nv], sub.freq[:2 * nv], sub.present_mask[:nv], sub.flags[:nv], K, 0.001, 200, False)
        cpu_geno_s = time.perf_counter() - t0
        ctx.call_isolated_device(nv, d_pos.data_ptr(), d_vo.data_ptr(), d_ao.data_ptr(), d_pool.data_ptr(), d_freq.data_ptr(), d_pm.data_ptr(),
                                 d_fl.data_ptr(), 0.001, 200, False, d_cov.data_ptr(), d_g1.data_ptr(), d_g2.data_ptr(), d_gq.data_ptr(), d_st.data_ptr(), d_probs.data_ptr(), d_goff.data_ptr())
        torch.cuda.synchronize()
        ok_gt = bool(np.array_equal(d_g1[:nv].cpu().numpy(), og1) and np.array_equal(d_g2[:nv].cpu().numpy(), og2)
                     and np.array_equal(d_gq[:nv].cpu().numpy(), ogq) and np.array_equal(d_cov[:2 * nv].cpu().numpy().view(np.uint32), ocov))
        parity_sample = {"bf_counters_equal": ok_bf, "map_values_equal": ok_map, "gt_gq_cov_equal": ok_gt, "rows": ns, "variants": nv}
        # the same loop on every host core the box gives this job (atomic, commuting counter adds;
        # tests/test_synth_cpu.py shows the results equal the single-threaded ones) -- after the parity check,
        # because it adds to the same oracle counters
        cores = max(1, min(len(os.sched_getaffinity(0)), 64))
        ns_all = int(min(len(hi), ns * max(1, cores // 2)))
        t0 = time.perf_counter()
        ocapi.kmc_scan_packed_mt(octx, obf, omap, hi[:ns_all], lo[:ns_all], cnt[:ns_all], K, R, cores)
        cpu_all_s = time.perf_counter() - t0
        cpu_baseline = {"value": ns / cpu_scan_s, "unit": "kmers/s", "cores": 1, "kind": "port",
                        "sample": "first %d rows of rank 0's table through oracle/malva_oracle.c (single thread, as the reference); "
                                  "%d variants through its loop-B restatement" % (ns, nv),
                        "variants_per_s": nv / cpu_geno_s,
                        "all_cores": {"value": ns_all / cpu_all_s, "unit": "kmers/s", "cores": cores, "kind": "port",
                                      "sample": "first %d rows, the same loop on %d threads sharing one index (atomic adds)" % (ns_all, cores)}}

    traffic = None
    tname = "traffic_scan_filter12.json" if compact else "traffic_scan_filter.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and not args.scan_ablate:
        # HBM bytes per launch cannot be counted from inside this process: it comes from the rocprofv3 PMC passes
        # of tools/profile_gpu.sh on this same command (FETCH_SIZE calibrated on the kernel's own 8-byte-per-lane
        # stream, WRITE_SIZE as read), committed with the profile it was derived from.
        t = json.load(open(tpath))
        if t.get("units_per_launch") == n_rows and t.get("bf_bits") == bf_bits:
            traffic = t["hbm_bytes_per_launch"]
    if rank == 0:
        total_rows = int(args.kmers) if args.strong else n_rows * world
        total_kmers = total_rows * args.steps
        total_vars = n_vars_total * args.steps
        rows_per_launch = min(n_rows, 1 << 27)      # mg_kmc_scan_device walks the table in chunks of 2^27 rows; the first is timed
        achieved = SCAN_BYTES_PER_KMER * rows_per_launch / (filt_ms * 1e-3) / 1e9
        scan_ms_sum = filt_ms + probe_ms + hits_ms         # the three kernels of one scan chunk, back to back on one stream
        achieved_scan = SCAN_BYTES_PER_KMER * rows_per_launch / (scan_ms_sum * 1e-3) / 1e9
        spec = "%d,%d" % (K, R) if (K, R) in ((35, 43), (35, 63)) else "0,0"
        out = {
            "metric": "KMC k-mers scanned/sec (whole call step: scan + counter all-reduce + genotyping), k=%d r=%d" % (K, R),
            "value": total_kmers / elapsed,
            "unit": "kmers/s",
            "variants_per_s": total_vars / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s%s: %.3g KMC k-mers per GPU (table of %.3g rows sharded by rows) against a panel of %.3g isolated biallelic SNPs, "
                                   "k=%d r=%d b=%d, table resident in HBM" % (
                                       "C3" if (n_vars_total, args.b, K, R) == (1000000, 4, 35, 43) else "custom",
                                       " per GPU" if args.grow_panel or world == 1 else (" whole table / N, fixed panel" if args.strong else " table x N, fixed panel"),
                                       n_rows, total_rows, n_vars_total, K, R, args.b),
                       "kmers_per_gpu": n_rows, "kmers_total": total_rows, "panel_variants": n_vars_total, "variants_genotyped_per_gpu": n_vars,
                       "k": K, "ref_k": R, "bf_bits": bf_bits,
                       "table_layout": "12-byte rows (count << 2r | r-mer)" if compact else "SoA hi[] lo[] cnt[] (20 B/row)",
                       "parallelism": "table rows x%d (%s), panel genotyping split x%d, index replicated" % (world, "strong" if args.strong else "weak", world),
                       "summary_bitmaps": not args.no_summary,
                       "exchange": ("none" if world == 1 else "%s over %d counters%s" % (
                           exchange, n_bf + n_map, ", 16-bit packed when exact (%d of %d steps)" % (sum(packed_steps), len(packed_steps)) if packed_steps else ""))},
            "roofline": {"kernel": ("scan_ticket_sort_kernel<%s> + scan_ticket_gate_kernel (ticket form, %d gate slices; the two passes summed)" % (spec, ctx.get_option("scan_tickets"))
                                    if ctx.get_option("scan_tickets") else
                                    "scan_bin_kernel<%s,4> + scan_bin_gate_kernel<%s> (partitioned second level, %d slices)" % (spec, spec, ctx.get_option("scan_bins"))
                                    if ctx.get_option("scan_bins") else ("scan_filter12_kernel<%s>" if compact else "scan_filter_kernel<%s,2>") % spec), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": "profiles/%s (rocprofv3 PMC)" % tname if traffic else None,
                         "algorithmic_bytes_per_launch": SCAN_BYTES_PER_KMER * rows_per_launch, "bytes_per_unit": SCAN_BYTES_PER_KMER,
                         "units_per_launch": rows_per_launch, "avg_launch_ms": filt_ms},
            # the whole H10 loop (filter + probe + hit kernels, summed): the fraction SURVEY 8(d)'s 44 B/k-mer budget is about
            "roofline_scan": {"kernels": ("scan_ticket_sort + scan_ticket_gate + scan_probe + scan_hits" if ctx.get_option("scan_tickets") else
                                          "scan_filter + scan_probe + scan_hits"), "bound": "hbm", "achieved": achieved_scan, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved_scan / HBM_PEAK_GBS, "ms": scan_ms_sum, "bytes_per_unit": SCAN_BYTES_PER_KMER,
                              "units_per_launch": rows_per_launch},
            "kernels_ms": {"scan_filter": filt_ms, "scan_probe": probe_ms, "scan_hits": hits_ms, "call_isolated": geno_ms_avg,
                           "gate_open_rows": n_open, "bf_hit_rows": n_hits},
            "genotype_roofline": {"achieved": GENO_BYTES_PER_SNP * n_vars / (geno_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "bytes_per_unit": GENO_BYTES_PER_SNP},
            "calls": called,
            "cpu_baseline": cpu_baseline,
            "parity_sample": parity_sample,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
