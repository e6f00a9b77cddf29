#!/usr/bin/env python3
"""bench.py -- the malva-geno `call` hot path on N MI355X GPUs of one node.

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM
(SURVEY.md 8(d)):

    1. KMC scan (main.cpp:482-500) of this rank's shard of the k-mer table
    2. exchange: one sum all-reduce of the counter vector over RCCL (N > 1 only)
    3. the record loop (main.cpp:522-579) on this rank's run of the panel's records: block cut, signature
       enumeration + lookup + coverage, likelihoods, GT / GQ

Workloads (`--workload`, BASELINE.json's configs):
    c3 (default)  1e8 KMC k-mers + 1e6 isolated biallelic SNPs, k35 r43 b4: the "1xMI355X HBM-roofline run"; the record
                  loop is the fused lone-variant path (mg_call_isolated_device)
    c4            whole genome: 3e9 k-mers + 8e7 SNPs as SURVEY 8(d) draws them (38 nt mean spacing, a tenth in clusters
                  of <= 4 within 17 nt, 3.1e9-nt genome in 24 sequences), k35 r43 b16; the table is cut over the ranks
                  (strong scaling); the record loop runs on the resident panel: mg_cut_blocks_device ->
                  mg_cover_blocks_device -> mg_genotype_device
    c5            indel / MNP-heavy clusters (<= 6 records, <= 3 ALTs, insertions up to k and beyond, 8 samples half
                  unphased), k35 r63 b8, diploid AND haploid (two indexes, two measurements), same resident record loop

Scaling with N GPUs: the KMC table is what shards (north_star).  c3 / c5: every rank scans `--kmers` rows of a table N
times as large (weak scaling of the metric's unit) against a fixed panel; c4 (or `--strong`): `--kmers` is the whole
table.  The panel's index is replicated, its counters are all-reduced, its record loop is split N ways (at block
boundaries) with no collective.  Every N > 1 line of the default workload also carries `strong_c4`: the whole-genome
configuration cut N ways, measured right after.

Launch:  python bench.py --gpus N [--steps K --warmup W]     N > 1 without a launcher: the ranks are started from here,
                                                               before this process touches a GPU
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCAN_BYTES_PER_KMER = 44      # 20 B streamed + 3 probes x 8 B
GENO_BYTES_PER_SNP = 128      # isolated biallelic SNP
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def blocks_bytes(n_vars, n_sig_kmers, n_genotypes):
    """SURVEY 8(d), general form: 64 + 36 K + 8 G bytes per variant with K signature k-mers and G genotypes"""
    return 64 * n_vars + 36 * n_sig_kmers + 8 * n_genotypes


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c3", "c4", "c5"], default="c3")
    ap.add_argument("--kmers", type=float, default=None, help="k-mer table rows per GPU (c3, c5: 1e8) or in all (c4, --strong: 3e9)")
    ap.add_argument("--variants", type=float, default=None, help="panel records (c3: 1e6 isolated SNPs; c4: 8e7 clustered SNPs)")
    ap.add_argument("--clusters", type=float, default=2.6e5, help="c5: clusters of 1..6 records (2.6e5 clusters ~ 9e5 records)")
    ap.add_argument("--b", type=int, default=None, help="filter size in units of 2^33 bits (malva-geno -b); default per workload: 4 / 16 / 8")
    ap.add_argument("--k", type=int, default=35, help="signature k-mer length (malva-geno -k)")
    ap.add_argument("--r", type=int, default=None, help="context k-mer length of the KMC table (malva-geno -r); c5: 63, else 43")
    ap.add_argument("--contigs", type=int, default=24, help="c4: sequences the genome is cut into (positions stay inside int32)")
    ap.add_argument("--table", choices=["auto", "host", "device"], default="auto",
                    help="c3: where the synthetic table is drawn: host = numpy (malva_amd.synth.kmer_table, SURVEY 8(d)), device = random rows drawn on the "
                         "GPU with the windows around 20 %% / 7.5 of the rows' worth of variant sites planted (minutes -> seconds); auto: device above 2e8 rows")
    ap.add_argument("--layout", choices=["compact", "soa"], default="compact",
                    help="table layout in HBM: compact = 12-byte rows (count << 2r | r-mer; needs 33 <= r <= 44; packing is outside the timed step and "
                         "reported as pack_rows_ms), soa = {hi[], lo[], cnt[]} 20 B/row")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --kmers is the WHOLE table, sharded over the ranks (implied by --workload c4)")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1: native = mg_counters_allreduce (RCCL inside libmalva_hip.so); torch = torch.distributed over the aliased vector")
    ap.add_argument("--cpu-sample", type=float, default=5e6, help="rows of the table the CPU oracle scans (0 = skip)")
    ap.add_argument("--cpu-variants", type=float, default=2e5)
    ap.add_argument("--sustained-s", type=float, default=2.0, help="length of the sustained leg (back-to-back steps after the timed region; 0 = skip)")
    ap.add_argument("--no-strong-c4", action="store_true", help="N > 1, default workload: skip the strong_c4 leg")
    ap.add_argument("--no-c4-leg", action="store_true", help="N = 1, default workload: skip the whole-genome leg (c4_whole in the JSON line)")
    ap.add_argument("--c4-leg-kmers", type=float, default=3e9)
    ap.add_argument("--c4-leg-variants", type=float, default=8e7)
    ap.add_argument("--c4-parity-records", type=float, default=5e5, help="records per slice of the whole-genome parity check (three slices: head, middle, tail; 0 = skip)")
    ap.add_argument("--c4-parity-rows", type=float, default=8e6, help="rows of each slice's own k-mer table (its donor's windows + random rows)")
    ap.add_argument("--no-c5-leg", action="store_true", help="N = 1, default workload: skip the reduced C5 leg (general_blocks_c5 in the JSON line)")
    ap.add_argument("--c5-leg-clusters", type=float, default=6e4, help="clusters of the reduced C5 leg (6e4 ~ 2.1e5 records)")
    ap.add_argument("--c5-leg-kmers", type=float, default=2e7)
    ap.add_argument("--plant-records", type=float, default=None, help="c5: records whose donor windows are planted in the table (host loop; default: 20 %% of the rows' worth)")
    ap.add_argument("--strong-c4-kmers", type=float, default=3e9)
    ap.add_argument("--strong-c4-variants", type=float, default=8e7)
    ap.add_argument("--no-summary", action="store_true", help="A/B: disable the cache-resident gate")
    ap.add_argument("--no-graph", action="store_true", help="N = 1, c3: launch every step kernel by kernel instead of replaying one captured step")
    ap.add_argument("--grow-panel", action="store_true", help="c3: panel of N x --variants SNPs instead of a fixed one")
    ap.add_argument("--pack16-min-mb", type=float, default=32.0,
                    help="torch exchange: counter vectors of at least this size try the 16-bit packed all-reduce")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="A/B: mg_set_option before the index is built")
    ap.add_argument("--scan-ablate", type=int, default=0, help="profiling only (results invalid): filter-kernel ablation mask")
    ap.add_argument("--launch-check", action="store_true",
                    help="print this rank's launch environment as one JSON line and exit before anything touches a GPU (tests of the self-launch)")
    ap.add_argument("--launch-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests of the self-launch: this rank exits 3 at once, the others would wait a minute
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and reduce over gloo: exercises the multi-rank code path on a 1-GPU box (numbers meaningless)")
    args = ap.parse_args(argv)
    if args.workload == "c4":
        args.strong = True
    d = {"c3": (1e8, 1e6, 4, 43), "c4": (3e9, 8e7, 16, 43), "c5": (1e8, None, 8, 63)}[args.workload]
    args.kmers = d[0] if args.kmers is None else args.kmers
    args.variants = d[1] if args.variants is None else args.variants
    args.b = d[2] if args.b is None else args.b
    args.r = d[3] if args.r is None else args.r
    return args


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as child processes.  This process has not imported torch
    or touched a GPU (and never does); rank 0's stdout -- the JSON line -- is passed through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    # all ranks are watched together: the first one that fails takes the others down with it (they would otherwise sit in a
    # barrier or a collective until the backend's timeout), and the launcher returns its code
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
            if rc:      # give the terminated ranks a moment, then make sure
                deadline = time.time() + 10
                while live and time.time() < deadline:
                    live = [q for q in live if q.poll() is None]
                    time.sleep(0.05)
                for q in live:
                    q.kill()
                for q in live:
                    q.wait()
                live = []
    return rc


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


class Job:
    """One workload on this rank: context, index, resident table and panel, and `step()`."""

    def __init__(self, workload, args, rank, world, local, torch, dist, haploid=False, kmers=None, variants=None, b=None, strong=None, clusters=None,
                 plant_records=None):
        # Building a job is local work (panel, index, table) followed, for N > 1, by collectives (the exchange's bring-up).  A rank
        # that fails in the local part must not leave its peers waiting in those: every rank reports, and all give up together.
        err = None
        try:
            self._build(workload, args, rank, world, local, torch, dist, haploid, kmers, variants, b, strong, clusters, plant_records)
        except Exception as e:      # noqa: BLE001 -- re-raised below, on every rank
            err = e
        if world > 1:
            t = torch.tensor([0 if err is None else 1], dtype=torch.int32, device="cpu" if args.rehearse_on_one_gpu else torch.device("cuda", local))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if int(t.item()) and err is None:
                err = RuntimeError("another rank could not build the %s job" % workload)
        if err is not None:
            raise err
        if world > 1:
            self._bring_up_exchange()

    def _build(self, workload, args, rank, world, local, torch, dist, haploid, kmers, variants, b, strong, clusters, plant_records):
        from malva_amd import BF_ALT, BF_CTX, Context, synth
        from malva_amd.dist import alias_int32, shard_range
        self.workload, self.args, self.rank, self.world, self.torch, self.dist = workload, args, rank, world, torch, dist
        self.haploid = haploid
        self.K = K = args.k
        self.R = R = {"c5": 63}.get(workload, 43) if workload != args.workload else args.r
        self.b = b if b is not None else args.b
        self.bf_bits = self.b << 33
        self.strong = args.strong if strong is None else strong
        kmers = args.kmers if kmers is None else kmers
        variants = args.variants if variants is None else variants
        self.dev = dev = torch.device("cuda", local)
        self.total_rows = int(kmers) * (1 if self.strong else world)
        if self.strong:
            a_, b_ = shard_range(int(kmers), rank, world)
            self.n_rows = b_ - a_
        else:
            self.n_rows = int(kmers)
        self.flat = workload != "c3"
        t0 = time.time()
        if workload == "c3":
            self.n_vars_total = int(variants) * (world if args.grow_panel else 1)
            self.panel = synth.snp_panel(self.n_vars_total, seed=20261003)
            genome = self.panel.genome
        elif workload == "c4":
            self.panel = synth.clustered_snp_panel(int(variants), seed=20261004, n_contigs=args.contigs, k=K)
            self.n_vars_total = self.panel.n
            genome = self.panel.genome
        else:
            self.panel = synth.indel_panel(int(args.clusters if clusters is None else clusters), seed=20261005, k=K)
            self.n_vars_total = self.panel.n
            genome = self.panel.genome
        log(rank, "%s panel: %d records on a %.3g-base genome (%.1fs)" % (workload, self.n_vars_total, genome.size, time.time() - t0))
        self.ctx = ctx = Context(K, R, self.bf_bits, device=local)
        self.stream = torch.cuda.current_stream(dev)
        ctx.set_stream(self.stream.cuda_stream)
        if args.no_summary:
            ctx.set_option("use_summary", 0)
        if args.scan_ablate:
            ctx.set_option("scan_ablate", args.scan_ablate)
        for kv in args.opt:
            name, value = kv.split("=")
            ctx.set_option(name, int(value))
        # ---- index (untimed) --------------------------------------------------------------------------------------
        t0 = time.time()
        ctx.reference_upload(genome)
        if not self.flat:
            sig, _ = synth.snp_signature_rows(self.panel, K)
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
            ctx.ref_scan_resident(0, genome.size)    # main.cpp:383-401 on the reference already in HBM
        else:
            from malva_amd.resident import ResidentPanel
            whole = ResidentPanel(self.panel, dev, haploid=haploid)
            ovf = whole.index(ctx)                  # cut + extract_kmers + add_kmers_to_bf on the device (main.cpp:309-370)
            if int(ovf.sum()):
                raise SystemExit("bench: %d records exceeded a device capacity at index time (the CLI's host enumerator would take them): "
                                 "not a measurement of the device path" % int(ovf.sum()))
            del whole
            ctx.bf_finalize(BF_ALT)
            for cb, cl in zip(self.panel.contig_base, self.panel.contig_len):
                ctx.ref_scan_resident(int(cb), int(cl))
        ctx.bf_finalize(BF_CTX)
        _, n_alt, _ = ctx.bf_info(BF_ALT)
        _, n_ctx, _ = ctx.bf_info(BF_CTX)
        self.index_s = time.time() - t0
        log(rank, "index: %d bf bits set, %d context bits set, %d map insertion rows (%.1fs)" % (n_alt, n_ctx, ctx.counters_size()[1], self.index_s))
        # ---- this rank's shard of the table and run of the records ------------------------------------------------------
        t0 = time.time()

        def dev_i64(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
        if not self.flat:
            v0, v1 = shard_range(self.n_vars_total, rank, world)
            self.n_vars = v1 - v0
            p = self.panel
            self.sub = sub = synth.Panel(genome=p.genome, pos=p.pos[v0:v1], var_allele_off=(p.var_allele_off[v0:v1 + 1] - p.var_allele_off[v0]),
                                         allele_off=p.allele_off[2 * v0:2 * v1 + 1] - p.allele_off[2 * v0], pool=p.pool[2 * v0:2 * v1],
                                         freq=p.freq[2 * v0:2 * v1], present_mask=p.present_mask[v0:v1], flags=p.flags[v0:v1], donor_gt=p.donor_gt[v0:v1])
            device_table = args.table == "device" or (args.table == "auto" and self.n_rows > 200_000_000)
            if device_table:
                plant = max(1, min(self.n_vars, int(self.n_rows * 0.2 / 7.5)))      # 5 windows x 1.5 haplotypes per planted variant -> 20 % of the rows
                tb = synth.device_table(sub, self.n_rows, K, R, 777 + rank, dev, plant_variants=plant)
                self.d_hi, self.d_lo, self.d_cnt = tb["d_hi"], tb["d_lo"], tb["d_cnt"]
            else:
                hi, lo, cnt = synth.kmer_table(sub, self.n_rows, K, R, seed=777 + rank)
                self.d_hi, self.d_lo = dev_i64(hi), dev_i64(lo)
                self.d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
            self.d_pos = dev_i64(sub.pos.astype(np.uint64))
            self.d_vo = torch.from_numpy(sub.var_allele_off.astype(np.uint32).view(np.int32)).to(dev)
            self.d_ao = torch.from_numpy(sub.allele_off.astype(np.uint32).view(np.int32)).to(dev)
            self.d_pool = torch.from_numpy(np.ascontiguousarray(sub.pool)).to(dev)
            self.d_freq = torch.from_numpy(np.ascontiguousarray(sub.freq)).to(dev)
            self.d_pm = dev_i64(sub.present_mask)
            self.d_fl = torch.from_numpy(np.ascontiguousarray(sub.flags)).to(dev)
            na = int(sub.var_allele_off[-1])
            self.d_cov = torch.zeros(na, dtype=torch.int32, device=dev)
            self.d_g1 = torch.zeros(self.n_vars, dtype=torch.int32, device=dev)
            self.d_g2 = torch.zeros(self.n_vars, dtype=torch.int32, device=dev)
            self.d_gq = torch.zeros(self.n_vars, dtype=torch.int32, device=dev)
            self.d_st = torch.zeros(self.n_vars, dtype=torch.uint8, device=dev)
            self.d_goff = dev_i64((3 * np.arange(self.n_vars + 1)).astype(np.uint64))       # biallelic diploid: 3 genotypes per variant
            self.d_probs = torch.zeros(3 * self.n_vars, dtype=torch.float64, device=dev)   # normalised likelihoods (GTS) + workspace
            self.n_genotypes = 3 * self.n_vars
            # the record loop of the step is the GENERAL resident path (block cut on the device, panel GT gathered by the lone tier,
            # likelihoods): the same records as a FlatPanel.  The fused lone-variant entry (mg_call_isolated_device, fed masks the host
            # precomputed) stays as a named extra: kernels_ms.record_loop_isolated
            from malva_amd.resident import ResidentPanel
            self.rp = ResidentPanel(synth.flat_from_snp_panel(sub), dev, haploid=False)
        else:
            from malva_amd.resident import ResidentPanel
            cuts = self.panel.split_points(world)
            v0, v1 = cuts[rank], cuts[rank + 1]
            self.v0, self.n_vars = v0, v1 - v0
            self.sub = sub = self.panel.slice(v0, v1)
            self.rp = ResidentPanel(sub, dev, haploid=haploid)
            self.n_genotypes = self.rp.n_gt
            plant = max(1, min(self.n_vars, int(self.n_rows * 0.2 / 7.5)))
            if plant_records is None and args.plant_records is not None:
                plant_records = args.plant_records
            if plant_records is not None:
                plant = max(1, min(plant, int(plant_records)))
            tb = synth.device_table_flat(sub, self.n_rows, K, R, 777 + rank, dev, plant_records=plant)
            self.d_hi, self.d_lo, self.d_cnt = tb["d_hi"], tb["d_lo"], tb["d_cnt"]
            log(rank, "table: %d of the rows are the donor's windows around %d records" % (tb["n_site"], plant))
        log(rank, "table: %d rows per GPU (%.1fs)" % (self.n_rows, time.time() - t0))
        self.compact = args.layout == "compact" and 33 <= R <= 44
        self.d_rows = None
        self.pack_ms = None
        if self.compact:                     # the table as it stays resident: packed once, outside the timed region (and timed: pack_rows_ms)
            self.d_rows = torch.zeros(ctx.kmc_rows_bytes(self.n_rows) // 4, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.kmc_pack_rows_device(self.d_hi.data_ptr(), self.d_lo.data_ptr(), self.d_cnt.data_ptr(), self.n_rows, self.d_rows.data_ptr())
            e1.record()
            e1.synchronize()
            self.pack_ms = e0.elapsed_time(e1)
        self.d_counters = None
        if world > 1:                                                         # (a context whose vector has been handed out keeps its counters there alone:
            cptr, self.n_bf, self.n_map = ctx.counters_view()                 #  one GPU does not ask for it, and its lookups read the records' own copies)
            self.d_counters = alias_int32(cptr, self.n_bf + self.n_map, dev)  # [bf counters | map counters], reduced in place: no export/import copies
        else:
            self.n_bf, self.n_map = ctx.counters_size()
        self.exchange = "none"
        self.native = False
        self.packed_steps = []

    def _bring_up_exchange(self):
        args, rank, world, torch, dist, ctx, dev = self.args, self.rank, self.world, self.torch, self.dist, self.ctx, self.dev
        self.exchange = "torch.distributed all_reduce(sum,int32), in place"
        if args.exchange == "native" and not args.rehearse_on_one_gpu:
            # RCCL inside the library: rank 0's ncclUniqueId travels over the process group, then every step's exchange is
            # mg_counters_allreduce on the library's stream.  If the library cannot bring RCCL up on this node the run
            # still measures (torch's RCCL, same collective) and the JSON line says so.
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
                self.exchange = "mg_counters_allreduce: ncclAllReduce(sum,uint32) inside libmalva_hip.so, in place"
            else:
                self.exchange += " (native RCCL init failed on some rank: see stderr)"
        self.native = self.exchange.startswith("mg_counters_allreduce")

    # ---- the step ------------------------------------------------------------------------------------------------------
    def scan(self, n):
        if self.compact:
            self.ctx.kmc_scan_rows_device(self.d_rows.data_ptr(), n)
        else:
            self.ctx.kmc_scan_device(self.d_hi.data_ptr(), self.d_lo.data_ptr(), self.d_cnt.data_ptr(), n)

    def call(self):
        """the record loop on this rank's records: mg_cut_blocks_device -> mg_cover_blocks_device -> mg_genotype_device (GT and GQ, as
        `malva-geno call` without -v prints them; the normalised likelihood lists of -v are timed apart: kernels_ms.genotype_verbose)"""
        self.rp.call_step(self.ctx, probs=False)

    def call_isolated(self, n=None):
        """c3 only: the fused lone-variant entry on the first n records"""
        n = self.n_vars if n is None else n
        self.ctx.call_isolated_device(n, self.d_pos.data_ptr(), self.d_vo.data_ptr(), self.d_ao.data_ptr(), self.d_pool.data_ptr(), self.d_freq.data_ptr(),
                                      self.d_pm.data_ptr(), self.d_fl.data_ptr(), 0.001, 200, False, self.d_cov.data_ptr(), self.d_g1.data_ptr(),
                                      self.d_g2.data_ptr(), self.d_gq.data_ptr(), self.d_st.data_ptr(), self.d_probs.data_ptr(), self.d_goff.data_ptr())

    def step(self):
        from malva_amd.dist import allreduce_counters_, allreduce_counters_packed_
        self.ctx.counters_reset()
        self.scan(self.n_rows)
        if self.world > 1 and self.native:
            # the exchange on its own stream (RCCL inside the library: 16-bit packed when no partial counter can carry), the
            # record loop's block cut -- which needs no counters -- beside it, then the rest of the record loop behind it
            self.ctx.counters_allreduce_begin()
            self.rp.cut(self.ctx)
            self.ctx.counters_allreduce_end()
            self.rp.cover(self.ctx)
            self.rp.genotype(self.ctx, probs=False)
            return
        elif self.world > 1:
            # a large vector is worth halving on the wire; for a small one the guard's extra round trip costs more
            if 4.0 * self.d_counters.numel() >= self.args.pack16_min_mb * (1 << 20):
                self.packed_steps.append(allreduce_counters_packed_(self.d_counters))
            else:
                allreduce_counters_(self.d_counters)
        self.call()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def capture_step(self):
        """one step captured into a HIP graph (every launch of a step goes to the context's stream; after the warm-up steps the library
        allocates nothing): the C3 step is three dozen launches of which two dozen are the record loop's, a few microseconds apart.
        Returns the replay callable, or None where capture is not on (N > 1: the exchange's second stream and RCCL stay outside graphs;
        the whole-genome and C5 steps are milliseconds of kernels, not of launches) or fails."""
        torch = self.torch
        self.launch = "stream"
        if self.world > 1 or self.workload != "c3" or self.args.no_graph:
            return None
        if self.ctx.get_option("record_counters_live"):     # (a replay repeats its kernels' arguments, the counter copies' epoch among them: only where the vectors are the counters)
            return None
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(self.stream)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                self.ctx.set_stream(side.cuda_stream)
                self.step()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=side):
                    self.step()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 -- reported; the steps are then launched one by one as before
            log(self.rank, "step not captured into a graph (%s): plain launches" % (repr(e)[:200]))
            self.ctx.set_stream(self.stream.cuda_stream)
            torch.cuda.synchronize()
            return None
        self.ctx.set_stream(self.stream.cuda_stream)     # (what follows the timed region launches on the usual stream again)
        self._graph, self._graph_stream = graph, side    # kept alive as long as the job
        self.launch = "hipGraph replay of one captured step"
        return graph.replay

    def timed(self, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks"""
        torch = self.torch
        for _ in range(warmup):
            self.step()
        run = self.capture_step() or self.step
        torch.cuda.synchronize()
        self.barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        self.barrier()
        elapsed = time.perf_counter() - t_start
        self.stream_ms = None
        if run != self.step:        # the same K steps launched one by one: for the record, and as the check that a replayed step leaves what a launched one does
            def state():
                r = self.rp.results()
                return [r[k].copy() for k in ("cov", "g1", "g2", "gq", "overflow")]
            after_replay = state()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            stream_elapsed = time.perf_counter() - t0
            self.stream_ms = 1e3 * stream_elapsed / steps
            self.replay_equals_launched = all(np.array_equal(a, b) for a, b in zip(after_replay, state()))
            if not self.replay_equals_launched:      # never seen; the replayed timing is then not a measurement of this step
                log(self.rank, "a replayed step left other results than a launched one: the plain launches are the timing")
                self.launch = "stream (graph replay discarded: results differed)"
                elapsed = stream_elapsed
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.args.rehearse_on_one_gpu else self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def sustained(self, seconds):
        """back-to-back steps for at least `seconds` (no host synchronisation inside a burst): the sustained-clock figure"""
        torch = self.torch
        if seconds <= 0:
            return None
        torch.cuda.synchronize()
        self.barrier()
        n, t0 = 0, time.perf_counter()
        burst = 8
        while True:
            for _ in range(burst):
                self.step()
            n += burst
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
        el = time.perf_counter() - t0
        clock = None
        try:
            clock = int(torch.cuda.clock_rate(self.dev))
        except Exception:   # noqa: BLE001 -- amdsmi not importable: the clock is simply not reported
            pass
        if self.world > 1:
            t = torch.tensor([el / n], dtype=torch.float64, device="cpu" if self.args.rehearse_on_one_gpu else self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return {"ms_per_step": 1e3 * float(t.item()), "steps": n, "seconds": el, "sclk_mhz": clock}
        return {"ms_per_step": 1e3 * el / n, "steps": n, "seconds": el, "sclk_mhz": clock}

    def kernel_times(self, reps):
        """per-kernel durations outside the timed region, same launches: HIP events on the launch stream"""
        torch, ctx = self.torch, self.ctx
        scan_ms, call_ms, cut_ms, geno_ms, genov_ms, blk, iso_ms = [], [], [], [], [], [], []
        for _ in range(reps):
            ctx.counters_reset()
            self.scan(self.n_rows)
            scan_ms.append(ctx.scan_stats())
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            self.rp.cut(ctx)
            ev[1].record()
            self.rp.cover(ctx)
            ev[2].record()
            self.rp.genotype(ctx, probs=False)
            ev[3].record()
            self.rp.genotype(ctx, probs=True)          # (-v: the normalised lists too; outside the record loop's figure)
            ev4 = torch.cuda.Event(enable_timing=True)
            ev4.record()
            ev4.synchronize()
            cut_ms.append(ev[0].elapsed_time(ev[1]))
            geno_ms.append(ev[2].elapsed_time(ev[3]))
            genov_ms.append(ev[3].elapsed_time(ev4))
            blk.append(ctx.blocks_stats())
            call_ms.append(ev[0].elapsed_time(ev[3]))
            if not self.flat:       # the fused lone-variant entry on the same counters (it ends the records' copies' validity: last)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.call_isolated()
                e1.record()
                e1.synchronize()
                iso_ms.append(e0.elapsed_time(e1))
        out = {"scan_filter": float(np.mean([m[0] for m in scan_ms])), "scan_probe": float(np.mean([m[1] for m in scan_ms])),
               "scan_hits": float(np.mean([m[2] for m in scan_ms])), "record_loop": float(np.mean(call_ms)),
               "gate_open_rows": int(scan_ms[-1][3]), "bf_hit_rows": int(scan_ms[-1][4])}
        out.update({"cut_blocks": float(np.mean(cut_ms)), "tier1_lone": float(np.mean([b[0] for b in blk])),
                    "tier2_flat": float(np.mean([b[1] for b in blk])), "tier3_workgroup": float(np.mean([b[2] for b in blk])),
                    "genotype": float(np.mean(geno_ms)), "genotype_verbose": float(np.mean(genov_ms)), "general_records": blk[-1][3], "lone_signature_kmers": blk[-1][4],
                    "general_signature_kmers": blk[-1][5], "tier3_records": blk[-1][6]})
        if iso_ms:
            out["record_loop_isolated"] = float(np.mean(iso_ms))
        if self.world > 1 and self.native:      # the exchange by itself (collective + pack / unpack), and the counter-free work that ran beside it
            ex = []
            for _ in range(reps):
                ctx.counters_reset()
                self.scan(self.n_rows)
                ctx.counters_allreduce_begin()
                ctx.counters_allreduce_end()
                ex.append(ctx.exchange_stats())
            out["exchange_ms"] = float(np.mean([e[0] for e in ex]))
            out["exchange_packed_16bit"] = bool(ex[-1][1])
            out["overlapped_ms"] = out["cut_blocks"]
            out["overlap_note"] = "what runs beside the exchange is the block cut alone: tier 1's lookups need the summed counters, and the general list tier 2 walks comes out of tier 1"
        return out

    def results(self):
        r = self.rp.results()
        return r["g1"], r["g2"], r["gq"], int(r["overflow"].sum())

    def close(self):
        self.ctx.close()
        for name in list(self.__dict__):
            if name.startswith("d_") or name in ("rp", "panel", "sub"):
                self.__dict__[name] = None
        self.torch.cuda.empty_cache()


def cpu_leg(job, log_rank):
    """The oracle (a CPU port, single thread as the reference) on a bounded sample of the same workload, on rank 0 at N = 1:
    timing, and parity of the sample through the device path."""
    from malva_amd import BF_ALT, BF_CTX, synth
    from oracle import capi as ocapi
    args, ctx, K, R, torch = job.args, job.ctx, job.K, job.R, job.torch
    ns = int(min(args.cpu_sample, job.n_rows))
    hi = job.d_hi[:ns].cpu().numpy().view(np.uint64)
    lo = job.d_lo[:ns].cpu().numpy().view(np.uint64)
    cnt = job.d_cnt[:ns].cpu().numpy().view(np.uint32)
    log(log_rank, "cpu baseline: importing the device-built filter bits into the oracle ...")
    obf, octx, omap = ocapi.BF(job.bf_bits), ocapi.BF(job.bf_bits), ocapi.KMAP()
    _, _, words, _ = ctx.bf_export(BF_ALT)
    obf.load_words(words)
    _, _, words, _ = ctx.bf_export(BF_CTX)
    octx.load_words(words)
    del words
    obf.switch_mode(); octx.switch_mode()
    nv = int(min(args.cpu_variants, job.n_vars))
    full_index = True
    if not job.flat:
        stride = (K + 1 + 7) // 8 * 8
        sigs, _ = synth.snp_signature_rows(job.panel, K)
        refrows = np.zeros((job.n_vars_total, stride), dtype=np.uint8)
        refrows[:, :K] = sigs[0::2]
        ocapi.add_kmers(obf, omap, refrows, np.ones(job.n_vars_total, dtype=np.uint8))
        del sigs, refrows
    else:
        # the exact map's keys: the oracle's own enumeration of the panel (its bf inserts go to a sink: the bits were imported).
        # Bounded: beyond 2e6 records only the head of the panel is indexed and the leg is a timing, not a parity check.
        p = job.panel
        n_idx = p.n if p.n <= 2_000_000 else 2_000_000
        full_index = n_idx == p.n
        head = p.slice(0, n_idx)
        off, bc = ocapi.cut_blocks(head.pos, head.ref_size, head.min_size, head.contig_id, K)
        sink = ocapi.BF(1 << 16)
        ocapi.index_blocks(sink, omap, p.genome, p.contig_base[bc], p.contig_len[bc], off, head.pos, head.ref_size, head.min_size, head.present,
                           head.var_allele_off, head.allele_off, head.pool, head.canon, head.gt, head.n_samples, job.haploid, K)
    t0 = time.perf_counter()
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
    cpu_scan_s = time.perf_counter() - t0
    parity = None
    sub = job.sub
    if full_index:      # parity of the same sample through the device path
        ctx.counters_reset()
        job.scan(ns)
        ctx.synchronize()
        _, _, _, counts = ctx.bf_export(BF_ALT)
        keys, vals = ctx.map_export()
        parity = {"bf_counters_equal": bool(np.array_equal(counts, obf.counts())),
                  "map_values_equal": dict(zip(keys, (int(v) for v in vals))) == dict(omap.items()), "rows": ns}
    if not job.flat:
        t0 = time.perf_counter()
        ocov, og1, og2, ogq = ocapi.call_isolated(obf, omap, job.panel.genome, sub.pos[:nv], sub.allele_off[:2 * nv + 1], sub.var_allele_off[:nv + 1],
                                                  sub.pool[:2 * nv], sub.freq[:2 * nv], sub.present_mask[:nv], sub.flags[:nv], K, 0.001, 200, False)
        cpu_geno_s = time.perf_counter() - t0
        from malva_amd.resident import ResidentPanel
        hp = ResidentPanel(synth.flat_from_snp_panel(synth.head(sub, nv)), job.dev, haploid=False)      # the general resident path, as the step runs it
        hp.call_step(ctx)
        r = hp.results()
        parity["gt_gq_cov_equal"] = bool(np.array_equal(r["g1"], og1) and np.array_equal(r["g2"], og2) and np.array_equal(r["gq"], ogq)
                                         and np.array_equal(r["cov"], ocov) and not r["overflow"].any())
        job.call_isolated(nv)                                                                          # and the fused lone-variant entry
        torch.cuda.synchronize()
        parity["isolated_entry_equal"] = bool(np.array_equal(job.d_g1[:nv].cpu().numpy(), og1) and np.array_equal(job.d_g2[:nv].cpu().numpy(), og2)
                                              and np.array_equal(job.d_gq[:nv].cpu().numpy(), ogq)
                                              and np.array_equal(job.d_cov[:2 * nv].cpu().numpy().view(np.uint32), ocov))
        what = "%d variants through its loop-B restatement (lone variants)" % nv
    else:
        nv = sub.split_points(max(1, sub.n // max(nv, 1)))[1] if nv < sub.n else sub.n      # a whole number of blocks
        head = sub.slice(0, nv)
        t0 = time.perf_counter()
        off, bc = ocapi.cut_blocks(head.pos, head.ref_size, head.min_size, head.contig_id, K)
        stats = {}
        ocov = ocapi.cover_blocks(obf, omap, sub.genome, sub.contig_base[bc], sub.contig_len[bc], off, head.pos, head.ref_size, head.min_size, head.present,
                                  head.var_allele_off, head.allele_off, head.pool, head.canon, head.gt, head.n_samples, job.haploid, K, stats=stats)
        og1, og2, ogq = ocapi.genotype_panel(ocov, head.freq, head.var_allele_off, 0.001, 200, job.haploid)
        cpu_geno_s = time.perf_counter() - t0
        if full_index:
            from malva_amd.resident import ResidentPanel
            hp = ResidentPanel(head, job.dev, haploid=job.haploid)
            hp.call_step(ctx)
            r = hp.results()
            parity["gt_gq_cov_equal"] = bool(np.array_equal(r["g1"], og1) and np.array_equal(r["g2"], og2) and np.array_equal(r["gq"], ogq)
                                             and np.array_equal(r["cov"], ocov) and np.array_equal(r["blk_var_off"], off) and not r["overflow"].any())
        what = "%d records (%d blocks, %d signature k-mers) through its block path: cut, extract_kmers, set_coverages, genotype" % (
            nv, len(off) - 1, stats["kmers"])
    if parity is not None:
        parity["variants"] = nv
    # the same scan loop on every host core the box gives this job (atomic, commuting counter adds) -- after
    # the parity check, because it adds to the same oracle counters
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    ns_all = int(min(job.n_rows, ns * max(1, cores // 2)))
    hi = job.d_hi[:ns_all].cpu().numpy().view(np.uint64)
    lo = job.d_lo[:ns_all].cpu().numpy().view(np.uint64)
    cnt = job.d_cnt[:ns_all].cpu().numpy().view(np.uint32)
    t0 = time.perf_counter()
    ocapi.kmc_scan_packed_mt(octx, obf, omap, hi, lo, cnt, K, R, cores)
    cpu_all_s = time.perf_counter() - t0
    base = {"value": ns / cpu_scan_s, "unit": "kmers/s", "cores": 1, "kind": "port",
            "sample": "first %d rows of rank 0's table through oracle/malva_oracle.c (single thread, as the reference); %s%s" % (
                ns, what, "" if full_index else "; exact map of the panel's first 2e6 records only: a timing, no parity"),
            "variants_per_s": nv / cpu_geno_s,
            "all_cores": {"value": ns_all / cpu_all_s, "unit": "kmers/s", "cores": cores, "kind": "port",
                          "sample": "first %d rows, the same loop on %d threads sharing one index (atomic adds)" % (ns_all, cores)}}
    return base, parity


def c4_slice_parity(job, log_rank):
    """Parity of the whole-genome job at its full index: three slices of the panel (head, middle, tail -- whole blocks; the
    middle and tail lie beyond position 2^24, where are_near runs in float) through the oracle.  The oracle cannot build the
    8e7-record index or scan 3e9 rows in the time a bench has, and does not have to: `bf` and `context_bf` are the device's
    bits imported whole (their parity is the index tests' business), the exact map holds the slice's own keys (the oracle's
    enumeration of the slice), and both sides scan the slice's OWN k-mer table -- the donor's windows around every record of
    the slice plus random rows -- from zeroed counters, through the form of the scan the step uses.  Compared: every bf
    counter, and cut / coverage / GT / GQ of every record of the slice."""
    from malva_amd import BF_ALT, BF_CTX, synth
    from malva_amd.resident import ResidentPanel
    from oracle import capi as ocapi
    args, ctx, K, R, torch, p = job.args, job.ctx, job.K, job.R, job.torch, job.panel
    want = int(args.c4_parity_records)
    if want <= 0:
        return None
    t_all = time.perf_counter()
    log(log_rank, "c4 parity: importing the device-built filter bits into the oracle ...")
    obf, octx = ocapi.BF(job.bf_bits), ocapi.BF(job.bf_bits)
    _, _, words, _ = ctx.bf_export(BF_ALT)
    obf.load_words(words)
    _, _, words, _ = ctx.bf_export(BF_CTX)
    octx.load_words(words)
    del words
    obf.switch_mode(); octx.switch_mode()
    cuts = sorted(set(p.split_points(max(3, p.n // max(want, 1)))))      # (whole blocks; a cut that found no block boundary nearby repeats its neighbour)
    picks = sorted({0, (len(cuts) - 1) // 2, len(cuts) - 2})
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    res = {"slices": [], "all_equal": True, "records": 0, "rows": 0}
    cpu_s = 0.0
    for q in picks:
        a, b = cuts[q], cuts[q + 1]
        sl = p.slice(a, b)
        t0 = time.perf_counter()
        off, bc = ocapi.cut_blocks(sl.pos, sl.ref_size, sl.min_size, sl.contig_id, K)
        omap, sink = ocapi.KMAP(), ocapi.BF(1 << 16)
        ocapi.index_blocks(sink, omap, p.genome, p.contig_base[bc], p.contig_len[bc], off, sl.pos, sl.ref_size, sl.min_size, sl.present, sl.var_allele_off, sl.allele_off,
                           sl.pool, sl.canon, sl.gt, sl.n_samples, job.haploid, K)
        cpu_s += time.perf_counter() - t0
        nr = int(args.c4_parity_rows)
        tb = synth.device_table_flat(sl, nr, K, R, 4242 + q, job.dev, plant_records=sl.n)
        ctx.counters_reset()
        if job.compact:
            d_rows = torch.zeros(ctx.kmc_rows_bytes(nr) // 4, dtype=torch.int32, device=job.dev)
            ctx.kmc_pack_rows_device(tb["d_hi"].data_ptr(), tb["d_lo"].data_ptr(), tb["d_cnt"].data_ptr(), nr, d_rows.data_ptr())
            ctx.kmc_scan_rows_device(d_rows.data_ptr(), nr)
        else:
            ctx.kmc_scan_device(tb["d_hi"].data_ptr(), tb["d_lo"].data_ptr(), tb["d_cnt"].data_ptr(), nr)
        form = "sub-slice form, %d pieces" % ctx.get_option("scan_subs") if ctx.get_option("scan_subs") else "ticket form" if ctx.get_option("scan_tickets") else "direct form"
        rp = ResidentPanel(sl, job.dev, haploid=job.haploid)
        rp.call_step(ctx)                      # (reads the records' copies: the vectors stay lazy until the export below)
        r = rp.results()
        hi = tb["d_hi"].cpu().numpy().view(np.uint64); lo = tb["d_lo"].cpu().numpy().view(np.uint64); cnt = tb["d_cnt"].cpu().numpy().view(np.uint32)
        t0 = time.perf_counter()
        obf.counts()[:] = 0
        ocapi.kmc_scan_packed_mt(octx, obf, omap, hi, lo, cnt, K, R, cores)
        ocov = ocapi.cover_blocks(obf, omap, p.genome, p.contig_base[bc], p.contig_len[bc], off, sl.pos, sl.ref_size, sl.min_size, sl.present, sl.var_allele_off,
                                  sl.allele_off, sl.pool, sl.canon, sl.gt, sl.n_samples, job.haploid, K)
        og1, og2, ogq = ocapi.genotype_panel(ocov, sl.freq, sl.var_allele_off, 0.001, 200, job.haploid)
        cpu_s += time.perf_counter() - t0
        _, _, _, counts = ctx.bf_export(BF_ALT)
        eq = {"bf_counters_equal": bool(np.array_equal(counts, obf.counts())), "cuts_equal": bool(np.array_equal(r["blk_var_off"], off)),
              "cov_equal": bool(np.array_equal(r["cov"], ocov)), "gt_gq_equal": bool(np.array_equal(r["g1"], og1) and np.array_equal(r["g2"], og2) and np.array_equal(r["gq"], ogq)),
              "no_overflow": not bool(r["overflow"].any())}
        ok = all(eq.values())
        res["slices"].append({"records": [int(a), int(b)], "max_pos": int(sl.pos.max()), "beyond_2_24": bool(sl.pos.max() > (1 << 24)), "rows": nr, "site_rows": int(tb["n_site"]),
                              "covered_alleles": int((ocov > 0).sum()), "non_reference_calls": int(((og1 > 0) | (og2 > 0)).sum()), "scan_form": form, **eq})
        res["all_equal"] = res["all_equal"] and ok
        res["records"] += int(b - a)
        res["rows"] += nr
        del tb, rp
    res["oracle_cpu_s"] = cpu_s
    res["seconds"] = time.perf_counter() - t_all
    ctx.counters_reset()
    return res


def traffic_of(job):
    """HBM bytes per launch of the filter kernel cannot be counted from inside this process: it comes from the rocprofv3 PMC
    passes of tools/profile_gpu.sh on this same command, committed with the profile it was derived from"""
    tname = "traffic_scan_filter12.json" if job.compact else "traffic_scan_filter.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and not job.args.scan_ablate:
        t = json.load(open(tpath))
        if t.get("units_per_launch") == job.n_rows and t.get("bf_bits") == job.bf_bits:
            return t["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 PMC)" % tname
    return None, None


def record(job, elapsed, steps, warmup, kt, sustained):
    """the JSON fields of one measured job (rank 0)"""
    args, ctx, K, R, world = job.args, job.ctx, job.K, job.R, job.world
    total_kmers = job.total_rows * steps
    total_vars = job.n_vars_total * steps
    rows_per_launch = min(job.n_rows, 1 << 27)      # the scan walks the table in launch groups of 2^27 rows; mg_scan_stats times every one of them and
                                                    # gives the durations per group of this many rows, averaged over all groups (= rocprofv3's per-kernel averages)
    n_groups = (job.n_rows + (1 << 27) - 1) >> 27
    filt, scan_sum = kt["scan_filter"], kt["scan_filter"] + kt["scan_probe"] + kt["scan_hits"]
    ach_filter = SCAN_BYTES_PER_KMER * rows_per_launch / (filt * 1e-3) / 1e9
    ach_scan = SCAN_BYTES_PER_KMER * rows_per_launch / (scan_sum * 1e-3) / 1e9
    spec = "%d,%d" % (K, R) if (K, R) in ((35, 43), (35, 63)) else "0,0"
    if ctx.get_option("scan_subs"):
        fname = "scan_sub_sort_kernel<%s> + scan_sub_gate_kernel (sub-slice form, %d pieces of the gate; the two passes summed)" % (spec, ctx.get_option("scan_subs"))
        sname = "scan_sub_sort + scan_sub_gate + scan_sub_probe (probe and hit pass in one kernel)"
    elif ctx.get_option("scan_tickets"):
        fname = "scan_ticket_sort_kernel<%s> + scan_ticket_gate_kernel (ticket form, %d gate slices; the two passes summed)" % (spec, ctx.get_option("scan_tickets"))
        sname = "scan_ticket_sort + scan_ticket_gate + scan_probe + scan_hits"
    elif ctx.get_option("scan_bins"):
        fname = "scan_bin_kernel<%s,4> + scan_bin_gate_kernel<%s> (partitioned second level, %d slices)" % (spec, spec, ctx.get_option("scan_bins"))
        sname = "scan_bin + scan_bin_gate + scan_probe + scan_hits"
    else:
        fname = ("scan_filter12_kernel<%s>" if job.compact else "scan_filter_kernel<%s,2>") % spec
        sname = "%s + scan_probe_kernel + scan_hits_kernel" % fname
    traffic, tsrc = traffic_of(job)
    # the three kernels' counted HBM bytes (profiles/traffic_scan_c3.json: FETCH_SIZE / WRITE_SIZE passes at HEAD), when this IS that workload
    scan_traffic, scan_tsrc = None, None
    t3 = os.path.join(ROOT, "profiles", "traffic_scan_c3.json")
    if job.compact and os.path.exists(t3) and not job.args.scan_ablate and not ctx.get_option("scan_tickets") and not ctx.get_option("scan_bins") and not ctx.get_option("scan_subs"):
        t = json.load(open(t3))
        if t.get("units_per_launch") == job.n_rows and t.get("bf_bits") == job.bf_bits and (K, R) == (35, 43):
            scan_traffic, scan_tsrc = t["hbm_bytes_per_launch"], "profiles/traffic_scan_c3.json (rocprofv3 PMC)"
    t4 = os.path.join(ROOT, "profiles", "traffic_scan_c4.json")     # the whole-genome table: FETCH_SIZE / WRITE_SIZE passes over the sub-slice form's three kernels
    if job.workload == "c4" and job.compact and os.path.exists(t4) and not job.args.scan_ablate and ctx.get_option("scan_subs"):
        t = json.load(open(t4))
        if t.get("rows") == job.n_rows and t.get("bf_bits") == job.bf_bits and t.get("panel_variants") == job.n_vars_total:
            scan_traffic, scan_tsrc = t["hbm_bytes_per_launch"], "profiles/traffic_scan_c4.json (rocprofv3 PMC, per launch group of %d rows)" % t["units_per_launch"]
    # the whole H10 loop (filter + probe + hit kernels, summed): the fraction SURVEY 8(d)'s 44 B/k-mer budget is about
    roof_scan = {"kernel": sname, "bound": "hbm", "achieved": ach_scan, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_scan / HBM_PEAK_GBS,
                 "traffic": scan_traffic, "traffic_source": scan_tsrc, "algorithmic_bytes_per_launch": SCAN_BYTES_PER_KMER * rows_per_launch,
                 "bytes_per_unit": SCAN_BYTES_PER_KMER, "units_per_launch": rows_per_launch, "avg_launch_ms": scan_sum,
                 "launch_groups": n_groups,
                 "note": "the kernels of one scan launch group, back to back on one stream, averaged over the table's %d groups (roofline_filter: the passes "
                         "in front of the probe kernel alone)" % n_groups}
    roof_filter = {"kernel": fname, "bound": "hbm", "achieved": ach_filter, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_filter / HBM_PEAK_GBS,
                   "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": SCAN_BYTES_PER_KMER * rows_per_launch,
                   "bytes_per_unit": SCAN_BYTES_PER_KMER, "units_per_launch": rows_per_launch, "avg_launch_ms": filt}
    wl = {"c3": "C3", "c4": "C4 (whole genome, clustered panel as SURVEY 8(d) draws it)", "c5": "C5 (indel / MNP clusters, %s)" % ("haploid" if job.haploid else "diploid")}[job.workload]
    if job.workload == "c3" and (job.n_vars_total, job.b, K, R) != (1000000, 4, 35, 43):
        wl = "custom (C3 shape)"
    out = {
        "metric": "KMC k-mers scanned/sec (whole call step: scan + counter all-reduce + record loop: cut, signatures, coverage, likelihoods), k=%d r=%d" % (K, R),
        "value": total_kmers / elapsed,
        "unit": "kmers/s",
        "variants_per_s": total_vars / elapsed,
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "launch": getattr(job, "launch", "stream"),                      # how the timed steps were issued: kernel by kernel, or one captured step replayed
        "ms_per_step_stream": getattr(job, "stream_ms", None),           # (graph replay only) the same steps launched kernel by kernel
        "replay_equals_launched": getattr(job, "replay_equals_launched", None),   # ... and whether both left the same coverages, GT, GQ
        "higher_is_better": True,
        "scaling": "strong" if job.strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "%s: %.3g KMC k-mers per GPU (table of %.3g rows %s) against a panel of %.3g records, k=%d r=%d b=%d, table and panel resident in HBM" % (
                       wl, job.n_rows, job.total_rows, "cut over the ranks" if job.strong else "= rows per GPU x N", job.n_vars_total, K, R, job.b),
                   "kmers_per_gpu": job.n_rows, "kmers_total": job.total_rows, "panel_variants": job.n_vars_total, "variants_genotyped_per_gpu": job.n_vars,
                   "k": K, "ref_k": R, "bf_bits": job.bf_bits, "haploid": job.haploid,
                   "table_layout": "12-byte rows (count << 2r | r-mer), packed outside the step" if job.compact else "SoA hi[] lo[] cnt[] (20 B/row)",
                   "parallelism": "table rows x%d (%s), record loop split x%d at block boundaries, index replicated" % (world, "strong" if job.strong else "weak", world),
                   "summary_bitmaps": not args.no_summary,
                   "exchange": ("none" if world == 1 else "%s over %d counters%s" % (
                       job.exchange, job.n_bf + job.n_map, ", 16-bit packed when exact (%d of %d steps)" % (sum(job.packed_steps), len(job.packed_steps)) if job.packed_steps else "")),
                   "comm_ranks": (ctx.comm_info()[1] if job.native else (world if world > 1 else 0))},
        "roofline": roof_scan,
        "roofline_filter": roof_filter,
        "kernels_ms": kt,
        "pack_rows_ms": job.pack_ms,
        # the same step with the SoA -> 12-byte-row conversion counted in (no pipeline stage produces compact rows today: see DESIGN.md)
        "ms_per_step_incl_pack": (1e3 * elapsed / steps + job.pack_ms) if job.pack_ms is not None else None,
        "value_incl_pack": (job.total_rows / (elapsed / steps + job.pack_ms * 1e-3)) if job.pack_ms is not None else None,
        "index_build_s": job.index_s,
        "sustained": sustained,
    }
    if job.flat:
        n_sig = kt["lone_signature_kmers"] + kt["general_signature_kmers"]
        alg = blocks_bytes(job.n_vars, n_sig, job.n_genotypes)
        blocks_traffic, blocks_tsrc = None, None      # counted HBM bytes of one record loop (tools/traffic_blocks.sh), when this IS that workload
        for tname in ("traffic_blocks_%s.json" % job.workload, "traffic_blocks_%s_leg.json" % job.workload):   # (the workload at its full size; at the size of the default line's leg)
            tb = os.path.join(ROOT, "profiles", tname)
            if blocks_traffic is None and os.path.exists(tb) and world == 1:
                t = json.load(open(tb))
                if t.get("panel_variants") == job.n_vars_total and t.get("units_per_launch") == job.n_vars:
                    blocks_traffic, blocks_tsrc = t["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 PMC%s)" % (
                        tname, "; mean of the diploid and the haploid job's loops" if t.get("haploid_included") else "")
        loop_ms = kt["record_loop"]
        out["roofline_blocks"] = {"kernel": "cut_flags + flag_scatter + panel_lone + fw_walk + fw_snp + fw_order + fw_chain (+ fw_picks<true> / fw_eval for what it lists) + fw_slide + cover_blocks<0> + genotype (the record loop, main.cpp:522-579)",
                                  "bound": "hbm", "achieved": alg / (loop_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": blocks_traffic, "traffic_source": blocks_tsrc, "algorithmic_bytes_per_launch": alg,
                                  "bytes_per_unit": "64 + 36 K + 8 G (SURVEY 8(d))", "units_per_launch": job.n_vars, "signature_kmers": n_sig,
                                  "genotypes": job.n_genotypes, "avg_launch_ms": loop_ms, "variants_per_s_record_loop_alone": job.n_vars / (loop_ms * 1e-3)}
        if loop_ms > scan_sum * n_groups:     # the record loop is the larger part of this workload's step
            out["roofline"], out["roofline_scan"] = out["roofline_blocks"], roof_scan
    else:
        out["roofline_blocks_c3"] = {"kernel": "cut_flags + flag_scatter + panel_lone<false> + genotype (the general resident record loop on isolated SNPs)",
                                     "avg_launch_ms": kt["record_loop"], "isolated_entry_ms": kt.get("record_loop_isolated")}
        out["genotype_roofline"] = {"achieved": GENO_BYTES_PER_SNP * job.n_vars / (kt["record_loop"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": GENO_BYTES_PER_SNP * job.n_vars / (kt["record_loop"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_unit": GENO_BYTES_PER_SNP}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    if args.launch_fail_rank >= 0:
        if int(os.environ.get("RANK", "0")) == args.launch_fail_rank:
            sys.exit(3)
        time.sleep(60)
        return
    if args.launch_check:
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"launch_check": True, "world": int(os.environ.get("WORLD_SIZE", "1")), "master": os.environ.get("MASTER_ADDR"),
                              "port": int(os.environ.get("MASTER_PORT", "0")), "torch_imported": "torch" in sys.modules}))
        return
    # stdout carries exactly one line, the JSON: whatever a library prints there (gloo's rank banner, RCCL warnings)
    # goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from malva_amd.dist import rank_world

    rank, world = rank_world()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import datetime
        patience = datetime.timedelta(minutes=10)       # (a rank that died leaves the others in a collective: fail in minutes, not after the backends' half hour)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", timeout=patience)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=patience)
    # one explicit stream for the library's kernels AND torch's work on the aliased counters (a NULL handle would
    # mean the library's private stream: torch's default stream has handle 0)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)

    def measure(job, cpu):
        elapsed = job.timed(args.steps, args.warmup)
        kt = job.kernel_times(max(3, min(args.steps, 10)))
        sus = job.sustained(args.sustained_s)
        g1, g2, gq, n_ovf = job.results()
        out = None
        if rank == 0:
            out = record(job, elapsed, args.steps, args.warmup, kt, sus)
            if job.haploid:
                out["calls"] = {"0": int(np.sum(g1 == 0)), "non-reference": int(np.sum(g1 > 0))}
            else:
                out["calls"] = {"0/0": int(np.sum((g1 == 0) & (g2 == 0))), "het": int(np.sum(g1 != g2)), "hom-alt": int(np.sum((g1 == g2) & (g1 > 0)))}
            out["overflow_records"] = n_ovf
            if n_ovf:
                raise SystemExit("bench: %d records exceeded a device capacity at call time: not a measurement of the device path" % n_ovf)
            out["cpu_baseline"], out["parity_sample"] = (None, None)
            if cpu and world == 1 and args.cpu_sample > 0:
                out["cpu_baseline"], out["parity_sample"] = cpu_leg(job, rank)
        return out

    job = Job(args.workload, args, rank, world, local, torch, dist, haploid=False)
    out = measure(job, cpu=True)
    job.close()
    if args.workload == "c5":      # config C5 names both modes: the haploid index and step, same panel arrays
        hj = Job("c5", args, rank, world, local, torch, dist, haploid=True)
        hout = measure(hj, cpu=True)
        hj.close()
        if rank == 0:
            out["haploid"] = {k_: hout[k_] for k_ in ("value", "variants_per_s", "ms_per_step", "kernels_ms", "roofline", "roofline_blocks", "sustained",
                                                      "calls", "cpu_baseline", "parity_sample", "index_build_s")}
    if world == 1 and args.workload == "c3" and not args.no_c5_leg and not args.scan_ablate:
        # BASELINE config C5's shape at a fifth of its bench size, so that the default line carries the general-block path too
        # (`--workload c5` is the full measurement): diploid and haploid, parity of everything against the oracle
        leg = {}
        for hap in (False, True):
            lj = Job("c5", args, rank, world, local, torch, dist, haploid=hap, kmers=args.c5_leg_kmers, b=8, strong=False, clusters=args.c5_leg_clusters,
                     plant_records=5e4)
            el = lj.timed(args.steps, args.warmup)
            lkt = lj.kernel_times(3)
            _, _, _, n_ovf = lj.results()
            rec = record(lj, el, args.steps, args.warmup, lkt, None)
            cpu, par = cpu_leg(lj, rank) if args.cpu_sample > 0 else (None, None)
            leg["haploid" if hap else "diploid"] = {
                "ms_per_step": rec["ms_per_step"], "variants_per_s": rec["variants_per_s"], "value": rec["value"], "kernels_ms": lkt, "roofline_blocks": rec["roofline_blocks"],
                "overflow_records": n_ovf, "parity_sample": par, "cpu_variants_per_s": cpu and cpu["variants_per_s"], "config": rec["config"]["workload"]}
            lj.close()
        out["general_blocks_c5"] = leg
    if world == 1 and args.workload == "c3" and not args.no_c4_leg and not args.scan_ablate:
        # BASELINE config C4 (whole genome: 3e9 k-mers against the 8e7-record index, b=16) on this one GPU -- the per-GPU rate of
        # the 8-GPU configuration too, whose index is replicated: step, kernel times, roofline of the scan and of the record
        # loop, and parity of three slices of the panel at the full index
        cj = Job("c4", args, rank, world, local, torch, dist, kmers=args.c4_leg_kmers, variants=args.c4_leg_variants, b=16, strong=True)
        cst = max(2, min(args.steps, 5))
        el = cj.timed(cst, 1)
        ckt = cj.kernel_times(2)
        _, _, _, n_ovf = cj.results()
        crec = record(cj, el, cst, 1, ckt, None)
        out["c4_whole"] = {k_: crec[k_] for k_ in ("value", "variants_per_s", "ms_per_step", "steps", "warmup", "n_gpus", "scaling", "config", "kernels_ms", "roofline", "roofline_filter",
                                                   "roofline_blocks", "pack_rows_ms", "ms_per_step_incl_pack", "index_build_s")}
        out["c4_whole"]["roofline_scan"] = crec.get("roofline_scan", crec["roofline"])
        out["c4_whole"]["overflow_records"] = n_ovf
        out["c4_whole"]["parity_slices"] = c4_slice_parity(cj, rank) if args.cpu_sample > 0 else None
        cj.close()
    # (under --rehearse-on-one-gpu the leg runs only at explicitly reduced sizes: N whole-genome indexes do not share one GPU)
    if world > 1 and args.workload == "c3" and not args.no_strong_c4 and (not args.rehearse_on_one_gpu or args.strong_c4_kmers < 1e9):
        # north_star's whole-genome claim on the same ranks: the 3e9-row table cut N ways against the 8e7-SNP index
        def all_ok(ok):
            """every rank abandons the leg together: a failure on one rank only (an out-of-memory, say) must not leave its peers inside a collective"""
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if args.rehearse_on_one_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        sj, err = None, None
        try:
            sj = Job("c4", args, rank, world, local, torch, dist, kmers=args.strong_c4_kmers, variants=args.strong_c4_variants, b=16, strong=True)
        except Exception as e:      # noqa: BLE001 -- reported in the line, never silent
            err = "%s: %s" % (type(e).__name__, e)
        if all_ok(sj is not None):
            st = max(2, min(args.steps, 5))
            el = skt = n_ovf = None
            try:
                el = sj.timed(st, 1)        # (its collectives are matched on every rank: all of them got here)
                skt = sj.kernel_times(3)
                _, _, _, n_ovf = sj.results()
            except Exception as e:      # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, e)
            if rank == 0:
                if err is None:
                    srec = record(sj, el, st, 1, skt, None)
                    out["strong_c4"] = {k_: srec[k_] for k_ in ("value", "variants_per_s", "ms_per_step", "n_gpus", "scaling", "config", "kernels_ms", "roofline", "roofline_blocks")}
                    out["strong_c4"]["roofline_scan"] = srec.get("roofline_scan", srec["roofline"])
                    out["strong_c4"]["overflow_records"] = n_ovf
                    out["strong_c4"]["one_gpu_reference"] = "the c4_whole leg of this driver's N = 1 line (same job on one GPU); no speed-up is computed here"
                else:
                    out["strong_c4"] = {"error": err}
        elif rank == 0:
            out["strong_c4"] = {"error": err or "the job could not be built on some rank (see stderr)"}
        if sj is not None:
            sj.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
