"""the KMC database read directly.  The product side is malva_amd/host/kmc_db.hpp (file layout) +
kmc_decode_kernel (records -> table rows on the device) behind mg_kmc_set_lut / mg_kmc_scan_records; the checker is
oracle/kmc_db.py, an independent reading of the same published layout (PARITY UNPINNED: no database written by KMC
itself is available; see its header)."""
import os
import random

import numpy as np
import pytest

from gpu_util import build_index_pair, map_values_by_key
from malva_amd import BF_ALT, Context, synth
from oracle import capi as ocapi
from oracle import kmc_db
from test_gpu_cli import run_cli
import vcf_synth
from oracle import pipeline

pytestmark = pytest.mark.gpu


def _random_kmers(n, k, seed, lo=1, hi=300):
    rng = random.Random(seed)
    out = {}
    while len(out) < n:
        out[bytes(rng.choice(b"ACGT") for _ in range(k))] = rng.randint(lo, hi)
    return list(out.items())


@pytest.mark.parametrize("k,p,bins,cs,n", [(43, 7, 3, 1, 3000),        # sparse table: tiles span thousands of empty prefixes
                                           (43, 3, 2, 2, 200000),      # dense table: every tile inside its LDS window
                                           (43, 7, 2, 1, 400000),
                                           (63, 7, 2, 4, 50000), (41, 5, 5, 1, 70000), (21, 1, 1, 3, 9000)])
def test_device_decode_equals_the_listing(tmp_path, k, p, bins, cs, n):
    items = _random_kmers(n, k, 1000 * k + p)
    path = str(tmp_path / "db")
    kmc_db.write_db(path, items, k, lut_prefix_len=p, n_bins=bins, counter_size=cs)
    db = kmc_db.KmcDb(path)
    whi, wlo, wcnt = db.table()
    with Context(min(k, 35), k, 1 << 20) as ctx:
        ctx.kmc_set_lut(db.lut, db.lut_prefix_len, db.suffix_bytes, db.counter_size, db.min_count, db.max_count, db.total)
        hi, lo, cnt = ctx.kmc_decode_records(db.records)
        assert np.array_equal(hi, whi) and np.array_equal(lo, wlo) and np.array_equal(cnt, wcnt)
        # a run that starts in the middle of the database (what each device of a multi-GPU call gets)
        a = n // 3 + 5
        hi, lo, cnt = ctx.kmc_decode_records(db.records[a:], first_record=a)
        assert np.array_equal(hi, whi[a:]) and np.array_equal(lo, wlo[a:]) and np.array_equal(cnt, wcnt[a:])


def test_lut_contract_is_checked(tmp_path):
    path = str(tmp_path / "db")
    kmc_db.write_db(path, _random_kmers(100, 43, 3), 43)
    db = kmc_db.KmcDb(path)
    from malva_amd.capi import MalvaError
    with Context(35, 43, 1 << 20) as ctx:
        with pytest.raises(MalvaError):                          # records before the table
            buf = np.zeros(16, dtype=np.uint8)
            ctx._ck(ctx._L.mg_kmc_scan_records(ctx.h, buf.ctypes.data, 1, 0))
        bad = db.lut.copy()
        bad[5] = bad[-1] + 10**6
        with pytest.raises(MalvaError):
            ctx.kmc_set_lut(bad, db.lut_prefix_len, db.suffix_bytes, db.counter_size, 2, 255, db.total)
    with Context(35, 45, 1 << 20) as ctx:                       # -r 45 against a 43-mer database
        with pytest.raises(MalvaError):
            ctx.kmc_set_lut(db.lut, db.lut_prefix_len, db.suffix_bytes, db.counter_size, 2, 255, db.total)


def test_scan_from_records_equals_scan_from_rows_and_oracle(tmp_path):
    k, ref_k, bits = 35, 43, 1 << 20
    panel = synth.snp_panel(3000, 91)
    hi, lo, cnt = synth.kmer_table(panel, 150000, k, ref_k, 92)
    # as a database: distinct k-mers with counts 1..255 (min_count 2: the count-1 records are skipped by the listing)
    seen, items = set(), []
    text = synth.unpack_ascii(hi, lo, ref_k)
    for row, c in zip(text, cnt):
        km = bytes(row[:ref_k])
        if km not in seen:
            seen.add(km)
            items.append((km, int(c) % 255 + 1))
    path = str(tmp_path / "db")
    kmc_db.write_db(path, items, ref_k, n_bins=4)
    db = kmc_db.KmcDb(path)
    thi, tlo, tcnt = db.table()
    with Context(k, ref_k, bits) as ctx:
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        ocapi.kmc_scan_packed(octx, obf, omap, thi, tlo, tcnt, k, ref_k)
        ctx.kmc_set_lut(db.lut, db.lut_prefix_len, db.suffix_bytes, db.counter_size, db.min_count, db.max_count, db.total)
        ctx.kmc_scan_records(db.records)
        _, _, _, counts = ctx.bf_export(BF_ALT)
        assert np.array_equal(counts, obf.counts()) and counts.any()
        assert map_values_by_key(ctx) == dict(omap.items())
        ctx.counters_reset()
        ctx.kmc_scan(thi, tlo, tcnt)                               # the SoA form of the same listing
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], counts)


def test_cli_reads_the_database_like_the_text_dump(tmp_path):
    seed, k, ref_k = 31, 35, 43
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=False, k=k, n_clusters=60, vcf_strip_chr=True)
    dump = str(tmp_path / "dump.kmers")
    vcf_synth.donor_table(contigs, records, ref_k, seed, dump + ".txt")
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(dump + ".txt")]
    args = ["-k", str(k), "-r", str(ref_k), "-b", "1", "-p", "-v", prefix + ".fa", prefix + ".vcf"]
    run_cli(["index"] + args + [dump])
    from_text = run_cli(["call"] + args + [dump])
    dbp = str(tmp_path / "sample")
    kmc_db.write_db(dbp, kmers, ref_k, n_bins=3, min_count=1)
    assert not os.path.exists(dbp + ".txt")
    from_db = run_cli(["call"] + args + [dbp])
    assert from_db == from_text
    assert sum(1 for l in from_db.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 20
    import torch
    share = {} if torch.cuda.device_count() >= 2 else {"MALVA_GENO_SHARE_DEVICE": "1"}
    assert run_cli(["call", "--gpus", "2"] + args + [dbp], env=dict(os.environ, **share)) == from_text
    # the oracle pipeline fed with the oracle's own listing of the database agrees as well
    opt = pipeline.Options(haploid=False, verbose=True, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmc_db.KmcDb(dbp).kmers(), opt)
    strip = lambda s: "\n".join(";".join(p for p in l.split(";") if not p.startswith("GTS=")) if "GTS=" in l else l for l in s.split("\n"))
    assert strip(from_db) == strip(want)
