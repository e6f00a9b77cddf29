#!/usr/bin/env python3
"""Generates tests/golden/sars_cov2.oracle.malva.vcf.gz: the full expected output of BASELINE config C1
(example/reference_sarsCov2.fasta + example/sars_cov2.vcf.gz, -1 -k 35 -r 43 -b 1 -f AF, the haploid example's reads
counted by oracle/kmc_standin.py as the sample) through oracle/pipeline.py -- the CPU restatement of
main.cpp:251-594.  Run once in the build container (pure Python over 15,154 records x 27,934 samples: slow);
the committed file is what tests/test_gpu_configs.py compares `bin/malva-geno` with, byte for byte.
Provenance of the two calls this output must contain (17747 C>T 1:94, 17858 A>G 1:100): the compiled reference,
SURVEY.md section 8(c) item 3."""
import gzip
import os
import sys
import time

import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> the repository root
sys.path.insert(0, ROOT)
from oracle import kmc_standin, model, pipeline  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")

# The panel's three giant blocks (up to 8,724 variants x 27,934 samples) take the pure-Python model hours on one core.
# VB.extract_kmers enumerates a block's variants independently of one another (oracle/model.py: extract_one), so a large
# block is spread over forked worker processes here -- the restated logic runs unchanged, one variant at a time.
_JOB = {}


def _work(vis):
    vb, reference, haploid = _JOB["vb"], _JOB["reference"], _JOB["haploid"]
    return [(vi, vb.extract_one(vi, reference, haploid)) for vi in vis]


def _parallel_extract(self, reference, haploid):
    n = len(self.variants)
    workers = max(1, min(int(os.environ.get("GEN_WORKERS", "7")), n // 32))
    if workers == 1:
        return {vi: self.extract_one(vi, reference, haploid) for vi in range(n)}
    _JOB.update(vb=self, reference=reference, haploid=haploid)
    chunks = [list(range(i, n, workers * 4)) for i in range(workers * 4)]        # interleaved: chain counts vary along a block
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_work, chunks)
    out = {}
    for part in parts:
        out.update(part)
    print("  block of %d variants enumerated by %d processes (%.0f s since start)" % (n, workers, time.time() - T0), file=sys.stderr, flush=True)
    return {vi: out[vi] for vi in range(n)}


T0 = time.time()
model.VB.extract_kmers = _parallel_extract

# index and call each decode the whole panel (850 MB of GT text): decode it once per (samples, key) and hand the records out again
_records = model.VCFReader.records
_decoded = {}


def _records_once(self, freq_key="AF", uniform=False):
    key = (self.path if hasattr(self, "path") else id(self), tuple(self.keep), freq_key, uniform)
    if key not in _decoded:
        _decoded[key] = list(_records(self, freq_key, uniform))
        print("  panel decoded: %d records (%.0f s since start)" % (len(_decoded[key]), time.time() - T0), file=sys.stderr, flush=True)
    for v in _decoded[key]:
        v.coverages = []
        yield v


model.VCFReader.records = _records_once


def main(verbose):
    opt = pipeline.Options(haploid=True, verbose=verbose, k=35, ref_k=43, bf_size=1 << 33, freq_key="AF")
    t0 = time.time()
    fa, vcf = os.path.join(G, "reference_sarsCov2.fasta"), os.path.join(G, "sars_cov2.vcf.gz")
    idx = pipeline.index(fa, vcf, opt)
    print("index: %.0f s" % (time.time() - t0), file=sys.stderr, flush=True)
    kmers = list(kmc_standin.count_fastq(os.path.join(G, "haploid.fq"), 43))
    t0 = time.time()
    out = pipeline.call(fa, vcf, idx, kmers, opt)
    print("call: %.0f s, %d lines" % (time.time() - t0, out.count("\n")), file=sys.stderr, flush=True)
    name = "sars_cov2.oracle.malva%s.vcf.gz" % (".verbose" if verbose else "")
    with gzip.GzipFile(os.path.join(G, name), "wb", compresslevel=9, mtime=0) as fh:
        fh.write(out.encode())
    print("wrote", name, file=sys.stderr)


if __name__ == "__main__":
    main("-v" in sys.argv)
