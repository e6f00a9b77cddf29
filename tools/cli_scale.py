#!/usr/bin/env python3
"""End-to-end timing of bin/malva-geno on a C3-shaped input that goes through FILES (FASTA, VCF, k-mer dump), i.e.
including the host side the bench.py step leaves out: VCF parsing, block building, batching, VCF output.
usage (on a GPU box): python tools/cli_scale.py [--variants 1e6] [--kmers 2e6] [--samples 2] [--dir /tmp/cli_scale]
Prints the CLI's own phase lines and variants/s for `index` and `call`."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from malva_amd import synth  # noqa: E402

K, R = 35, 43


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=float, default=1e6)
    ap.add_argument("--kmers", type=float, default=2e6)
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--spacing", type=int, default=100, help="bases between SNPs (below k/2 + 1 = 18 neighbours share k-mers: general blocks)")
    ap.add_argument("--b", type=int, default=4)
    ap.add_argument("--dir", default="/tmp/cli_scale")
    ap.add_argument("--no-run", action="store_true", help="only write the input files")
    ap.add_argument("--c5", type=int, default=0, metavar="CLUSTERS",
                    help="config C5 shape instead: CLUSTERS clusters of SNPs / MNPs / indels / multi-allelic records (tests/vcf_synth.py), "
                         "k=35 r=63 b=8, run haploid and diploid")
    args = ap.parse_args()
    if args.c5:
        return c5(args)
    n, nk = int(args.variants), int(args.kmers)
    os.makedirs(args.dir, exist_ok=True)
    fa, vcf, prefix = (os.path.join(args.dir, x) for x in ("ref.fa", "panel.vcf", "sample.kmercount"))
    t0 = time.time()
    panel = synth.snp_panel(n, seed=20261003, spacing=args.spacing)
    g = panel.genome.tobytes().decode()
    with open(fa, "w") as fh:
        fh.write(">1\n")
        for i in range(0, len(g), 1 << 20):
            fh.write(g[i:i + (1 << 20)] + "\n")
    rng = np.random.default_rng(5)
    # phased diploid samples; sample 0 carries 0|1 everywhere so that both alleles are present in the panel; the
    # others carry the ALT allele with the record's AF (a panel is mostly 0|0)
    af = panel.freq[1::2]
    ref = panel.pool[0::2].tobytes().decode()
    alt = panel.pool[1::2].tobytes().decode()
    with open(vcf, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"af\">\n"
                 "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##contig=<ID=1,length=%d>\n" % len(g))
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(args.samples)) + "\n")
        step = max(1, (64 << 20) // (4 * args.samples))
        for a in range(0, n, step):
            b = min(n, a + step)
            gts = (rng.random((b - a, args.samples, 2)) < af[a:b, None, None]).astype(np.uint8)
            gts[:, 0, 0], gts[:, 0, 1] = 0, 1
            cells = np.empty((b - a, args.samples, 4), dtype=np.uint8)   # "a|b\t"
            cells[:, :, 0] = 48 + gts[:, :, 0]
            cells[:, :, 1] = ord("|")
            cells[:, :, 2] = 48 + gts[:, :, 1]
            cells[:, :, 3] = ord("\t")
            flat = cells.reshape(b - a, -1)
            out = []
            for i in range(a, b):
                out.append("1\t%d\t.\t%s\t%s\t.\t.\tAF=%.4f\tGT\t" % (panel.pos[i] + 1, ref[i], alt[i], af[i]))
                out.append(flat[i - a, :-1].tobytes().decode())
                out.append("\n")
            fh.write("".join(out))
    hi, lo, cnt = synth.kmer_table(panel, nk, K, R, seed=777)
    rows = synth.unpack_ascii(hi, lo, R)
    with open(prefix + ".txt", "w") as fh:
        for i in range(0, nk, 100000):
            fh.write("".join("%s\t%d\n" % (rows[j, :R].tobytes().decode(), cnt[j]) for j in range(i, min(nk, i + 100000))))
    print("inputs: %d SNPs x %d samples (%.0f MB VCF), %d k-mers as text, %.0f s" %
          (n, args.samples, os.path.getsize(vcf) / 1e6, nk, time.time() - t0), flush=True)
    if args.no_run:
        return
    common = ["-k", str(K), "-r", str(R), "-b", str(args.b), "-f", "AF", fa, vcf, prefix]
    for sub in ("index", "call"):
        t0 = time.time()
        with open(os.path.join(args.dir, "out.vcf"), "w") as so:
            r = subprocess.run([os.path.join(ROOT, "bin", "malva-geno"), sub] + common, stdout=so, stderr=subprocess.PIPE, text=True,
                               env=dict(os.environ, MALVA_GENO_TIMERS="1"))
        dt = time.time() - t0
        if r.returncode:
            print(r.stderr[-2000:])
            raise SystemExit("malva-geno %s failed" % sub)
        phases = [l for l in r.stderr.split("\n") if ("Execution Time" in l and "000 variants]" not in l) or "on the device" in l or "on the host" in l or "/timer]" in l]
        print("== malva-geno %s: %.2f s wall = %.3g variants/s\n   %s" % (sub, dt, n / dt, "\n   ".join(phases)), flush=True)
    nrec = sum(1 for l in open(os.path.join(args.dir, "out.vcf")) if not l.startswith("#"))
    print("records written by call: %d" % nrec)


def c5(args):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vcf_synth
    os.makedirs(args.dir, exist_ok=True)
    for haploid in (True, False):
        prefix = os.path.join(args.dir, "c5h" if haploid else "c5d")
        t0 = time.time()
        contigs, records = vcf_synth.make_case(prefix, 77, n_clusters=args.c5, haploid=haploid, n_samples=8, k=K, vcf_strip_chr=True)
        vcf_synth.donor_table(contigs, records, 63, 77, prefix + ".kmers.txt")
        n = len(records)
        print("C5 %s inputs: %d records in %d clusters, %d-base genome, %d k-mers, %.0f s" %
              ("haploid" if haploid else "diploid", n, args.c5, sum(len(s_) for s_ in contigs.values()),
               sum(1 for _ in open(prefix + ".kmers.txt")), time.time() - t0), flush=True)
        common = ["-k", str(K), "-r", "63", "-b", "8", "-p"] + (["-1"] if haploid else []) + [prefix + ".fa", prefix + ".vcf", prefix + ".kmers"]
        for sub in ("index", "call"):
            t0 = time.time()
            with open(prefix + ".out.vcf", "w") as so:
                r = subprocess.run([os.path.join(ROOT, "bin", "malva-geno"), sub] + common, stdout=so, stderr=subprocess.PIPE, text=True)
            dt = time.time() - t0
            if r.returncode:
                print(r.stderr[-2000:])
                raise SystemExit("malva-geno %s failed" % sub)
            notes = [l for l in r.stderr.split("\n") if "on the host" in l or "on the device" in l]
            print("== malva-geno %s: %.2f s wall = %.3g records/s %s" % (sub, dt, n / dt, " ".join(notes)), flush=True)


if __name__ == "__main__":
    main()
