#!/usr/bin/env python3
"""Test infrastructure (not collected by pytest; lives here because it uses the oracle). One-off soak: many random clustered panels through bin/malva-geno (GPU) vs the oracle pipeline.
usage: python tests/soak_cli.py [first_seed] [n]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vcf_synth  # noqa: E402
from oracle import pipeline  # noqa: E402

BIN = os.path.join(ROOT, "bin", "malva-geno")
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + n):
    if os.environ.get("SOAK_ONLY_FAR") and seed % 8 != 7:
        continue
    haploid, dense, verbose = bool(seed & 1), bool(seed & 2), bool(seed & 4)
    k, ref_k = [(35, 43), (31, 41), (35, 63), (25, 33)][(seed >> 3) & 3]
    with tempfile.TemporaryDirectory() as d:
        prefix = os.path.join(d, "c")
        far = seed % 8 == 7 and not os.environ.get("SOAK_NO_FAR")   # one case in eight: a 35 Mb contig, clusters beyond position 2^25 (are_near in float)
        if far:
            seq, records, _ = vcf_synth.make_far_case(prefix, seed, k=k, n_clusters=150, n_samples=3 + seed % 4, haploid=haploid)
            contigs = {"1": seq}
        else:
            contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=k, n_clusters=30 if dense else 80, dense=dense,
                                                   n_samples=3 + seed % 6, vcf_strip_chr=True)
        table = os.path.join(d, "t")
        vcf_synth.donor_table(contigs, records, ref_k, seed, table + ".txt")
        opt = pipeline.Options(haploid=haploid, verbose=verbose, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
        idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
        kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
        want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
        args = ["-k", str(k), "-r", str(ref_k), "-b", "1", "-p"] + (["-1"] if haploid else []) + (["-v"] if verbose else [])
        args += [prefix + ".fa", prefix + ".vcf", table]
        env = dict(os.environ, MALVA_GENO_CUT_BATCH=str(1 + seed % 13)) if seed % 3 == 0 else dict(os.environ)   # block-cut batch seams everywhere
        if seed % 5 in (1, 2):       # the sample columns decoded on the device (what a panel of >= 1024 samples gets by itself)
            env.update(MALVA_GENO_GT_DEVICE="1", MALVA_GENO_VCF_POOL="1")
        elif seed % 5 == 3:          # the reader's thread pool over blocks of whole lines (what a file of >= 8 MB gets by itself)
            env.update(MALVA_GENO_VCF_POOL="1")
        r = subprocess.run([BIN, "index"] + args, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-500:]
        r = subprocess.run([BIN, "call"] + args, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-500:]
        ok = r.stdout == want
        nonref = sum(1 for l in want.split("\n") if l and not l.startswith("#") and not l.split("\t")[-1].startswith(("0:", "0/0:")))
        print("seed %d far=%d haploid=%d dense=%d verbose=%d k=%d r=%d records=%d nonref=%d %s %s" % (
            seed, far, haploid, dense, verbose, k, ref_k, want.count("\n"), nonref, "OK" if ok else "MISMATCH",
            ("(host blocks)" if "enumerated on the host" in r.stderr else "") + (" gt-device" if env.get("MALVA_GENO_GT_DEVICE") == "1" else "") +
            (" pool" if env.get("MALVA_GENO_VCF_POOL") == "1" else "")), flush=True)
        if not ok:
            bad += 1
            for a, b in zip(r.stdout.split("\n"), want.split("\n")):
                if a != b:
                    print("  got :", a[:300]); print("  want:", b[:300]); break
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
