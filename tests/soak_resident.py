#!/usr/bin/env python3
"""Test infrastructure (not collected by pytest; lives here because it uses the oracle).  One-off soak of the resident record loop:
many random C5-shaped panels -- 1 to 70 samples (every lane-group width of tier 2, and panels wider than a wave), any share of
unphased genotypes, haploid and diploid, dense and sparse genotype layout, with tier 2's kernels switched on and off at random --
through index + scan + cut / cover / genotype on the device against the C oracle (tests/test_gpu_resident.py: run_recipe).
usage: python tests/soak_resident.py [first_seed] [n]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from malva_amd import synth  # noqa: E402
from test_gpu_resident import run_recipe  # noqa: E402

first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    n_samples = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 70]))
    haploid = bool(seed & 1)
    sparse = bool(seed & 2) and n_samples >= 4
    unphased = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
    hom_ref = float(rng.choice([0.2, 0.45, 0.8, 0.97]))
    k, ref_k = [(35, 63), (35, 43), (31, 41), (21, 29)][(seed >> 2) & 3]
    options = [("use_chain_kernel", int(rng.integers(0, 4) > 0)), ("use_chain_order", int(rng.integers(0, 3) > 0)), ("use_snp_kernel", int(rng.integers(0, 3) > 0)),
               ("use_packed_pool", int(rng.integers(0, 3) > 0))]
    what = "seed %d: %d samples, %s, %s genotypes, unphased %.1f, hom_ref %.2f, k%d r%d, %s" % (
        seed, n_samples, "haploid" if haploid else "diploid", "sparse" if sparse else "dense", unphased, hom_ref, k, ref_k, " ".join("%s=%d" % o for o in options))
    try:
        panel = synth.indel_panel(2500, seed=seed, k=k, n_samples=n_samples, unphased_frac=unphased, hom_ref=hom_ref)
        run_recipe(panel, k, ref_k, haploid, 1 << 26, n_rows=150_000, plant=1_500, min_general=500, sparse=sparse, options=options)
        print("ok   " + what, flush=True)
    except AssertionError as e:
        where = traceback.extract_tb(e.__traceback__)[-1]
        if "plant //" in (where.line or ""):     # (the recipe's own sanity floor on how much the planted windows cover: comes after the comparisons; a panel of one haploid sample is below it)
            print("ok   " + what + "  (few covered alleles)", flush=True)
            continue
        bad += 1
        print("FAIL " + what + " :: line %d: %s %s" % (where.lineno, where.line, str(e)[:200]), flush=True)
print("%d panels, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
