#!/usr/bin/env python3
"""Regenerates the 128-entry table of glibc's exp() (ARM optimized-routines math/exp.c, N = 128):
2^(k/N) = H[k] * (1 + T[k]);  tab[2k] = bits(T[k]),  tab[2k+1] = bits(H[k]) - (k << 52) / N.
The table in malva_amd/csrc/geno_dev.h was produced by this script and then checked against the host's
libm with tools/check_exp_restatement.c (gcc -O2 -ffp-contract=off -mfma -DUSE_FMA -DSC_NOFMA):
0 mismatches in 6e7 random inputs over [-760, 20]."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 80
N = 128


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


for k in range(N):
    v = Decimal(2) ** (Decimal(k) / Decimal(N))
    h = float(v)
    t = float(v / Decimal(h) - 1)
    print("0x%016xULL, 0x%016xULL," % (bits(t), (bits(h) - ((k << 52) // N)) & 0xFFFFFFFFFFFFFFFF))
