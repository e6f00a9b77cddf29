"""malva_amd -- MI355X-native k-mer matching and genotyping hot path of malva-geno.

The product is the HIP library `malva_amd/lib/libmalva_hip.so` (C ABI in
include/malva_hip.h) and the C++17 driver `bin/malva-geno`.  This package is
the thin ctypes binding used by the tests and by bench.py; it has no CPU
implementation of anything and raises if the library is missing.
"""
from .capi import MalvaError, Context, BF_ALT, BF_CTX, library_path  # noqa: F401
