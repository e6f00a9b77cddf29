"""malva_amd/host/index_file.hpp (the product's reader/writer of the reference's .malvax.zst container, and of this
build's compact .hipz) against oracle/index_file.py, an independent restatement of BF::operator>>/<<, KMAP::operator>>/<<
and sdsl's int_vector serialisation.  Host code only (`malva-geno index-convert`): runs without a GPU.
FORMAT UNPINNED: no index written by the reference binary exists to compare with (oracle/index_file.py header)."""
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import capi as ocapi
from oracle import index_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "malva-geno")


def _index(bits, n, seed, irregular=True):
    rng = random.Random(seed)
    ctx, bf, kmap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
    for _ in range(n):
        ctx.add_key(bytes(rng.choice(b"ACGT") for _ in range(43)))
        km = bytes(rng.choice(b"ACGT") for _ in range(35))
        bf.add_key(km)
        kmap.add_key(bytes(rng.choice(b"ACGT") for _ in range(35)))
    if irregular:
        kmap.add_key(b"ACGTNACGTACGTACGTACGTACGTACGTACGTACG")       # an N survives canonicalisation
        kmap.add_key(b"ACGWACGTT")                                    # KMAP::canonical cuts at the NUL: key "AACGT"
    bf.switch_mode()
    ctx.switch_mode()
    for i, km in enumerate(k for k, _ in kmap.items()):
        kmap.increment(km, (i * 7919) % 100000 - 5)                   # values, some of them negative
    return ctx, bf, kmap


@pytest.mark.parametrize("bits,compress", [((1 << 22) + 77, True), (1 << 20, False), ((1 << 33) + 0, True)])
def test_reference_container_round_trips_through_the_host_code(tmp_path, bits, compress):
    if not os.path.exists(BIN):
        pytest.fail("bin/malva-geno not built: run `make cli`")
    ctx, bf, kmap = _index(bits, 400, bits % 1000)
    vcf = str(tmp_path / "p.vcf")
    zst = vcf + ".c43.k35.malvax.zst"
    index_file.write_index(zst, ctx, bf, kmap, compress=compress)     # an uncompressed file is accepted like zstd::ifstream does
    env = dict(os.environ, MALVA_GENO_BF_BITS=str(bits))
    run = lambda fmt: subprocess.run([BIN, "index-convert", "-k", "35", "-r", "43", "x.fa", vcf, fmt], env=env, capture_output=True, text=True, timeout=600)
    r = run("hipz")
    assert r.returncode == 0, r.stderr
    os.remove(zst)
    r = run("zst")                                                      # and back, now written by the product
    assert r.returncode == 0, r.stderr
    assert open(zst, "rb").read(4) == b"\x28\xb5\x2f\xfd"              # one zstd stream
    filters, keys = index_file.read_index(zst)
    for (mode, size, pos, counts), src in zip(filters, (ctx, bf)):
        assert (mode, size) == (1, bits)
        assert np.array_equal(pos, src.set_positions())
        assert np.array_equal(counts, src.counts()) and counts.size == src.nset
    assert keys == dict(kmap.items()) and b"AACGT" in keys


def test_wrong_size_and_truncation_are_refused(tmp_path):
    ctx, bf, kmap = _index(1 << 20, 50, 3, irregular=False)
    vcf = str(tmp_path / "p.vcf")
    zst = vcf + ".c43.k35.malvax.zst"
    index_file.write_index(zst, ctx, bf, kmap)
    run = lambda bits: subprocess.run([BIN, "index-convert", "x.fa", vcf, "hipz"], env=dict(os.environ, MALVA_GENO_BF_BITS=str(bits)),
                                      capture_output=True, text=True, timeout=600)
    r = run(1 << 21)
    assert r.returncode != 0 and "another -b" in r.stderr
    data = open(zst, "rb").read()
    open(zst, "wb").write(data[:len(data) // 2])
    r = run(1 << 20)
    assert r.returncode != 0 and "ERROR" in r.stderr


def test_compact_container_versions_and_corruption(tmp_path):
    """the compact container, versions 3 and 4 (sections of independently compressed chunks; 4 adds the tie to the panel): a flipped
    byte or a cut file is an error, never a crash or a silent index; a version 2 file (one gzip stream, what earlier builds wrote) is
    still read"""
    import gzip
    import struct
    bits = 1 << 22
    ctx, bf, kmap = _index(bits, 3000, 11)
    vcf = str(tmp_path / "p.vcf")
    zst, hipz = vcf + ".c43.k35.malvax.zst", vcf + ".c43.k35.malvax.hipz"
    index_file.write_index(zst, ctx, bf, kmap)
    env = dict(os.environ, MALVA_GENO_BF_BITS=str(bits))
    run = lambda fmt: subprocess.run([BIN, "index-convert", "-k", "35", "-r", "43", "x.fa", vcf, fmt], env=env, capture_output=True, text=True, timeout=600)
    assert run("hipz").returncode == 0
    good = open(hipz, "rb").read()
    assert good[:8] == b"MGHIPX4\n"                                   # version 4 = version 3 + the panel's size and modification time behind the header
    os.remove(zst)
    open(hipz, "wb").write(b"MGHIPX3\n" + good[8:80] + good[96:])      # the same index as version 3 (no tie): still read
    assert run("zst").returncode == 0
    assert index_file.read_index(zst)[1] == dict(kmap.items())
    os.remove(zst)
    for at in (len(good) // 3, len(good) - 9, 40):                    # inside a chunk, near the end, inside the header
        bad = bytearray(good)
        bad[at] ^= 0x5A
        open(hipz, "wb").write(bytes(bad))
        r = run("zst")
        assert r.returncode != 0 and "ERROR" in r.stderr, at
    open(hipz, "wb").write(good[:len(good) * 2 // 3])
    assert run("zst").returncode != 0
    # version 2, written here from the published layout: MAGIC, k, ref_k, bits, 2 x (mode, n, pos[], cnt[]), n_keys, stride, rows, vals
    stride = 136
    items = sorted(kmap.items())
    with gzip.open(hipz, "wb", compresslevel=1) as f:
        f.write(b"MGHIPX2\n" + struct.pack("<3Q", 35, 43, bits))
        for src in (ctx, bf):
            pos = src.set_positions()
            f.write(struct.pack("<2Q", 1, pos.size) + pos.astype("<u8").tobytes() + src.counts().astype("<u2").tobytes())
        f.write(struct.pack("<2Q", len(items), stride))
        f.write(b"".join(k_.ljust(stride, b"\0") for k_, _ in items))
        f.write(np.array([v for _, v in items], dtype="<i4").tobytes())
    r = run("zst")
    assert r.returncode == 0, r.stderr
    filters, keys = index_file.read_index(zst)
    assert keys == dict(kmap.items())
    for (mode, size, pos, counts), src in zip(filters, (ctx, bf)):
        assert np.array_equal(pos, src.set_positions()) and np.array_equal(counts, src.counts())
