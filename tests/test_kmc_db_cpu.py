"""oracle/kmc_db.py: the KMC database layout restated (writer used by the GPU tests, reader = the checker of the
product's reader).  No fixture written by KMC itself exists in the reference checkout: PARITY UNPINNED, see the
module's header.  What is checked here is internal consistency of the restatement and the listing rules it encodes."""
import random
import struct

import numpy as np
import pytest

from oracle import kmc_db


def _random_kmers(n, k, seed, lo=1, hi=300):
    rng = random.Random(seed)
    out = {}
    while len(out) < n:
        out[bytes(rng.choice(b"ACGT") for _ in range(k))] = rng.randint(lo, hi)
    return list(out.items())


@pytest.mark.parametrize("k,p,bins,cs", [(43, 7, 3, 1), (43, 3, 1, 2), (63, 7, 2, 4), (41, 5, 5, 1), (21, 1, 2, 3)])
def test_write_then_list(tmp_path, k, p, bins, cs):
    items = _random_kmers(3000, k, k * 10 + p)
    path = str(tmp_path / "db")
    n = kmc_db.write_db(path, items, k, lut_prefix_len=p, n_bins=bins, counter_size=cs, min_count=2, max_count=255)
    db = kmc_db.KmcDb(path)
    assert (db.k, db.lut_prefix_len, db.counter_size, db.total, db.min_count, db.max_count) == (k, p, cs, n, 2, 255)
    assert db.lut.size == bins * 4 ** p and db.lut[0] == 0 and np.all(np.diff(db.lut.astype(np.int64)) >= 0)
    listed = db.kmers()
    want = {km: min(c, 256 ** cs - 1) for km, c in items if 2 <= min(c, 256 ** cs - 1) <= 255}   # counters saturate at their width
    assert dict(listed) == want and len(listed) == len(want)          # ReadNextKmer skips counts outside [min, max]
    # listing order: bin after bin, ascending inside a bin -> at most `bins` descents
    keys = [km for km, _ in listed]
    assert sum(1 for a, b in zip(keys, keys[1:]) if b < a) <= bins - 1
    hi, lo, cnt = db.table()
    assert cnt.size == n and int((cnt > 0).sum()) == len(want)


def test_header_fields_sit_where_the_kmc_api_reads_them(tmp_path):
    path = str(tmp_path / "db")
    kmc_db.write_db(path, _random_kmers(10, 43, 1, 2, 9), 43, n_bins=1, max_count=(3 << 32) | 255)
    pre = open(path + ".kmc_pre", "rb").read()
    assert pre[:4] == b"KMCP" and pre[-4:] == b"KMCP"
    assert struct.unpack_from("<I", pre, len(pre) - 12)[0] == 0x200          # my_fseek(file_pre, -12, SEEK_END)
    hsize = pre[-8]                                                             # fgetc at -8
    assert struct.unpack_from("<I", pre, len(pre) - 8)[0] == hsize
    h0 = len(pre) - 8 - hsize                                                   # my_fseek(-(header_offset + 8), SEEK_END)
    assert struct.unpack_from("<7I", pre, h0)[:4] == (43, 0, 1, 7)
    assert kmc_db.KmcDb(path).max_count == (3 << 32) | 255
    suf = open(path + ".kmc_suf", "rb").read()
    assert suf[:4] == b"KMCS" and suf[-4:] == b"KMCS" and len(suf) == 8 + 10 * 10
