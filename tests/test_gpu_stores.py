"""H3-H9: the Bloom filter (bits, rank, u16 counters) and the exact map on the
device vs the oracle: same bits, same counters, same answers, including the
collisions a small filter forces, counter wrap-around, and non-ACGT k-mers."""
import numpy as np
import pytest

from malva_amd import BF_ALT, BF_CTX, Context, MalvaError
from malva_amd.capi import rows_of
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu


def _random_kmers(rng, n, k, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [bytes(rng.choice(a, size=k)) for _ in range(n)]


@pytest.mark.parametrize("size", [4099, 1 << 16, 1 << 33])
def test_bf_insert_test_finalize_counts(size):
    rng = np.random.default_rng(size % 1000)
    k = 35
    ins = _random_kmers(rng, 3000, k) + _random_kmers(rng, 200, k, b"ACGTN") + [b"ACGWACGTTACGWACGTTACGWACGTTACGWACGT"]
    probe = ins[::3] + _random_kmers(rng, 3000, k)
    ctx = Context(k, 43, size)
    obf = ocapi.BF(size)
    for km in ins:
        obf.add_key(km)
    ctx.bf_insert(BF_ALT, rows_of(ins))
    assert np.array_equal(ctx.bf_test(BF_ALT, rows_of(probe)), np.array([obf.test_key(p) for p in probe]))
    # write mode: get_count is 0, increment refuses (BF::increment returns false)
    assert not ctx.bf_get_count(BF_ALT, rows_of(probe[:10])).any()
    with pytest.raises(MalvaError):
        ctx.bf_increment(BF_ALT, rows_of(probe[:10]), np.ones(10, dtype=np.uint32))
    obf.switch_mode()
    ctx.bf_finalize(BF_ALT)
    assert ctx.bf_info(BF_ALT) == (size, obf.nset, 1)
    inc = probe * 3
    cnts = rng.integers(1, 256, size=len(inc)).astype(np.uint32)
    for km, c in zip(inc, cnts):
        obf.increment(km, int(c))
    ctx.bf_increment(BF_ALT, rows_of(inc), cnts)
    got = ctx.bf_get_count(BF_ALT, rows_of(probe))
    assert np.array_equal(got, np.array([obf.get_count(p) for p in probe], dtype=np.uint16))
    mode, sz, words, counts = ctx.bf_export(BF_ALT)
    assert mode == 1 and sz == size
    assert np.array_equal(words, obf.words())
    assert np.array_equal(counts, obf.counts())
    ctx.close()


def test_bf_counter_wraps_mod_65536():
    ctx = Context(35, 43, 1 << 20)
    km = [b"ACGTACGTACGTACGTACGTACGTACGTACGTACG"]
    ctx.bf_insert(BF_ALT, rows_of(km))
    ctx.bf_finalize(BF_ALT)
    ctx.bf_increment(BF_ALT, rows_of(km * 300), np.full(300, 255, dtype=np.uint32))
    assert int(ctx.bf_get_count(BF_ALT, rows_of(km))[0]) == 10964
    ctx.close()


def test_bf_import_roundtrip():
    rng = np.random.default_rng(5)
    ins = _random_kmers(rng, 500, 35)
    a = Context(35, 43, 1 << 18)
    a.bf_insert(BF_CTX, rows_of(ins))
    a.bf_finalize(BF_CTX)
    a.bf_increment(BF_CTX, rows_of(ins), np.arange(500, dtype=np.uint32))
    mode, size, words, counts = a.bf_export(BF_CTX)
    b = Context(35, 43, 1 << 18)
    b.bf_import(BF_CTX, mode, size, words, counts)
    assert np.array_equal(b.bf_get_count(BF_CTX, rows_of(ins)), a.bf_get_count(BF_CTX, rows_of(ins)))
    a.close(); b.close()


def test_kmap_semantics_and_growth():
    rng = np.random.default_rng(9)
    k = 35
    ctx = Context(k, 43, 1 << 16)
    om = ocapi.KMAP()
    keys = _random_kmers(rng, 4000, k)
    keys += [k_[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")) for k_ in keys[:500]]   # reverse complements: same key
    keys += keys[:300]                                                                     # duplicates inside one batch
    irregular = [b"ACGTNACGTNACGTNACGTNACGTNACGTNACGTN", b"ACGWACGTTACGWACGTTACGWACGTTACGWACGT"]
    for km in keys + irregular:
        om.add_key(km)
    ctx.map_insert(rows_of(keys[:1000]))
    ctx.map_insert(rows_of(keys[1000:] + irregular))     # second batch forces a rehash
    assert ctx.map_size() == len(om)
    probe = keys[::7] + _random_kmers(rng, 2000, k) + irregular + [b"ACGTNACGTNACGTNACGTNACGTNACGTNACGTA"]
    assert np.array_equal(ctx.map_test(rows_of(probe)), np.array([om.test_key(p) for p in probe]))
    inc = probe * 2
    cnts = rng.integers(1, 1 << 30, size=len(inc)).astype(np.int32)
    for km, c in zip(inc, cnts):
        om.increment(km, int(c))
    ctx.map_increment(rows_of(inc), cnts)
    assert np.array_equal(ctx.map_get_count(rows_of(probe)), np.array([om.get_count(p) for p in probe], dtype=np.int32))
    gk, gv = ctx.map_export()
    assert dict(zip(gk, (int(v) for v in gv))) == dict(om.items())
    # add_key on an existing key resets it to 0 (kmap.hpp:111)
    ctx.map_insert(rows_of(probe[:5]))
    for p in probe[:5]:
        om.add_key(p)
    assert np.array_equal(ctx.map_get_count(rows_of(probe[:20])), np.array([om.get_count(p) for p in probe[:20]], dtype=np.int32))
    ctx.close()


def test_argument_errors_are_reported():
    ctx = Context(35, 43, 1 << 16)
    with pytest.raises(MalvaError):
        ctx.bf_insert(7, rows_of([b"ACGT"]))
    with pytest.raises(MalvaError):
        ctx.kmc_scan(np.zeros(1, np.uint64), np.zeros(1, np.uint64), np.ones(1, np.uint32))   # filters not finalised
    ctx.close()
    with pytest.raises(MalvaError):
        Context(35, 20, 1 << 16)          # ref_k < k
