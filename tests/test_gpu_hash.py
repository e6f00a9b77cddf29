"""K1: canonical + XXH3 + hash % size on the device vs the oracle, bit exact."""
import numpy as np
import pytest

from malva_amd import BF_ALT, BF_CTX, Context, synth
from malva_amd.capi import rows_of
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu


def _oracle_index(kmers, size):
    return np.array([ocapi.lib().mo_bf_hash(k) % size for k in kmers], dtype=np.uint64)


@pytest.mark.parametrize("size", [1 << 20, 1000003, 3 << 33, (1 << 33) + 12345])
def test_ascii_index_all_lengths(size):
    rng = np.random.default_rng(1)
    kmers = []
    for L in list(range(1, 129)) * 3:
        alpha = b"ACGT" if L % 3 else b"ACGTNacgtnWMRY"
        kmers.append(bytes(rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=L)))
    kmers += [b"ACGTACGTACGTACGTACGTACGTACGTACGTACG", b"A" * 35, b"T" * 43, b"GATTACA" * 9, b"ACGWACGTT", b"ACGTNACGT"]
    ctx = Context(35, 43, size)
    got = ctx.bf_index(BF_ALT, rows_of(kmers, 136))
    assert np.array_equal(got, _oracle_index(kmers, size))
    ctx.close()


def test_known_answer_slot():
    ctx = Context(35, 43, 1 << 33)
    got = ctx.bf_index(BF_CTX, rows_of([b"ACGTACGTACGTACGTACGTACGTACGTACGTACG"]))
    assert int(got[0]) == 638663402
    ctx.close()


@pytest.mark.parametrize("klen", [17, 24, 31, 32, 33, 35, 43, 48, 63, 64])
def test_packed_index_matches_ascii_oracle(klen):
    rng = np.random.default_rng(klen)
    n = 5000
    rows = synth.BASES[rng.integers(0, 4, size=(n, klen))]
    rows[0, :] = ord("A"); rows[1, :] = ord("T"); rows[2, :] = ord("C")
    rows[3, :] = np.frombuffer((b"ACGT" * 16)[:klen], dtype=np.uint8)          # palindromic for even k
    hi, lo = synth.pack_ascii(rows)
    size = (1 << 35) if klen % 2 else 1000003
    ctx = Context(klen, klen, size)
    got = ctx.packed_index(BF_ALT, hi, lo, klen)
    want = _oracle_index([bytes(r) for r in rows], size)
    assert np.array_equal(got, want)
    ctx.close()
