"""Stand-in for the `kmc` counting step of the MALVA pipeline script.

TEST INFRASTRUCTURE ONLY.  The reference shells out to the third-party KMC
binary (`kmc -m<mem> -k<ref_k> -t1 -fm`, MALVA:107), which is not part of the
reference checkout and is not installed here.  Its published counting
semantics with those flags: canonical k-mers, windows containing a non-ACGT
symbol skipped, k-mers seen fewer than 2 times dropped (-ci2 default), counts
capped at 255 (-cs255 default).  This generates the *input* k-mer stream for
the end-to-end golden test; it is not part of the path being restated.
"""
from collections import Counter

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def canonical_acgt(kmer: bytes) -> bytes:
    rc = kmer.translate(_COMP)[::-1]
    return kmer if kmer < rc else rc


def count_fastq(path: str, k: int, ci: int = 2, cs: int = 255):
    """-> sorted list of (canonical k-mer bytes, count)"""
    counts = Counter()
    with open(path, "rb") as fh:
        for i, line in enumerate(fh):
            if i % 4 != 1:
                continue
            seq = line.strip().upper()
            for p in range(len(seq) - k + 1):
                w = seq[p:p + k]
                if w.strip(b"ACGT"):
                    # contains a symbol outside ACGT
                    if any(c not in b"ACGT" for c in w):
                        continue
                counts[canonical_acgt(w)] += 1
    return sorted((km, min(c, cs)) for km, c in counts.items() if c >= ci)
