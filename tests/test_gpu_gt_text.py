"""mg_decode_gt_text: the panel's genotypes decoded from the VCF's sample columns on the device, against the oracle's reading
of the same records (oracle/model.py VcfReader._genotypes: variant.hpp:158-211 over what bcf_get_genotypes returns):
ploidy 1 records (whose second allele is the NEXT sample's), mixed ploidy, missing alleles, multi-digit allele numbers,
GT behind other FORMAT keys, short records, a sample subset, haploid mode."""
import numpy as np
import pytest

from malva_amd import Context
from oracle import model

pytestmark = pytest.mark.gpu


def _write(path, records, n_samples, fmt_of=lambda i: "GT", pad=False):
    """pad: records with fewer sample columns than the header are completed with "." -- what the product reads them as;
    the oracle's reader (like htslib) has no opinion on such a record, so it is given the completed one"""
    if pad:
        records = [(list(c) if c is not None else []) + ["."] * (n_samples - (len(c) if c is not None else 0)) for c in records]
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"af\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(n_samples)) + "\n")
        for i, cells in enumerate(records):
            fh.write("1\t%d\t.\tA\tC,G,T\t.\t.\tAF=0.1,0.1,0.1\t%s\t%s\n" % (100 + 40 * i, fmt_of(i), "\t".join(cells)) if cells is not None else
                     "1\t%d\t.\tA\tC,G,T\t.\t.\tAF=0.1,0.1,0.1\t%s\n" % (100 + 40 * i, fmt_of(i)))


def _spans(path):
    """what the host side of the decode does: per record line, where its sample columns start and end, and where GT sits in FORMAT"""
    raw = open(path, "rb").read()
    off, ln, gi = [], [], []
    at = 0
    for line in raw.split(b"\n"):
        if line and not line.startswith(b"#"):
            cols = line.split(b"\t", 9)
            gi.append(cols[8].split(b":").index(b"GT"))
            if len(cols) > 9:
                off.append(at + len(line) - len(cols[9]))
                ln.append(len(cols[9]))
            else:
                off.append(at + len(line))
                ln.append(0)
        at += len(line) + 1
    return raw, np.array(off, np.uint64), np.array(ln, np.uint32), np.array(gi, np.int32)


def _expected(path, samples_file, haploid):
    rd = model.VCFReader(path, samples_file or "-")
    words, masks, mx = [], [], []
    for v in rd.records():
        w = []
        m = 0
        big = 0
        for (a1, a2), ph in zip(v.genotypes, v.phasing):
            w.append((a1 & 127) | (1 << 14) if haploid else (a1 & 127) | (a2 & 127) << 7 | int(ph) << 14)
            m |= 1 << (a1 & 63)
            if not haploid:
                m |= 1 << (a2 & 63)
            big = max(big, a1, a2)
        words.append(np.array(w, np.uint16))
        masks.append(m)
        mx.append(big)
    return words, masks, mx


def _dense(n_keep, dflt, sp_off, ss, sg):
    out = []
    for r in range(len(sp_off) - 1):
        w = np.full(n_keep, dflt, np.uint16)
        e0, e1 = int(sp_off[r]), int(sp_off[r + 1])
        assert np.all(np.diff(ss[e0:e1].astype(np.int64)) > 0)          # ascending samples inside a record
        assert np.all(sg[e0:e1] != dflt)                                 # only the words that differ from the default
        w[ss[e0:e1]] = sg[e0:e1]
        out.append(w)
    return out


@pytest.mark.parametrize("haploid", [False, True])
@pytest.mark.parametrize("subset", [False, True])
def test_decode_matches_the_oracles_reading(tmp_path, haploid, subset):
    rng = np.random.default_rng(17 + haploid + 2 * subset)
    n_samples = 1500
    forms = ["0", "1", ".", "0|0", "0|1", "1|0", "0/0", "0/1", "2|3", "./.", ".|1", "1/.", "0|1|2", "3", "12|0", "0/10", "", "1/2/3", "0|0|0"]
    records = []
    for i in range(60):
        kind = i % 6
        if kind == 0:    # ploidy 1 throughout: the second allele of every sample is its neighbour's
            cells = list(rng.choice(["0", "1", ".", "2", "3"], size=n_samples, p=[0.8, 0.1, 0.03, 0.04, 0.03]))
        elif kind == 1:  # diploid phased, nearly all 0|0
            cells = list(rng.choice(["0|0", "0|1", "1|0", "1|1", "2|0"], size=n_samples, p=[0.9, 0.04, 0.03, 0.02, 0.01]))
        elif kind == 2:  # unphased
            cells = list(rng.choice(["0/0", "0/1", "1/1", "./."], size=n_samples, p=[0.85, 0.08, 0.05, 0.02]))
        elif kind == 3:  # everything at once, mixed ploidy
            cells = list(rng.choice(forms, size=n_samples))
        elif kind == 4:  # a record with fewer sample columns than the header
            cells = list(rng.choice(["0|1", "1|1", "0|0"], size=int(rng.integers(1, n_samples))))
        else:            # GT behind / between other keys (see fmt_of), other sub-fields present or cut short
            cells = list(rng.choice(forms[:12], size=n_samples))
        records.append(cells)
    records.append(None)                      # no sample column at all
    records.append([""])                      # a ninth tab and nothing behind it
    records.append(["0|1"] * n_samples)
    path = str(tmp_path / "p.vcf")

    def fmt_of(i):
        return ["GT", "GT:DP", "DP:GT", "DP:GQ:GT:PL"][i % 4] if i % 6 == 5 else "GT"

    # records whose FORMAT puts GT behind other keys carry those keys' values in front of it
    recs = []
    for i, cells in enumerate(records):
        f = fmt_of(i)
        if cells is None or f == "GT":
            recs.append(cells)
            continue
        keys = f.split(":")
        gi = keys.index("GT")
        out = []
        for j, g in enumerate(cells):
            parts = ["%d" % (j % 50)] * len(keys)
            parts[gi] = g
            if j % 7 == 3:
                parts = parts[:max(1, gi)] if gi else parts[:1]       # the column stops short (before GT when it is not first)
            out.append(":".join(parts))
        recs.append(out)
    _write(path, recs, n_samples, fmt_of)
    _write(path + ".padded", recs, n_samples, fmt_of, pad=True)
    samples_file = None
    keep = None
    if subset:
        picked = sorted(rng.choice(n_samples, size=400, replace=False))
        samples_file = str(tmp_path / "keep.txt")
        open(samples_file, "w").write("".join("S%d\n" % i for i in picked))
        keep = np.zeros(n_samples, np.uint8)
        keep[picked] = 1
    want, wmask, wmax = _expected(path + ".padded", samples_file, haploid)
    raw, off, ln, gi = _spans(path)
    assert len(want) == len(off) == len(recs)
    with Context(35, 43, 1 << 20) as ctx:
        dflt, sp_off, ss, sg, mask, mx = ctx.decode_gt_text(raw, off, ln, gi, n_samples, keep, haploid)
        got = _dense(400 if subset else n_samples, dflt, sp_off, ss, sg)
        assert dflt in (0, 1 << 14) and (not haploid or dflt == 1 << 14)
        for r, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), (r, np.flatnonzero(g != w)[:5], g[g != w][:5], w[g != w][:5])
        assert [int(x) for x in mask] == [m & ((1 << 64) - 1) for m in wmask]
        assert [int(x) for x in mx] == wmax
        # an unphased panel: 0/0 becomes the default word
        if not haploid:
            sel = [r for r in range(len(recs)) if r % 6 == 2]
            dflt2, sp2, ss2, sg2, _, _ = ctx.decode_gt_text(raw, off[sel], ln[sel], gi[sel], n_samples, keep, False)
            assert dflt2 == 0
            for g, r in zip(_dense(400 if subset else n_samples, dflt2, sp2, ss2, sg2), sel):
                assert np.array_equal(g, want[r])
