"""Random clustered VCF + FASTA fixtures for the host-enumerator and CLI tests: SNPs, MNPs,
insertions (some >= k), deletions, multi-allelic sites, overlapping records, phased and unphased
genotypes, a non-present record, records near contig ends, two contigs, N/IUPAC in the reference."""
import numpy as np


def make_case(path_prefix, seed, n_clusters=40, haploid=False, n_samples=5, k=35, vcf_strip_chr=False, dense=False, second_freq_key=None):
    """second_freq_key: INFO carries AF and a second key (e.g. EUR_AF, BASELINE config C2's `-f EUR_AF`) with OTHER values,
    sometimes zero where AF is not (so is_present differs between the keys) and sometimes before AF in the INFO column"""
    rng = np.random.default_rng(seed)
    contigs = {}
    scale = max(1, n_clusters // 40)
    for name, length in (("1", 9000 * scale), ("chr2", 4000 * scale)):
        g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
        for p in rng.integers(0, length, size=6):
            g[p] = rng.choice(np.frombuffer(b"NRY", dtype=np.uint8))
        contigs[name] = bytes(g).decode()
    records = []
    for name, seq in contigs.items():
        centres = np.sort(rng.choice(np.arange(10, len(seq) - 10), size=n_clusters if name == "1" else n_clusters // 3, replace=False))
        used = set()
        for c in centres:
            for _ in range(int(rng.integers(6, 16)) if dense else int(rng.integers(1, 6))):
                pos = int(c + rng.integers(0, 22 if dense else 28))
                if pos in used or pos >= len(seq) - 70:
                    continue
                used.add(pos)
                kind = rng.integers(0, 10)
                ref_len = 1 if kind < 6 else int(rng.integers(1, 9))
                ref = seq[pos:pos + ref_len]
                if any(ch not in "ACGT" for ch in ref):
                    continue
                n_alt = 1 if rng.random() < 0.75 else int(rng.integers(2, 4))
                alts = []
                while len(alts) < n_alt:
                    r = rng.random()
                    alen = 1 if r < 0.55 else (int(rng.integers(2, 12)) if r < 0.95 else int(rng.integers(k, k + 8)))
                    a = "".join(rng.choice(list("ACGT"), size=alen))
                    if a != ref and a not in alts:
                        alts.append(a)
                if rng.random() < 0.04:
                    alts.append("<DEL>")          # symbolic: dropped by the reader
                records.append((name, pos, ref, alts))
    records.sort(key=lambda r: (list(contigs).index(r[0]), r[1]))
    samples = ["S%d" % i for i in range(n_samples)]
    lines = ["##fileformat=VCFv4.2", '##INFO=<ID=AF,Number=A,Type=Float,Description="af">'] + \
            (['##INFO=<ID=%s,Number=A,Type=Float,Description="second af">' % second_freq_key] if second_freq_key else []) + \
            ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">'] + \
            ["##contig=<ID=%s,length=%d>" % (n, len(s)) for n, s in contigs.items()]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    for i, (name, pos, ref, alts) in enumerate(records):
        real = [a for a in alts if not a.startswith("<")]
        if i % 23 == 5:
            af = [0.0] * len(alts)               # f[0] == 1 -> not present
        else:
            w = rng.dirichlet(np.ones(len(alts) + 1))[1:]
            af = [round(float(x), 4) for x in w]
        gts = []
        for _ in samples:
            if haploid:
                gts.append(str(int(rng.integers(0, len(real) + 1))) if rng.random() > 0.03 else ".")
            else:
                a, b = int(rng.integers(0, len(real) + 1)), int(rng.integers(0, len(real) + 1))
                sep = "|" if rng.random() < 0.7 else "/"
                gts.append("%d%s%d" % (a, sep, b) if rng.random() > 0.03 else "./.")
        qual = "." if i % 3 else "%d" % (10 + i % 90)
        if vcf_strip_chr and name.startswith("chr"):
            name = name[3:]                     # matches the FASTA id only under -p/--strip-chr
        info = "AF=%s" % ",".join("%g" % x for x in af)
        if second_freq_key:
            w2 = rng.dirichlet(np.ones(len(alts) + 1))[1:]
            af2 = [0.0] * len(alts) if i % 17 == 3 else [round(float(x), 4) for x in w2]
            second = "%s=%s" % (second_freq_key, ",".join("%g" % x for x in af2))
            info = second + ";" + info if i % 2 else info + ";DP=%d;" % (i % 50) + second
        lines.append("%s\t%d\t%s\t%s\t%s\t%s\t.\t%s\tGT\t%s" % (name, pos + 1, "." if i % 4 else "rs%d" % i, ref, ",".join(alts), qual,
                                                               info, "\t".join(gts)))
    with open(path_prefix + ".vcf", "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(path_prefix + ".fa", "w") as fh:
        for n, s in contigs.items():
            fh.write(">%s some description\n" % n)
            for a in range(0, len(s), 60):
                fh.write(s[a:a + 60].lower() if (a // 60) % 7 == 3 else s[a:a + 60])
                fh.write("\n")
    return contigs, records


def donor_table(contigs, records, ref_k, seed, path):
    """a donor carrying a random allele of every record; `KMER count` dump of the canonical ref_k-mers
    seen >= 2 times in 12x tiled 'reads' of the two donor haplotypes"""
    from collections import Counter
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    counts = Counter()
    for hap in range(2):
        for name, seq in contigs.items():
            out, last = [], 0
            for (cn, pos, ref, alts) in records:
                real = [a for a in alts if not a.startswith("<")]
                if cn != name or pos < last:
                    continue
                pick = int(rng.integers(0, len(real) + 1))
                out.append(seq[last:pos]); out.append(ref if pick == 0 else real[pick - 1])
                last = pos + len(ref)
            out.append(seq[last:])
            donor = "".join(out).encode()
            depth = int(rng.integers(3, 9))
            for p in range(len(donor) - ref_k + 1):
                w = donor[p:p + ref_k]
                if w.strip(b"ACGT"):
                    continue
                rc = w.translate(comp)[::-1]
                counts[min(w, rc)] += depth
    with open(path, "w") as fh:
        for km, c in sorted(counts.items()):
            if c >= 2:
                fh.write("%s\t%d\n" % (km.decode(), min(c, 255)))


def make_far_case(path_prefix, seed, k=35, base=1 << 25, span=1_500_000, n_clusters=400, n_samples=4, haploid=False):
    """One contig longer than 2^25 bases with clusters of SNPs and short indels BEYOND position 2^25, where the
    reference's `are_near` (float arithmetic, var_block.hpp:417-423) stops agreeing with integer arithmetic: gaps
    between neighbours are drawn around the k/2 threshold.  Returns (contig sequence, records, pairs) with `pairs` =
    the number of adjacent record pairs on which the float test and the exact one disagree."""
    import struct
    rng = np.random.default_rng(seed)
    length = base + span
    g = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
    seq = bytes(g).decode()
    half = (k + 1) // 2
    records = []
    pos = base + 1000
    for _ in range(n_clusters):
        pos += int(rng.integers(200, (span - 4000) // n_clusters))
        p = pos
        for _ in range(int(rng.integers(2, 5))):
            kind = rng.integers(0, 10)
            ref_len = 1 if kind < 7 else int(rng.integers(2, 7))
            ref = seq[p:p + ref_len]
            alen = 1 if rng.random() < 0.7 else int(rng.integers(1, 7))
            alt = "".join(rng.choice(list("ACGT"), size=alen))
            if alt == ref:
                alt = ("A" if ref[0] != "A" else "C") + alt[1:]
            records.append(("1", p, ref, [alt]))
            min_size = min(ref_len, len(alt))
            p += ref_len - min_size - 1 + half + int(rng.integers(-5, 6))      # the next one lands around this one's reach
            p = max(p, records[-1][1] + ref_len)                                # (never overlapping: keeps the chains simple)
    f32 = lambda x: struct.unpack("f", struct.pack("f", float(x)))[0]
    pairs = 0
    for (_, p1, r1, a1), (_, p2, _, _) in zip(records, records[1:]):
        lhs = p1 + len(r1) - min(len(r1), len(a1[0])) - 1
        pairs += (f32(f32(lhs) + f32(half)) >= f32(p2)) != (lhs + half >= p2)
    samples = ["S%d" % i for i in range(n_samples)]
    with open(path_prefix + ".fa", "w") as fh:
        fh.write(">1\n")
        for i in range(0, length, 1 << 20):
            fh.write(seq[i:i + (1 << 20)] + "\n")
    with open(path_prefix + ".vcf", "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"af\">\n"
                 "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##contig=<ID=1,length=%d>\n" % length)
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for name, p, ref, alts in records:
            if haploid:
                gts = [str(int(rng.integers(0, 2))) for _ in samples]
            else:
                gts = ["%d%s%d" % (rng.integers(0, 2), "|" if rng.random() < 0.8 else "/", rng.integers(0, 2)) for _ in samples]
            if not any("1" in g_ for g_ in gts):
                gts[0] = "1" if haploid else "1|0"
            fh.write("%s\t%d\t.\t%s\t%s\t.\t.\tAF=%.3f\tGT\t%s\n" % (name, p + 1, ref, ",".join(alts), 0.1 + 0.5 * rng.random(), "\t".join(gts)))
    return seq, records, pairs
