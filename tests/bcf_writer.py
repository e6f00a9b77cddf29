"""VCF text -> BCF2 (VCF/BCF specification v4.3, section 6), test infrastructure: an independent writer for the product's
BCF reader (malva_amd/host/io.hpp LineReader), which restates the same published layout.  The reference reads BCF through
htslib (third party, absent): PARITY UNPINNED, as for the other third-party formats.  Only what a genotype panel holds is
encoded: INFO of any declared type, FORMAT GT."""
import gzip
import struct
import zlib

MISSING = {1: -128, 2: -32768, 3: -(1 << 31)}
EOV = {1: -127, 2: -32767, 3: -(1 << 31) + 1}
FMT = {1: "<b", 2: "<h", 3: "<i"}


def _desc(n, t):
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + _typed_ints([n])


def _int_type(vals):
    lo, hi = min(vals, default=0), max(vals, default=0)
    for t, (a, b) in ((1, (-120, 127)), (2, (-32760, 32767)), (3, (-(1 << 31) + 8, (1 << 31) - 1))):
        if lo >= a and hi <= b:
            return t
    raise ValueError("integer out of range")


def _typed_ints(vals):
    t = _int_type([v for v in vals if v is not None])
    return _desc(len(vals), t) + b"".join(struct.pack(FMT[t], MISSING[t] if v is None else v) for v in vals)


def _typed_str(s):
    b = s.encode()
    return _desc(len(b), 7) + b


def _attr(line, key):
    inner = line[line.index("<") + 1:line.rindex(">")]
    out, cur, q = [], "", False
    for ch in inner:
        if ch == '"':
            q = not q
        if ch == "," and not q:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    for kv in out:
        if kv.startswith(key + "="):
            return kv[len(key) + 1:].strip('"')
    return None


def _bgzf(data, out):
    for a in range(0, len(data), 0xFF00):
        raw = data[a:a + 0xFF00]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1) + body + struct.pack("<II", zlib.crc32(raw), len(raw)))
    out.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))      # the BGZF end-of-file block


def vcf_to_bcf(vcf_path, bcf_path, with_idx=False):
    opener = gzip.open if open(vcf_path, "rb").read(2) == b"\x1f\x8b" else open
    lines = [l.rstrip("\r\n") for l in opener(vcf_path, "rt")]
    header = [l for l in lines if l.startswith("##")]
    chrom_line = next(l for l in lines if l.startswith("#CHROM"))
    records = [l for l in lines if l and not l.startswith("#")]
    n_sample = max(0, len(chrom_line.split("\t")) - 9)
    contigs = [_attr(l, "ID") for l in header if l.startswith("##contig=")]
    for r in records:                                             # BCF names contigs by index: every one must be declared
        c = r.split("\t", 1)[0]
        if c not in contigs:
            contigs.append(c)
            header.append("##contig=<ID=%s>" % c)
    if not any(l.startswith("##FILTER=<ID=PASS") for l in header):
        header.insert(1, '##FILTER=<ID=PASS,Description="All filters passed">')
    dic, types = ["PASS"], {}
    for l in header:
        for kind in ("INFO", "FORMAT", "FILTER"):
            if l.startswith("##%s=" % kind):
                i = _attr(l, "ID")
                if i not in dic:
                    dic.append(i)
                if kind != "FILTER":
                    types[(kind, i)] = _attr(l, "Type")
    if with_idx:                                                   # the explicit form htslib writes: IDX= on every dictionary line
        def idx(l):
            for kind in ("INFO", "FORMAT", "FILTER"):
                if l.startswith("##%s=" % kind):
                    return l[:-1] + ",IDX=%d>" % dic.index(_attr(l, "ID"))
            if l.startswith("##contig="):
                return l[:-1] + ",IDX=%d>" % contigs.index(_attr(l, "ID"))
            return l
        header = [idx(l) for l in header]
    text = ("\n".join(header + [chrom_line]) + "\n").encode() + b"\0"
    body = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)
    for r in records:
        c = r.split("\t")
        chrom, pos, vid, ref, alt, qual, flt, info = c[:8]
        alleles = [ref] + ([] if alt == "." else alt.split(","))
        infos = [] if info == "." else info.split(";")
        fmt_keys = c[8].split(":") if len(c) > 8 else []
        assert fmt_keys in ([], ["GT"]), "the test writer encodes FORMAT GT only"
        shared = struct.pack("<iii", contigs.index(chrom), int(pos) - 1, len(ref))
        shared += struct.pack("<I", 0x7F800001) if qual == "." else struct.pack("<f", float(qual))
        shared += struct.pack("<II", (len(alleles) << 16) | len(infos), (len(fmt_keys) << 24) | (n_sample if fmt_keys else 0))
        shared += _typed_str("" if vid == "." else vid)
        for a in alleles:
            shared += _typed_str(a)
        shared += _desc(0, 0) if flt == "." else _typed_ints([dic.index(x) for x in flt.split(";")])
        for kv in infos:
            key, _, val = kv.partition("=")
            shared += _typed_ints([dic.index(key)])
            ty = types.get(("INFO", key), "String")
            if ty == "Flag" or not _:
                shared += _desc(0, 0)
            elif ty == "Integer":
                shared += _typed_ints([None if x == "." else int(x) for x in val.split(",")])
            elif ty == "Float":
                vals = val.split(",")
                shared += _desc(len(vals), 5) + b"".join(struct.pack("<I", 0x7F800001) if x == "." else struct.pack("<f", float(x)) for x in vals)
            else:
                shared += _typed_str(val)
        indiv = b""
        if fmt_keys:
            gts = []
            for s in c[9:9 + n_sample]:
                g = s.split(":")[0]
                toks, cur, phased = [], "", [0]
                for ch in g:
                    if ch in "|/":
                        toks.append(cur); cur = ""; phased.append(1 if ch == "|" else 0)
                    else:
                        cur += ch
                toks.append(cur)
                gts.append([(0 if t in (".", "") else (int(t) + 1) << 1) | p for t, p in zip(toks, phased)])
            ploidy = max(len(g) for g in gts)
            t = _int_type([v for g in gts for v in g] + [0])
            indiv = _typed_ints([dic.index("GT")]) + _desc(ploidy, t)
            for g in gts:
                indiv += b"".join(struct.pack(FMT[t], v) for v in g) + struct.pack(FMT[t], EOV[t]) * (ploidy - len(g))
        body += struct.pack("<II", len(shared), len(indiv)) + shared + indiv
    with open(bcf_path, "wb") as out:
        _bgzf(bytes(body), out)
