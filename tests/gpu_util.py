"""Shared helpers for the -m gpu parity tests (HIP path vs CPU oracle)."""
import numpy as np

from malva_amd import synth
from oracle import capi as ocapi


def rows_bytes(rows):
    """uint8 [n, stride] -> list of bytes up to the NUL"""
    return [bytes(r).split(b"\0", 1)[0] for r in rows]


def pad_rows(rows, stride=None):
    """uint8 [n, L] -> uint8 [n, stride] NUL padded"""
    n, L = rows.shape
    stride = stride or (L + 1 + 7) // 8 * 8
    out = np.zeros((n, stride), dtype=np.uint8)
    out[:, :L] = rows
    return out


def build_index_pair(ctx, panel, k, ref_k, bf_bits, sig_rows=None, valid=None):
    """Build the same index on the device context and in the oracle from an isolated panel:
    allele 0 signatures -> exact map, the others -> bf; finalise; reference scan; finalise."""
    if sig_rows is None:
        sig_rows, valid = synth.signature_rows(panel, k)
    na = sig_rows.shape[0]
    is_ref = np.zeros(na, dtype=np.uint8)
    is_ref[panel.var_allele_off[:-1]] = 1
    rows = pad_rows(sig_rows[valid])
    isr = is_ref[valid]
    obf, octx, omap = ocapi.BF(bf_bits), ocapi.BF(bf_bits), ocapi.KMAP()
    ocapi.add_kmers(obf, omap, rows, isr)
    obf.switch_mode()
    ocapi.ref_scan(obf, octx, panel.genome.tobytes(), k, ref_k)
    octx.switch_mode()
    from malva_amd import BF_ALT, BF_CTX
    ctx.map_insert(rows[isr == 1])
    ctx.bf_insert(BF_ALT, rows[isr == 0])
    ctx.bf_finalize(BF_ALT)
    ctx.ref_scan(panel.genome.tobytes())
    ctx.bf_finalize(BF_CTX)
    return obf, octx, omap


def map_values_by_key(ctx):
    keys, vals = ctx.map_export()
    return dict(zip(keys, (int(v) for v in vals)))
