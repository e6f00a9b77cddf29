"""A panel resident in HBM, and one step of the record loop over it through the *_device entry points.

PyTorch is the allocator here (device tensors keep the arrays alive and give their addresses); everything that is
computed is computed by libmalva_hip.so.  Used by bench.py and the -m gpu tests."""
import numpy as np

from .capi import PanelDev, sparse_genotypes


class ResidentPanel:
    """The arrays of malva_amd.synth.FlatPanel uploaded once; `.dev` is the mg_panel_dev to hand to the library."""

    def __init__(self, panel, device, haploid=False, sparse=False, sp_default=1 << 14):
        """sparse: the genotypes go up as the entries other than 0|0 phased (mg_panel_dev.sp_*) instead of the dense matrix"""
        import torch
        self.torch = torch
        self.dev_t = torch.device("cuda", device) if isinstance(device, int) else device
        self.n = panel.n
        self.n_slots = int(panel.var_allele_off[-1])
        self.n_samples = int(panel.n_samples)
        self.haploid = haploid
        up = self._up
        self.t = {
            "contig_base": up(panel.contig_base, np.uint64), "contig_len": up(panel.contig_len, np.uint32), "contig_id": up(panel.contig_id, np.uint32),
            "pos": up(panel.pos, np.int32), "ref_size": up(panel.ref_size, np.uint32), "min_size": up(panel.min_size, np.uint32),
            "present": up(panel.present, np.uint8), "var_allele_off": up(panel.var_allele_off, np.uint32), "allele_off": up(panel.allele_off, np.uint32),
            "pool": up(panel.pool, np.uint8), "canon": up(panel.canon, np.uint8), "gt": up(np.ascontiguousarray(panel.gt).reshape(-1), np.uint16),
            "freq": up(panel.freq, np.float32),
        }
        A = np.diff(panel.var_allele_off.astype(np.int64))
        goff = np.zeros(self.n + 1, dtype=np.uint64)
        goff[1:] = np.cumsum(A if haploid else A * (A + 1) // 2)
        self.n_gt = int(goff[-1])
        self.t["gt_off"] = up(goff, np.uint64)
        z = lambda n, dt: torch.zeros(max(int(n), 1), dtype=dt, device=self.dev_t)
        self.blk_var_off, self.var_block, self.n_blocks = z(self.n + 1, torch.int32), z(self.n, torch.int32), z(1, torch.int64)
        self.cov, self.overflow = z(self.n_slots, torch.int32), z(self.n, torch.uint8)
        self.g1, self.g2, self.gq, self.status = z(self.n, torch.int32), z(self.n, torch.int32), z(self.n, torch.int32), z(self.n, torch.uint8)
        self.probs = z(self.n_gt, torch.float64)
        d = PanelDev()
        d.n_vars, d.n_contigs, d.n_samples = self.n, len(panel.contig_len), self.n_samples
        for name in ("contig_base", "contig_len", "contig_id", "pos", "ref_size", "min_size", "present", "var_allele_off", "allele_off", "pool", "canon", "gt"):
            setattr(d, name, self.t[name].data_ptr())
        d.pool_bytes = int(len(panel.pool))
        if sparse:
            so, ss, sg = sparse_genotypes(panel.gt, self.n_samples, sp_default)
            d.sp_default = sp_default
            self.t["sp_off"], self.t["sp_sample"], self.t["sp_gt"] = up(so, np.uint32), up(ss, np.uint32), up(sg, np.uint16)
            d.gt = None
            d.sp_off, d.sp_sample, d.sp_gt = (self.t[x].data_ptr() for x in ("sp_off", "sp_sample", "sp_gt"))
        self.dev = d

    def _up(self, a, dt):
        """numpy array -> device tensor of the same bytes (torch has no unsigned 16/32/64-bit tensors worth the name)"""
        torch = self.torch
        a = np.ascontiguousarray(a, dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        view = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[a.dtype.itemsize]
        if a.dtype == np.float32:
            view = np.float32
        return torch.from_numpy(a.view(view)).to(self.dev_t)

    # one step of loop B (main.cpp:522-579) with everything resident
    def cut(self, ctx):
        ctx.cut_blocks_device(self.dev, self.blk_var_off.data_ptr(), self.var_block.data_ptr(), self.n_blocks.data_ptr())

    def cover(self, ctx):
        ctx.cover_blocks_device(self.dev, self.blk_var_off.data_ptr(), self.var_block.data_ptr(), self.n_blocks.data_ptr(), self.haploid,
                                self.cov.data_ptr(), self.overflow.data_ptr())

    def genotype(self, ctx, error_rate=0.001, max_cov=200, probs=True):
        ctx.genotype_device(self.cov.data_ptr(), self.t["freq"].data_ptr(), self.t["var_allele_off"].data_ptr(), self.n, error_rate, max_cov, self.haploid,
                            self.g1.data_ptr(), self.g2.data_ptr(), self.gq.data_ptr(), self.status.data_ptr(),
                            self.probs.data_ptr() if probs else None, self.t["gt_off"].data_ptr() if probs else None)

    def call_step(self, ctx, error_rate=0.001, max_cov=200, probs=True):
        """probs=False: GT and GQ only, what `malva-geno call` prints without -v (var_block.hpp:366-394)"""
        self.cut(ctx)
        self.cover(ctx)
        self.genotype(ctx, error_rate, max_cov, probs)

    def index(self, ctx):
        """cut + extract_kmers + add_kmers_to_bf (main.cpp:309-370); returns the overflow flags (host array)"""
        self.cut(ctx)
        ctx.index_blocks_device(self.dev, self.blk_var_off.data_ptr(), self.var_block.data_ptr(), self.n_blocks.data_ptr(), self.haploid, self.overflow.data_ptr())
        return self.overflow[:self.n].cpu().numpy()

    def results(self):
        n, na = self.n, self.n_slots
        self.torch.cuda.synchronize()
        nb = int(self.n_blocks.item())
        return {"blk_var_off": self.blk_var_off[:nb + 1].cpu().numpy().view(np.uint32), "cov": self.cov[:na].cpu().numpy().view(np.uint32),
                "overflow": self.overflow[:n].cpu().numpy(), "g1": self.g1[:n].cpu().numpy(), "g2": self.g2[:n].cpu().numpy(),
                "gq": self.gq[:n].cpu().numpy(), "status": self.status[:n].cpu().numpy(), "probs": self.probs[:self.n_gt].cpu().numpy()}
