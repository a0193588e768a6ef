"""Thin helper for the GPU tests: call C-ABI stage functions with torch tensors."""
import ctypes as C

import numpy as np
import torch

from dmpfold2_amd import _lib
from dmpfold2_amd.predict import Engine


class Stages:
    def __init__(self, state_dict, max_L=128, max_N=512, device="cuda:0"):
        self.eng = Engine(device, max_L, max_N)
        self.eng.set_weights({k: torch.from_numpy(np.array(v)) for k, v in state_dict.items()})
        self.lib = self.eng.lib
        self.dev = self.eng.device

    def _s(self):
        return self.eng.stream()

    def f32(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def to(self, a, dtype=torch.float32):
        return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(self.dev).contiguous()

    def call(self, name, *args):
        conv = []
        for a in args:
            if isinstance(a, torch.Tensor):
                assert a.is_contiguous() and a.device.type == "cuda"
                conv.append(a.data_ptr())
            else:
                conv.append(a)
        rc = getattr(self.lib, name)(self.eng.ctx, *conv, self._s())
        _lib.check(rc)
        return rc

    # ---- stages ----------------------------------------------------------------------
    def msa_weights(self, alnmat):
        m = self.to(alnmat, torch.uint8)
        n, L = m.shape
        w = self.f32(n)
        self.call("dmp_msa_weights", m, n, L, w)
        return w

    def cov_build(self, alnmat, w):
        m = self.to(alnmat, torch.uint8)
        n, L = m.shape
        cov = self.f32(21 * L, 21 * L)
        self.call("dmp_cov_build", m, w, n, L, cov)
        return cov

    def spd_inverse(self, a):
        a = a.clone()
        self.call("dmp_spd_inverse", a, a.shape[0])
        return a

    def dca_contacts(self, inv, L):
        out = self.f32(L, L)
        self.call("dmp_dca_contacts", inv, L, out)
        return out

    def dca_features(self, alnmat):
        m = self.to(alnmat, torch.uint8)
        n, L = m.shape
        out = self.f32(L, L, 442)
        self.call("dmp_dca_features", m, n, L, out)
        return out

    def gru_vertical(self, alnmat):
        m = self.to(alnmat, torch.uint8)
        n, L = m.shape
        out = self.f32(L, 512)
        self.call("dmp_gru_vertical", m, n, L, out)
        return out

    def gru_bidir(self, which, x):
        T = x.shape[0]
        out = self.f32(T, 512)
        self.call("dmp_gru_bidir", which, x, T, out)
        return out

    def stem_static(self, mat1d, inv, contacts):
        L = mat1d.shape[1]
        z0 = self.f32(384, L, L)
        self.call("dmp_stem_static", mat1d, inv, contacts, L, z0)
        return z0

    def stem_update(self, z0, dmap):
        L = dmap.shape[0]
        x = self.f32(128, L, L)
        self.call("dmp_stem_update", z0, dmap, L, x)
        return x

    def conv(self, block, x):
        L = x.shape[-1]
        u = self.f32(128, L, L)
        st = torch.empty((128, 2), dtype=torch.float64, device=self.dev)
        self.call("dmp_block_conv5x5_maxout", block, x, L, u, st)
        return u, st

    def norm(self, block, u, st, x):
        L = x.shape[-1]
        out = self.f32(128, L, L)
        self.call("dmp_block_norm_scse_residual", block, u, st, x, L, out)
        return out

    def conv_bwd(self, block, x, du, idx=None):
        L = x.shape[-1]
        dx, dw, db = self.f32(128, L, L), self.f32(512, 128, 5, 5), self.f32(512)
        self.call("dmp_block_conv5x5_maxout_bwd", block, x, du, idx, L, dx, dw, db)
        return dx, dw, db

    def conv_winners(self, block, x):
        """the float32 forward of a training step: maxout output + the winner of every quadruple (uint8)"""
        L = x.shape[-1]
        u = self.f32(128, L, L)
        idx = torch.empty((128, L, L), dtype=torch.uint8, device=self.dev)
        self.call("dmp_block_conv5x5_maxout_winners", block, x, L, u, idx)
        return u, idx

    def norm_bwd(self, block, u, dout):
        L = u.shape[-1]
        du, dparams = self.f32(128, L, L), self.f32(2433)
        self.call("dmp_block_norm_scse_residual_bwd", block, u, dout, L, du, dparams)
        return du, dparams

    def stem_winners(self, z0, dmap):
        L = dmap.shape[0]
        u = self.f32(128, L, L)
        idx = torch.empty((128, L, L), dtype=torch.uint8, device=self.dev)
        self.call("dmp_stem_maxout_winners", z0, dmap, L, u, idx)
        return u, idx

    def stem_bwd(self, u, idx, dy, mat1d, dmap):
        L = dmap.shape[0]
        dw, dparams, dmat1d = self.f32(384, 955), self.f32(640), self.f32(512, L)
        self.call("dmp_stem_bwd", u, idx, dy, mat1d, dmap, L, dw, dparams, dmat1d)
        return dw, dparams, dmat1d

    def head_bwd(self, x, g):
        L = x.shape[-1]
        dx, dparams = self.f32(128, L, L), self.f32(258)
        self.call("dmp_head_conv_bwd", x, g, L, dx, dparams)
        return dx, dparams

    def head_gram(self, x):
        L = x.shape[-1]
        conf, M = self.f32(L), self.f32(L, L)
        self.call("dmp_head_gram", x, L, conf, M)
        return conf, M

    def trunk_pass(self, z0, dmap):
        L = dmap.shape[0]
        conf, M = self.f32(L), self.f32(L, L)
        self.call("dmp_trunk_pass", z0, dmap, L, conf, M)
        return conf, M

    def conv_ms(self, z0, dmap, passes=1):
        """Mean duration (ms) of the 5x5 convolution launches of `passes` trunk passes, from the HIP events the library
        records around every launch (dmp_profile_enable / dmp_profile_conv_intervals)."""
        L = dmap.shape[0]
        cap = 16 * passes
        _lib.check(self.lib.dmp_profile_enable(self.eng.ctx, 1, cap))
        for _ in range(passes):
            self.trunk_pass(z0, dmap)
        torch.cuda.synchronize()
        a, b, n = (C.c_float * cap)(), (C.c_float * cap)(), C.c_int()
        _lib.check(self.lib.dmp_profile_conv_intervals(self.eng.ctx, self.eng.ctx, a, b, cap, C.byref(n)))
        _lib.check(self.lib.dmp_profile_enable(self.eng.ctx, 0, 0))
        return sum(b[i] - a[i] for i in range(n.value)) / max(1, n.value)

    def eigh_top8(self, M):
        L = M.shape[0]
        out = self.f32(L, 8)
        self.call("dmp_eigh_top8", M, L, out)
        return out

    def coords_from_mds(self, mat1d, mds):
        L = mds.shape[0]
        ca = self.f32(L, 3)
        self.call("dmp_coords_from_mds", mat1d, mds, L, ca)
        return ca

    def pair_distances(self, ca, clamp=1):
        L = ca.shape[0]
        d = self.f32(L, L)
        self.call("dmp_pair_distances", ca, L, clamp, d)
        return d

    def refine(self, ca, steps):
        ca = ca.clone()
        self.call("dmp_refine_coords", ca, ca.shape[0], steps)
        return ca

    def backbone(self, ca, logit):
        L = ca.shape[0]
        coords, conf = self.f32(L, 5, 3), self.f32(L)
        self.call("dmp_ca_to_backbone", ca, logit, L, coords, conf)
        return coords, conf
