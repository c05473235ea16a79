"""GAT attention aggregate (all heads fused) over the sl_gat_* kernels:
the per-head ``_aggregate_attention`` loop of shaDow/layers.py:560-582,612-619."""
import torch

from . import _lib, ops
from ._lib import check


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_self, z_neigh, attention, adj, act_code, heads):
        z_self, z_neigh = ops._f32c(z_self).contiguous(), ops._f32c(z_neigh).contiguous()
        att = attention.detach().float().contiguous()
        ops._need_cuda(z_self, z_neigh, att)
        n, F = z_self.shape
        dev = z_self.device
        c = adj.csr
        hn = torch.empty(n, F, device=dev)
        u_s = torch.empty(n, heads, device=dev); u_n = torch.empty(n, heads, device=dev)
        mx = torch.empty(n, heads, device=dev); den = torch.empty(n, heads, device=dev)
        nagg = torch.empty(n, F, device=dev)
        w = adj.edge_w
        check(_lib.load().sl_gat_fwd(c.indptr.data_ptr(), c.indices.data_ptr(), w.data_ptr() if w is not None else None,
                                     z_self.data_ptr(), z_neigh.data_ptr(), att.data_ptr(), act_code, n, F, heads,
                                     hn.data_ptr(), u_s.data_ptr(), u_n.data_ptr(), mx.data_ptr(), den.data_ptr(),
                                     nagg.data_ptr(), ops._stream(z_self)))
        ctx.save_for_backward(z_self, z_neigh, att, hn, u_s, u_n, mx, den, nagg)
        ctx.adj, ctx.meta = adj, (act_code, heads, attention.shape)
        return nagg

    @staticmethod
    def backward(ctx, dnagg):
        z_self, z_neigh, att, hn, u_s, u_n, mx, den, nagg = ctx.saved_tensors
        act_code, heads, att_shape = ctx.meta
        adj = ctx.adj
        c = adj.csr
        n, F = z_self.shape
        dev = z_self.device
        dnagg = ops._f32c(dnagg).contiguous()
        ti, tx, tp = c.transposed
        work = torch.empty(2 * c.e * heads + n * heads + 4, device=dev)
        dzs = torch.empty_like(z_self); dzn = torch.empty_like(z_neigh)
        datt = torch.empty(2, F, device=dev)
        w = adj.edge_w
        check(_lib.load().sl_gat_bwd(c.indptr.data_ptr(), c.indices.data_ptr(), ti.data_ptr(), tx.data_ptr(), tp.data_ptr(),
                                     w.data_ptr() if w is not None else None, z_self.data_ptr(), z_neigh.data_ptr(),
                                     att.data_ptr(), act_code, n, c.e, F, heads, hn.data_ptr(), u_s.data_ptr(),
                                     u_n.data_ptr(), mx.data_ptr(), den.data_ptr(), nagg.data_ptr(), dnagg.data_ptr(),
                                     work.data_ptr(), dzs.data_ptr(), dzn.data_ptr(), datt.data_ptr(), ops._stream(dnagg)))
        return dzs, dzn, datt.reshape(att_shape), None, None, None


def gat_aggregate(adj: "ops.NormAdj", z_self, z_neigh, attention, act: str, heads: int):
    """N = softmax_row(lrelu(a_s.act(z_self)) + lrelu(a_n.act(z_neigh))) @ act(z_neigh), per head."""
    return _GatAggregate.apply(z_self, z_neigh, attention, adj, ops.ACT_CODE[act], int(heads))
