"""Which stage of the GAT forward differs between the paired-Linear tail path and the node-pass path, bit for bit?"""
import numpy as np
import torch
from shadow_gnn_amd import _lib, layers, ops, ops_gat
from tests.test_layers_gpu import _bench_scale_batch

DEV = "cuda:0"
lib = _lib.load()


def cmp(name, a, b):
    a, b = a.detach().float(), b.detach().float()
    ne = int((a != b).sum())
    print(f"{name:28s} differing {ne:9d} / {a.numel():9d}   max |diff| {float((a - b).abs().max()):.3e}")


for act in ("elu", "relu"):
    print("== act", act)
    b, X, labels, F0, C = _bench_scale_batch("gat", 96, F0=100)
    csr = ops.DeviceCSR(b.indptr, b.indices, subg_off=b.subg_node_off, subg_edge_off=b.subg_edge_off, max_subg_nodes=b.counts["max_subg_nodes"])
    torch.manual_seed(3)
    lay = layers.GAT(100, 256, dropout=0.0, act=act, norm="norm_feat", mulhead=4).to(DEV)
    with torch.no_grad():
        lay.scale.add_(0.1 * torch.randn_like(lay.scale)); lay.offset.add_(0.1 * torch.randn_like(lay.offset))
    Xd = X.to(DEV)
    na = ops.NormAdj(csr, edge_w=ops.dropedge_mask(csr, 0.1))
    n, F, H = Xd.shape[0], 256, 4
    with torch.no_grad():
        # path A: paired Linear with the GAT tail
        ops.GAT_PAIR_TAIL = True
        pre = ops.GatPre(lay.attention, ops.ACT_CODE[act], H)
        zsA, hnA = ops.linear_pair(Xd, lay.f_lin[0], lay.f_lin[1], gat=pre)
        assert pre.filled
        # path B: plain paired Linear + node pass
        zsB, znB = ops.linear_pair(Xd, lay.f_lin[0], lay.f_lin[1], gat=None)
        cmp("z_self", zsA, zsB)
        hnB = torch.empty(n, F, device=DEV); usB = torch.empty(n, H, device=DEV); unB = torch.empty(n, H, device=DEV)
        mxB = torch.empty(n, H, device=DEV); denB = torch.empty(n, H, device=DEV); naggB = torch.empty(n, F, device=DEV)
        att = lay.attention.detach().float().contiguous()
        w = na.edge_w
        _lib.check(lib.sl_gat_fwd(csr.indptr.data_ptr(), csr.indices.data_ptr(), w.data_ptr() if w is not None else None, zsB.data_ptr(), znB.data_ptr(),
                                  att.data_ptr(), ops.ACT_CODE[act], n, F, H, hnB.data_ptr(), usB.data_ptr(), unB.data_ptr(), mxB.data_ptr(), denB.data_ptr(),
                                  naggB.data_ptr(), ops._stream(zsB)))
        cmp("hn (gemm tail vs node pass)", hnA, hnB); cmp("u_s", pre.u_s, usB); cmp("u_n", pre.u_n, unB)
        # row pass alone on path A's inputs
        mxA = torch.empty(n, H, device=DEV); denA = torch.empty(n, H, device=DEV); naggA = torch.empty(n, F, device=DEV)
        _lib.check(lib.sl_gat_fwd_rows(csr.indptr.data_ptr(), csr.indices.data_ptr(), w.data_ptr() if w is not None else None, hnA.data_ptr(), pre.u_s.data_ptr(),
                                       pre.u_n.data_ptr(), n, F, H, mxA.data_ptr(), denA.data_ptr(), naggA.data_ptr(), ops._stream(zsB)))
        cmp("nagg rows(A) vs fwd(B)", naggA, naggB); cmp("den", denA, denB)
        sc = lay.scale.reshape(2, F).contiguous().float(); of = lay.offset.reshape(2, F).contiguous().float()
        for drop in ((0.0, 0), (0.3, 12345)):
            outS = ops._an_fwd([naggA, zsA], [None, None], (0, ops.ACT_CODE[act]), sc, of, 64, 0.5, drop)
            mxT = torch.empty(n, H, device=DEV); denT = torch.empty(n, H, device=DEV); naggT = torch.empty(n, F, device=DEV); outT = torch.empty(n, F, device=DEV)
            amT = torch.empty(n, device=DEV)
            _lib.check(lib.sl_gat_fwd_tail(csr.indptr.data_ptr(), csr.indices.data_ptr(), w.data_ptr() if w is not None else None, hnA.data_ptr(), pre.u_s.data_ptr(),
                                           pre.u_n.data_ptr(), zsA.data_ptr(), ops.ACT_CODE[act], sc.data_ptr(), of.data_ptr(), n, F, H, 0.5, float(drop[0]), int(drop[1]),
                                           mxT.data_ptr(), denT.data_ptr(), naggT.data_ptr(), outT.data_ptr(), amT.data_ptr(), ops._stream(zsB)))
            cmp(f"nagg tail vs rows p={drop[0]}", naggT, naggA); cmp(f"out tail vs act_norm p={drop[0]}", outT, outS)
            am = ops.get_row_amax(outS)
            if am is not None:
                cmp("row amax", amT, am)
torch.cuda.synchronize()
