"""Top-k approximate PPR tables on the GPU (the reference's
ParallelSampler::preproc_ppr_approximate, ParallelSampler.cpp:237-344).

The push runs in libshadow_hip.so (sg_ppr_push, one wavefront per target,
bit-exact with the reference's smallest-id-first push order); the final
ordering by (-score, id) and the cut to k entries (.cpp:320-339) are three
stable device sorts over the flat touched list."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


PUSH_MODE = {"ordered": 0, "fifo": 1}


def ppr_approximate_device(hs, targets, k: int, alpha: float = 0.85, epsilon: float = 1e-5,
                           hash_slots: int = 1 << 15, num_waves: int = 2048, chunk: int = 1 << 16,
                           order: str = "ordered"):
    """Returns (len[T] uint32, neigh[T,k] uint32, score[T,k] float32) for `targets`
    (numpy), computed on hs.device.  `hs` is a HipSampler (owner of the CSR in HBM).
    ``order``: "ordered" reproduces the reference's tables bit for bit (smallest pending id first);
    "fifo" pushes in discovery order -- same error bound, several times faster, tables agree with the
    reference's within the approximation error (use it when the cache files need not be interchangeable)."""
    mode = PUSH_MODE[order]
    lib = _lib.load()
    dev = hs.device
    targets = np.ascontiguousarray(np.asarray(targets).reshape(-1), dtype=np.uint32)
    T = targets.size
    out_len = np.zeros(T, dtype=np.uint32)
    out_nb = np.full((T, k), 0xFFFFFFFF, dtype=np.uint32)
    out_sc = np.zeros((T, k), dtype=np.float32)
    d_ip, d_ix = lib.sg_device_indptr(hs._h), lib.sg_device_indices(hs._h)
    N = hs.num_nodes()
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        for c0 in range(0, T, chunk):
            tg = targets[c0:c0 + chunk]
            Tc = tg.size
            d_t = torch.from_numpy(tg.view(np.int32)).to(dev)
            cap = max(1 << 20, Tc * 1024)
            while True:
                waves = max(4, min(num_waves, ((Tc + 3) // 4) * 4))
                work = torch.empty(waves * hash_slots * 21 + 256, dtype=torch.uint8, device=dev)
                cnt = torch.empty(Tc, dtype=torch.int32, device=dev)
                off = torch.empty(Tc, dtype=torch.int64, device=dev)
                o_node = torch.empty(cap, dtype=torch.int32, device=dev)
                o_score = torch.empty(cap, dtype=torch.float32, device=dev)
                total, flags = C.c_uint64(), C.c_uint32()
                rc = lib.sg_ppr_push(d_ip, d_ix, N, d_t.data_ptr(), Tc, alpha, epsilon, hash_slots, waves,
                                     work.data_ptr(), work.numel(), cnt.data_ptr(), off.data_ptr(),
                                     o_node.data_ptr(), o_score.data_ptr(), cap, C.byref(total), C.byref(flags), mode,
                                     stream)
                if rc == _lib.SG_ERR_CAPACITY:
                    if flags.value & 1:
                        hash_slots *= 4
                        if hash_slots > (1 << 24):
                            check(rc)
                    if flags.value & 2:
                        cap = int(total.value) + 1024
                    continue
                check(rc)
                break
            n_ent = int(total.value)
            cnt64 = cnt.long()
            # segment id of every entry, then order by (segment, -score, id) with stable sorts
            seg_of_target = torch.argsort(off)                      # targets in output order
            seg = torch.repeat_interleave(seg_of_target, cnt64[seg_of_target])
            node = o_node[:n_ent].long() & 0xFFFFFFFF
            score = o_score[:n_ent]
            o1 = torch.sort(node, stable=True).indices
            o2 = torch.sort(score[o1], descending=True, stable=True).indices
            o12 = o1[o2]
            o3 = torch.sort(seg[o12], stable=True).indices
            perm = o12[o3]
            seg_s, node_s, score_s = seg[perm], node[perm], score[perm]
            start = torch.cumsum(cnt64, 0) - cnt64                   # per target, in target order (segments sorted by id)
            rank = torch.arange(n_ent, device=dev) - start[seg_s]
            keep = rank < k
            tsel, rsel = seg_s[keep].cpu().numpy(), rank[keep].cpu().numpy()
            out_nb[c0 + tsel, rsel] = node_s[keep].cpu().numpy().astype(np.uint32)
            out_sc[c0 + tsel, rsel] = score_s[keep].cpu().numpy()
            out_len[c0:c0 + Tc] = np.minimum(cnt.cpu().numpy().astype(np.int64), k).astype(np.uint32)
    return out_len, out_nb, out_sc
