"""Minibatch orchestration with the reference's API surface
(shaDow/minibatch.py:94-495): ``OneBatchSubgraph`` record and
``MinibatchShallowExtractor`` with epoch_start_reset / shuffle_entity /
one_batch / is_end_epoch / epoch_end_reset.

Fast path: one sampler call returns the whole batch in block-diagonal form in
HBM (no host pool of per-subgraph objects, no scipy, no host<->device copies):
features are gathered by a HIP kernel from the device-resident feature matrix,
the adjacency is an ``ops.DeviceCSR`` that layer 0 recognises.  The next batch
is sampled on a side stream while the current one trains."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import ops
from .sampler import DeviceBatch, HipSampler, SamplerConfig, SubgraphCache

REUSABLE_SAMPLER = {"ppr"}          # CONFIG_TEMPLATE.yml:16-17 (algorithm.sampler.deterministic)

TRAIN, VALID, TEST = 0, 1, 2          # graph_engine.frontend mode constants
STR2MODE = {"train": TRAIN, "valid": VALID, "test": TEST}
MODE2STR = {TRAIN: "train", VALID: "valid", TEST: "test"}


@dataclass
class OneBatchSubgraph:
    """data returned by one minibatch (shaDow/minibatch.py:94-140)"""
    adj_ens: List[Any]
    feat_ens: List[torch.Tensor]
    label: torch.Tensor
    size_subg_ens: Optional[torch.Tensor]
    target_ens: List[Any]
    feat_aug_ens: Optional[List[Dict[str, Any]]]
    idx_raw: Optional[List[Any]] = None
    tail_ens: Optional[List[Any]] = None      # per-branch target-only-tail plan (tail.py), built while prefetching

    @property
    def num_ens(self):
        return len(self.adj_ens)

    @property
    def batch_size(self):
        return int(self.target_ens[0].numel() if torch.is_tensor(self.target_ens[0]) else self.target_ens[0].size)

    def __post_init__(self):
        assert len(self.feat_ens) == self.num_ens and len(self.target_ens) == self.num_ens
        if self.size_subg_ens is not None:
            assert self.size_subg_ens.shape[0] == self.num_ens
        if self.feat_aug_ens is not None:
            assert len(self.feat_aug_ens) == self.num_ens

    def pop_idx_raw(self):
        ret, self.idx_raw = self.idx_raw, None
        return ret

    def to_dict(self, keys=None):
        if keys is None:
            keys = self.__dataclass_fields__
        return {k: getattr(self, k) for k in keys}


def hop2onehot(hop: torch.Tensor, dim_1hot_vec: int) -> torch.Tensor:
    """EntityEncoding.hop2onehot_vec (frontend/graph.py:134-147) on the device:
    hop h in [0, dim-2] -> column h+1, unreachable (0xFFFFFFFF, i.e. -1 as int32)
    and hops >= 255 -> column 0, other hops -> all-zero row."""
    h = hop.long()
    n = h.numel()
    col = torch.where((h >= 0) & (h <= dim_1hot_vec - 2), h + 1, torch.full_like(h, -1))
    col = torch.where((h < 0) | (h >= 255), torch.zeros_like(h), col)
    out = torch.zeros(n, dim_1hot_vec, dtype=torch.float32, device=hop.device)
    valid = col >= 0
    out[torch.nonzero(valid).squeeze(1), col[valid]] = 1.0
    return out


class MinibatchShallowExtractor:
    """Node-task minibatch loop over a device-resident graph (fast path of
    shaDow/minibatch.py:143-495).

    adjs:        {mode: (indptr, indices)} uint32 CSR per mode (numpy, or int32 torch tensors on the device)
    entity_set:  {mode: array of root node ids}
    sampler_config: dict like one entry of the reference's sampler section, e.g.
                 {"method": "khop", "depth": 2, "budget": 20, "add_self_edge": False}
    """
    def __init__(self, adjs, entity_set, sampler_config: Dict[str, Any], aug_feats, feat_full: torch.Tensor,
                 label_full: torch.Tensor, batch_size: int, device, seed_cpp: int = -1,
                 rank: int = 0, world_size: int = 1, prefetch: bool = True, nocache_modes=()):
        self.device = torch.device(device)
        self.aug_feats = set(aug_feats)
        self.raw_entity_set = {m: np.asarray(v) for m, v in entity_set.items()}
        self.feat_full = feat_full.to(self.device)
        self.label_full = label_full.to(self.device)
        self.batch_size_global = int(batch_size)
        self.rank, self.world_size = rank, world_size
        assert batch_size % world_size == 0, "global batch must divide evenly over the ranks"
        self.batch_size = {m: batch_size // world_size for m in (TRAIN, VALID, TEST)}
        cfg = dict(sampler_config)
        method = cfg.pop("method")
        self.sampler_cfg = SamplerConfig(
            method=method, num_roots=1, depth=int(cfg.get("depth", 2)), budget=int(cfg.get("budget", -1)),
            k=int(cfg.get("k", 0)), threshold=float(cfg.get("threshold", 0.0)),
            add_self_edge=bool(cfg.get("add_self_edge", False)),
            aug=tuple(sorted(self.aug_feats & {"hops", "pprs", "drnls"})))
        self.graph_sampler = {}
        self._adjs = adjs
        self._seed = seed_cpp
        self.entity_epoch = {m: None for m in (TRAIN, VALID, TEST)}
        self.label_epoch = {m: None for m in (TRAIN, VALID, TEST)}
        self.idx_entity_evaluated = {m: 0 for m in (TRAIN, VALID, TEST)}
        self.end_epoch = {m: False for m in (TRAIN, VALID, TEST)}
        self.batch_num = -1
        self.dim_1hot_hop, self.dim_1hot_ppr, self.dim_1hot_drnl = 5 + 2, 1, 25 + 1   # minibatch.py:246-248
        self.prefetch = prefetch
        # > 0: every batch carries a target-only-tail plan for a model of that many layers (DeepGNN.prune_tail)
        self.tail_plan_layers = 0
        self.tail_plan_square = False      # True for GAT stacks: prepare the square form of every level instead
        self._side = torch.cuda.Stream(device=self.device) if prefetch else None
        self._inflight = {}
        # record -> reuse of sampled subgraphs for deterministic samplers (minibatch.py:306-339, :403-426)
        self.nocache_modes = set(nocache_modes)
        self.record_subgraphs = {}
        self.cache_subg = {}
        self._roots_dev = {}
        self._cursor = {m: 0 for m in (TRAIN, VALID, TEST)}
        self._hwm = {m: [1, 1] for m in (TRAIN, VALID, TEST)}       # largest batch seen (nodes, edges)

    # ------------------------------------------------------------------ API
    def get_aug_dim(self, aug_type):
        return getattr(self, f'dim_1hot_{aug_type[:-1]}')

    def epoch_start_reset(self, epoch, mode):
        self.batch_num = -1
        if mode not in self.graph_sampler:
            ip, ix = self._adjs[mode]
            self.graph_sampler[mode] = hs = HipSampler(ip, ix, device=self.device, seed=self._seed)
            reusable = self.sampler_cfg.method in REUSABLE_SAMPLER and mode not in self.nocache_modes
            self.record_subgraphs[mode] = "record" if reusable else ("noncache" if mode in self.nocache_modes else "none")
            if reusable:
                self.cache_subg[mode] = SubgraphCache(hs.num_nodes(), self.device)

    def shuffle_entity(self, mode, perm=None):
        """YOU MUST CALL THIS BEFORE STARTING ANY EPOCH (minibatch.py:269-280).  Every
        rank draws the same permutation and keeps its own slice of each global batch."""
        ent = self.raw_entity_set[mode]
        if perm is None:
            perm = np.random.permutation(ent.size)
        ent = ent[perm]
        B, G, r = self.batch_size_global, self.world_size, self.rank
        nfull = (ent.size // B) * B
        body = ent[:nfull].reshape(-1, G, B // G)[:, r, :].reshape(-1)
        tail = ent[nfull:]
        per = -(-tail.size // G)                      # ceil: the last, smaller global batch
        mine = np.concatenate([body, tail[r * per:(r + 1) * per]])
        self.entity_epoch[mode] = mine
        self.label_epoch[mode] = self.label_full[torch.as_tensor(mine.astype(np.int64), device=self.device)]
        self.graph_sampler[mode].shuffle_targets(mine.astype(np.uint32))
        self._roots_dev[mode] = torch.as_tensor(mine.astype(np.uint32).view(np.int32)).to(self.device)
        self._cursor[mode] = 0
        self.idx_entity_evaluated[mode] = 0
        self.end_epoch[mode] = False
        self._inflight.pop(mode, None)

    def is_end_epoch(self, mode):
        return self.end_epoch[mode]

    def epoch_end_reset(self, mode, drop_full_graph: bool = False):
        """After the first full epoch of a deterministic sampler the recorded subgraphs are
        reused (minibatch.py:326-334); ``drop_full_graph`` is the reference's optm_level 'high'
        (:335-341): the full CSR is freed once nothing samples from it any more."""
        self.end_epoch[mode] = False
        if self.record_subgraphs.get(mode) == "record" and not self.cache_subg[mode].is_empty():
            self.record_subgraphs[mode] = "reuse"
            if drop_full_graph:
                self.drop_full_graph_info(mode)

    def drop_full_graph_info(self, mode):
        self.graph_sampler[mode].drop_full_graph_info()

    def disable_cache(self, mode):
        """minibatch.py:490-492: stop recording / reusing subgraphs of this mode."""
        self.nocache_modes.add(mode)
        if mode in self.record_subgraphs:
            self.record_subgraphs[mode] = "noncache"
            self._inflight.pop(mode, None)

    # ------------------------------------------------------------- batching
    def _tail_plan(self, subgs, adj, targets):
        """Target-only-tail plan of this batch (tail.py) on the prefetch stream: its host syncs wait for the
        side stream only, the training stream keeps running the previous step.
        Stream-ordered allocation: without this, blocks of the side-stream pool are only reused after _launch's
        side.wait_stream(main), i.e. after every consumer on the training stream.  The plan allocates on the
        side stream BEFORE that point, so the sampler outputs (and the plan) are recorded on the training
        stream: the allocator then defers their reuse until the training stream is done with them."""
        from . import tail
        if self._side is None:
            return tail.build_tail_plan(adj, targets, self.tail_plan_layers, eager_transpose=("square" if self.tail_plan_square else True))
        main = torch.cuda.current_stream(self.device)
        for t in (subgs.node, subgs.indptr, subgs.indices, subgs.edge_id, subgs.target, subgs.subg_node_off,
                  subgs.subg_edge_off, subgs.ppr, subgs.hop, subgs.drnl):
            if t is not None and t.is_cuda:
                t.record_stream(main)
        with torch.cuda.stream(self._side):
            levels = tail.build_tail_plan(adj, targets, self.tail_plan_layers, eager_transpose=("square" if self.tail_plan_square else True))
        main.wait_stream(self._side)
        for lv in levels:
            for t in lv.tensors():
                t.record_stream(main)
        return levels

    def _launch(self, mode):
        hs = self.graph_sampler[mode]
        reuse = self.record_subgraphs.get(mode) == "reuse"
        bs = min(self.batch_size[mode], self._roots_dev[mode].numel() - self._cursor[mode])

        def go():
            if reuse:
                roots = self._roots_dev[mode][self._cursor[mode]:self._cursor[mode] + bs]
                hn, he = self._hwm[mode]
                self.cache_subg[mode].collate_async(roots, hn + hn // 8 + 64, he + he // 8 + 64,
                                                    want_hop="hops" in self.aug_feats)
            else:
                hs.sample_async(self.sampler_cfg, self.batch_size[mode])
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                go()
        else:
            go()
        self._cursor[mode] += bs
        self._inflight[mode] = "reuse" if reuse else "sample"

    def _collect(self, mode) -> DeviceBatch:
        hs = self.graph_sampler[mode]
        kind = self._inflight[mode]

        def go():
            if kind == "reuse":
                return self.cache_subg[mode].finish()
            b = hs.finish()
            if self.record_subgraphs.get(mode) == "record":
                self.cache_subg[mode].record(b)                      # minibatch.py:407-412
            return b
        if self._side is not None:
            with torch.cuda.stream(self._side):
                b = go()
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        else:
            b = go()
        self._inflight.pop(mode, None)
        self._hwm[mode] = [max(self._hwm[mode][0], b.num_nodes), max(self._hwm[mode][1], b.num_edges)]
        return b

    def one_batch(self, mode=TRAIN, ret_raw_idx=False) -> OneBatchSubgraph:
        remaining = self.entity_epoch[mode].shape[0] - self.idx_entity_evaluated[mode]
        batch_size_ = min(remaining, self.batch_size[mode])
        launch_next = False
        if mode not in self._inflight:
            self._launch(mode)
        subgs = self._collect(mode)
        assert subgs.num_subgraphs == batch_size_, (subgs.num_subgraphs, batch_size_)
        i0 = self.idx_entity_evaluated[mode]
        self.idx_entity_evaluated[mode] += batch_size_
        self.batch_num += 1
        if self.idx_entity_evaluated[mode] >= self.entity_epoch[mode].shape[0]:
            self.idx_entity_evaluated[mode] = 0
            self.end_epoch[mode] = True
            if self.record_subgraphs.get(mode) != "reuse":
                assert self.graph_sampler[mode].get_idx_root() == 0      # samplers_ensemble.py:298-301
        elif self.prefetch:
            launch_next = True
        adj = ops.DeviceCSR(subgs.indptr, subgs.indices, subg_off=subgs.subg_node_off,
                            subg_edge_off=subgs.subg_edge_off, max_subg_nodes=subgs.counts["max_subg_nodes"])
        tail_plan = self._tail_plan(subgs, adj, subgs.target) if self.tail_plan_layers > 0 else None
        if launch_next:
            self._launch(mode)        # overlap the next sampler call with this batch's training
        feat = ops.gather_rows(self.feat_full, subgs.node)           # minibatch.py:469
        label = self.label_epoch[mode][i0:i0 + batch_size_]
        feat_aug = {}
        # entity encodings (frontend/graph.py:134-172) as per-node bit masks; the model's augmentation
        # Linear consumes them fused (ops.onehot_linear_add), .dense() gives the reference's matrix
        if "hops" in self.aug_feats:
            feat_aug["hops"] = ops.OneHotCodes(ops.encode_codes("hops", subgs.hop, self.dim_1hot_hop), self.dim_1hot_hop)
        if "pprs" in self.aug_feats:
            feat_aug["pprs"] = ops.OneHotCodes(ops.encode_codes("pprs", subgs.ppr, self.dim_1hot_ppr), self.dim_1hot_ppr)
        if "drnls" in self.aug_feats:
            feat_aug["drnls"] = ops.OneHotCodes(ops.encode_codes("drnls", subgs.drnl, self.dim_1hot_drnl), self.dim_1hot_drnl)
        size_subg = subgs.size_subg.unsqueeze(0)
        ret = OneBatchSubgraph([adj], [feat], label, size_subg, [subgs.target], [feat_aug])
        if tail_plan is not None:
            ret.tail_ens = [tail_plan]
        ret.device_batch = subgs
        if ret_raw_idx:
            ret.idx_raw = [subgs.node]
        return ret
