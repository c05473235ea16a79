"""Minibatch orchestration with the reference's API surface
(shaDow/minibatch.py:94-495): ``OneBatchSubgraph`` record and
``MinibatchShallowExtractor`` with epoch_start_reset / shuffle_entity /
one_batch / is_end_epoch / epoch_end_reset / disable_cache.

Fast path: one sampler call returns the whole batch in block-diagonal form in
HBM (no host pool of per-subgraph objects, no scipy, no host<->device copies):
features are gathered by a HIP kernel from the device-resident feature matrix,
the adjacency is an ``ops.DeviceCSR`` that layer 0 recognises.  The next batch
is sampled on a side stream while the current one trains.

Data parallel (one process per GPU): ``plan_epoch`` gives every rank the SAME
number of steps per epoch; a rank whose share of the last, smaller global batch
is empty still takes the step (with an empty batch record) so that the gradient
all-reduce stays collective, and every batch carries ``loss_weight`` = its share
B_r / B_t of the step's global batch so that a SUM all-reduce yields the gradient
of the mean loss over the global batch (CE is a mean over roots,
shaDow/models.py:166)."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import os
import weakref

import numpy as np
import torch

from . import ops
from .sampler import DeviceBatch, HipSampler, SamplerConfig, SubgraphCache

REUSABLE_SAMPLER = {"ppr"}          # CONFIG_TEMPLATE.yml:16-17 (algorithm.sampler.deterministic)

TRAIN, VALID, TEST = 0, 1, 2          # graph_engine.frontend mode constants
STR2MODE = {"train": TRAIN, "valid": VALID, "test": TEST}
MODE2STR = {TRAIN: "train", VALID: "valid", TEST: "test"}
_MODES = (TRAIN, VALID, TEST)


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
    loss_weight: float = 1.0                  # this rank's share of the step's global batch (data parallel)

    @property
    def num_ens(self):
        return len(self.adj_ens)

    @property
    def batch_size(self):
        return int(self.target_ens[0].numel() if torch.is_tensor(self.target_ens[0]) else np.size(self.target_ens[0]))

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


def plan_epoch(order: np.ndarray, batch_global: int, world: int, rank: int,
               static_partition: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Split one epoch's entity order over the ranks.  ``order``: positions into the mode's raw entity
    set, in epoch order (identical on every rank).  Returns (mine, local_sizes[T], global_sizes[T]):
    the positions this rank samples, in its own order, how many of them it takes at each of the
    T = ceil(E / B) steps (possibly 0 at the end), and the size of every step's global batch.

    Default: global batch t is ``order[t*B:(t+1)*B]`` -- exactly the single-process batch -- cut into
    ``world`` contiguous slices whose sizes differ by at most one.
    ``static_partition``: position p always belongs to rank p % world (needed by the per-rank subgraph
    cache: a root recorded in epoch 1 must come back to the same rank); a rank spreads its share evenly
    over the T steps, so a global batch is the union of the ranks' local batches."""
    order = np.asarray(order).reshape(-1)
    E, B, G = int(order.size), int(batch_global), int(world)
    assert B >= 1 and G >= 1 and 0 <= rank < G
    T = -(-E // B) if E else 0
    if not static_partition:
        sizes = np.minimum(B, E - B * np.arange(T, dtype=np.int64))          # B, ..., B, tail
        base, rem = sizes // G, sizes % G
        local_all = base[None, :] + (np.arange(G)[:, None] < rem[None, :])      # [G, T]
        start = B * np.arange(T, dtype=np.int64) + rank * base + np.minimum(rank, rem)
        mine = (np.concatenate([order[s:s + c] for s, c in zip(start, local_all[rank])])
                if T else order[:0])
        return mine, local_all[rank].astype(np.int64), sizes.astype(np.int64)
    owner = order % G
    per_rank = np.bincount(owner, minlength=G).astype(np.int64)
    t = np.arange(max(T, 1), dtype=np.int64)[None, :]
    local_all = per_rank[:, None] // max(T, 1) + (t < (per_rank[:, None] % max(T, 1)))
    local_all = local_all[:, :T]
    return order[owner == rank], local_all[rank].astype(np.int64), local_all.sum(0).astype(np.int64)


def _parse_sampler_section(section: Dict[str, Any]) -> Tuple[int, Dict[str, Any]]:
    """The reference's ``sampler_config_ensemble`` = {"batch_size": B, "configs": [{"method": m, key: [v], ...}]}
    (shaDow/minibatch.py:213-221, :344-358).  One branch only: subgraph ensembles are outside the hot path."""
    cfgs = list(section["configs"])
    if len(cfgs) != 1:
        raise NotImplementedError("subgraph ensembles (several sampler configs) are outside the hot path built here")
    flat = {}
    for key, val in cfgs[0].items():
        if key != "method" and isinstance(val, (list, tuple)):
            if len(val) != 1:
                raise NotImplementedError("subgraph ensembles (several values per sampler key) are outside the hot path")
            val = val[0]
        flat[key] = val
    if flat.get("method") == "full":
        raise NotImplementedError("the 'full' (no sampling) mode is a preprocessing path of the reference, not the hot path")
    return int(section["batch_size"]), flat


def _as_uint32_csr(adj):
    """scipy CSR (the reference's type) or an (indptr, indices) pair -> what HipSampler takes."""
    if isinstance(adj, (tuple, list)):
        return adj[0], adj[1]
    return np.ascontiguousarray(adj.indptr, dtype=np.uint32), np.ascontiguousarray(adj.indices, dtype=np.uint32)


class MinibatchShallowExtractor:
    """Node-task minibatch loop over a device-resident graph.  The constructor takes the reference's
    argument list (shaDow/minibatch.py:154-174) plus keyword-only placement arguments; ``on_device``
    is the short form used by bench.py and the tests.

    NOTE on batch_size (as in the reference): the number of target nodes per gradient update -- with
    ``world_size`` ranks it is the GLOBAL batch; every rank samples its share of it."""

    def __init__(self, name_data, dir_data, adjs, entity_set, sampler_config_ensemble, aug_feats, percent_per_epoch,
                 feat_full: torch.Tensor, label_full: torch.Tensor, dim_feat_raw: int, is_transductive: bool,
                 parallelism: int = 0, full_tensor_on_gpu: bool = True, bin_adj_files=None, nocache_modes=frozenset(),
                 optm_level: str = "high", seed_cpp: int = -1, metrics_profile=None, *,
                 device=None, rank: Optional[int] = None, world_size: Optional[int] = None, prefetch: bool = True):
        if isinstance(entity_set.get(TRAIN), dict):
            raise NotImplementedError("link prediction (pos / neg edge sets) is outside the hot path built here")
        if not full_tensor_on_gpu:
            raise ValueError("the HIP path keeps the feature matrix in HBM (the reference's --full_tensor_on_gpu)")
        if rank is None or world_size is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            rank = dist.get_rank() if on else 0
            world_size = dist.get_world_size() if on else 1
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None and self.device.type == "cuda":
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.name_data, self.dir_data = name_data, dir_data
        self.is_transductive, self.optm_level = bool(is_transductive), optm_level
        self.dim_feat_raw = int(dim_feat_raw)
        self.prediction_task = "node"
        self.aug_feats = set(aug_feats)
        self.raw_entity_set = {m: np.asarray(v) for m, v in entity_set.items()}
        self.feat_full = feat_full.to(self.device)
        self.label_full = label_full.to(self.device)
        self.rank, self.world_size = int(rank), int(world_size)
        self.batch_size_global, scfg = _parse_sampler_section(sampler_config_ensemble)
        self.sampler_cfg = SamplerConfig(
            method=scfg["method"], num_roots=1, depth=int(scfg.get("depth", 2)), budget=int(scfg.get("budget", -1)),
            k=int(scfg.get("k", 0)), threshold=float(scfg.get("threshold", 0.0)),
            add_self_edge=bool(scfg.get("add_self_edge", False)),
            aug=tuple(sorted(self.aug_feats & {"hops", "pprs", "drnls"})))
        self._ppr_args = dict(alpha=float(scfg.get("alpha", 0.85)), epsilon=float(scfg.get("epsilon", 1e-5)))
        # the largest local share of a global batch (what one sampler call is sized for)
        self.batch_size = {m: -(-self.batch_size_global // self.world_size) for m in _MODES}
        self.percent_per_epoch = {m: 1.0 for m in _MODES}
        for key, val in (percent_per_epoch or {}).items():
            self.percent_per_epoch[STR2MODE[key] if isinstance(key, str) else key] = float(val)
        self.num_ensemble = 1
        self.graph_sampler: Dict[int, HipSampler] = {}
        self._adjs, self._bin_adj_files, self._seed = adjs, bin_adj_files, seed_cpp
        self.seed_cpp = seed_cpp
        self.entity_epoch = {m: None for m in _MODES}
        self.label_epoch = {m: None for m in _MODES}
        self.idx_entity_evaluated = {m: 0 for m in _MODES}
        self.end_epoch = {m: False for m in _MODES}
        self.batch_num = -1
        self.dim_1hot_hop, self.dim_1hot_ppr, self.dim_1hot_drnl = 5 + 2, 1, 25 + 1   # minibatch.py:246-248
        self.prefetch = bool(prefetch)
        # True: batches carry ops.LazyRows(feat_full, node) instead of the gathered matrix -- layer 0 of GCN / GraphSAGE
        # then reads feat_full[node] inside its aggregation kernel (everything else materialises it on first use)
        self.lazy_features = False
        # > 0: every batch carries a target-only-tail plan for a model of that many layers (DeepGNN.prune_tail)
        self.tail_plan_layers = 0
        self.tail_plan_square = False      # True for GAT stacks: prepare the square form of every level instead
        # True: every batch carries the row sets of the row-sparse top-layer backward (tail.TopBackwardPlan; node tasks whose
        # read-out takes the roots' rows of the last GraphSAGE layer), built on the prefetch stream like the tail plan
        self.top_backward_plan = False
        self.top_plan_filter = True        # (the plan's filtered transposed structure is wanted: attach_model narrows it to its one reader)
        # > 0: every batch carries up to that many nested levels of a row-sparse backward pass instead (GAT stacks on deep
        # subgraphs: tail.build_backward_levels)
        self.backward_levels = 0
        # (priority -1 = high: when the deferred sampler call meets the GEMMs of the step, its workgroups take the CU slots
        #  as they free up instead of queueing behind the GEMM's grid; SHADOW_PREFETCH_PRIORITY=0 for a normal stream)
        prio = int(os.environ.get("SHADOW_PREFETCH_PRIORITY", "-1"))
        self._side = torch.cuda.Stream(device=self.device, priority=prio) if self.prefetch else None
        # prefetch launched when the consumer reaches ops.fire_deferred (see ops.DEFER_POINT: once the GNN layers of the forward
        # pass are enqueued) instead of at once: it then runs in the nearly idle read-out / loss stretch of the step instead of
        # on its HBM-bound head (round-3 A/B); SHADOW_DEFER_PREFETCH=0 restores the immediate launch
        self.defer_prefetch = os.environ.get("SHADOW_DEFER_PREFETCH", "1") != "0"
        self._inflight: Dict[int, Tuple[str, int, int]] = {}   # mode -> (kind, roots in the call, epoch cursor at its start)
        # Sampler calls that cover several steps (HipSampler.sample_multi_async -> sg_sample_multi): the pipeline's four
        # dependent kernel launches cost ~0.08 ms whatever the call's size -- a third of a 1 024-root call -- and the draws are
        # keyed on the subgraph's serial number, so S steps' batches from one call are bit-identical to S calls.  The batches
        # of a call wait in ``_ready`` until their step comes.  1 = one call per step (the reference's rhythm).
        # Default 4 = what bench.py times (the published configuration is the library's default).
        self.steps_per_call = max(1, min(int(os.environ.get("SHADOW_SAMPLER_STEPS_PER_CALL", "4")), 16))
        self._ready: Dict[int, list] = {m: [] for m in _MODES}  # collected batches of later steps, in step order
        # record -> reuse of sampled subgraphs for deterministic samplers (minibatch.py:306-339, :403-426)
        self.nocache_modes = set(nocache_modes)
        self.record_subgraphs: Dict[int, str] = {}
        self.cache_subg: Dict[int, SubgraphCache] = {}
        self._roots_dev: Dict[int, torch.Tensor] = {}
        self._mine_pos: Dict[int, np.ndarray] = {}            # this rank's positions into the raw entity set, epoch order
        self._recorded: Dict[int, np.ndarray] = {}            # per position of the raw entity set: subgraph is in the cache
        self._cursor = {m: 0 for m in _MODES}                 # roots handed to the sampler so far this epoch
        self._step = {m: 0 for m in _MODES}                   # steps returned by one_batch this epoch
        self._launched = {m: 0 for m in _MODES}               # steps whose sampler call has been issued
        self._local_sizes = {m: np.zeros(0, dtype=np.int64) for m in _MODES}
        self._global_sizes = {m: np.zeros(0, dtype=np.int64) for m in _MODES}
        self._hwm = {m: [1, 1] for m in _MODES}               # largest batch seen (nodes, edges)
        self.wait_s = 0.0                                      # host seconds blocked on sampler read-backs (see _collect)

    @classmethod
    def on_device(cls, adjs, entity_set, sampler_config: Dict[str, Any], aug_feats, feat_full: torch.Tensor,
                  label_full: torch.Tensor, batch_size: int, device, seed_cpp: int = -1, rank: int = 0,
                  world_size: int = 1, prefetch: bool = True, nocache_modes=(), percent_per_epoch=None):
        """Short form: ``sampler_config`` is one flat sampler entry, e.g.
        {"method": "khop", "depth": 2, "budget": 20, "add_self_edge": False}; ``batch_size`` is the global batch."""
        section = {"batch_size": int(batch_size), "configs": [dict(sampler_config)]}
        return cls(None, None, adjs, entity_set, section, aug_feats, percent_per_epoch, feat_full, label_full,
                   int(feat_full.shape[1]), True, nocache_modes=set(nocache_modes), seed_cpp=seed_cpp, device=device,
                   rank=rank, world_size=world_size, prefetch=prefetch)

    # ------------------------------------------------------------------ API
    def attach_model(self, model):
        """Let the extractor prepare what ``model``'s training step will ask of every TRAIN batch: the row sets of the row-sparse
        top-layer backward pass (node task, residue none + centre pooling: GraphSAGE stacks take a tail.TopBackwardPlan, GAT
        stacks two nested levels), built on the prefetch stream.  Without it the model builds them inside its forward pass
        (two host syncs on the training stream per step).  Returns self."""
        from . import layers as _layers
        self._attached_model = weakref.ref(model)
        ok = bool(ops.SPARSE_TOP_BWD and getattr(model, "prediction_task", None) == "node" and len(model.conv_layers) == 1
                  and model._tail_prunable(0) and not getattr(model, "prune_tail", False))
        convs = list(model.conv_layers[0]) if ok else []
        sage = ok and all(isinstance(md, _layers.GraphSAGE) for md in convs)
        gat = ok and all(isinstance(md, _layers.GAT) for md in convs)
        self.top_backward_plan = bool(sage or gat)
        self.backward_levels = 2 if gat else 0
        # the plan's filtered transposed structure has one reader: the A^T dZn of the layer below the row-sparse pass at
        # hidden widths in (128, 256] (ops._at_dzn_on_rows)
        self.top_plan_filter = bool(sage and any(128 < int(md.dim_out) <= 256 for md in convs))
        return self

    def get_aug_dim(self, aug_type):
        return getattr(self, f'dim_1hot_{aug_type[:-1]}')

    def _static_partition(self, mode) -> bool:
        """Per-rank subgraph caches need a fixed root -> rank map across epochs."""
        return self.world_size > 1 and self.record_subgraphs.get(mode) in ("record", "reuse")

    def epoch_start_reset(self, epoch, mode):
        self.batch_num = -1
        if mode in self.graph_sampler:
            return
        files = (self._bin_adj_files or {}).get(mode) if isinstance(self._bin_adj_files, dict) else None
        if files and files.get("indptr") and files.get("indices"):
            # the reference's cpp/adj_*_{indptr,indices}.bin (loader.py:63-96): read straight into HBM
            hs = HipSampler(device=self.device, seed=self._seed, path_indptr=files["indptr"], path_indices=files["indices"])
        else:
            ip, ix = _as_uint32_csr(self._adjs[mode])
            hs = HipSampler(ip, ix, device=self.device, seed=self._seed)
        self.graph_sampler[mode] = hs
        if mode in self.nocache_modes:
            self.record_subgraphs[mode] = "noncache"
        elif self.sampler_cfg.method in REUSABLE_SAMPLER:
            self.record_subgraphs[mode] = "record"
            self.cache_subg[mode] = SubgraphCache(hs.num_nodes(), self.device)
        else:
            self.record_subgraphs[mode] = "none"

    def prepare_ppr(self, mode, order: str = "ordered"):
        """PPRSamplingCpp.preproc (frontend/samplers_cpp.py:166-186): the top-k table of the mode's entity set,
        read from the reference's cache files when ``dir_data`` names them, computed on the GPU otherwise
        (and written back in the same format)."""
        import glob
        import os
        from .ppr import ppr_approximate_device
        hs = self.graph_sampler[mode]
        k, a, eps = self.sampler_cfg.k, self._ppr_args["alpha"], self._ppr_args["epsilon"]
        f_nb = f_sc = None
        if self.dir_data and self.name_data and not self.dir_data.get("is_adj_changed", False):
            folder = f"{self.dir_data['local']}/{self.name_data}/ppr_float"
            tag = f"{'transductive' if self.is_transductive else 'inductive'}_{MODE2STR[mode]}_{a}_{eps}"
            os.makedirs(folder, exist_ok=True)
            f_nb, f_sc = f"{folder}/neighs_{tag}_{k}.bin", f"{folder}/scores_{tag}_{k}.bin"
            for cand in sorted(glob.glob(f"{folder}/neighs_{tag}_*.bin")):         # any stored k' >= k serves
                k_meta = int(cand.rsplit("_", 1)[1][:-4])
                sc = f"{folder}/scores_{tag}_{k_meta}.bin"
                if k_meta >= k and os.path.isfile(sc):
                    hs.load_ppr_bin(cand, sc, k, a, eps)
                    return
        targets = np.unique(self.raw_entity_set[mode]).astype(np.uint32)
        ln, nb, sc = ppr_approximate_device(hs, targets, k, a, eps, order=order)
        hs.set_ppr(targets, ln, nb, sc)
        if f_nb is not None:
            hs.save_ppr_bin(f_nb, f_sc, k, a, eps)

    def _drain(self, mode):
        """Finish and discard a prefetched sampler call (its outputs are dropped); returns its root count."""
        bs = sum(b.num_subgraphs for b in self._ready[mode])           # batches of a multi-step call not consumed yet
        self._ready[mode] = []
        if mode not in self._inflight:
            return bs
        n = self._inflight[mode][1]
        bs += sum(n) if isinstance(n, (list, tuple)) else n
        self._collect(mode, discard=True)
        return bs

    def shuffle_entity(self, mode, perm=None):
        """YOU MUST CALL THIS BEFORE STARTING ANY EPOCH (minibatch.py:269-280).  Every rank works from the
        same permutation (rank 0's draw is broadcast when none is given) and keeps its own share of each
        global batch (``plan_epoch``)."""
        self._drain(mode)
        raw = self.raw_entity_set[mode]
        if perm is None:
            perm = np.random.permutation(raw.size)
            from . import dist as sdist
            if self.world_size > 1 or sdist.collectives_on():       # (one rank under SHADOW_DIST_FORCE_INIT=1: the RCCL smoke path)
                perm = sdist.broadcast_array(perm.astype(np.int64), src=0, device=self.device)
        perm = np.asarray(perm).reshape(-1)
        if self.percent_per_epoch[mode] < 1.0:
            perm = perm[:int(np.ceil(self.percent_per_epoch[mode] * perm.size))]
        mine_pos, local, glob = plan_epoch(perm, self.batch_size_global, self.world_size, self.rank,
                                           static_partition=self._static_partition(mode))
        mine = raw[mine_pos]
        self._mine_pos[mode] = mine_pos
        self._local_sizes[mode], self._global_sizes[mode] = local, glob
        self.entity_epoch[mode] = mine
        self.label_epoch[mode] = self.label_full[torch.as_tensor(mine.astype(np.int64), device=self.device)]
        if mine.size:
            self.graph_sampler[mode].shuffle_targets(mine.astype(np.uint32))
            self._set_root_cursor(mode, 0)               # (an epoch abandoned half-way leaves the cursor mid-list)
        self._roots_dev[mode] = torch.as_tensor(mine.astype(np.uint32).view(np.int32)).to(self.device)
        self._cursor[mode] = self._step[mode] = self._launched[mode] = 0
        self.idx_entity_evaluated[mode] = 0
        self.end_epoch[mode] = False

    def _set_root_cursor(self, mode, pos: int):
        """Put the sampler's sequential root cursor (ParallelSampler::_get_roots_p, .cpp:456-468) at ``pos`` of the
        installed target list: hand out (and drop) root ranges until it stands there.  Only the RNG serial moves on."""
        hs = self.graph_sampler[mode]
        total, cur = hs.num_nodes_target(), hs.get_idx_root()
        if cur > pos:                                    # run to the end of the list: the cursor wraps to 0
            hs.next_roots(1, total - cur)
            cur = 0
        if pos > cur:
            hs.next_roots(1, pos - cur)
        assert hs.get_idx_root() == pos % max(total, 1)

    def is_end_epoch(self, mode):
        return self.end_epoch[mode]

    def num_steps(self, mode):
        """Steps per epoch -- the same number on every rank."""
        return int(self._global_sizes[mode].size)

    def epoch_end_reset(self, mode, drop_full_graph: bool = False):
        """After the first full epoch of a deterministic sampler the recorded subgraphs are
        reused (minibatch.py:326-334); ``drop_full_graph`` is the reference's optm_level 'high'
        (:335-341): the full CSR is freed once nothing samples from it any more.  The switch needs the
        cache to hold this rank's whole share of the entity set (an epoch over a ``percent_per_epoch``
        sub-sample keeps recording)."""
        self.end_epoch[mode] = False
        if self.record_subgraphs.get(mode) == "record":
            done = self._recorded.get(mode)
            share = np.arange(self.raw_entity_set[mode].size) % self.world_size == self.rank
            if done is not None and share.any() and bool(done[share].all()):
                self.record_subgraphs[mode] = "reuse"
                if drop_full_graph:
                    self.drop_full_graph_info(mode)

    def drop_full_graph_info(self, mode):
        self.graph_sampler[mode].drop_full_graph_info()

    def disable_cache(self, mode):
        """minibatch.py:490-492: stop recording / reusing subgraphs of this mode.  A prefetched call is
        finished and dropped, and the sampler's root cursor is put back where the epoch stands."""
        self.nocache_modes.add(mode)
        if mode not in self.record_subgraphs:
            return
        dropped = self._drain(mode)
        if dropped or self.record_subgraphs[mode] == "reuse":
            self._cursor[mode] -= dropped
            self._launched[mode] = self._step[mode]
            mine = self.entity_epoch[mode]
            if mine is not None and mine.size:
                self.graph_sampler[mode].shuffle_targets(mine.astype(np.uint32))
                self._set_root_cursor(mode, self._cursor[mode])    # cursor := roots already consumed this epoch
        self.record_subgraphs[mode] = "noncache"

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

    def _top_plan(self, subgs, adj):
        """tail.TopBackwardPlan of this batch, on the prefetch stream (its two host syncs wait for the side stream only);
        stream bookkeeping as in _tail_plan."""
        from . import tail
        if self._side is None:
            return (tail.build_backward_levels(adj, subgs.target, max_levels=self.backward_levels) if self.backward_levels > 0
                    else tail.TopBackwardPlan(adj, subgs.target, want_filter=self.top_plan_filter))
        main = torch.cuda.current_stream(self.device)
        # (as in _tail_plan: the plan allocates on the side stream BEFORE the next _launch orders that stream behind the
        #  training stream -- every sampler output of this batch must therefore be recorded on the training stream, or a block
        #  the training stream still reads could be handed to the next batch's plan)
        for t in (subgs.node, subgs.indptr, subgs.indices, subgs.edge_id, subgs.target, subgs.subg_node_off,
                  subgs.subg_edge_off, subgs.ppr, subgs.hop, subgs.drnl):
            if t is not None and t.is_cuda:
                t.record_stream(main)
        w0 = tail._SYNC_WAIT[0]
        with torch.cuda.stream(self._side):
            # (a collated batch's roots ascend: one per subgraph, the subgraphs in order)
            plan = (tail.build_backward_levels(adj, subgs.target, max_levels=self.backward_levels, targets_ascending=int(self.sampler_cfg.num_roots) == 1)
                    if self.backward_levels > 0 else tail.TopBackwardPlan(adj, subgs.target, want_filter=self.top_plan_filter))
        self.wait_s += tail._SYNC_WAIT[0] - w0                      # (the levels' size read-backs: blocked on the prefetch stream, as _collect)
        main.wait_stream(self._side)
        for t in ([x for lv in plan for x in lv.tensors()] if isinstance(plan, list) else plan.tensors()):
            t.record_stream(main)
        if isinstance(plan, list):                                # (the batch CSR's transpose, built with the levels)
            for t in (*(adj._t or ()), adj._edge_row):
                if t is not None:
                    t.record_stream(main)
        self.wait_s += getattr(plan, "sync_wait_s", 0.0)          # (the plan's size read-back: blocked on the prefetch stream, as _collect)
        return plan

    def _launch(self, mode):
        """Issue the sampler call of the next un-launched step (nothing to issue for an empty share)."""
        t = self._launched[mode]
        bs = int(self._local_sizes[mode][t])
        if bs == 0:
            self._launched[mode] = t + 1
            return
        hs = self.graph_sampler[mode]
        reuse = self.record_subgraphs.get(mode) == "reuse"
        c0 = self._cursor[mode]
        # several steps from one call: the following steps' shares, up to the first empty one (ragged epoch tails)
        sizes = [bs]
        if not reuse and self.steps_per_call > 1:
            for x in self._local_sizes[mode][t + 1:t + self.steps_per_call]:
                if int(x) == 0:
                    break
                sizes.append(int(x))
        multi = len(sizes) > 1
        self._launched[mode] = t + len(sizes)
        bs = sum(sizes)

        def go():
            if reuse:
                hn, he = self._hwm[mode]
                self.cache_subg[mode].collate_async(self._roots_dev[mode][c0:c0 + bs], hn + hn // 8 + 64, he + he // 8 + 64,
                                                    want_hop="hops" in self.aug_feats)
            elif multi:
                hs.sample_multi_async(self.sampler_cfg, sizes)
            else:
                hs.sample_async(self.sampler_cfg, bs)
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                go()
        else:
            go()
        self._cursor[mode] = c0 + bs
        self._inflight[mode] = ("reuse" if reuse else ("multi" if multi else "sample"), sizes if multi else bs, c0)

    def _collect(self, mode, discard: bool = False) -> Optional[DeviceBatch]:
        hs = self.graph_sampler[mode]
        kind, bs, c0 = self._inflight[mode]
        main = torch.cuda.current_stream(self.device)

        def before_rerun():
            # a capacity re-run allocates new outputs on the prefetch stream: blocks freed by the previous
            # batch may still be read by the training stream, so order the re-run behind it
            if self._side is not None:
                self._side.wait_stream(main)

        def record(b, lo, hi):
            if not discard and self.record_subgraphs.get(mode) == "record":
                self.cache_subg[mode].record(b)                      # minibatch.py:407-412
                if mode not in self._recorded:
                    self._recorded[mode] = np.zeros(self.raw_entity_set[mode].size, dtype=bool)
                self._recorded[mode][self._mine_pos[mode][lo:hi]] = True

        def go():
            if kind == "reuse":
                return self.cache_subg[mode].finish(on_retry=before_rerun)
            if kind == "multi":          # the batches of several steps: the first one is returned, the others wait their turn
                bl = hs.finish_multi(on_retry=before_rerun)
                lo = c0
                for b_, n_ in zip(bl, bs):
                    record(b_, lo, lo + n_)
                    lo += n_
                return bl
            b = hs.finish(on_retry=before_rerun)
            record(b, c0, c0 + bs)
            return b
        import time as _time
        t_wait = _time.perf_counter()
        try:
            if self._side is not None:
                with torch.cuda.stream(self._side):
                    b = go()
                main.wait_stream(self._side)
            else:
                b = go()
        finally:
            self._inflight.pop(mode, None)
            # host time spent here is (almost all) BLOCKED on the sampler's count read-back, not busy: bench.py subtracts
            # it from the enqueue time of a step to report how much of the step the host is really working
            self.wait_s += _time.perf_counter() - t_wait
        if discard:
            return None
        if kind == "multi":
            for b_ in b:
                self._hwm[mode] = [max(self._hwm[mode][0], b_.num_nodes), max(self._hwm[mode][1], b_.num_edges)]
            self._ready[mode] = list(b[1:])
            return b[0]
        self._hwm[mode] = [max(self._hwm[mode][0], b.num_nodes), max(self._hwm[mode][1], b.num_edges)]
        return b

    def _empty_batch(self, mode) -> OneBatchSubgraph:
        """This rank has no root in the step's global batch: an empty record the model still steps on
        (zero gradient into the all-reduce)."""
        dev = self.device
        z32 = torch.zeros(0, dtype=torch.int32, device=dev)
        ret = OneBatchSubgraph([None], [self.feat_full[:0]], self.label_full[:0],
                               torch.zeros(1, 0, dtype=torch.int32, device=dev), [z32], [{}], loss_weight=0.0)
        ret.device_batch = None
        return ret

    def one_batch(self, mode=TRAIN, ret_raw_idx=False) -> OneBatchSubgraph:
        t = self._step[mode]
        assert t < self._local_sizes[mode].size, "epoch exhausted: call shuffle_entity before the next epoch"
        batch_size_ = int(self._local_sizes[mode][t])
        if batch_size_ > 0 and self._ready[mode]:
            subgs = self._ready[mode].pop(0)               # sampled by an earlier step's multi-step call
        else:
            if self._launched[mode] <= t:
                self._launch(mode)
            subgs = self._collect(mode) if batch_size_ > 0 else None
        assert subgs is None or subgs.num_subgraphs == batch_size_, (subgs.num_subgraphs, batch_size_)
        i0 = self.idx_entity_evaluated[mode]
        self.idx_entity_evaluated[mode] += batch_size_
        self._step[mode] = t + 1
        self.batch_num += 1
        last = self._step[mode] >= self._local_sizes[mode].size
        if last:
            assert self.idx_entity_evaluated[mode] == self.entity_epoch[mode].shape[0]
            self.idx_entity_evaluated[mode] = 0
            self.end_epoch[mode] = True
            if self.record_subgraphs.get(mode) != "reuse" and self.entity_epoch[mode].size:
                assert self.graph_sampler[mode].get_idx_root() == 0      # samplers_ensemble.py:298-301
        weight = batch_size_ / float(self._global_sizes[mode][t])
        if subgs is None:
            if not last and self.prefetch and self._launched[mode] == t + 1:
                self._launch(mode)
            return self._empty_batch(mode)
        # (a top-k PPR subgraph holds k nodes, its root's row up to k - 1 of them: the bound that sends the 256-float aggregations
        #  of such batches to the pipelined CSR kernel, ops.DeviceCSR; the rows of a node-induced k-hop subgraph have no such bound)
        cfg = self.sampler_cfg
        bound = int(cfg.k) if (cfg.method == "ppr" and int(cfg.num_roots) == 1) else 0
        adj = ops.DeviceCSR(subgs.indptr, subgs.indices, subg_off=subgs.subg_node_off,
                            subg_edge_off=subgs.subg_edge_off, max_subg_nodes=subgs.counts["max_subg_nodes"], row_entries_bound=bound)
        tail_plan = self._tail_plan(subgs, adj, subgs.target) if self.tail_plan_layers > 0 else None
        if (self.top_backward_plan and mode == TRAIN and self.tail_plan_layers == 0 and adj.n >= ops.SPARSE_TOP_BWD_MIN_ROWS
                and ops.SPARSE_TOP_BWD):                  # (evaluation batches: no backward pass will ask for the row sets)
            plan = self._top_plan(subgs, adj)
            if isinstance(plan, list):
                subgs.target._shd_bwd_levels = plan
            else:
                subgs.target._shd_top_plan = plan
        if not last and self.prefetch:
            if self.defer_prefetch:
                # ... issued when the consumer reaches ops.fire_deferred, so that it stays off the HBM-bound head of the step
                # (feature gather, layer-0 aggregation); if the consumer never gets there, the next one_batch launches it
                t1 = t + 1
                ops.defer((id(self), mode),
                          lambda: self._launch(mode) if (self._launched[mode] == t1 and self._step[mode] == t1) else None)
            elif self._launched[mode] == t + 1:
                self._launch(mode)        # overlap the next sampler call with this batch's training
        feat = (ops.LazyRows(self.feat_full, subgs.node) if self.lazy_features
                else ops.gather_rows(self.feat_full, subgs.node))    # minibatch.py:469
        label = self.label_epoch[mode][i0:i0 + batch_size_]
        feat_aug = {}
        # entity encodings (frontend/graph.py:134-172) as per-node bit masks; the model's augmentation
        # Linear consumes them fused (ops.onehot_linear_add), .dense() gives the reference's matrix
        for name, src, dim in (("hops", subgs.hop, self.dim_1hot_hop), ("pprs", subgs.ppr, self.dim_1hot_ppr),
                               ("drnls", subgs.drnl, self.dim_1hot_drnl)):
            if name in self.aug_feats:
                feat_aug[name] = ops.OneHotCodes(ops.encode_codes(name, src, dim), dim)
        size_subg = subgs.size_subg.unsqueeze(0)
        ret = OneBatchSubgraph([adj], [feat], label, size_subg, [subgs.target], [feat_aug], loss_weight=weight)
        if tail_plan is not None:
            ret.tail_ens = [tail_plan]
        ret.device_batch = subgs
        ret.extractor = weakref.ref(self)       # (DeepGNN.step attaches itself on its first training batch: attach_model)
        if ret_raw_idx:
            ret.idx_raw = [subgs.node]
        return ret
