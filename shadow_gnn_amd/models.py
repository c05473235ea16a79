"""DeepGNN with the reference's API surface (shaDow/models.py:16-237): same
constructor arguments, layer registry keys, parameter names (the checkpoint
contract), ``forward`` signature and ``step`` return dict -- running on the
HIP layers of ``shadow_gnn_amd.layers``.

Data parallel use: construct with ``grad_sync=dist.GradSync(...)``; every
``step`` then scales its loss by the batch's share of the global batch
(``OneBatchSubgraph.loss_weight``), sums the flat fp32 gradient buffer over the
ranks (RCCL, overlapped with backward) and runs clip + Adam on identical values
(DESIGN.md section "multi-GPU")."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import layers, ops, tail
from .minibatch import TRAIN, OneBatchSubgraph

_FORWARD_FIELDS = ("feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens")
GRAD_CLIP_NORM = 5.0                       # shaDow/models.py:225


@dataclass(frozen=True)
class _Arch:
    """The architecture section of a training configuration (config_train/**.yml, 'architecture'), parsed once."""
    aggr: str
    num_gnn_layers: int
    num_cls_layers: int
    dim_hid: int
    heads: int
    act: str
    layer_norm: str
    aug_ops: str
    residue: str
    pooling: str
    sigmoid_loss: bool
    branch_sharing: bool

    @classmethod
    def parse(cls, cfg: Dict[str, Any]) -> "_Arch":
        return cls(aggr=cfg["aggr"], num_gnn_layers=int(cfg["num_layers"]), num_cls_layers=int(cfg.get("num_cls_layers", 1)),
                   dim_hid=int(cfg["dim"]), heads=int(cfg.get("heads", 1)), act=cfg["act"],
                   layer_norm=cfg.get("layer_norm", "norm_feat"), aug_ops=cfg.get("feature_augment_ops", "sum"),
                   residue=str(cfg.get("residue", "none")).lower(),
                   pooling=str(cfg.get("pooling", "center")).split("-")[0].lower(),
                   sigmoid_loss=cfg.get("loss", "softmax") == "sigmoid", branch_sharing=bool(cfg.get("branch_sharing", False)))


class DeepGNN(nn.Module):
    NAME2CLS = {
        "mlp": layers.MLP,
        "gcn": layers.GCN,
        "sage": layers.GraphSAGE,
        "gat": layers.GAT,
    }

    def __init__(self, dim_feat_raw: int, dim_feat_smooth: int, dim_label_raw: int, dim_label_smooth: int,
                 arch_gnn: Dict[str, Any], aug_feat, num_ensemble: int, train_params: Dict[str, Any],
                 prediction_task: str, grad_sync=None):
        super().__init__()
        if prediction_task not in ("link", "node"):
            raise ValueError(f"prediction_task {prediction_task!r}")
        if num_ensemble != 1:
            raise NotImplementedError("subgraph ensembles (EnsembleAggregator) are outside the hot path built here")
        if dim_feat_raw > dim_feat_smooth:
            raise ValueError("smoothed feature width below the raw width")
        spec = _Arch.parse(arch_gnn)
        if spec.aggr not in self.NAME2CLS:
            raise NotImplementedError(f"aggr {spec.aggr!r} not provided (have {sorted(self.NAME2CLS)})")
        self._spec = spec
        # public attributes of the reference's model object
        self.prediction_task, self.num_ensemble = prediction_task, num_ensemble
        self.num_gnn_layers, self.num_cls_layers = spec.num_gnn_layers, spec.num_cls_layers
        self.dim_hid, self.mulhead, self.branch_sharing = spec.dim_hid, spec.heads, spec.branch_sharing
        self.feat_aug_ops, self.sigmoid_loss = spec.aug_ops, spec.sigmoid_loss
        self.dropout, self.dropedge, self.lr = train_params["dropout"], train_params["dropedge"], train_params["lr"]
        self.type_feature_augment = aug_feat
        self.num_classes, self.dim_label_in, self.dim_feat_in = dim_label_raw, dim_label_smooth, dim_feat_smooth
        # modules; the attribute names below are the checkpoint's key prefixes
        branches = [self._build_branch(spec) for _ in range(num_ensemble)]
        aug = [b[0] for b in branches if b[0] is not None]
        self.aug_layers = nn.ModuleList(aug) if aug else []
        self.conv_layers = nn.ModuleList(b[1] for b in branches)
        self.res_pool_layers = nn.ModuleList(b[2] for b in branches)
        self.ensembler = layers.EnsembleDummy()
        self.classifier = self._build_classifier(spec)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=self.lr)
        self.fuse_dropout = True       # fold each layer's input dropout into the producing kernel where possible
        # Exact dead-row elimination for residue 'none' + centre pooling (tail.py): the last layers are computed
        # only on the rows the roots depend on.  Off by default: the reference computes every row.
        self.prune_tail = False
        self.grad_sync = grad_sync

    # ------------------------------------------------------------ construction
    def _build_branch(self, spec: _Arch):
        """(augmentation Linears | None, conv stack, read-out) of one subgraph branch."""
        aug, widen = None, 0
        if len(self.type_feature_augment) > 0:
            # 'sum' adds the encoding into the raw features; anything else appends dim_hid columns
            out = self.dim_feat_in if spec.aug_ops == "sum" else spec.dim_hid
            widen = 0 if spec.aug_ops == "sum" else out
            aug = nn.ModuleList(nn.Linear(width, out) for _name, width in self.type_feature_augment)
        widths = [self.dim_feat_in + self.dim_label_in + widen] + [spec.dim_hid] * (spec.num_gnn_layers - 1)
        layer_cls = self.NAME2CLS[spec.aggr]
        convs = nn.Sequential(*[layer_cls(w, spec.dim_hid, dropout=self.dropout, act=spec.act, norm=spec.layer_norm,
                                          mulhead=spec.heads) for w in widths])
        readout = layers.ResPool(spec.dim_hid, spec.dim_hid, spec.num_gnn_layers, spec.residue, spec.pooling,
                                 dropout=self.dropout, act=spec.act, args_pool={}, prediction_task=self.prediction_task)
        return aug, convs, readout

    def _build_classifier(self, spec: _Arch):
        """MLP head: hidden layers keep the model's activation / dropout, the last one maps to the classes with
        the identity; feature normalisation only for the node task."""
        norm = "norm_feat" if self.prediction_task == "node" else "none"
        hidden = [dict(dim_out=spec.dim_hid, act=spec.act, dropout=self.dropout)] * (spec.num_cls_layers - 1)
        heads = hidden + [dict(dim_out=self.num_classes, act="I", dropout=0.0)]
        return nn.Sequential(*[self.NAME2CLS["mlp"](dim_in=spec.dim_hid, norm=norm, **kw) for kw in heads])

    # ------------------------------------------------------------------ loss
    def _loss(self, preds, labels):
        """Mean CE over the roots on class indices (one-hot label rows are arg-maxed), or the reference's
        class-summed BCE for multi-label data (shaDow/models.py:151-166)."""
        if self.sigmoid_loss:
            if preds.shape != labels.shape:
                raise ValueError(f"sigmoid loss needs label rows of {preds.shape[1]} classes")
            return F.binary_cross_entropy_with_logits(preds, labels.to(preds.dtype)) * preds.shape[1]
        index = labels.argmax(dim=1) if labels.dim() == 2 else labels
        return F.cross_entropy(preds, index)

    # --------------------------------------------------------------- forward
    def _augment(self, i, feat, encodings):
        """Entity encodings -> Linear -> added to / appended to the features (shaDow/models.py:183-192)."""
        for ia, (kind, _width) in enumerate(self.type_feature_augment):
            enc, lin = encodings[kind], self.aug_layers[i][ia]
            whole = self.dim_feat_in == feat.shape[1]
            if isinstance(enc, ops.OneHotCodes):
                if self.feat_aug_ops == "sum" and enc.dim <= 16 and whole:
                    # fused one-hot + Linear + add: no [n, dim] matrix, one pass over the features
                    feat = ops.onehot_linear_add(feat, enc.codes, lin)
                    continue
                enc = enc.dense()
            emb = lin(enc)
            if self.feat_aug_ops != "sum":
                feat = torch.cat([feat, emb], dim=1)
            elif whole:
                feat = feat + emb
            else:           # smoothed-label columns ride behind the features and are not augmented
                feat = torch.cat([feat[:, :self.dim_feat_in] + emb, feat[:, self.dim_feat_in:]], dim=1)
        return feat

    def _run_branch(self, i, feat, adj, tgt, sizes, dropedge, levels):
        """Conv stack + read-out of branch i; ``levels``: target-only-tail plan (possibly empty)."""
        convs = list(self.conv_layers[i])
        num_full = len(convs) - len(levels)
        state = (feat, adj, False, dropedge)
        outs = []
        self._plan_dropout_fusion(i)
        if not levels and ops.SAGE_STACK and self._stack_plan.get(i):
            emb = self._run_stack(self._stack_plan[i], convs, feat, adj, tgt, dropedge)
            if emb is not None:
                return emb
        for md in convs[:num_full]:
            state = md(state, sizes_subg=sizes)
            outs.append(state[0])
            dropped = md.take_dropped_out() if hasattr(md, 'take_dropped_out') else None
            if dropped is not None:       # dual mode: the read-out keeps the plain output, the next layer
                state = (dropped,) + tuple(state[1:])     # gets the one its input dropout was applied to
        if not levels:
            return self.res_pool_layers[i](outs, tgt, sizes)
        # target-only tail: each remaining layer on the rows the roots depend on; the last one yields the
        # root rows in target order, which is all that residue 'none' + centre pooling reads
        x, adj_norm = ops.dense_rows(state[0]), state[1]
        if num_full == 0:
            first = convs[0]
            adj_norm = (first.norm_adj(adj, False, dropedge, x.device) if hasattr(first, 'norm_adj')
                        else first._adj_norm(adj, False, x.device, dropedge=dropedge))
        for md, level in zip(convs[num_full:], levels):
            x = md.forward_rows(x, adj_norm, level)
        return x

    def _run_stack(self, kind, convs, feat, adj, tgt, dropedge):
        """The whole GraphSAGE / GCN stack + the read-out's row select as one node (ops._SageStack / ops._GcnStack: one C call per
        direction; large GraphSAGE training batches: the two top layers' backward row-sparse from the node, the rest one C call),
        or None when this batch goes layer by layer (a layer 0 that gathers inside its aggregation kernel, frozen parameters)."""
        first = convs[0]
        n = int(feat.shape[0])
        if not feat.is_cuda:
            return None
        # the table of ops.step_path decides (this branch's stack passed the static preconditions: `kind` is set)
        F = first.f_lin_self.weight.shape[0] if kind == 'sage' else first.f_lin.weight.shape[0]
        blocks = getattr(adj, "spmm_blocks", None)
        path = ops.step_path(kind, n, F, len(convs), bool(self.training and torch.is_grad_enabled()), "center", stackable=True,
                             blockdiag=blocks is not None and blocks[0] is not None)
        if path.forward != "stack":
            return None                                      # (kernel by kernel, or the layer-by-layer nodes' row-sparse top pass)
        sparse_top = path.backward == "stack+sparse-top" and ops.sparse_top_stack_usable(adj, convs)
        lazy = isinstance(feat, ops.LazyRows)
        if lazy and ops.FUSE_GATHER_INTO_SPMM:
            return None
        plist = self.__dict__.get('_stack_params')
        if plist is None or plist[0] is not convs[0]:
            plist = self.__dict__['_stack_params'] = (convs[0], [p for md in convs for p in md.parameters()])
        if torch.is_grad_enabled() and not all(p.requires_grad for p in plist[1]):
            return None
        # (from here on the step's random draws are taken in the layer-by-layer order: drop-edge, input dropout, output dropouts)
        adj_norm = first.norm_adj(adj, False, dropedge, feat.device)
        if lazy:
            x0, _seed = feat.gather_dropped(first._in_p())
        else:
            x0 = first.in_dropout(feat)
        if kind == 'sage':
            plan = None
            if sparse_top:
                # (training batches large enough for the row-sparse backward of the two top layers: the row sets the minibatch
                #  extractor built on its prefetch stream, or -- hand-made batches -- built here with two host syncs)
                plan = getattr(tgt, "_shd_top_plan", None)
                if plan is None or not plan.matches(adj_norm.csr, int(tgt.numel())):
                    from . import tail
                    plan = tail.TopBackwardPlan(adj_norm.csr, tgt)
                if not plan.matches(adj_norm.csr, int(tgt.numel())):
                    plan = None                              # (a multigraph root row: the dense pass)
            emb = ops.sage_stack(x0, adj_norm, convs, tgt, plan)
        else:
            emb = ops.gcn_stack(x0, adj_norm, convs, tgt)
        assert emb is not None
        return emb

    def forward(self, mode, feat_ens, adj_ens, target_ens, size_subg_ens, feat_aug_ens, dropedge, tail_ens=None):
        return self._head(self._embed(mode, feat_ens, adj_ens, target_ens, size_subg_ens, feat_aug_ens, dropedge, tail_ens))

    def _head(self, embs):
        emb_subg_ens = [F.normalize(emb, p=2, dim=1) for emb in embs]          # shaDow/models.py:200
        pred_subg = self.classifier(self.ensembler(emb_subg_ens))
        ops.fire_deferred("fwd")                     # (the extractor's deferred prefetch: see ops.defer)
        return pred_subg, emb_subg_ens

    def _embed(self, mode, feat_ens, adj_ens, target_ens, size_subg_ens, feat_aug_ens, dropedge, tail_ens=None):
        """The branches' read-out rows (before the L2 normalisation): augmentation, conv stack, ResPool per branch."""
        embs = []
        for i, feat in enumerate(feat_ens):
            tgt = torch.as_tensor(target_ens[i], device=feat.device).long()
            if getattr(target_ens[i], "_shd_top_plan", None) is not None:      # (minibatch: row sets of the row-sparse top-layer backward)
                tgt._shd_top_plan = target_ens[i]._shd_top_plan
            if getattr(target_ens[i], "_shd_bwd_levels", None) is not None:    # (... of a GAT stack's row-sparse backward, tail.build_backward_levels)
                tgt._shd_bwd_levels = target_ens[i]._shd_bwd_levels
            if self.dim_label_in > 0 and mode == TRAIN:
                feat = ops.dense_rows(feat)
                feat[tgt, -self.dim_label_in:] = 0            # a root never sees its own label (models.py:181-182)
            if len(self.type_feature_augment) > 0:
                feat = self._augment(i, ops.dense_rows(feat), feat_aug_ens[i])
            adj_i, levels = adj_ens[i], []
            if self.prune_tail and self._tail_prunable(i):
                adj_i = layers._as_device_csr(adj_i, feat.device)
                planned = tail_ens[i] if tail_ens is not None else None      # built by the minibatch while prefetching
                levels = planned if planned is not None else tail.build_tail_plan(adj_i, tgt, len(self.conv_layers[i]))
                assert len(levels) <= len(self.conv_layers[i])
            emb = self._run_branch(i, feat, adj_i, tgt, size_subg_ens[i], dropedge, levels)
            if i + 1 == len(feat_ens):
                ops.fire_deferred("body")                # (the big kernels of the forward pass are enqueued)
            embs.append(emb)
        return embs

    def _fused_head(self, embs, index):
        """(loss, softmax(preds), [normalised embeddings]) from ops.node_head (csrc/head.hip: one kernel forward, two backward) when
        the head is the default one -- a single branch, the one-layer classifier with feature normalisation, softmax loss on class
        indices -- else None."""
        if not ops.FUSED_HEAD or self.sigmoid_loss or len(embs) != 1 or len(self.classifier) != 1 or index is None:
            return None
        cls = self.classifier[0]
        if not (type(cls) is layers.MLP and isinstance(self.ensembler, layers.EnsembleDummy) and cls.norm == 'norm_feat' and cls.act is None
                and cls.act_name == 'I' and (cls.dropout <= 0 or not self.training)
                and ops.node_head_usable(embs[0], cls.f_lin, cls.scale, cls.offset, index)):
            return None
        loss, _preds, prob, xn = ops.node_head(embs[0], cls.f_lin, cls.scale, cls.offset, index)
        ops.fire_deferred("fwd")
        return loss, prob, [xn]

    def _tail_prunable(self, i):
        rp = self.res_pool_layers[i]
        return (rp.type_res == 'none' and rp.type_pool == 'center' and self.prediction_task == 'node'
                and all(isinstance(md, (layers.GCN, layers.GraphSAGE, layers.GAT)) for md in self.conv_layers[i]))

    def _plan_dropout_fusion(self, i):
        """Layer l+1's input dropout (shaDow/layers.py:430,471,601) is applied by layer l's own act_norm
        kernel.  Residue 'none' with centre pooling only consumes the LAST layer's output (layers.py:159-163),
        so the kernel writes the dropped tensor alone; every other read-out (residue 'concat' / 'max', the
        mean / max / sort poolings) also reads the plain output of every layer, and the kernel then writes both
        from the same pass (dual mode).  Same distribution, no [n, F] mask tensor, no separate dropout pass;
        evaluation keeps nn.Dropout (identity)."""
        layers_i = list(self.conv_layers[i])
        rp = self.res_pool_layers[i]
        fuse_ok = self.training and self.fuse_dropout
        # the read-out takes the roots' rows of the LAST layer only (row-sparse gradient hand-overs apply) ...
        center_only = rp.type_res == 'none' and rp.type_pool == 'center'
        # ... or also reads the plain output of every lower layer (residue concat / max / ...): those layers then have two readers.
        # (Residue 'none' with a pooled read-out reads every row of the last layer but nothing of the layers below: they stay
        #  single-output and chain like under centre pooling -- round 6; rounds 1 - 5 ran them dual with an unread plain copy.)
        dual = rp.type_res != 'none'
        # (the plan only depends on these: nn.Module attribute writes cost ~2.5 us each, 25 of them per step otherwise)
        # (layer identities and the task are part of it: a swapped conv layer or a changed read-out re-plans)
        key = (fuse_ok, dual, center_only, ops.CHAIN_DUAL, rp.type_pool, self.prediction_task, tuple((id(md), float(getattr(md, 'dropout', 0.0))) for md in layers_i))
        plans = self.__dict__.setdefault('_fusion_plan_keys', {})
        if plans.get(i) == key:
            return
        plans[i] = key
        for l, md in enumerate(layers_i):
            if not hasattr(md, 'out_dropout'):
                continue
            nxt = layers_i[l + 1] if l + 1 < len(layers_i) else None
            fuse = (fuse_ok and nxt is not None and hasattr(nxt, 'input_pre_dropped') and nxt.dropout > 0
                    and md.can_fuse_out_dropout())
            md.out_dropout = nxt.dropout if fuse else 0.0
            md.out_dual = bool(fuse and dual)
            # nothing but the next layer reads this output (the read-out takes the last layer only): consecutive GraphSAGE
            # nodes may chain their backward passes (ops.ChainLink)
            # (dual-output layers chain too -- ops.CHAIN_DUAL -- when the read-out's pooling node hands the plain gradient over)
            md.chain_next = bool((not dual or (ops.CHAIN_DUAL and fuse and rp.type_pool in rp.POOLED)) and isinstance(md, layers.GraphSAGE)
                                 and isinstance(nxt, layers.GraphSAGE))
            # ... and the LAST layer's output only by the read-out's row select: its gradient travels as (rows, values)
            # (node tasks: one root per subgraph, so the selected rows are distinct)
            # ... or only by a pooled read-out (ops.pool_and_roots: pooled rows + root rows from one node): its gradient may arrive as a
            # table of one row per subgraph and root (ops.POOL_GRAD_TABLE)
            md.pool_only = bool(nxt is None and rp.type_pool in rp.POOLED and getattr(rp, 'dim_in', 1) != 0 and isinstance(md, layers.GraphSAGE))
            md.roots_only = bool(center_only and nxt is None and isinstance(md, (layers.GraphSAGE, layers.GAT))
                                 and self.prediction_task == 'node')
            # GAT below GAT, nothing else reads this output and the next layer's input dropout is fused into it (or absent): the
            # next layer may hand its input gradient down as (rows, values) -- the row-sparse top pass of ops_gat._GatTail
            md.rows_next = bool(center_only and isinstance(md, layers.GAT) and isinstance(nxt, layers.GAT)
                                and (fuse or float(getattr(nxt, 'dropout', 0.0)) == 0.0 or not self.training))
            if nxt is not None and hasattr(nxt, 'input_pre_dropped'):
                nxt.input_pre_dropped = bool(fuse)
        if layers_i and hasattr(layers_i[0], 'input_pre_dropped'):
            layers_i[0].input_pre_dropped = False
        # the whole stack as one autograd node (ops._SageStack): GraphSAGE layers only, nothing but the next layer / the row select
        # reads a layer's output, and every inner input dropout is applied by the producing layer's kernel (or is the identity)
        kind = ''
        if (center_only and self.prediction_task == 'node' and rp.dim_in == 0 and layers_i
                and all((not self.training) or md.dropout <= 0 or md.input_pre_dropped for md in layers_i[1:])):
            if all(type(md) is layers.GraphSAGE for md in layers_i) and ops.sage_stack_usable(layers_i):
                kind = 'sage'
            elif all(type(md) is layers.GCN for md in layers_i) and ops.gcn_stack_usable(layers_i):
                kind = 'gcn'
        self._stack_plan[i] = kind

    @property
    def _stack_plan(self):
        return self.__dict__.setdefault('_stack_plan_d', {})

    def invalidate_fusion_plan(self):
        """Forget the cached dropout / chaining plan (call after editing planned layer attributes by hand)."""
        self.__dict__.pop('_fusion_plan_keys', None)
        self.__dict__.pop('_stack_plan_d', None)
        self.__dict__.pop('_stack_params', None)

    def train(self, mode: bool = True):
        self.invalidate_fusion_plan()
        return super().train(mode)

    def predict(self, preds):
        return torch.sigmoid(preds) if self.sigmoid_loss else F.softmax(preds, dim=1)

    # ------------------------------------------------------------------ step
    def _begin_update(self):
        if not self.training:                     # (nn.Module.train() walks the whole module tree: only on a mode change)
            self.train()
        if self.grad_sync is not None:
            self.grad_sync.zero()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _finish_update(self):
        """Gradient exchange (data parallel), clip by global norm, Adam -- on every rank, also one whose share of
        the global batch was empty: the parameters stay identical everywhere."""
        if self.grad_sync is not None:
            self.grad_sync.all_reduce(self.parameters())
        if hasattr(self.optimizer, "clip_step_"):        # optim.FlatAdam: clip + Adam on the flat buffers, two launches
            self.optimizer.clip_step_(GRAD_CLIP_NORM)
            return
        torch.nn.utils.clip_grad_norm_(self.parameters(), GRAD_CLIP_NORM)
        self.optimizer.step()

    def _empty_result(self, batch_data):
        dev = self.classifier[0].f_lin.weight.device
        preds = torch.zeros(0, self.num_classes, device=dev)
        return {'batch_size': 0, 'loss': torch.zeros((), device=dev), 'labels': preds.long(), 'preds': preds,
                'emb_ens': [torch.zeros(0, self.dim_hid, device=dev)]}

    def step(self, mode, status, batch_data: OneBatchSubgraph, loss_scale: float = 1.0):
        """One training step (mode TRAIN, status 'running': forward, loss, backward, [all-reduce,] clip, Adam)
        or one evaluation pass; returns the reference's result dict (shaDow/models.py:209-237)."""
        if status not in ('running', 'final'):
            raise ValueError(f"status {status!r}")
        training = mode == TRAIN and status == 'running'
        if training:
            # a batch of MinibatchShallowExtractor knows where it came from: from the next batch on the extractor prepares what this
            # model's step asks of every TRAIN batch on its prefetch stream (the benchmarked path is the library default)
            ex = getattr(batch_data, "extractor", None)
            ex = ex() if ex is not None else None
            if ex is not None:
                at = getattr(ex, "_attached_model", None)
                if at is None or at() is not self:
                    ex.attach_model(self)
        if batch_data.batch_size == 0:          # a rank without roots in this global batch (minibatch.plan_epoch)
            if training:
                self._begin_update()
                self._finish_update()
            return self._empty_result(batch_data)
        fwd = {k: getattr(batch_data, k) for k in _FORWARD_FIELDS}
        fwd["feat_ens"] = list(fwd["feat_ens"])          # the step consumes the record (features are augmented)
        fwd["tail_ens"] = getattr(batch_data, "tail_ens", None)
        labels = batch_data.label
        # class indices as they come for the softmax loss (no one-hot -> argmax round trip); the returned record carries
        # one-hot rows like the reference's label matrix
        index = labels.to(torch.int64) if (labels.dim() == 1 and not self.sigmoid_loss) else None
        if labels.dim() == 1 and self.num_classes > 1:
            labels = F.one_hot(labels.to(torch.int64), num_classes=self.num_classes)
        probs = None
        if training:
            self._begin_update()
            embs = self._embed(mode, dropedge=self.dropedge, **fwd)
            head = self._fused_head(embs, index)
            if head is not None:
                loss, probs, emb_ens = head
            else:
                preds, emb_ens = self._head(embs)
                loss = self._loss(preds, labels if index is None else index)
            weight = loss_scale * (getattr(batch_data, "loss_weight", 1.0) if self.grad_sync is not None else 1.0)
            (loss if weight == 1.0 else loss * weight).backward()
            self._finish_update()
        else:
            if self.training:
                self.eval()
            with torch.no_grad():
                embs = self._embed(mode, dropedge=0., **fwd)
                head = self._fused_head(embs, index)
                if head is not None:
                    loss, probs, emb_ens = head
                else:
                    preds, emb_ens = self._head(embs)
                    loss = self._loss(preds, labels if index is None else index)
        if probs is None:
            probs = self.predict(preds)
        assert probs.shape[0] == labels.shape[0]
        return {'batch_size': probs.shape[0], 'loss': loss, 'labels': labels, 'preds': probs, 'emb_ens': emb_ens}

    def __str__(self):
        return f"model name: {type(self).__name__}"
