"""DeepGNN with the reference's API surface (shaDow/models.py:16-237): same
constructor arguments, layer registry keys, parameter names (the checkpoint
contract), ``forward`` signature and ``step`` return dict -- running on the
HIP layers of ``shadow_gnn_amd.layers``.

Data parallel use: construct with ``grad_sync=dist.GradSync(...)`` and every
``step`` all-reduces ONE flattened fp32 gradient bucket over RCCL before the
clip + Adam update (DESIGN.md section "multi-GPU")."""
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import layers, ops, tail
from .minibatch import TRAIN, OneBatchSubgraph


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
        assert prediction_task in {'link', 'node'}
        if num_ensemble != 1:
            raise NotImplementedError("subgraph ensembles (EnsembleAggregator) are outside the hot path built here")
        self.prediction_task = prediction_task
        self.num_gnn_layers = arch_gnn["num_layers"]
        self.num_cls_layers = arch_gnn.get("num_cls_layers", 1)
        self.dropout, self.dropedge = train_params["dropout"], train_params['dropedge']
        self.mulhead = int(arch_gnn.get("heads", 1))
        self.branch_sharing = arch_gnn.get('branch_sharing', False)
        self.type_feature_augment = aug_feat
        assert dim_feat_raw <= dim_feat_smooth
        self.num_classes = dim_label_raw
        self.dim_label_in = dim_label_smooth
        self.dim_feat_in = dim_feat_smooth
        self.dim_hid = arch_gnn['dim']
        act, layer_norm = arch_gnn['act'], arch_gnn.get('layer_norm', 'norm_feat')
        self.feat_aug_ops = arch_gnn.get('feature_augment_ops', 'sum')
        aug_layers, conv_layers, res_pool_layers = [], [], []
        for i in range(num_ensemble):
            dim_aug_add = 0
            if len(self.type_feature_augment) > 0:
                _dim_aug_out = self.dim_feat_in if self.feat_aug_ops == 'sum' else self.dim_hid
                dim_aug_add += 0 if self.feat_aug_ops == 'sum' else _dim_aug_out
                aug_layers.append(nn.ModuleList(
                    nn.Linear(_dim, _dim_aug_out) for _, _dim in self.type_feature_augment))
            convs = []
            for j in range(self.num_gnn_layers):
                dim_in = (self.dim_feat_in + self.dim_label_in + dim_aug_add) if j == 0 else self.dim_hid
                if arch_gnn['aggr'] not in DeepGNN.NAME2CLS:
                    raise NotImplementedError(f"aggr {arch_gnn['aggr']!r} not provided (have {sorted(DeepGNN.NAME2CLS)})")
                convs.append(DeepGNN.NAME2CLS[arch_gnn['aggr']](
                    dim_in, self.dim_hid, dropout=self.dropout, act=act, norm=layer_norm, mulhead=self.mulhead))
            conv_layers.append(nn.Sequential(*convs))
            type_res = arch_gnn.get('residue', 'none').lower()
            type_pool = arch_gnn.get('pooling', 'center').split('-')[0].lower()
            res_pool_layers.append(layers.ResPool(
                self.dim_hid, self.dim_hid, self.num_gnn_layers, type_res, type_pool, dropout=self.dropout,
                act=act, args_pool={}, prediction_task=self.prediction_task))
        if len(aug_layers) > 0:
            self.aug_layers = nn.ModuleList(aug_layers)
        else:
            self.aug_layers = []
        self.conv_layers = nn.ModuleList(conv_layers)
        self.res_pool_layers = nn.ModuleList(res_pool_layers)
        self.ensembler = layers.EnsembleDummy()
        _norm_type = 'norm_feat' if self.prediction_task == 'node' else 'none'
        classifier = []
        for i in range(self.num_cls_layers):
            if i < self.num_cls_layers - 1:
                _kwargs = {'dim_out': self.dim_hid, 'act': act, 'dropout': self.dropout}
            else:
                _kwargs = {'dim_out': self.num_classes, 'act': 'I', 'dropout': 0.}
            _kwargs.update({'dim_in': self.dim_hid, 'norm': _norm_type})
            classifier.append(DeepGNN.NAME2CLS['mlp'](**_kwargs))
        self.classifier = nn.Sequential(*classifier)
        self.lr = train_params["lr"]
        self.sigmoid_loss = arch_gnn.get("loss", "softmax") == "sigmoid"
        self.optimizer = torch.optim.Adam(self.parameters(), lr=self.lr)
        self.fuse_dropout = True       # fold each layer's input dropout into the producing kernel where possible
        # Exact dead-row elimination for residue 'none' + centre pooling (tail.py): the last layers are computed
        # only on the rows the roots depend on.  Off by default: the reference computes every row.
        self.prune_tail = False
        self.num_ensemble = num_ensemble
        self.grad_sync = grad_sync

    def _loss(self, preds, labels):
        if self.sigmoid_loss:
            assert preds.shape == labels.shape
            return torch.nn.BCEWithLogitsLoss()(preds, labels.type(preds.dtype)) * preds.shape[1]
        if len(labels.shape) == 2:
            labels = torch.max(labels, dim=1)[1]
        return torch.nn.CrossEntropyLoss()(preds, labels)

    def forward(self, mode, feat_ens, adj_ens, target_ens, size_subg_ens, feat_aug_ens, dropedge, tail_ens=None):
        num_ensemble = len(feat_ens)
        emb_subg_ens = []
        for i in range(num_ensemble):
            tgt = torch.as_tensor(target_ens[i], device=feat_ens[i].device).long()
            if self.dim_label_in > 0 and mode == TRAIN:
                feat_ens[i][tgt, -self.dim_label_in:] = 0
            if len(self.type_feature_augment) > 0:
                for ia, (ta, _dim) in enumerate(self.type_feature_augment):
                    enc = feat_aug_ens[i][ta]
                    if isinstance(enc, ops.OneHotCodes):
                        if (self.feat_aug_ops == 'sum' and enc.dim <= 16
                                and self.dim_feat_in == feat_ens[i].shape[1]):
                            # fused one-hot + Linear + add: no [n, dim] matrix, one pass over the features
                            feat_ens[i] = ops.onehot_linear_add(feat_ens[i], enc.codes, self.aug_layers[i][ia])
                            continue
                        enc = enc.dense()
                    feat_aug_emb = self.aug_layers[i][ia](enc)
                    if self.feat_aug_ops == 'sum':
                        # (the reference adds in place into the gathered features, models.py:189)
                        if self.dim_feat_in == feat_ens[i].shape[1]:
                            feat_ens[i] = feat_ens[i] + feat_aug_emb
                        else:
                            feat_ens[i] = torch.cat([feat_ens[i][:, :self.dim_feat_in] + feat_aug_emb,
                                                     feat_ens[i][:, self.dim_feat_in:]], dim=1)
                    else:
                        feat_ens[i] = torch.cat([feat_ens[i], feat_aug_emb], dim=1)
            xjk = []
            convs = list(self.conv_layers[i])
            adj_i = adj_ens[i]
            levels = []
            if self.prune_tail and self._tail_prunable(i):
                adj_i = layers._as_device_csr(adj_i, feat_ens[i].device)
                if tail_ens is not None and tail_ens[i] is not None:
                    levels = tail_ens[i]             # built by the minibatch on its prefetch stream
                    assert len(levels) <= len(convs)
                else:
                    levels = tail.build_tail_plan(adj_i, tgt, len(convs))
            num_full = len(convs) - len(levels)
            xmd = (feat_ens[i], adj_i, False, dropedge)
            self._plan_dropout_fusion(i)
            for md in convs[:num_full]:
                xmd = md(xmd, sizes_subg=size_subg_ens[i])
                xjk.append(xmd[0])
                dropped = md.take_dropped_out() if hasattr(md, 'take_dropped_out') else None
                if dropped is not None:       # dual mode: the read-out keeps the plain output, the next layer
                    xmd = (dropped,) + tuple(xmd[1:])     # gets the one its input dropout was applied to
            if levels:
                # target-only tail: each remaining layer on the rows the roots depend on; the last one yields the
                # root rows in target order, which is all that residue 'none' + centre pooling reads
                x, adj_norm = xmd[0], xmd[1]
                if num_full == 0:
                    adj_norm = (convs[0].norm_adj(adj_i, False, dropedge, x.device) if hasattr(convs[0], 'norm_adj')
                                else convs[0]._adj_norm(adj_i, False, x.device, dropedge=dropedge))
                for md, level in zip(convs[num_full:], levels):
                    x = md.forward_rows(x, adj_norm, level)
                emb_subg_i = x
            else:
                emb_subg_i = self.res_pool_layers[i](xjk, tgt, size_subg_ens[i])
            emb_subg_i = F.normalize(emb_subg_i, p=2, dim=1)
            emb_subg_ens.append(emb_subg_i)
        emb_ensemble = self.ensembler(emb_subg_ens)
        pred_subg = self.classifier(emb_ensemble)
        return pred_subg, emb_subg_ens

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
        dual = not (rp.type_res == 'none' and rp.type_pool == 'center')
        for l, md in enumerate(layers_i):
            if not hasattr(md, 'out_dropout'):
                continue
            nxt = layers_i[l + 1] if l + 1 < len(layers_i) else None
            fuse = (fuse_ok and nxt is not None and hasattr(nxt, 'input_pre_dropped') and nxt.dropout > 0
                    and md.can_fuse_out_dropout())
            md.out_dropout = nxt.dropout if fuse else 0.0
            md.out_dual = bool(fuse and dual)
            if nxt is not None and hasattr(nxt, 'input_pre_dropped'):
                nxt.input_pre_dropped = bool(fuse)
        if layers_i and hasattr(layers_i[0], 'input_pre_dropped'):
            layers_i[0].input_pre_dropped = False

    def predict(self, preds):
        return torch.sigmoid(preds) if self.sigmoid_loss else F.softmax(preds, dim=1)

    def step(self, mode, status, batch_data: OneBatchSubgraph, loss_scale: float = 1.0):
        assert status in ['running', 'final']
        args_forward_common = batch_data.to_dict(
            {"feat_ens", "adj_ens", "target_ens", "size_subg_ens", "feat_aug_ens"})
        # the step consumes the batch record (features are augmented in place in the reference)
        args_forward_common["feat_ens"] = list(args_forward_common["feat_ens"])
        args_forward_common["tail_ens"] = getattr(batch_data, "tail_ens", None)
        label_targets = batch_data.label
        if len(label_targets.shape) == 1 and self.num_classes > 1:
            label_targets = F.one_hot(label_targets.to(torch.int64), num_classes=self.num_classes)
        if mode == TRAIN and status == 'running':
            self.train()
            if self.grad_sync is not None:
                self.grad_sync.zero()
            else:
                self.optimizer.zero_grad(set_to_none=True)
            preds, emb_ens = self(mode, dropedge=self.dropedge, **args_forward_common)
            loss = self._loss(preds, label_targets)
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
            if self.grad_sync is not None:
                self.grad_sync.all_reduce(self.parameters())
            torch.nn.utils.clip_grad_norm_(self.parameters(), 5)
            self.optimizer.step()
        else:
            self.eval()
            with torch.no_grad():
                preds, emb_ens = self(mode, dropedge=0., **args_forward_common)
                loss = self._loss(preds, label_targets)
        assert preds.shape[0] == label_targets.shape[0]
        return {'batch_size': preds.shape[0], 'loss': loss, 'labels': label_targets,
                'preds': self.predict(preds), 'emb_ens': emb_ens}

    def __str__(self):
        return f"model name: {type(self).__name__}"
