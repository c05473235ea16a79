"""GAT attention aggregate (all heads fused) over the sl_gat_* kernels:
the per-head ``_aggregate_attention`` loop of shaDow/layers.py:560-582,612-619."""
import torch

from . import _lib, ops
from ._lib import check


# True: hn = act(z_neigh) is not materialised -- the edge kernels apply the activation to the z_neigh rows they gather (one [n, F]
# tensor less kept per layer, one write per forward and one read per backward pass less).  Measured on the depth-3 products
# batches with elu (round 5, same box): gat_fwd 0.463 -> 0.592 ms, gat_bwd 0.948 -> 1.213 ms, step 11.59 -> 12.94 ms -- the gathers
# are issue-bound, four expm1f per gathered float4 cost more than the [n, F] stream they save.  Off; kept as the memory option.
RECOMPUTE_HN = False


# The row pass of the forward also applies the layer's act + per-head normalisation + branch average + output dropout to the
# aggregate it holds in registers (sl_gat_fwd_tail, round 6; equal to sl_gat_fwd_rows + sl_act_norm_fwd to rounding -- the compiler
# fuses multiply-add pairs differently in the two kernels).  False: the two separate launches (tests compare the two).
FUSED_FWD_TAIL = True


# The layer below a row-sparse backward pass receives its output gradient on a few rows (depth-3 products batches: 10 % of them).
# True: its act + norm backward leaves the aggregate's gradient COMPACT and the attention backward reads it through the rows' map
# (sl_gat_bwd_map: edges of rows without a gradient are passed over) -- no zero-filled [n, F] tensor, nine gathers in ten not
# issued; bit-identical to the expanded form (False; tests compare the two).
MAP_ROWS_GRADIENT = True


def _hn_buffer(n, F, dev):
    return None if RECOMPUTE_HN else torch.empty(n, F, device=dev)


def _p(t):
    return t.data_ptr() if t is not None and t.numel() else None


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_self, z_neigh, attention, adj, act_code, heads):
        z_self, z_neigh = ops._f32c(z_self).contiguous(), ops._f32c(z_neigh).contiguous()
        att = attention.detach().float().contiguous()
        ops._need_cuda(z_self, z_neigh, att)
        n, F = z_self.shape
        dev = z_self.device
        c = adj.csr
        hn = _hn_buffer(n, F, dev)
        u_s = torch.empty(n, heads, device=dev); u_n = torch.empty(n, heads, device=dev)
        mx = torch.empty(n, heads, device=dev); den = torch.empty(n, heads, device=dev)
        nagg = torch.empty(n, F, device=dev)
        w = adj.edge_w
        # algorithmic bytes (GAT row): structure + (edge mask) + read z_self, z_neigh, write the aggregate
        # + the per-node scores / softmax statistics of every head
        nbytes = 4 * (n + 1) + 4 * c.e + (4 * c.e if w is not None else 0) + 3 * 4 * n * F + 4 * 4 * n * heads
        with ops._timed(f"gat_fwd_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_fwd(c.indptr.data_ptr(), c.indices.data_ptr(), w.data_ptr() if w is not None else None,
                                         z_self.data_ptr(), z_neigh.data_ptr(), att.data_ptr(), act_code, n, F, heads,
                                         _p(hn), u_s.data_ptr(), u_n.data_ptr(), mx.data_ptr(), den.data_ptr(),
                                         nagg.data_ptr(), ops._stream(z_self)))
        ctx.save_for_backward(z_self, z_neigh, att, hn if hn is not None else att.new_empty(0), u_s, u_n, mx, den, nagg)
        ctx.adj, ctx.meta = adj, (act_code, heads, attention.shape)
        ops.fire_deferred()               # (the step's first aggregation is enqueued: see ops.defer)
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
        work = torch.empty(n * heads + 4096 * F + 4, device=dev)
        dzs = torch.empty_like(z_self); dzn = torch.empty_like(z_neigh)
        datt = torch.empty(2, F, device=dev)
        w = adj.edge_w
        # t pass + column pass: the transposed CSR structure; the incoming gradient (twice) and the aggregate, z_neigh and hn in, dz_self
        # (zeros) and dz_neigh out
        nbytes = 4 * (n + 1) + 8 * c.e + (4 * c.e if w is not None else 0) + 7 * 4 * n * F + 7 * 4 * n * heads
        with ops._timed(f"gat_bwd_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_bwd(c.indptr.data_ptr(), c.indices.data_ptr(), ti.data_ptr(), tx.data_ptr(), tp.data_ptr(),
                                         w.data_ptr() if w is not None else None, z_self.data_ptr(), z_neigh.data_ptr(),
                                         att.data_ptr(), act_code, n, c.e, F, heads, _p(hn), u_s.data_ptr(),
                                         u_n.data_ptr(), mx.data_ptr(), den.data_ptr(), nagg.data_ptr(), dnagg.data_ptr(), None,
                                         work.data_ptr(), dzs.data_ptr(), dzn.data_ptr(), datt.data_ptr(), 0, None,
                                         ops._stream(dnagg)))
        return dzs, dzn, datt.reshape(att_shape), None, None, None


class _GatTail(torch.autograd.Function):
    """Attention aggregate AND the layer's act + feature normalisation (shaDow/layers.py:612-625) as one autograd node:
        out = out_scale * (norm_0(N) + norm_1(act(z_self))),   N = the attention aggregate of (z_self, z_neigh).
    z_self feeds both the attention scores and the normalised output; as two nodes autograd adds its two gradient shares
    with a separate pass over [n, F].  Here the act_norm backward writes its share into the buffer the attention
    backward then ADDS to (sl_gat_bwd, accumulate_dz_self), and the two kernels leave the row maxima of the final
    (dz_self | dz_neigh) behind for the K-concatenated input-gradient product of ops._LinearPair."""
    @staticmethod
    def forward(ctx, z_self, z_neigh, attention, scale, offset, adj, act_code, heads, seg, out_scale, drop, link_roots=None, pair=None, pre=None):
        z_self, z_neigh = ops._f32c(z_self).contiguous(), ops._f32c(z_neigh).contiguous()
        att = attention.detach().float().contiguous()
        ops._need_cuda(z_self, z_neigh, att, scale, offset)
        n, F = z_self.shape
        dev = z_self.device
        c = adj.csr
        mx = torch.empty(n, heads, device=dev); den = torch.empty(n, heads, device=dev)
        nagg = torch.empty(n, F, device=dev)
        w = adj.edge_w
        out = None
        if pre is not None:
            # ``z_neigh`` IS hn = act(z_neigh) and the per-node terms are in hand (the paired Linear's kernel left them, ops.GatPre):
            # the row pass alone; the pre-activation does not exist -- the backward kernels take its derivative from hn
            hn, u_s, u_n = z_neigh, pre.u_s, pre.u_n
            z_neigh = att.new_empty(0)
            if FUSED_FWD_TAIL and not ops._is_dual(drop) and int(seg) * heads == F:
                # ... and the act + norm + average + dropout of the layer in the same pass: the aggregate goes to memory once (for the
                # backward pass) instead of out and back in
                out = torch.empty(n, F, device=dev)
                amax = torch.empty(n, device=dev) if n >= ops.AMAX_HANDOVER_ROWS else None
                sc = scale.reshape(2, F).contiguous().float()
                of = offset.reshape(2, F).contiguous().float()
                # structure (+ mask); hn and z_self in, the aggregate and the layer output out; per-node scores / softmax statistics
                nbytes = 4 * (n + 1) + 4 * c.e + (4 * c.e if w is not None else 0) + 4 * 4 * n * F + 4 * 4 * n * heads
                with ops._timed(f"gat_fwd_tail_F{F}_H{heads}", nbytes, dev):
                    check(_lib.load().sl_gat_fwd_tail(c.indptr.data_ptr(), c.indices.data_ptr(), w.data_ptr() if w is not None else None,
                                                      hn.data_ptr(), u_s.data_ptr(), u_n.data_ptr(), z_self.data_ptr(), act_code,
                                                      sc.data_ptr(), of.data_ptr(), n, F, heads, float(out_scale), float(drop[0]), int(drop[1]),
                                                      mx.data_ptr(), den.data_ptr(), nagg.data_ptr(), out.data_ptr(),
                                                      amax.data_ptr() if amax is not None else None, ops._stream(z_self)))
                if amax is not None:
                    ops.set_row_amax(out, amax)
                _GatTail.fused_tail_calls += 1
            else:
              nbytes = 4 * (n + 1) + 4 * c.e + (4 * c.e if w is not None else 0) + 2 * 4 * n * F + 4 * 4 * n * heads
              with ops._timed(f"gat_fwd_F{F}_H{heads}", nbytes, dev):
                check(_lib.load().sl_gat_fwd_rows(c.indptr.data_ptr(), c.indices.data_ptr(), w.data_ptr() if w is not None else None,
                                                  hn.data_ptr(), u_s.data_ptr(), u_n.data_ptr(), n, F, heads, mx.data_ptr(), den.data_ptr(),
                                                  nagg.data_ptr(), ops._stream(z_self)))
            _GatTail.pre_calls += 1
        else:
          hn = _hn_buffer(n, F, dev)
          u_s = torch.empty(n, heads, device=dev); u_n = torch.empty(n, heads, device=dev)
          nbytes = 4 * (n + 1) + 4 * c.e + (4 * c.e if w is not None else 0) + 3 * 4 * n * F + 4 * 4 * n * heads
          with ops._timed(f"gat_fwd_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_fwd(c.indptr.data_ptr(), c.indices.data_ptr(), w.data_ptr() if w is not None else None,
                                         z_self.data_ptr(), z_neigh.data_ptr(), att.data_ptr(), act_code, n, F, heads,
                                         _p(hn), u_s.data_ptr(), u_n.data_ptr(), mx.data_ptr(), den.data_ptr(),
                                         nagg.data_ptr(), ops._stream(z_self)))
        sc = scale.reshape(2, F).contiguous().float()
        of = offset.reshape(2, F).contiguous().float()
        # reference order: f_norm([neigh, self]) -> scale[0] = neigh (identity: the aggregate is activated already), scale[1] = self
        if out is None:
            out = ops._an_fwd([nagg, z_self], [None, None], (0, act_code), sc, of, seg, out_scale, drop)
        ctx.save_for_backward(z_self, z_neigh, att, hn if hn is not None else att.new_empty(0), u_s, u_n, mx, den, nagg, sc, of)
        ctx.adj, ctx.meta = adj, (act_code, heads, attention.shape, seg, out_scale, drop, scale.shape, offset.shape)
        ctx.link_roots = None
        ctx.pair = pair                                       # (ops.PairLink of the node that produced z_self and z_neigh, or None)
        # (the selected-rows act_norm backward lives in the vector kernel: float4 rows, power-of-two head slices)
        if link_roots is not None and not ops._is_dual(drop) and _lib.load().sl_act_norm_vector_layout(F, int(seg)):
            link_roots.published = True                       # (only a row-selecting read-out reads `out`, see ops.RootsLink)
            link_roots.csr = c                                # (ops.select_roots may build the row sets of the row-sparse pass from it)
            link_roots.want_levels = True
            ctx.link_roots = link_roots
        ctx.set_materialize_grads(False)
        ops.fire_deferred()               # (the step's first aggregation is enqueued: see ops.defer)
        return out

    sparse_top_calls = 0
    fused_tail_calls = 0     # forward passes whose row pass carried the act + norm tail (sl_gat_fwd_tail)
    pre_calls = 0            # forward passes that took hn / u_s / u_n from the paired Linear's kernel (tests assert on it)

    @staticmethod
    def _rows_backward(ctx, lr, level, rest):
        """A layer whose output gradient lives on a few rows (``lr.rows32`` / ``lr.grad`` -- the roots under a row-selecting
        read-out for the top layer, the rows T the layer above handed down for the one below): the gradient of the aggregate and
        the normalised branch's share of dz_self live on those rows, the attention backward touches THEIR edges only, and
        dz_self / dz_neigh are non-zero on those rows / on their inputs ``level.in_ids_full`` (tail.build_backward_levels).  The
        ordinary kernels run on the COMPACT problem -- the saved tensors gathered on the input set, the rows as a square CSR over
        it (RectLevel.square) -- and the two gradients go to the paired Linear's node on that set (ops.PairLink) together with the
        levels that are left; the forward pass is untouched."""
        z_self, z_neigh, att, hn, u_s, u_n, mx, den, nagg, sc, of = ctx.saved_tensors
        act_code, heads, att_shape, seg, out_scale, drop, sshape, oshape = ctx.meta
        adj = ctx.adj
        n, F = z_self.shape
        dev = z_self.device
        f32 = dict(dtype=torch.float32, device=dev)
        Tl, sidx, t, r = level.in_ids_full, level.self_idx, level.m_in, level.r
        # act + norm backward on the given rows: Z and this layer's output-dropout mask are addressed through the row ids, the
        # two gradients stay compact ([r, F])
        dn_r, dzs_r = torch.empty(r, F, **f32), torch.empty(r, F, **f32)
        _dz, dsc, dof, _ = ops._an_bwd([nagg, z_self], [None, None], (0, act_code), sc, of, seg, out_scale, (lr.grad,), [True, True], False, drop,
                                       dz_out=[dn_r, dzs_r], row_idx=lr.rows32, dz_compact=True)
        lr.release()
        csr_c, order = level.square
        E = csr_c.e
        ti, tx, tp = csr_c.transposed
        w = adj.edge_w.index_select(0, level.edge_pos.index_select(0, order)) if adj.edge_w is not None else None
        # the saved tensors on the input set, the two gradients scattered into zeroed tensors over it: two launches (ops.rows_multi;
        # rounds 5 - 6a: eight index_select, two zeros, two index_copy_)
        like = lambda x: torch.empty(t, x.shape[1], **f32)
        src = [z_self, u_s, u_n, mx, den, nagg] + ([z_neigh] if z_neigh.numel() else []) + ([hn] if hn.numel() else [])
        out = [like(x) for x in src]
        dn_c, dzs_c = torch.empty(t, F, **f32), torch.empty(t, F, **f32)
        ops.rows_multi([("gather", x, y) for x, y in zip(src, out)] + [("clear", None, dn_c), ("clear", None, dzs_c)], Tl, t)
        ops.rows_multi([("scatter", dn_r, dn_c), ("scatter", dzs_r, dzs_c)], sidx, r)
        zs_c, us_c, un_c, mx_c, den_c, na_c = out[:6]
        zn_c = out[6] if z_neigh.numel() else None              # (no pre-activation kept: hn carries the derivative, ops.GatPre)
        hn_c = out[-1] if hn.numel() else None
        dzn_c = torch.empty(t, F, **f32)
        datt = torch.empty(2, F, **f32)
        work = torch.empty(t * heads + 4096 * F + 4, **f32)
        nbytes = 4 * (t + 1) + 8 * E + (4 * E if w is not None else 0) + 6 * 4 * t * F + 7 * 4 * t * heads
        with ops._timed(f"gat_bwd_rows_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_bwd(csr_c.indptr.data_ptr(), csr_c.indices.data_ptr(), ti.data_ptr(), tx.data_ptr(), tp.data_ptr(),
                                         w.data_ptr() if w is not None else None, zs_c.data_ptr(), _p(zn_c),
                                         att.data_ptr(), act_code, t, E, F, heads, _p(hn_c), us_c.data_ptr(),
                                         un_c.data_ptr(), mx_c.data_ptr(), den_c.data_ptr(), na_c.data_ptr(), dn_c.data_ptr(), None,
                                         work.data_ptr(), dzs_c.data_ptr(), dzn_c.data_ptr(), datt.data_ptr(), 1, None, ops._stream(dn_c)))
        pair = ctx.pair
        pair.rows32, pair.dza, pair.dzb, pair.levels = level.in32, dzs_c, dzn_c, list(rest)
        pair.row_map = level.in_map32
        pair.dummy = ops.placeholder(n, F, dev)
        pair.filled = True
        _GatTail.sparse_top_calls += 1
        return (pair.dummy, pair.dummy, datt.reshape(att_shape), dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, None,
                None, None, None)

    mapped_calls = 0         # backward passes that read a row-sparse incoming gradient through its map (sl_gat_bwd_map)

    @staticmethod
    def _mapped_backward(ctx, lr):
        """The dense backward of a layer whose output gradient lives on the rows ``lr.rows32`` (``lr.row_map``: their positions):
        the act + norm backward runs on those rows and leaves d aggregate, the normalised branch's share of dz_self, their row
        maxima and t COMPACT; dz_self is expanded (the paired Linear's products read it densely), the aggregate's gradient is not --
        the column walk finds a row's through the map and passes over the edges of rows that have none."""
        z_self, z_neigh, att, hn, u_s, u_n, mx, den, nagg, sc, of = ctx.saved_tensors
        act_code, heads, att_shape, seg, out_scale, drop, sshape, oshape = ctx.meta
        adj = ctx.adj
        c = adj.csr
        n, F = z_self.shape
        dev = z_self.device
        f32 = dict(dtype=torch.float32, device=dev)
        rows32, row_map, grad = lr.rows32, lr.row_map, lr.grad
        m = int(rows32.numel())
        dn_c, dzs_c = torch.empty(m, F, **f32), torch.empty(m, F, **f32)
        want_amax = n >= ops.AMAX_HANDOVER_ROWS
        amax_c = torch.empty(m, 1, **f32) if want_amax else None
        t_c = torch.empty(m, heads, **f32)
        _dz, dsc, dof, _ = ops._an_bwd([nagg, z_self], [None, None], (0, act_code), sc, of, seg, out_scale, (grad,), [True, True], False, drop,
                                       dz_out=[dn_c, dzs_c], row_idx=rows32, dz_compact=True, dz0_amax=amax_c, amax_branch=1, t_out=(0, t_c))
        lr.release()
        dzs = torch.zeros(n, F, **f32)
        amax = torch.zeros(n, 1, **f32) if want_amax else None
        ops.rows_multi([("scatter", dzs_c, dzs)] + ([("scatter", amax_c, amax)] if want_amax else []), rows32.long(), m)
        ti, tx, tp = c.transposed
        work = torch.empty(4096 * F + n * heads + 4, **f32)
        dzn = torch.empty(n, F, **f32)
        datt = torch.empty(2, F, **f32)
        w = adj.edge_w
        # the transposed structure (+ the mask through its permutation) and the rows' map; hn in, dz_neigh out; the [n, heads] scores /
        # statistics; the gradient rows that exist
        nbytes = 4 * (n + 1) + 8 * c.e + (4 * c.e if w is not None else 0) + 4 * n + 2 * 4 * n * F + 5 * 4 * n * heads + 4 * m * (F + heads)
        with ops._timed(f"gat_bwd_map_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_bwd_map(c.indptr.data_ptr(), c.indices.data_ptr(), ti.data_ptr(), tx.data_ptr(), tp.data_ptr(),
                                             w.data_ptr() if w is not None else None, z_self.data_ptr(), _p(z_neigh),
                                             att.data_ptr(), act_code, n, c.e, F, heads, _p(hn), u_s.data_ptr(),
                                             u_n.data_ptr(), mx.data_ptr(), den.data_ptr(), nagg.data_ptr(), dn_c.data_ptr(), t_c.data_ptr(),
                                             row_map.data_ptr(), work.data_ptr(), dzs.data_ptr(), dzn.data_ptr(), datt.data_ptr(), 1,
                                             amax.data_ptr() if amax is not None else None, ops._stream(dn_c)))
        if amax is not None:
            amax = amax.reshape(n)
            ops.set_row_amax(dzs, amax); ops.set_row_amax(dzn, amax)
        _GatTail.mapped_calls += 1
        return (dzs, dzn, datt.reshape(att_shape), dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, None, None, None, None)

    @staticmethod
    def backward(ctx, *dout):
        z_self, z_neigh, att, hn, u_s, u_n, mx, den, nagg, sc, of = ctx.saved_tensors
        act_code, heads, att_shape, seg, out_scale, drop, sshape, oshape = ctx.meta
        adj = ctx.adj
        c = adj.csr
        n, F = z_self.shape
        dev = z_self.device
        # act + norm backward: d aggregate and the normalised branch's share of dz_self (on the read-out's rows only when it
        # handed over (rows, values): the other rows of both are zero)
        lr, rows = ctx.link_roots, None
        if lr is not None and lr.filled:
            g = dout[0]
            if g is None or g.data_ptr() != lr.dummy.data_ptr() or tuple(g.stride()) != (0, 0):
                raise RuntimeError("sparse read-out gradient: the layer's output has a consumer besides the read-out")
            lv = lr.levels[0] if lr.levels else None
            if (ops.SPARSE_TOP_BWD and lv is not None and ctx.pair is not None and n >= ops.SPARSE_TOP_BWD_MIN_ROWS and not ops._is_dual(drop)
                    and lv.r == int(lr.rows32.numel()) and lv.in_ids_full is not None):
                return _GatTail._rows_backward(ctx, lr, lv, lr.levels[1:])
            rows, dout = lr.rows32, (lr.grad,)
        # (the row maxima asked for are those of the SECOND branch's gradient -- dz_self's, which the attention backward below no
        #  longer reads on the rows it leaves alone; rounds 5 - 6a passed the branches self-first instead and flipped the [2, F]
        #  scale / offset rows and their gradients: four small kernels per layer)
        vec = bool(_lib.load().sl_act_norm_vector_layout(F, int(seg)) and int(seg) * heads == F)
        if rows is not None and MAP_ROWS_GRADIENT and vec and lr.row_map is not None and lr.row_map.numel() == n:
            return _GatTail._mapped_backward(ctx, lr)
        amax = torch.empty(n, device=dev) if n >= ops.AMAX_HANDOVER_ROWS else None
        if amax is not None and rows is not None:
            amax.zero_()                 # (rows the read-out gradient does not reach: dz_self = 0)
        # (... and t_i = dN_i . N_i per head for the attention backward, while both rows are in registers; with the read-out's row
        #  list the other rows' dN is zero: t cleared first)
        #  (the act + norm vector kernel only: other head widths leave t to sl_gat_bwd's own row-wise pre-pass)
        tdot = None
        if _lib.load().sl_act_norm_vector_layout(F, int(seg)) and int(seg) * heads == F:
            tdot = (torch.zeros if rows is not None else torch.empty)(n, heads, device=dev)
        (dnagg, dzs), dsc, dof, _ = ops._an_bwd([nagg, z_self], [None, None], (0, act_code), sc, of, seg,
                                                out_scale, dout, [True, True], False, drop, row_idx=rows, dz0_amax=amax, amax_branch=1,
                                                t_out=(0, tdot) if tdot is not None else None)
        if rows is not None:
            lr.release()
        ti, tx, tp = c.transposed
        work = torch.empty(4096 * F + n * heads + 4, device=dev)
        dzn = torch.empty_like(z_self)
        datt = torch.empty(2, F, device=dev)
        w = adj.edge_w
        # (round 5: both CSR structures; the incoming gradient, the aggregate, z_neigh and hn in, dz_neigh out -- dz_self / z_self are
        #  no longer touched: the attention's share of dz_self is exactly zero, see gat_row_bwd_kernel)
        # (round 6: one edge walk -- the transposed structure (+ the mask through its permutation); the incoming gradient and hn in,
        #  dz_neigh out; the [n, heads] scores / statistics / t)
        nbytes = 4 * (n + 1) + 8 * c.e + (4 * c.e if w is not None else 0) + (3 if (RECOMPUTE_HN or not z_neigh.numel()) else 4) * 4 * n * F + 7 * 4 * n * heads
        with ops._timed(f"gat_bwd_F{F}_H{heads}", nbytes, dev):
            check(_lib.load().sl_gat_bwd(c.indptr.data_ptr(), c.indices.data_ptr(), ti.data_ptr(), tx.data_ptr(), tp.data_ptr(),
                                         w.data_ptr() if w is not None else None, z_self.data_ptr(), _p(z_neigh),
                                         att.data_ptr(), act_code, n, c.e, F, heads, _p(hn), u_s.data_ptr(),
                                         u_n.data_ptr(), mx.data_ptr(), den.data_ptr(), nagg.data_ptr(), dnagg.data_ptr(), tdot.data_ptr() if tdot is not None else None,
                                         work.data_ptr(), dzs.data_ptr(), dzn.data_ptr(), datt.data_ptr(), 1,
                                         amax.data_ptr() if amax is not None else None, ops._stream(dnagg)))
        if amax is not None:              # (ONE array for both gradients: the maximum over the pair of rows)
            ops.set_row_amax(dzs, amax); ops.set_row_amax(dzn, amax)
        return (dzs, dzn, datt.reshape(att_shape), dsc.reshape(sshape), dof.reshape(oshape), None, None, None, None, None, None, None, None, None)


def gat_tail_usable(x, F: int, heads: int) -> bool:
    """Shapes gat_tail takes (the caller may then ask the paired Linear for hn / u_s / u_n: ops.GatPre)."""
    heads = int(heads)
    return bool(torch.is_tensor(x) and x.is_cuda and F <= 256 and F % heads == 0 and _fused_slice(F // heads))


def gat_tail(adj: "ops.NormAdj", z_self, z_neigh, attention, act: str, heads: int, scale, offset, seg: int, out_scale: float,
             out_dropout: float = 0.0, dual: bool = False, roots_only: bool = False):
    """out_scale * (norm(gat_aggregate(...)) + norm(act(z_self))) as one node (see _GatTail); None when the shape needs the
    padded multi-launch aggregate (the caller then composes gat_aggregate and ops.act_norm)."""
    n, F = z_self.shape
    heads = int(heads)
    pre = getattr(z_neigh, "_shd_gat_pre", None)
    pre = pre if (pre is not None and pre.filled) else None
    if not gat_tail_usable(z_self, F, heads):
        if pre is not None:
            raise RuntimeError("gat_tail: the paired Linear already left hn in z_neigh's place but the fused tail does not take this shape")
        return None
    drop = ops._drop_arg(out_dropout, F, seg, dual)
    link = ops.RootsLink() if (roots_only and not dual and ops.ROOTS_SPARSE_GRAD) else None
    pair = getattr(z_self, "_shd_pair", None)
    if pair is None or getattr(z_neigh, "_shd_pair", None) is not pair:
        pair = None                     # (the two inputs are not the two outputs of ONE paired Linear)
    res = _GatTail.apply(z_self, z_neigh, attention, scale, offset, adj, ops.ACT_CODE[act], heads, int(seg), float(out_scale), drop, link, pair, pre)
    if link is not None and link.published and torch.is_tensor(res):
        res._shadow_roots = link
    return res


def _fused_slice(D: int) -> bool:
    """Head widths the fused kernels take: 4 * 2^k floats."""
    return D >= 4 and D % 4 == 0 and ((D // 4) & (D // 4 - 1)) == 0


def gat_aggregate(adj: "ops.NormAdj", z_self, z_neigh, attention, act: str, heads: int):
    """N = softmax_row(lrelu(a_s.act(z_self)) + lrelu(a_n.act(z_neigh))) @ act(z_neigh), per head.

    The fused kernels take up to 256 columns of heads that are 4 * 2^k wide (the reference's dim 256 / 4 heads).
    Other shapes -- dim 512 / 4 heads and dim 800 / 4 heads in config_train/{products,papers100M}/leaderboard --
    run as several launches over groups of heads, each head zero-padded to the next such width: heads are
    independent, and a zero column contributes nothing (act(0) = 0 for every supported activation, its attention
    weight is padded with 0)."""
    code, heads = ops.ACT_CODE[act], int(heads)
    n, F = z_self.shape
    D = F // heads
    assert D * heads == F
    if F <= 256 and _fused_slice(D):
        return _GatAggregate.apply(z_self, z_neigh, attention, adj, code, heads)
    Dp = 4
    while Dp < D:
        Dp *= 2
    if Dp > 256:
        raise NotImplementedError(f"GAT head width {D} > 256 is not provided")
    per = max(1, 256 // Dp)                                   # heads per launch
    att = attention.reshape(2, heads, D)
    zs, zn = z_self.reshape(n, heads, D), z_neigh.reshape(n, heads, D)
    pad = (lambda t: torch.nn.functional.pad(t, (0, Dp - D))) if Dp != D else (lambda t: t)
    outs = []
    for h0 in range(0, heads, per):
        h1 = min(heads, h0 + per)
        o = _GatAggregate.apply(pad(zs[:, h0:h1]).reshape(n, (h1 - h0) * Dp), pad(zn[:, h0:h1]).reshape(n, (h1 - h0) * Dp),
                                pad(att[:, h0:h1]).contiguous(), adj, code, h1 - h0)
        outs.append(o.reshape(n, h1 - h0, Dp)[:, :, :D])
    return torch.cat(outs, dim=1).reshape(n, F)
