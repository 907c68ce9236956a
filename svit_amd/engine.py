"""Forward / backward schedule of the SViT backbone over the HIP kernels.

This is the MI355X-first replacement for what the reference leaves to autograd over ~60 ATen
ops per block (slowfast/models/attention.py:331-466,557-571; video_model_builder.py:315-375):
an explicit, fixed launch schedule -- every launch is a C-ABI call into libsvit_hip.so on
torch's current stream, activations live in the fp32 residual stream / bf16 GEMM operands laid
out token-major, and parameter gradients are accumulated straight into one flat fp32 buffer
(which is what the optimiser and the data-parallel all-reduce consume).  torch is used for
memory, streams and a handful of O(B*65*C) index/reduction ops on the special tokens.
"""
import math

import torch

from . import arch, hip, ops
from .input import U8Clips

HD = arch.HEAD_DIM
F32, BF16 = torch.float32, torch.bfloat16
SCALE = HD ** -0.5
LOG2E = 1.4426950408889634
# the pooled keys leave the pooling kernel multiplied by scale * log2(e) and the rel-pos columns of qa
# carry log2(e) * (q . R): qa . ka^T is then the attention score in the log2 domain and the fused
# attention kernels exponentiate it without a multiply (csrc/attn_fwd.hip)
K_SCALE = SCALE * LOG2E


def _align(n, a=8):
    return (n + a - 1) // a * a


class FlatParams:
    """One fp32 buffer for all parameters ([decayed | not decayed]), one for their grads, a bf16
    mirror for GEMM operands and a bf16 buffer of transposed Linear weights for dgrad."""

    def __init__(self, shapes, decay_of, device, rank_of):
        names = list(shapes)
        depth = 1 + max([int(n.split(".")[1]) for n in names if n.startswith("blocks.")] or [0])
        ranks = {n: rank_of(n, depth) for n in names}
        # [decayed | not decayed]; inside each group in gradient-readiness order
        order = sorted(names, key=lambda n: (not decay_of(n, shapes[n]), ranks[n]))
        self.slots, off = {}, 0
        self.n_decay = 0
        self.n_ranks = depth + 2
        # ready_ranges[r] = list of (begin, end) slices of the flat buffer final after rank r
        self.ready_ranges = [[] for _ in range(self.n_ranks)]
        for n in order:
            numel = int(math.prod(shapes[n]))
            self.slots[n] = (off, numel, tuple(shapes[n]))
            rr = self.ready_ranges[ranks[n]]
            if rr and rr[-1][1] == off:
                rr[-1] = (rr[-1][0], off + _align(numel))
            else:
                rr.append((off, off + _align(numel)))
            off += _align(numel)
            if decay_of(n, shapes[n]):
                self.n_decay = off
        self.total = off
        self.data = torch.zeros(off, device=device, dtype=F32)
        self.grad = torch.zeros(off, device=device, dtype=F32)
        # bf16 mirror (+ slack so that GEMM operands padded past a slot stay inside the buffer)
        self.w16 = torch.zeros(off + 96 * 96, device=device, dtype=BF16)
        # transposed copies of every 2-D Linear weight in the blocks
        self.t_slots, toff, table = {}, 0, []
        for n in order:
            if n.startswith("blocks.") and n.endswith(".weight") and len(shapes[n]) == 2:
                r, c = shapes[n]
                self.t_slots[n] = (toff, (c, r))
                table += [self.slots[n][0], toff, r, c, r]
                toff += _align(r * c)
        # per block: the three rel-pos tables transposed side by side, [96, Lpad] with 8-aligned
        # column sections (h | w | t) -- the B operand of the rel-pos backward GEMMs
        self.rel_slots = {}
        for n in order:
            if n.endswith("attn.rel_pos_h"):
                pre = n[:-len("attn.rel_pos_h")]
                rows = [shapes[pre + "attn.rel_pos_" + a][0] for a in "hwt"]
                offs, lpad = rel_sections(rows)
                self.rel_slots[pre] = (toff, lpad, offs)
                for a, o, r in zip("hwt", offs, rows):
                    table += [self.slots[pre + "attn.rel_pos_" + a][0], toff + o, r, HD, lpad]
                toff += _align(HD * lpad)
        # selector tables of the depthwise pooling weights (scalar operands of the LDS-tiled
        # stencils, csrc/pool.hip): [block * 3 + (q, k, v)][27][96] uint32, refreshed with the mirror
        pool_names = [n for n in names if ".attn.pool_" in n and n.endswith(".weight")]
        pool_names.sort(key=lambda n: (int(n.split(".")[1]), "qkv".index(n.split(".")[3][-1])))
        self.pool_sel_index = {n: i for i, n in enumerate(pool_names)}
        self.pool_sel = torch.zeros((max(1, len(pool_names)), 27 * 96), device=device, dtype=torch.int32)
        self.pool_sel_off = torch.tensor([self.slots[n][0] for n in pool_names] or [0], dtype=torch.int64,
                                         device=device)
        self.wT16 = torch.zeros(max(toff, 8), device=device, dtype=BF16)
        self.t_table = torch.tensor(table, dtype=torch.int64, device=device)
        self.n_t = len(table) // 5

    def view(self, buf, name):
        off, numel, shape = self.slots[name]
        return buf[off:off + numel].view(shape)

    def p(self, name):
        return self.view(self.data, name)

    def g(self, name):
        return self.view(self.grad, name)

    def w(self, name):
        """bf16 copy [out,in] of a weight."""
        off, numel, shape = self.slots[name]
        return self.w16[off:off + numel].view(shape)

    def wt(self, name):
        """bf16 transposed copy [in,out] of a 2-D weight."""
        off, shape = self.t_slots[name]
        return self.wT16[off:off + shape[0] * shape[1]].view(shape)

    def rel_cat_t(self, pre):
        """-> (bf16 [96, Lpad] transposed concatenation of the block's tables, Lpad, offsets)."""
        off, lpad, offs = self.rel_slots[pre]
        return self.wT16[off:off + HD * lpad].view(HD, lpad), lpad, offs

    def rel_tables32(self, pre):
        """fp32 [h+w+t rows, 96] view of the block's three (adjacent) rel-pos tables."""
        names = [pre + "attn.rel_pos_" + a for a in "hwt"]
        offs = [self.slots[n][0] for n in names]
        rows = [self.slots[n][2][0] for n in names]
        assert offs[1] == offs[0] + rows[0] * HD and offs[2] == offs[1] + rows[1] * HD
        return self.data[offs[0]:offs[0] + sum(rows) * HD].view(sum(rows), HD)

    def rel_cat(self, pre):
        """bf16 [Lpad96, 96] view of the block's three rel-pos tables (adjacent slots of the
        mirror; rows past h+w+t belong to other parameters and are never gathered),
        plus the row offsets of the three sections."""
        names = [pre + "attn.rel_pos_" + a for a in "hwt"]
        offs = [self.slots[n][0] for n in names]
        rows = [self.slots[n][2][0] for n in names]
        assert offs[1] == offs[0] + rows[0] * HD and offs[2] == offs[1] + rows[1] * HD
        total = rows[0] + rows[1] + rows[2]
        lpad = (total + 95) // 96 * 96
        return (self.w16[offs[0]:offs[0] + lpad * HD].view(lpad, HD), (0, rows[0], rows[0] + rows[1]))

    def sels(self, pre):
        """the three selector tables of a block's pool_q / pool_k / pool_v weights"""
        return [self.pool_sel[self.pool_sel_index[pre + "attn.pool_%s.weight" % r]] for r in "qkv"]

    # The LDS-tiled stride-1 stencil (csrc/pool.hip::pool_tiled_body) only pays on the largest
    # planes (measured, tools/bench_kernels.py pooltiled: 56x56 forward 111 -> 83 us; 28x28 and
    # below the same or slower: those launches are bound by per-workgroup latency chains, not by
    # the 27x re-reads -- DESIGN.md section 5)
    TILED_MIN_PLANE = 2048
    # round 3: the slab stencil (csrc/pool.hip::pool_slab_fwd_kernel: input slab resident in LDS, scalar
    # weights, LayerNorm as a second row-wise launch) on the small planes -- 14x14 and 7x7, 12 of 16 blocks
    SLAB_MAX_PLANE = 196

    def refresh_low_precision(self):
        ops.cast_bf16(self.data, self.w16[:self.total])
        if self.pool_sel_index:
            # selector tables = the SCALAR weight operands (s_load) of the forward stencils: pool_tiled_body<.., SW = true>
            # (56x56 stride-1 planes), pool_slab_fwd_kernel (blocks 14 / 15) -- csrc/pool.hip
            ops.pool_weight_sel(self.data, self.pool_sel_off, self.pool_sel)
        if self.n_t:    # from the mirror the cast above just wrote: half the bytes of the fp32 source
            ops.transpose_bf16_batched(self.w16, self.wT16, self.t_table, self.n_t, 256)


def rel_sections(rows):
    """column offsets (8-aligned) of the h / w / t sections and the 32-aligned total width."""
    off_w = _align(rows[0])
    off_t = off_w + _align(rows[1])
    return (0, off_w, off_t), _align(off_t + rows[2], 32)


def _rel_index(q_n, k_n):
    """dist.long() table of cal_rel_pos_* (attention.py:100-119,156-163), float32 arithmetic as
    in the reference so that truncation agrees for non-integer ratios."""
    q_ratio = max(k_n / q_n, 1.0)
    k_ratio = max(q_n / k_n, 1.0)
    d = torch.arange(q_n)[:, None] * q_ratio - torch.arange(k_n)[None, :] * k_ratio
    d = d + (k_n - 1) * k_ratio
    return d.long().to(torch.int32)


def _resize_matrix(rows_have, rows_need):
    """Dense [rows_need, rows_have] matrix of F.interpolate(mode='linear', align_corners=False)
    used by get_rel_pos (attention.py:68-81) when a table's length differs from 2*max(q,k)-1."""
    m = torch.zeros(rows_need, rows_have)
    scale = rows_have / rows_need
    for i in range(rows_need):
        pos = max((i + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(pos)), rows_have - 1)
        i1 = min(i0 + 1, rows_have - 1)
        lam = pos - i0
        m[i, i0] += 1.0 - lam
        m[i, i1] += lam
    return m


class Engine:
    def __init__(self, plan: arch.Plan, flat: FlatParams, red_group=4):
        self.plan, self.flat = plan, flat
        self.dev = flat.data.device
        self._rel_cache = {}
        self._interp_cache = {}     # (thw of the first block, save) -> the batched table-interpolation launch of a forward pass
        self._interp_out = {}       # block index -> (r32 or None, rcat bf16) of the forward pass that is running
        self.batch_interp = True    # False: one svit_table_interp launch per block (A/B: tools/diag/interp_ab.py)
        self._relq_cache = {}
        self.fused_scatter = True     # (False = the stand-alone rel-pos scatter launch; tests / tools flip the attribute)
        self.patch_w16 = torch.zeros((plan.embed_dim, 448), device=self.dev, dtype=BF16)
        self._tn = []           # weight-gradient GEMMs queued by the running block backward
        # a block's four second-stage reductions (LN2, pooled LN, conv wgrad, LN1) run as one
        # deferred launch: each producer gets its own region of this scratch (floats)
        # round 4: RED_GROUP consecutive blocks share ONE deferred launch (17 -> 5 reduce launches per step): every
        # block of a group owns its own copy of the scratch, the queue is run when the group's last block is done --
        # in front of the gradient-ready callback of the data-parallel buckets, which are four blocks wide as well
        # (1 = one launch per block, the round-3 schedule).  Scratch: RED_GROUP x 72 MB of fp32 partial rows.
        self.RED_GROUP = max(1, int(red_group))
        M1 = 1024 * 1024
        self._red_ws = torch.empty(self.RED_GROUP * 18 * M1, device=self.dev, dtype=F32)
        self._red_slot = 0
        self._wgrad_need, self._wgrad_big = {}, {}
        self._red_regions = {"ln2": (0, 4 * M1), "ln1": (4 * M1, 8 * M1), "pln": (8 * M1, 9 * M1),
                             "wgrad": (9 * M1, 18 * M1)}
        # the grouped weight-gradient GEMM leaves the dgrad chain: it runs on a second stream next
        # to the latency-bound kernels of the chain (its fp32 atomics execute at the memory side,
        # so nothing of the chain depends on it) and is joined only where gradients must be final
        # (all-reduce launch points, end of backward)
        self._side = None
        self._side_keep = []
        self._side_active = False
        # weight-gradient GEMMs on a side stream beside the dgrad chain: off (round-1 measurement: the TN GEMM beside
        # the chain cost more than it hid); tools set `engine.overlap_wgrad = True` to re-measure it
        self.overlap_wgrad = False
        self.attn_q_splits = 0      # 0 = heuristic; 1 = no query split in the dk/dv kernel
        # regression-diff mode: every reduction that normally meets in fp32 atomics (attention
        # dk/dv query splits, the row splits of the grouped weight-gradient GEMM) runs unsplit, so
        # the whole backward is bit-reproducible run to run (several times slower)
        self.deterministic = False
        self._capture_fork = None   # set by svit_amd/graph.py while it captures: (fn) -> None
        self._capture_join = None

    # ------------------------------------------------------------------ helpers ----------
    def refresh_weights(self):
        self.flat.refresh_low_precision()
        ops.pad_cast_rows(self.flat.p("patch_embed.proj.weight").view(self.plan.embed_dim, 441),
                          self.patch_w16)

    def _rel(self, blk, q_thw, k_thw):
        """index tables (+ resize matrices when needed) for a block at this resolution."""
        key = (blk.index, q_thw, k_thw)
        ent = self._rel_cache.get(key)
        if ent is None:
            idx = [_rel_index(q_thw[1], k_thw[1]), _rel_index(q_thw[2], k_thw[2]),
                   _rel_index(q_thw[0], k_thw[0])]
            need = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
            have = [blk.rel_sp_rows, blk.rel_sp_rows, blk.rel_t_rows]
            mats = [None if n == h else _resize_matrix(h, n).to(self.dev) for n, h in zip(need, have)]
            mcat = None
            if any(m is not None for m in mats):
                # one [lp96, h+w+t rows] matrix (resize blocks / identities on the diagonal, zero
                # pad rows) turns "interpolate three tables, concatenate, pad, cast" into a
                # single small matmul on the adjacent fp32 tables (the
                # per-block interpolation plumbing of the T' = 1 and 312^2 paths, batched)
                lp = (sum(need) + 95) // 96 * 96
                mcat = torch.zeros((lp, sum(have)), device=self.dev)
                r0 = c0 = 0
                for n, h, m in zip(need, have, mats):
                    mcat[r0:r0 + n, c0:c0 + h] = torch.eye(h, device=self.dev) if m is None else m
                    r0, c0 = r0 + n, c0 + h
            ent = ([t.contiguous().to(self.dev) for t in idx], mats, mcat, tuple(need))
            self._rel_cache[key] = ent
        return ent

    def _interp_tables(self, thw, save):
        """Rel-pos tables of every block whose grid differs from its tables' (odd crops, the T' = 1 frames pass, 312^2 test
        crops): ONE launch at the start of the pass (sixteen 6-us launches otherwise).  The descriptor table and the output
        buffers are cached per geometry; the fp32 results (the backward's operand) exist only when the pass saves."""
        if not self.batch_interp:
            self._interp_out = {}
            return
        key = (tuple(thw), bool(save))
        ent = self._interp_cache.get(key)
        if ent is None:
            entries, outs, cur = [], {}, tuple(thw)
            for blk in self.plan.blocks:
                sq, skv = blk.stride_q[1], blk.stride_kv[1]
                q_thw = (cur[0], arch.pooled(cur[1], sq), arch.pooled(cur[2], sq))
                k_thw = (cur[0], arch.pooled(cur[1], skv), arch.pooled(cur[2], skv))
                _, _, mcat, _ = self._rel(blk, q_thw, k_thw)
                if mcat is not None:
                    r32 = torch.empty((mcat.shape[0], HD), device=self.dev, dtype=F32) if save else None
                    r16 = torch.empty((mcat.shape[0], HD), device=self.dev, dtype=BF16)
                    entries.append((mcat, self.flat.rel_tables32("blocks.%d." % blk.index), r32, r16))
                    outs[blk.index] = (r32, r16)
                cur = q_thw
            jobs = ops.table_interp_jobs(entries, self.dev) if entries else None
            ent = (jobs, max((e[0].shape[0] for e in entries), default=0), outs, entries)
            self._interp_cache[key] = ent
        jobs, max_rows, outs, _ = ent
        if jobs is not None:
            ops.table_interp_batched(jobs, max_rows)
        self._interp_out = outs

    def _relq_map(self, blk, q_thw, k_thw, idx, rows_off, n_obj, extra):
        """i32 [Nq, extra]: the column of P = q . Rcat^T that (token, j) reads, -1 for cls / object
        rows and for the padding columns j >= kh + kw + kt (svit_gemm_args.relq_map)."""
        key = (blk.index, q_thw, k_thw, tuple(rows_off), n_obj, extra)
        m = self._relq_cache.get(key)
        if m is None:
            qt, qh, qw = q_thw
            kt, kh, kw = k_thw
            ih, iw, it = (t.cpu().view(n, k) for t, n, k in zip(idx, (qh, qw, qt), (kh, kw, kt)))
            Lq = qt * qh * qw
            body = torch.full((qt, qh, qw, extra), -1, dtype=torch.int32)
            body[..., :kh] = (rows_off[0] + ih).view(1, qh, 1, kh)
            body[..., kh:kh + kw] = (rows_off[1] + iw).view(1, 1, qw, kw)
            body[..., kh + kw:kh + kw + kt] = (rows_off[2] + it).view(qt, 1, 1, kt)
            m = torch.full((1 + Lq + n_obj, extra), -1, dtype=torch.int32)
            m[1:1 + Lq] = body.view(Lq, extra)
            m = m.contiguous().to(self.dev)
            self._relq_cache[key] = m
        return m

    def _tables(self, pre, mats):
        names = (pre + "attn.rel_pos_h", pre + "attn.rel_pos_w", pre + "attn.rel_pos_t")
        out = []
        for n, m in zip(names, mats):
            t = self.flat.p(n)
            out.append(t if m is None else (m @ t).contiguous())
        return out

    # ------------------------------------------------------------------ forward ----------
    def forward(self, video, drop_scales=None, save=True):
        """video f32 [B,3,Tx,S,S] (or U8Clips) -> (normed tokens f32 [B,N_last,C_last], saved-state dict)."""
        plan, f = self.plan, self.flat
        if isinstance(video, U8Clips):       # decoded uint8 frames + crop table (svit_amd/input.py)
            B, _, Tx = video.shape[:3]
            cols, (To, Ho, Wo) = ops.im2col_patch_u8(video)
        else:
            if video.dim() == 4:
                video = video.unsqueeze(2)
            video = video.contiguous().to(F32)
            B, _, Tx = video.shape[:3]
            cols, (To, Ho, Wo) = ops.im2col_patch(video)
        T = plan.num_frames // plan.patch_stride[0] if Tx > 1 else Tx  # from cfg, builder:322
        if To != T:
            raise hip.SvitHipError("clip has %d frames but cfg.DATA.NUM_FRAMES=%d" % (Tx, plan.num_frames))
        L, n_obj, C = To * Ho * Wo, Tx * plan.objects, plan.embed_dim
        N = 1 + L + n_obj
        x = torch.empty((B, N, C), device=self.dev, dtype=F32)
        ops.gemm_nt(cols, self.patch_w16, f.p("patch_embed.proj.bias"), hip.EPI_F32,
                    out=x.view(B * N, C), remap=(L, N, 1))
        ops.fill_special_tokens(x, f.p("cls_token"), f.p("object_queries"),
                                f.p("pos_embed_temporal"), L, Tx, plan.objects, Tx > 1)
        st = {"B": B, "Tx": Tx, "n_obj": n_obj, "L0": L, "cols": cols if save else None,
              "blocks": []}
        thw = (T, Ho, Wo)
        self._interp_tables(thw, save)
        hip.mark("stem")
        for blk in plan.blocks:
            ds = drop_scales[blk.index] if drop_scales is not None else None
            x, thw, sv = self._block_fwd(blk, x, thw, n_obj, ds, save)
            st["blocks"].append(sv)
            hip.mark("fwd%d" % blk.index)
        y16, y32, mean, rstd = ops.layernorm_fwd(x, f.p("norm.weight"), f.p("norm.bias"),
                                                 want_f32=True, want_bf16=False, save_stats=save)
        st.update(x_last=x if save else None, mean=mean, rstd=rstd, thw=thw)
        # the batched interpolation results belong to THIS pass: a stand-alone _block_fwd afterwards (tests, tools) interpolates
        # its own tables.  The fp32 / bf16 buffers are cached per (geometry, save) and the saved `tabs` of a pass are views of
        # them: at most ONE saved forward per geometry may be awaiting its backward (the step's passes all differ in geometry
        # or in `save`; two saved forwards of one geometry with a weight update between them would share the operand).
        self._interp_out = {}
        return y32, st

    def _block_fwd(self, blk, x, thw, n_obj, ds, save):
        f = self.flat
        pre = "blocks.%d." % blk.index
        B, N, C = x.shape
        Co, h = blk.dim_out, blk.heads
        sq, skv = blk.stride_q[1], blk.stride_kv[1]
        q_thw = (thw[0], arch.pooled(thw[1], sq), arch.pooled(thw[2], sq))
        k_thw = (thw[0], arch.pooled(thw[1], skv), arch.pooled(thw[2], skv))
        Nq = 1 + q_thw[0] * q_thw[1] * q_thw[2] + n_obj
        J = k_thw[0] + k_thw[1] + k_thw[2]
        DA = HD + 32 if J <= 32 else HD + 64
        if J > 64:
            raise hip.SvitHipError("key grid %s too large for the in-MFMA rel-pos bias" % (k_thw,))
        xn, _, mean1, rstd1 = ops.layernorm_fwd(x, f.p(pre + "norm1.weight"), f.p(pre + "norm1.bias"),
                                                save_stats=save)
        xn2d = xn.view(B * N, C)
        qkv = ops.gemm_nt(xn2d, f.w(pre + "attn.qkv.weight"), f.p(pre + "attn.qkv.bias"), hip.EPI_BF16)
        idx, mats, mcat, need = self._rel(blk, q_thw, k_thw)
        # rel-pos query side: P = q . Rcat^T on the matrix pipe, then per (query, j) the table row it needs --
        # inside the slab LayerNorm kernel where the q tensor takes that path, else as one GEMM launch whose
        # epilogue does the gather (the pooling entry point adds it); P itself is never stored
        if mcat is None:
            tabs = self._tables(pre, mats)
            rcat, rows_off = f.rel_cat(pre)
        else:   # interpolated tables (odd crops, T=1 frames pass): one small launch (fp32 for the backward + the bf16 operand)
            if blk.index in self._interp_out:             # (computed for all blocks at the start of the pass)
                r32, rcat = self._interp_out[blk.index]
            else:
                r32, rcat = ops.table_interp(mcat, f.rel_tables32(pre), want_f32=save)
            rows_off = (0, need[0], need[0] + need[1])
            tabs = None if r32 is None else [r32[rows_off[0]:rows_off[0] + need[0]], r32[rows_off[1]:rows_off[1] + need[1]],
                                             r32[rows_off[2]:rows_off[2] + need[2]]]
        pools = ops.pool_ln_fwd_qkv(
            qkv, [f.p(pre + "attn.pool_%s.weight" % r).view(HD, 27) for r in "qkv"],
            [f.p(pre + "attn.norm_%s.weight" % r) for r in "qkv"],
            [f.p(pre + "attn.norm_%s.bias" % r) for r in "qkv"],
            B, h, thw, n_obj, (sq, skv, skv), (DA, DA, HD), (0, 1, 0), save=save, out_scales=(1.0, K_SCALE, 1.0),
            sels=f.sels(pre) if (thw[1] * thw[2] >= f.TILED_MIN_PLANE or thw[1] * thw[2] <= f.SLAB_MAX_PLANE)
            else None,
            relq=(rcat.contiguous(), self._relq_map(blk, q_thw, k_thw, idx, rows_off, n_obj, DA - HD), LOG2E))
        qa, ka, v = pools[0][0], pools[1][0], pools[2][0]
        ctx, lse2 = ops.attn_fwd(qa, ka, v, SCALE, bias_cols=J)
        pool_idx = None
        if blk.has_proj:
            skip = ops.gemm_nt(xn2d, f.w(pre + "proj.weight"), f.p(pre + "proj.bias"), hip.EPI_F32)
            skip = skip.view(B, N, Co)
        else:
            skip = x
        if blk.pools_q:
            skip, pool_idx = ops.maxpool_fwd(skip, thw, n_obj)
        dpa, dpm = ds if ds is not None else (None, None)
        x1 = ops.gemm_nt(ctx.view(B * Nq, Co), f.w(pre + "attn.proj.weight"), f.p(pre + "attn.proj.bias"),
                         hip.EPI_RESID, aux=skip.view(B * Nq, Co), row_scale=dpa, rows_per_sample=Nq)
        x1 = x1.view(B, Nq, Co)
        xn2, _, mean2, rstd2 = ops.layernorm_fwd(x1, f.p(pre + "norm2.weight"), f.p(pre + "norm2.bias"),
                                                 save_stats=save)
        act, dact = ops.gemm_nt(xn2.view(B * Nq, Co), f.w(pre + "mlp.fc1.weight"),
                                f.p(pre + "mlp.fc1.bias"), hip.EPI_GELU, save=save)
        x2 = ops.gemm_nt(act, f.w(pre + "mlp.fc2.weight"), f.p(pre + "mlp.fc2.bias"), hip.EPI_RESID,
                         aux=x1.view(B * Nq, Co), row_scale=dpm, rows_per_sample=Nq).view(B, Nq, Co)
        sv = None
        if save:
            sv = dict(x=x, thw=thw, q_thw=q_thw, k_thw=k_thw, mean1=mean1, rstd1=rstd1, xn=xn2d,
                      qkv=qkv, pools=pools, tabs=tabs, idx=idx, mats=mats, ctx=ctx, lse2=lse2,
                      pool_idx=pool_idx, x1=x1, mean2=mean2, rstd2=rstd2, xn2=xn2, act=act,
                      dact=dact, dpa=dpa, dpm=dpm, Nq=Nq)
        return x2, q_thw, sv

    # ------------------------------------------------------------------ backward ---------
    def backward(self, st, dy, on_ready=None, ready_ranks=None):
        """dy: grad of the normed tokens f32 [B,N_last,C_last]; accumulates every parameter
        gradient into flat.grad.  on_ready(rank) is called when the gradients of readiness rank
        `rank` (arch.readiness_rank) are final -- the data-parallel wrapper launches the
        all-reduce of that slice there, overlapping it with the remaining backward.
        ready_ranks: the ranks at which on_ready really consumes gradients (None = every rank);
        side-stream wgrad work is joined only there and at the end."""
        def ready(rank):
            if on_ready is not None:
                if ready_ranks is None or rank in ready_ranks:
                    ops.reduce_flush()          # gradients must be final here: queued second-stage reductions ...
                    self._flush_tn()            # ... and queued weight-gradient GEMMs
                    self._join()
                on_ready(rank)
        plan, f = self.plan, self.flat
        depth = len(plan.blocks)
        last = st["blocks"][depth - 1]
        dx, dx16 = ops.layernorm_bwd(dy.contiguous(), st["x_last"], f.p("norm.weight"), st["mean"],
                                     st["rstd"], f.g("norm.weight"), f.g("norm.bias"),
                                     want_bf16=True, row_scale=last["dpm"],
                                     rows_per_sample=st["x_last"].shape[1])
        ready(0)
        hip.mark("bwd_norm")
        for blk in reversed(plan.blocks):
            below = st["blocks"][blk.index - 1] if blk.index > 0 else None
            try:
                # second-stage reductions: queued per group of RED_GROUP blocks, one launch when the group's last block is done
                G = self.RED_GROUP
                dx, dx16 = self._block_bwd(blk, st["blocks"][blk.index], dx, dx16, st["n_obj"], below,
                                           red_begin=blk.index % G == G - 1 or blk.index == depth - 1,
                                           red_end=blk.index % G == 0, red_slot=blk.index % G)
            except BaseException:
                # a failed launch / allocation inside the bracket: the queued second-stage
                # reductions and weight-gradient GEMMs hold pointers into this step's scratch
                ops.reduce_reset()
                self._tn = []
                raise
            hip.mark("bwd%d" % blk.index)
            ready(1 + (depth - 1 - blk.index))
        # block-0 input: [cls | patches | objects]
        B, Tx, L, O, C = st["B"], st["Tx"], st["L0"], plan.objects, plan.embed_dim
        ops.special_token_grads(dx, f.g("cls_token"), f.g("object_queries"),
                                f.g("pos_embed_temporal") if Tx > 1 else None, L, Tx, O, Tx > 1)
        dtok = ops.scale_cast(dx, gather=(L, 1))       # patch rows of every clip -> bf16 [B*L, C]
        ops.gemm_tn(dtok, st["cols"], f.g("patch_embed.proj.weight").view(C, 441),
                    splits=1 if self.deterministic else 0, dbias=f.g("patch_embed.proj.bias"))
        self._flush_tn()
        self._join()
        ready(depth + 1)

    def _rws(self, key):
        a, b = self._red_regions[key]
        o = self._red_slot * 18 * 1024 * 1024
        return self._red_ws[o + a:o + b]

    def _wgrad_ws(self, B, h, thw, n_obj, strides):
        """workspace of the fused pooling backward's partial rows for this block: the slot's fixed 9 M-float region (sized for
        8 clips of 16x224^2), or -- when the batch outgrows it (round 6: the product library has no streaming fallback any
        more) -- a buffer of the plan's size kept per scratch slot (the rows are read by the group's DEFERRED reduce)."""
        key = (B, h, tuple(thw), n_obj, tuple(strides))
        need = self._wgrad_need.get(key)
        if need is None:
            need = self._wgrad_need[key] = ops.pool_conv_bwd_workspace(B, h, thw, n_obj, strides)
        ws = self._rws("wgrad")
        if need <= ws.numel():
            return ws
        big = self._wgrad_big.get(self._red_slot)
        if big is None or big.numel() < need:
            big = self._wgrad_big[self._red_slot] = torch.empty(need, device=self.dev, dtype=F32)
        return big

    def _fork(self, fn, keep):
        """run fn() on the side stream, after everything enqueued so far on the current one"""
        if not self.overlap_wgrad:
            fn()
            return
        if self._capture_fork is not None:      # graph capture: becomes a side-stream replay item
            self._capture_fork(fn)
            self._side_active = True
            return
        main = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            fn()
        self._side_keep.append(keep)     # operands stay allocated until the join
        self._side_active = True

    def _join(self):
        if self._side_active and self._capture_join is not None:
            self._capture_join()
            self._side_active = False
            return
        if self._side_active:
            torch.cuda.current_stream(self.dev).wait_stream(self._side)
            self._side_keep.clear()
            self._side_active = False

    def _flush_tn(self, force=True):
        """queued weight-gradient GEMMs in one grouped launch (ops.gemm_tn_grouped).  Two blocks
        share a launch (<= 16 problems): the machine-wide accumulator flush is paid half as often."""
        if not force and len(self._tn) <= 8:
            return
        q, self._tn = self._tn, []
        if not q:
            return
        det = self.deterministic
        self._fork(lambda: ops.gemm_tn_grouped(q, ordered=det), q)

    def _linear_bwd(self, dy16, x16, wname, bname, need_dx, out=None, accumulate=False,
                    epilogue=hip.EPI_F32, aux=None):
        f = self.flat
        self._tn.append((dy16, x16, f.g(wname), f.g(bname)))     # flushed per block, grouped
        if not need_dx:
            return None
        return ops.gemm_nt(dy16, f.wt(wname), None, epilogue, out=out, aux=aux, accumulate=accumulate)

    def _block_bwd(self, blk, sv, dx2, dy, n_obj, below, red_begin=True, red_end=True, red_slot=0):
        """dx2: f32 grad of the block output; dy: bf16(DropPath_mlp * dx2) from the producing
        LayerNorm backward; below: saved state of block index-1 (its MLP DropPath scales the bf16
        copy of this block's input grad) or None for block 0."""
        f = self.flat
        pre = "blocks.%d." % blk.index
        B, Nq, Co = dx2.shape
        x = sv["x"]
        N, C, h = x.shape[1], x.shape[2], blk.heads
        M, Mq = B * N, B * Nq
        sq, skv = blk.stride_q[1], blk.stride_kv[1]
        thw, q_thw, k_thw = sv["thw"], sv["q_thw"], sv["k_thw"]
        # this block's second-stage reductions are queued (red_begin opens the queue, red_end runs it: backward() keeps
        # a group of RED_GROUP blocks in one queue, each block on its own scratch slot; a stand-alone call does both)
        self._red_slot = red_slot
        if red_begin:
            ops.reduce_defer(True)
        # ---- MLP branch: x2 = x1 + dp * fc2(gelu(fc1(LN2(x1)))) ------------------------------
        dy = dy.view(Mq, Co)
        dh = self._linear_bwd(dy, sv["act"], pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", True,
                              epilogue=hip.EPI_DGELU, aux=sv["dact"])
        dxn2 = self._linear_bwd(dh, sv["xn2"].view(Mq, Co), pre + "mlp.fc1.weight",
                                pre + "mlp.fc1.bias", True, epilogue=hip.EPI_BF16)   # bf16 like autocast
        dx1, dy = ops.layernorm_bwd(dxn2, sv["x1"], f.p(pre + "norm2.weight"), sv["mean2"],
                                    sv["rstd2"], f.g(pre + "norm2.weight"), f.g(pre + "norm2.bias"),
                                    dres=dx2, want_bf16=True, row_scale=sv["dpa"], rows_per_sample=Nq,
                                    ws=self._rws("ln2"))
        dy = dy.view(Mq, Co)
        # ---- attention branch: x1 = skip + dp * proj(ctx) ------------------------------------
        dctx = self._linear_bwd(dy, sv["ctx"].view(Mq, Co), pre + "attn.proj.weight",
                                pre + "attn.proj.bias", True, epilogue=hip.EPI_BF16)
        (qa, preq, mq, rq), (ka, prek, mk, rk), (v, prev, mv, rv) = sv["pools"]
        # rel-pos backward as GEMMs over the scattered matrix D [tokens, Lpad]
        tabs, mats = sv["tabs"], sv["mats"]
        names = (pre + "attn.rel_pos_h", pre + "attn.rel_pos_w", pre + "attn.rel_pos_t")
        if all(m is None for m in mats):
            rcat_t, lpad, offs = f.rel_cat_t(pre)
        else:   # interpolated tables (odd crops, T=1 frames pass): tiny torch plumbing
            offs, lpad = rel_sections([t.shape[0] for t in tabs])
            rcat = torch.zeros((lpad, HD), device=self.dev, dtype=F32)
            for o, t in zip(offs, tabs):
                rcat[o:o + t.shape[0]] = t
            rcat_t = rcat.t().contiguous().to(BF16)
        # the dq kernel writes D itself (its epilogue holds d(relq) of whole rows): no scatter launch
        if self.fused_scatter:
            dmap = self._relq_map(blk, q_thw, k_thw, sv["idx"], offs, n_obj, qa.shape[-1] - HD)
            dqa, dk, dv, D, dq_extra = ops.attn_bwd(
                qa, ka, v, sv["ctx"], dctx.view(B, Nq, Co), sv["lse2"], SCALE,
                q_splits=1 if self.deterministic else self.attn_q_splits, bias_cols=sum(sv["k_thw"]),
                reld=(dmap, lpad, LOG2E, rcat_t if rcat_t.is_contiguous() else None, "fold"))
        else:   # (measurements only: engine.fused_scatter = False)
            dqa, dk, dv = ops.attn_bwd(qa, ka, v, sv["ctx"], dctx.view(B, Nq, Co), sv["lse2"], SCALE,
                                       q_splits=1 if self.deterministic else self.attn_q_splits,
                                       bias_cols=sum(sv["k_thw"]))
            D = ops.relpos_scatter(dqa, sv["idx"], offs, lpad, B, h, q_thw, k_thw, n_obj, LOG2E)
            dq_extra = None
        qa2 = qa.view(B * h * Nq, qa.shape[-1])
        for n, m, t, o in zip(names, mats, tabs, offs):
            rows = t.shape[0]
            if m is None:
                self._tn.append((D[:, o:o + rows], qa2[:, :HD], f.g(n), None))
            else:
                d = torch.zeros_like(t)
                ops.gemm_tn(D[:, o:o + rows], qa2[:, :HD], d, splits=1 if self.deterministic else 0)
                f.g(n).add_(m.t() @ d)
        if dq_extra is None:      # wide tables (56x56 / 28x28 stages) or the A/B path: dq = D R as its own GEMM
            # (bf16 epilogue since round 5: the pooled-LN backward adds it to two other bf16 addends -- dq of the attention
            #  kernel, the residual-pooling dctx -- and reads half the bytes: 77 MB less each way at block 0)
            dq_extra = ops.gemm_nt(D, rcat_t, None, hip.EPI_BF16)
        elif isinstance(dq_extra, str):     # "folded": dqa[..., :96] already holds it (narrow tables)
            dq_extra = None
        Nk = ka.shape[2]
        dqkv = torch.empty_like(sv["qkv"])
        entries = []
        for r, pre_t, mean, rstd, nout, kw in (
                ("q", preq, mq, rq, Nq, dict(d_main=dqa, ld_main=qa.shape[-1], d_res=dctx,
                                             d_extra=dq_extra)),
                ("k", prek, mk, rk, Nk, dict(d_main=dk, ld_main=HD)),
                ("v", prev, mv, rv, Nk, dict(d_main=dv, ld_main=HD))):
            entries.append(((pre_t, mean, rstd, f.p(pre + "attn.norm_%s.weight" % r),
                             f.g(pre + "attn.norm_%s.weight" % r),
                             f.g(pre + "attn.norm_%s.bias" % r), B, h, nout), kw))
        dpres = ops.pool_ln_bwd_qkv(entries, ws=self._rws("pln"))   # q, k, v: one launch per stage
        strides = (sq, skv, skv)
        # conv dgrad + conv wgrad: one kernel with dpre in LDS for the small planes (blocks >= 4),
        # the two streaming launches otherwise (decided inside the library)
        ops.pool_conv_bwd_qkv(dpres, [f.p(pre + "attn.pool_%s.weight" % r).view(HD, 27) for r in "qkv"],
                              dqkv, sv["qkv"],
                              [f.g(pre + "attn.pool_%s.weight" % r).view(HD, 27) for r in "qkv"],
                              B, h, thw, n_obj, strides, ws=self._wgrad_ws(B, h, thw, n_obj, strides))
        # (bf16 unless the dim-change projection accumulates into it below)
        dxn = self._linear_bwd(dqkv, sv["xn"], pre + "attn.qkv.weight", pre + "attn.qkv.bias", True,
                               epilogue=hip.EPI_F32 if blk.has_proj else hip.EPI_BF16)
        # ---- skip path ------------------------------------------------------------------------
        dskip = dx1
        if blk.pools_q:       # (bf16 straight away when only the projection's backward GEMMs read it)
            dskip = ops.maxpool_bwd(dskip, sv["pool_idx"], thw, n_obj, bf16=blk.has_proj)
        if blk.has_proj:
            ds16 = dskip.view(M, Co) if blk.pools_q else ops.scale_cast(dskip.view(M, Co))
            self._linear_bwd(ds16, sv["xn"], pre + "proj.weight", pre + "proj.bias", True, out=dxn,
                             accumulate=True)
            dskip = None
        res = ops.layernorm_bwd(dxn, x, f.p(pre + "norm1.weight"), sv["mean1"], sv["rstd1"],
                                f.g(pre + "norm1.weight"), f.g(pre + "norm1.bias"), dres=dskip,
                                want_bf16=below is not None,
                                row_scale=below["dpm"] if below is not None else None,
                                rows_per_sample=N, ws=self._rws("ln1"))
        if red_end:
            ops.reduce_defer(False)     # runs the queued reductions (of the group)
        self._flush_tn(force=False)
        if below is None:
            return res.view(B, N, C), None
        return res[0].view(B, N, C), res[1]
