"""HIP-graph replay of the SViT training step (forward + loss + backward).

The eager path (svit_amd/model.py) issues ~750 kernel launches per step from Python; most of
them run for 5-50 us on an MI355X, so the host is as busy as the GPU and every hiccup on the
host shows up as an idle GPU (profiles/README.md: ~13 % idle at B=8).  The launch schedule is
static for a fixed input shape, so it is captured ONCE into HIP graphs and replayed:

    step = GraphedTrainStep(model, loss_fun, inputs, labels)
    for inputs, labels in loader:
        loss, preds = step(inputs, labels)      # static tensors, overwritten by every replay
        optimizer.step()                        # eager (lr / bias correction are host scalars)

What is inside the graph(s): bf16 weight refresh, grad-buffer memset, DropPath / dropout sampling
(torch's graph-safe Philox state), every backbone kernel, the head, `loss_fun`, the head's
autograd backward and the whole backbone backward.  Same arithmetic as the eager step
(tools/train_net.py:88-151 of the reference is the loop it replaces).

The capture is cut into segments wherever work leaves the main chain: (a) the engine's
grouped weight-gradient GEMM launches are replayed eagerly on a side stream next to the following segments -- inside one hipGraph the same fork/join ran slower
than serial on ROCm 7.2; (b) data parallel: after the segment that finalises a gradient bucket
the wrapper's `_on_ready` hook launches that bucket's all-reduce on RCCL's stream, overlapping
the next segment exactly like in the eager path (svit_amd/dp.py) -- no collective lives inside a
graph.
"""
import gc
import warnings

import torch

from . import hip


class GraphedTrainStep:
    def __init__(self, model, loss_fun, inputs, labels, warmup=2, frames_pass=False):
        """model: SViT or its DataParallel wrapper (train mode); loss_fun(preds, extra, labels) ->
        scalar; inputs: the reference's `inputs` list ([video f32 [B,3,T,S,S]]); labels: any
        tensor (or tuple/dict of tensors) `loss_fun` takes -- copied into static buffers.
        frames_pass: also run the reference's no-grad single-frame forward of every clip
        (tools/train_net.py:105-110) inside the graph; its outputs reach `loss_fun` as
        extra["frames_output"] = {"preds", "extra_preds"} (the consistency-loss operand)."""
        self.frames_pass = frames_pass
        self.wrapper = model
        self.core = model.module if hasattr(model, "module") else model
        if self.core.engine is None:
            raise hip.SvitHipError("GraphedTrainStep needs a finalized (on-GPU) SViT")
        if not self.core.training:
            raise RuntimeError("GraphedTrainStep captures the training step: call model.train() first")
        self.loss_fun = loss_fun
        self.dp = model if hasattr(model, "_on_ready") and (getattr(model, "world_size", 1) > 1 or
                                                           getattr(model, "force_collectives", False)) else None
        self.x = inputs[0].detach().clone().contiguous()
        self.labels = _tree_map(lambda t: t.detach().clone(), labels)
        self.segments = []          # replay items: ("graph", CUDAGraph) | ("side", fn) | ("join", None) | ("ready", ranks)
        self.loss = self.preds = self.extra = None
        self._keepalive = None
        self._capture(warmup)

    # ------------------------------------------------------------------------------------------
    def _body(self, boundary):
        core = self.core
        eng, flat = core.engine, core.flat
        x = self.x
        Tx = x.shape[2] if x.dim() == 5 else 1
        eng.refresh_weights()
        flat.grad.zero_()
        ds = core.sample_drop_scales(x.shape[0], x.device, Tx=Tx)      # (+ the head's dropout factors: one launch)
        head_keep, core._head_keep = core._head_keep, None
        with torch.no_grad():
            y, st = eng.forward(x, ds, save=True)
        n_obj = Tx * core.O
        frames_out = None
        if self.frames_pass and Tx > 1:
            with torch.no_grad():       # B*T single frames through the same kernels (T' = 1)
                xf = x.transpose(1, 2).flatten(0, 1).unsqueeze(2)
                fy, _ = eng.forward(xf, core.sample_drop_scales(xf.shape[0], x.device), save=False)
                ffeat = torch.cat((fy[:, :1], fy[:, -core.O:]), dim=1)
                fp, fe = core.head(ffeat, T=1)
                frames_out = {"preds": fp, "extra_preds": fe}
        if core.fused_head:
            # the head as one launch each way (csrc/head.hip): its backward writes the head's parameter
            # gradients straight into the flat buffer's views and returns d(tokens) whole -- no AccumulateGrad
            # node of a real parameter is involved (those may be pinned to another stream by an earlier eager
            # step, which a stream capture cannot follow), no slice / cat plumbing either
            yt = y.detach().requires_grad_(True)
            with torch.enable_grad():
                preds, extra = core.head_train(yt, Tx, dropout_keep=head_keep)
                if frames_out is not None:
                    extra = dict(extra)
                    extra["frames_output"] = frames_out
                loss = self.loss_fun(preds, extra, self.labels)
            dy, = torch.autograd.grad(loss, [yt])
            feat = yt
        else:
            feat = torch.cat((y[:, :1], y[:, -n_obj:]), dim=1).requires_grad_(True)
            # the head runs on fresh leaf aliases of its parameters and explicit autograd.grad instead
            # of loss.backward(): AccumulateGrad nodes of the real parameters may be pinned to another
            # stream by an earlier eager step, which a stream capture cannot follow
            named = list(core.head.named_parameters())
            alias = {n: p.detach().requires_grad_(True) for n, p in named}
            with torch.enable_grad():
                preds, extra = torch.func.functional_call(core.head, alias, (feat,), {"T": Tx})
                if frames_out is not None:
                    extra = dict(extra)
                    extra["frames_output"] = frames_out
                loss = self.loss_fun(preds, extra, self.labels)
            grads = torch.autograd.grad(loss, [feat] + [alias[n] for n, _ in named], allow_unused=True)
            with torch.no_grad():
                for (n, p), g in zip(named, grads[1:]):
                    if g is not None:
                        p.grad.add_(g)
                dfeat = grads[0]
                dy = torch.zeros_like(y)
                dy[:, :1] = dfeat[:, :1]
                dy[:, -n_obj:] = dfeat[:, 1:]
        with torch.no_grad():
            eng.backward(st, dy, on_ready=boundary,
                         ready_ranks=self.dp.launch_ranks() if self.dp is not None else None)
        return (loss.detach(), preds.detach(),
                {k: v.detach() for k, v in extra.items() if torch.is_tensor(v)}, (st, feat, dy))

    def _capture(self, warmup):
        core = self.core
        eng = core.engine
        core._attach_grads()                      # .grad views of the flat buffer, host side only
        dev = self.x.device
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):
            for _ in range(max(1, warmup)):       # lazy tables / library workspaces materialise here
                self._body(None)
        cap.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        pool = torch.cuda.graph_pool_handle()
        cuts = self.dp.launch_ranks() if self.dp is not None else set()
        # "thread_local": the process group's watchdog thread may query events while we capture
        mode = "thread_local"
        state = {"g": None}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(pool=pool, capture_error_mode=mode)

        def end():
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                state["g"].capture_end()
            if not any("Graph is empty" in str(x.message) for x in w):
                self.segments.append(("graph", state["g"]))

        def cut(item):
            """close the running capture, queue `item` for replay, open the next capture"""
            end()
            self.segments.append(item)
            begin()

        seen = []

        def boundary(rank):
            seen.append(rank)
            if rank in cuts:
                cut(("ready", list(seen)))
                seen.clear()

        eng._capture_fork = lambda fn: cut(("side", fn))
        eng._capture_join = lambda: cut(("join", None))
        try:
            with torch.cuda.stream(cap):
                begin()
                try:
                    self.loss, self.preds, self.extra, self._keepalive = self._body(
                        boundary if self.dp is not None else None)
                finally:
                    end()
        finally:
            eng._capture_fork = eng._capture_join = None
        torch.cuda.current_stream(dev).wait_stream(cap)
        torch.cuda.synchronize(dev)
        self._side = torch.cuda.Stream(device=dev)

    # ------------------------------------------------------------------------------------------
    def __call__(self, inputs, labels):
        x = inputs[0]
        if x.shape != self.x.shape:
            raise hip.SvitHipError("GraphedTrainStep was captured for input %s, got %s"
                                   % (tuple(self.x.shape), tuple(x.shape)))
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        _tree_copy(self.labels, labels)
        main = torch.cuda.current_stream(self.x.device)
        for kind, val in self.segments:
            if kind == "graph":
                val.replay()
            elif kind == "side":        # parameter-gradient work next to the following segments
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    val()
            elif kind == "join":
                main.wait_stream(self._side)
            else:                       # "ready": this bucket's gradients are final -> all-reduce
                for r in val:
                    self.dp._on_ready(r)
        return self.loss, (self.preds, self.extra)

    @property
    def static_inputs(self):
        """the buffers the captured step reads: a loader that makes them the TARGET of its host-to-device copy and
        passes them back to __call__ saves the device-to-device copy of every step (77 MB at B = 8, 16x224^2 fp32)"""
        return [self.x]

    @property
    def static_labels(self):
        return self.labels

    @property
    def n_graphs(self):
        return sum(1 for k, _ in self.segments if k == "graph")


def _tree_map(fn, obj):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _tree_map(fn, v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_map(fn, v) for v in obj)
    return obj


def _tree_copy(dst, src):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():     # (the static buffer itself was handed back: nothing to copy)
            dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _tree_copy(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _tree_copy(d, s)
