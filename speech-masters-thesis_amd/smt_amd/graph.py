"""hipGraph capture of a train step's forward + backward (torch.cuda.CUDAGraph = hipGraph on ROCm).

A train step of this build is ~1,000 launches.  For the TransformerLM at the reference's batch (8 x 258 tokens) the GPU
needs ~8 ms for them and a Python interpreter about as long to issue them, so the step follows whichever is slower on a
given host; for the VQ-VAE step (63 ms of kernels) the host matters less, but every synchronisation point restarts the
queue from empty.  One captured graph of forward + backward takes the host out of the loop.

What capture needs from a model: static input buffers; no host synchronisation inside forward / backward; and dropout keys
that live in DEVICE memory (``model.enable_device_keys()``): a graph freezes by-value kernel arguments, so by-value keys
would replay the same masks for ever.  Packed-weight copies (smt_amd.convops._PackCache) are refreshed by the first conv of
every training forward (one batched launch); that launch is captured with the rest and simply runs every replay.  Optimizer, scheduler, NaN guard, gradient clipping and the parameter EMA stay eager.
"""
import torch

from . import convops


def stale_autograd_graphs(params):
    """Names / indices of the parameters whose AccumulateGrad node is being kept alive by an autograd graph of an EARLIER
    iteration (a loss or an output with a grad_fn that the caller still holds).  Such a node stays pinned to the stream it
    was created on -- usually the default stream -- and a backward captured on another stream then drags the default stream
    into the capture: torch only warns ("The AccumulateGrad node's stream does not match ..."), hipStreamEndCapture
    segfaults (gpurun_out/amdlog.txt of round 2).  The node of a parameter is owned by the graphs that point at it and by
    nothing else, so: tag it through a throw-away graph, drop that graph, look again -- a tag that survived means somebody
    else holds the node.  No kernel runs and no model state moves."""
    token = object()
    stale = []
    for i, p in enumerate(params):
        if not p.requires_grad:
            continue
        p.expand_as(p).grad_fn.next_functions[0][0].metadata["smt_graph_probe"] = token
        if p.expand_as(p).grad_fn.next_functions[0][0].metadata.get("smt_graph_probe") is token:
            stale.append(i)
            del p.expand_as(p).grad_fn.next_functions[0][0].metadata["smt_graph_probe"]
    return stale


class GraphedStep:
    """``fn(*static_inputs) -> (loss_dict, metrics_dict)`` and ``loss_dict["loss"].backward()`` as one graph.

    ``replay(*inputs)`` copies the inputs into the static buffers, replays, and returns the (static) result tensors; the
    parameters' ``.grad`` tensors are static too, so an eager optimizer can read them afterwards.  Results are those of the
    eager step: same kernels, same masks (the host counter ``model._drop_seed`` is advanced alongside the device one).

    The caller must not keep a loss (or any tensor with a grad_fn) of an EARLIER eager iteration alive: that autograd graph
    pins the parameters' AccumulateGrad nodes to the stream they were created on, and a capture would crash inside
    hipStreamEndCapture.  The constructor checks (``stale_autograd_graphs``) and raises ``RuntimeError`` instead; it also
    turns torch's own warning about it, should it fire during the eager warm-up, into the same error.

    The graph holds raw pointers into the packed conv-weight copies of ``smt_amd.convops``: the entries captured are kept
    alive by this object, and ``replay`` refuses to run once they have been thrown away (``EMA.swap`` around validation,
    ``load_checkpoint``: ``convops.invalidate_packed_weights``) -- build a new GraphedStep afterwards."""

    def __init__(self, model, fn, inputs, zero_grad, warmup=3):
        assert model.training and all(t is None or t.is_cuda for t in inputs)
        import gc
        import warnings
        gc.collect()                           # drop unreachable autograd graphs of earlier iterations (see above)
        convops._pack_cache.prune()            # ... and the packed copies of models that no longer exist
        stale = stale_autograd_graphs(list(model.parameters()))
        if stale:
            raise RuntimeError(
                f"GraphedStep: {len(stale)} parameter(s) are still referenced by the autograd graph of an earlier iteration "
                "(a loss or output tensor with a grad_fn is alive); capturing a backward now would crash inside "
                "hipStreamEndCapture.  Delete those tensors (or .detach() what you keep) before building a GraphedStep.")
        self.model, self.fn = model, fn
        model.enable_device_keys(True)
        self.static = [None if t is None else t.clone() for t in inputs]
        device = next(t for t in self.static if t is not None).device
        side, main = torch.cuda.Stream(device=device), torch.cuda.current_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(warmup):            # eager warm-up off the capture stream: allocations, handles, workspaces
                zero_grad()
                loss_dict, _ = fn(*self.static)
                loss_dict["loss"].backward()
                del loss_dict, _               # no autograd graph of a warm-up pass may outlive it (see the class docstring)
            if convops._pack_cache.entries:
                convops._pack_cache.repack_all()   # builds the device-side pack table the captured launch will reuse
        main.wait_stream(side)
        mismatch = [w for w in caught if "AccumulateGrad node's stream does not match" in str(w.message)]
        for w in caught:
            if w not in mismatch:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if mismatch:
            raise RuntimeError("GraphedStep: " + str(mismatch[0].message))
        zero_grad()                            # set_to_none: backward inside the capture allocates .grad from the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        repacks = convops._pack_cache.repacks
        with torch.cuda.graph(self.graph):
            # a training forward of the VQ-VAE repacks the conv weights at its first conv (convops.training_forward): that
            # launch is captured here and runs on every replay; models without packed weights capture nothing for it
            self.loss_dict, self.metrics = fn(*self.static)
            self.loss_dict["loss"].backward()
        assert convops._pack_cache.repacks - repacks <= 1, "one batched repack per captured step"
        # everything the captured launches point at must outlive the graph: the packed copies and their device-side table
        self._pack_refs = (list(convops._pack_cache.entries.values()), convops._pack_cache.table)
        self._pack_epoch = convops.pack_epoch()
        model._drop_seed -= 1                  # capture ran the Python side of forward once without executing the device increment
        self.loss_dict = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in self.loss_dict.items()}

    def replay(self, *inputs):
        if convops.pack_epoch() != self._pack_epoch:
            raise RuntimeError("GraphedStep.replay: the packed conv-weight copies this graph was captured with have been "
                               "invalidated since (EMA.swap / load_checkpoint / invalidate_packed_weights); capture a new "
                               "GraphedStep")
        for dst, src in zip(self.static, inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        self.model._drop_seed += 1             # mirrors the device counter the graph advances
        self.graph.replay()
        return self.loss_dict, self.metrics


class GraphedTrainStep:
    """TransformerLM convenience wrapper: ``step(x, lens)`` = graphed forward + backward, eager optimizer + scheduler."""

    def __init__(self, model, optimizer, scheduler, x, lens, warmup=3):
        self.optimizer, self.scheduler = optimizer, scheduler
        self.core = GraphedStep(model, lambda a, b: model(a, b, None, None), [x, lens],
                                lambda: optimizer.zero_grad(set_to_none=True), warmup)

    def step(self, x, lens):
        loss_dict, _ = self.core.replay(x, lens)
        self.optimizer.step()
        self.scheduler.step()
        return loss_dict["loss"]
