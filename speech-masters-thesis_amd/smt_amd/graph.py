"""hipGraph capture of a train step's forward + backward (torch.cuda.CUDAGraph = hipGraph on ROCm).

A train step of this build is ~1,000 launches.  For the TransformerLM at the reference's batch (8 x 258 tokens) the GPU
needs ~8 ms for them and a Python interpreter about as long to issue them, so the step follows whichever is slower on a
given host; for the VQ-VAE step (63 ms of kernels) the host matters less, but every synchronisation point restarts the
queue from empty.  One captured graph of forward + backward takes the host out of the loop.

What capture needs from a model: static input buffers; no host synchronisation inside forward / backward; and dropout keys
that live in DEVICE memory (``model.enable_device_keys()``): a graph freezes by-value kernel arguments, so by-value keys
would replay the same masks for ever.  Packed-weight copies (smt_amd.convops._PackCache) are refreshed by a host-side
version check in eager mode; a graph cannot check, so the batched repack is captured at the top of the graph and simply
runs every replay (one launch).  Optimizer, scheduler, NaN guard, gradient clipping and the parameter EMA stay eager.
"""
import torch

from . import convops


class GraphedStep:
    """``fn(*static_inputs) -> (loss_dict, metrics_dict)`` and ``loss_dict["loss"].backward()`` as one graph.

    ``replay(*inputs)`` copies the inputs into the static buffers, replays, and returns the (static) result tensors; the
    parameters' ``.grad`` tensors are static too, so an eager optimizer can read them afterwards.  Results are those of the
    eager step: same kernels, same masks (the host counter ``model._drop_seed`` is advanced alongside the device one).

    The caller must not keep a loss (or any tensor with a grad_fn) of an EARLIER eager iteration alive: that autograd graph
    pins the parameters' AccumulateGrad nodes, which live on the default stream; the capture then has to synchronise with
    the default stream, and hipStreamEndCapture crashes (torch warns "The AccumulateGrad node's stream does not match")."""

    def __init__(self, model, fn, inputs, zero_grad, warmup=3):
        assert model.training and all(t is None or t.is_cuda for t in inputs)
        import gc
        gc.collect()                           # drop unreachable autograd graphs of earlier iterations (see above)
        self.model, self.fn = model, fn
        model.enable_device_keys(True)
        self.static = [None if t is None else t.clone() for t in inputs]
        device = next(t for t in self.static if t is not None).device
        side, main = torch.cuda.Stream(device=device), torch.cuda.current_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):          # eager warm-up off the capture stream: allocations, handles, workspaces
            for _ in range(warmup):
                zero_grad()
                loss_dict, _ = fn(*self.static)
                loss_dict["loss"].backward()
                del loss_dict, _               # no autograd graph of a warm-up pass may outlive it (see the class docstring)
            if convops._pack_cache.entries:
                convops._pack_cache.repack_all()   # builds the device-side pack table the captured launch will reuse
        main.wait_stream(side)
        zero_grad()                            # set_to_none: backward inside the capture allocates .grad from the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            if convops._pack_cache.entries:
                convops._pack_cache.repack_all()
            self.loss_dict, self.metrics = fn(*self.static)
            self.loss_dict["loss"].backward()
        model._drop_seed -= 1                  # capture ran the Python side of forward once without executing the device increment
        self.loss_dict = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in self.loss_dict.items()}

    def replay(self, *inputs):
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
