"""hipGraph capture of a train step (torch.cuda.CUDAGraph = hipGraph on ROCm).

The TransformerLM step is ~1,000 short launches; at the reference's batch (8 x 258 tokens) the GPU needs ~9 ms for them
and a Python interpreter about as long to issue them, so the step time follows whichever is slower on a given host.  One
captured graph of forward + backward removes the host from the loop.  What capture needs from the model: static input
buffers, no host synchronisation inside forward / backward (there is none), and dropout keys that live in DEVICE memory
(`TransformerLM.enable_device_keys`): a graph freezes by-value kernel arguments, so by-value keys would replay the same
masks for ever.  The optimizer and the scheduler stay eager (three launches).
"""
import torch


class GraphedTrainStep:
    """forward + backward of ``model(x, lens, None, None)`` as one graph; ``step(x, lens)`` copies the batch into the
    static buffers, replays, and runs optimizer / scheduler.  Losses and masks are bit-identical to the eager step."""

    def __init__(self, model, optimizer, scheduler, x, lens, warmup=3):
        assert model.training and x.is_cuda
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        model.enable_device_keys(True)
        self.x, self.lens = x.clone(), lens.clone()
        side, main = torch.cuda.Stream(device=x.device), torch.cuda.current_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):          # eager warm-up off the capture stream: allocations, handles, workspaces
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                out, _ = model(self.x, self.lens, None, None)
                out["loss"].backward()
        main.wait_stream(side)
        optimizer.zero_grad(set_to_none=True)  # backward inside the capture allocates the .grad tensors from the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out, metrics = model(self.x, self.lens, None, None)
            out["loss"].backward()
        model._drop_seed -= 1                  # capture ran the Python side of forward once without executing the device increment
        self.loss, self.accuracy = out["loss"].detach(), metrics["accuracy"]

    def step(self, x, lens):
        self.x.copy_(x, non_blocking=True)
        self.lens.copy_(lens, non_blocking=True)
        self.model._drop_seed += 1             # mirrors the device counter the graph advances
        self.graph.replay()
        self.optimizer.step()
        self.scheduler.step()
        return self.loss
