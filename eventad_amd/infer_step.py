"""Evaluation forward of EventADModel as ONE replayed hipGraph -- the serving-side analog of TrainStep.capture.

The reference's test loop calls `model(data, labels, testing=True)` per batch (utils/test.py:49-60, utils/fps.py:50-67): a few
hundred small dependent launches, whose issue and hand-over times are what a small batch waits for (batch 6 at 256x256 with
the live trunk: 2.9 ms eager for ~1.5 ms of kernels).  Nothing in the forward synchronises with the host, so for batches of
one shape (events N, boxes M, clips B: a shape bucket) it is captured once and replayed; a new batch is copied into the
graph's static input buffers by one launch in front of the replay.  Results are the padded device tables of
`EventADModel.predict_tensors` (logits f32[B,31,2], valid bool[B,31], targets i64[B,31]) in static buffers: valid until the
next replay.  Bitwise equal to the eager forward (tests/test_gpu_model.py)."""
import torch

from .train_step import TrainStep


class InferStep:
    def __init__(self, model):
        if next(model.gru_net_event.parameters()).device.type != "cuda":
            raise RuntimeError("InferStep: the forward pass runs on the GPU only (no CPU fallback); move the model to cuda")
        self.model = model
        self.graph, self.inputs, self.out = None, None, None
        self._retired = []                # superseded graphs stay alive: buffers first allocated under them stay mapped

    def _forward(self):
        return self.model.predict_tensors(self.inputs, self.inputs.bbox[:, 4])

    def capture(self, data, warmup=2):
        """data: a formatted batch on the device; every later batch must have its shapes."""
        self.model.eval()
        self.inputs = data.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._forward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the graph holds raw pointers into the native workspaces: one that has to grow later is retired, not freed
        nb = getattr(self.model.dagr_model.backbone, "_native", None)
        if nb is not None:
            nb.keep_superseded_workspaces = True
        self.model.native_head().keep_superseded_workspaces = True
        if self.graph is not None:
            self._retired.append(self.graph)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._forward()
        return self.graph

    def replay(self, batch=None):
        """batch: formatted, same shapes as the captured one (device or pinned host memory); None = the batch the buffers
        hold.  -> (logits, valid, targets), static device tensors."""
        if self.model.training:
            raise RuntimeError("InferStep.replay: the model was switched to training mode after the capture")
        if batch is not None:
            TrainStep.load_into(self.inputs, batch)
        self.graph.replay()
        return self.out
