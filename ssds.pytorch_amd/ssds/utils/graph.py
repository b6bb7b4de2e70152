"""hipGraph-captured inference: forward + decode + NMS of a detector recorded ONCE and replayed with a single
graph launch per batch (BASELINE config 5: "hipGraph-captured inference").

Everything on the hot path already runs without host synchronisation or allocation on the caller's stream (the
recorded plan is one C call, decode + NMS another, both enqueue kernels only), which is what makes the whole step
capturable: after capture a step is ``copy input -> hipGraphLaunch``.  ``torch.cuda.CUDAGraph`` is the hipGraph
binding on ROCm; the head outputs and the decoder's scratch are allocated from the graph's private pool during
capture and stay valid across replays."""
import torch


class GraphedInference(object):
    r"""Args:
        model:    detector in eval mode on a HIP device (``model(x) -> (loc, conf)``)
        decoder:  ``ssds.modeling.layers.decoder.Decoder``
        anchors:  OrderedDict{stride: [A,4]} from ``model_builder.create_anchors``
        example:  a batch with the shape / dtype / memory format every later batch will have
    ``__call__(x)`` copies ``x`` into the static input, replays the graph and returns the static output tensors
    (scores [B,D], boxes [B,D,4], classes [B,D]); they are overwritten by the next call."""

    def __init__(self, model, decoder, anchors, example, warmup=3):
        if model.training:
            raise ValueError("GraphedInference needs model.eval()")
        if not example.is_cuda:
            raise ValueError("GraphedInference needs a HIP device tensor")
        if getattr(decoder, "_tail", None) is not None:
            # a capture must end with every forked stream joined; the tail stream is a cross-step overlap by design
            raise ValueError("GraphedInference: call decoder.disable_tail_stream() first (a captured step is one graph)")
        self.model, self.decoder, self.anchors = model, decoder, anchors
        self.static_x = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):  # builds the plan, sizes every workspace, sets kernel attributes
                self._step()
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self._step()

    def _step(self):
        loc, conf = self.model(self.static_x)
        return self.decoder(loc, conf, self.anchors)

    def __call__(self, x):
        if x.shape != self.static_x.shape or x.dtype != self.static_x.dtype:
            raise ValueError("graph was captured for {} {}, got {} {}".format(
                tuple(self.static_x.shape), self.static_x.dtype, tuple(x.shape), x.dtype))
        if x.data_ptr() != self.static_x.data_ptr():
            self.static_x.copy_(x)
        self.graph.replay()
        return self.static_out
