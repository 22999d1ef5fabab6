"""Deployment wire contract of the shipped model.

The reference exports ``OstrackDist`` to ONNX with (``tracking/onnxexport.py:390-400``, consumed by
``tracking/video_onnx.py`` and by OpenCV's TrackerVit, ``readme.md:3-5``)

    inputs   'template' (1,3,128,128) fp32   'search' (1,3,256,256) fp32      (normalised crops)
    outputs  'output1' = score_map (1,1,16,16)   'output2' = size_map (1,2,16,16)   'output3' = offset_map (1,2,16,16)

(the export-time forward_head returns {'score_map','size_map','offset_map'} in that order, ``:313-319``; no
pred_boxes).  ``VitTrackSession`` answers the ``onnxruntime.InferenceSession`` calls those consumers make --
``get_inputs() / get_outputs() / run(output_names, {name: ndarray})`` -- on the MI355X library, so a consumer of
the ``.onnx`` file can switch by replacing the session object.  No ONNX file is read or written (neither ``onnx``
nor a public model file exists in this image); weights come from the reference's ``ckpt['net']`` layout."""
from __future__ import annotations

from collections import namedtuple

import numpy as np

NodeArg = namedtuple("NodeArg", ["name", "shape", "type"])

INPUT_NAMES = ("template", "search")
OUTPUT_NAMES = ("output1", "output2", "output3")
_OUTPUT_FIELDS = {"output1": "score_map", "output2": "size_map", "output3": "offset_map"}


def wire_signature(template_size=128, search_size=256, stride=16, batch=1):
    """(inputs, outputs) as NodeArg lists -- pure metadata, no GPU needed."""
    F = search_size // stride
    ins = [NodeArg("template", [batch, 3, template_size, template_size], "tensor(float)"),
           NodeArg("search", [batch, 3, search_size, search_size], "tensor(float)")]
    outs = [NodeArg("output1", [batch, 1, F, F], "tensor(float)"), NodeArg("output2", [batch, 2, F, F], "tensor(float)"),
            NodeArg("output3", [batch, 2, F, F], "tensor(float)")]
    return ins, outs


class VitTrackSession:
    def __init__(self, state_dict, template_size=128, search_size=256, channels=48, heads=1, depth=3, head_channels=32,
                 stride=16, batch=1):
        import torch
        from . import native
        self._torch = torch
        self._ins, self._outs = wire_signature(template_size, search_size, stride, batch)
        self.batch = batch
        self._m = native.Model(template_size, search_size, channels, heads, depth, head_channels, stride, max_batch=batch)
        self._m.load_state_dict(state_dict)
        self._z = torch.zeros(*self._ins[0].shape, device="cuda")
        self._x = torch.zeros(*self._ins[1].shape, device="cuda")
        self._graph, self._out = self._m.capture(self._z, self._x)

    @classmethod
    def from_checkpoint(cls, path, cfg, **kw):
        """``torch.load(path)['net']`` (lib/test/tracker/vit_dist.py:25) + geometry from the YAML config."""
        import torch
        from .config import geometry
        g = geometry(cfg)
        return cls(torch.load(path, map_location="cpu")["net"], g["template_size"], g["search_size"], g["channels"], g["heads"],
                   3, g["head_channels"], g["stride"], **kw)

    def get_inputs(self):
        return list(self._ins)

    def get_outputs(self):
        return list(self._outs)

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def run(self, output_names, input_feed: dict):
        """onnxruntime semantics: output_names None = all outputs in declaration order; unknown or missing input
        names and wrong shapes / dtypes raise (ORT raises InvalidArgument)."""
        torch = self._torch
        names = list(OUTPUT_NAMES) if output_names is None else list(output_names)
        for n in names:
            if n not in _OUTPUT_FIELDS:
                raise ValueError(f"Invalid output name: {n}")
        extra = set(input_feed) - set(INPUT_NAMES)
        if extra:
            raise ValueError(f"Invalid input name: {sorted(extra)[0]}")
        for arg, buf in zip(self._ins, (self._z, self._x)):
            if arg.name not in input_feed:
                raise ValueError(f"Required inputs ([{arg.name!r}]) are missing from input feed")
            a = input_feed[arg.name]
            a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
            if a.dtype != np.float32:
                raise ValueError(f"Unexpected input data type for {arg.name}: {a.dtype}, expected float32")
            if list(a.shape) != list(arg.shape):
                raise ValueError(f"Got invalid dimensions for input: {arg.name}: got {list(a.shape)}, expected {arg.shape}")
            buf.copy_(torch.from_numpy(np.ascontiguousarray(a)))
        self._graph.launch()
        torch.cuda.current_stream().synchronize()
        return [getattr(self._out, _OUTPUT_FIELDS[n]).cpu().numpy() for n in names]
