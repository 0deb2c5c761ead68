"""Import / export of the WebGL demo's JSON weight format.

Reference: ConditioneDyNCA/convert_models_to_webgl.ipynb cell 1 (`export_np_models_to_json`, `tile2d`) writes it,
docs/dynca.js:827-872 reads it.  Per layer the file holds ONE normalised texture for all exported models:

    params[n]   = [w | b]^T                      rows = inputs (+1 bias row), cols = outputs        ([rows, cols])
    cols padded to a multiple of 4, viewed as [rows, cols/4, 4] (RGBA texels), the n models tiled `layout = (w, h)`
    data        = (texture - min) / scale,  scale = max - min,  center = -min / scale
    value       = (data - center) * scale

`layers[0]` is w1/b1 (inputs ordered [x, Sx*x, Sy*x, L*x, conditioning]), `layers[1]` is w2/b2.
"""
import json
from typing import Dict, List, Sequence, Union

import numpy as np
import torch


def decode_layer(layer: Dict, index: int = 0) -> np.ndarray:
    """-> float32 [rows, cols] parameter matrix ([w | b]^T) of model `index`."""
    rows, cols = layer["shape"]
    wt = (cols + 3) // 4
    lw, lh = layer["layout"]
    a = np.asarray(layer["data_flatten"], dtype=np.float64).reshape(layer["data_shape"])
    if a.shape[0] != rows * lh or a.shape[1] != wt * lw or a.shape[2] != 4:
        raise ValueError(f"webgl layer: data_shape {a.shape} does not match shape {layer['shape']} x layout {layer['layout']}")
    if not 0 <= index < lw * lh:
        raise IndexError(f"webgl layer: model index {index} outside layout {layer['layout']}")
    r, c = index // lw, index % lw
    tile = a[r * rows:(r + 1) * rows, c * wt:(c + 1) * wt].reshape(rows, wt * 4)[:, :cols]
    return ((tile - layer["center"]) * layer["scale"]).astype(np.float32)


def load_dynca_weights(src: Union[str, Dict], index: int = 0) -> Dict[str, torch.Tensor]:
    """JSON file (or parsed dict) -> {'w1.weight' [fc,K,1,1], 'w1.bias', 'w2.weight' [C,fc,1,1], 'w2.bias'} plus the
    layer-0 flags 'pos_emb' / 'edge_conditioning'."""
    js = json.load(open(src)) if isinstance(src, str) else src
    l1, l2 = (decode_layer(l, index) for l in js["layers"][:2])
    out = {"w1.weight": torch.from_numpy(l1[:-1].T.copy())[:, :, None, None], "w1.bias": torch.from_numpy(l1[-1].copy())}
    if js["layers"][1].get("bias", True):
        out["w2.weight"] = torch.from_numpy(l2[:-1].T.copy())[:, :, None, None]
        out["w2.bias"] = torch.from_numpy(l2[-1].copy())
    else:
        out["w2.weight"] = torch.from_numpy(l2.T.copy())[:, :, None, None]
        out["w2.bias"] = torch.zeros(l2.shape[1])
    out["pos_emb"] = bool(js["layers"][0].get("pos_emb", False))
    out["edge_conditioning"] = bool(js["layers"][0].get("edge_conditioning", False))
    out["n_perception_scales"] = int(js.get("n_perception_scales", 1))     # every shipped video model: 2 (docs/dynca.js:288-355)
    return out


def load_dynca(src: Union[str, Dict], index: int = 0, c_out: int = 3, padding_mode: str = "circular",
               device: Union[str, torch.device] = "cuda", **kw):
    """Build an ncahip DyNCA with the weights of model `index`; the channel counts follow from the matrix shapes."""
    from .models.dynca import DyNCA
    w = load_dynca_weights(src, index)
    fc, k1 = w["w1.weight"].shape[:2]
    c_in = w["w2.weight"].shape[0]
    cond = "edges" if w["edge_conditioning"] else ("pos_emb" if w["pos_emb"] else "none")
    kw.setdefault("perception_scales", list(range(w["n_perception_scales"])))      # the file's n_perception_scales (dynca.js:288-355)
    m = DyNCA(c_in, c_out, fc_dim=fc, padding_mode=padding_mode, conditioning=cond, device=torch.device(device), **kw)
    if m.w1.weight.shape[1] != k1:
        raise ValueError(f"webgl model: layer 0 has {k1} inputs, DyNCA(c_in={c_in}, conditioning={cond!r}) expects {m.w1.weight.shape[1]}")
    with torch.no_grad():
        for k in ("w1.weight", "w1.bias", "w2.weight", "w2.bias"):
            m.get_parameter(k).copy_(w[k])
    return m


def _tile2d(a: np.ndarray, w: int) -> np.ndarray:
    n, th, tw = a.shape[:3]
    pad = (w - n) % w
    a = np.pad(a, [(0, pad)] + [(0, 0)] * (a.ndim - 1))
    h = len(a) // w
    return a.reshape(h, w, th, tw, 4).transpose(0, 2, 1, 3, 4).reshape(th * h, tw * w, 4)


def export_dynca_json(models: Sequence, model_names: Sequence[str], path: str = None) -> Dict:
    """ncahip / reference DyNCA modules (same architecture) -> the demo's JSON dict (written to `path` if given)."""
    assert len(models) == len(model_names) and len(models) > 0
    per_layer: List[List[np.ndarray]] = [[], []]
    for m in models:
        for i, conv in enumerate((m.w1, m.w2)):
            w = conv.weight.detach().float().cpu().numpy()[:, :, 0, 0]
            b = conv.bias.detach().float().cpu().numpy()[:, None]
            per_layer[i].append(np.concatenate([w, b], axis=1).T)      # [rows = in + 1, cols = out]
    out = {"model_names": list(model_names), "layers": [], "n_perception_scales": len(getattr(models[0], "perception_scales", [0]))}
    for i, mats in enumerate(per_layer):
        layer = np.stack(mats)                                          # [n, rows, cols]
        n, rows, cols = layer.shape
        layer = np.pad(layer, ((0, 0), (0, 0), (0, (4 - cols) % 4))).reshape(n, rows, -1, 4)
        wt = layer.shape[2]
        w = 1
        while w < n and w * wt < (n + w - 1) // w * rows:
            w += 1
        tex = _tile2d(layer, w)
        lo, hi = float(tex.min()), float(tex.max())
        scale = hi - lo if hi > lo else 1.0
        out["layers"].append({"scale": scale, "center": -lo / scale, "data_flatten": [float(v) for v in ((tex - lo) / scale).ravel()],
                              "data_shape": list(tex.shape), "shape": [rows, cols], "layout": [w, (n + w - 1) // w],
                              "pos_emb": i == 0 and getattr(models[0], "conditioning", "") == "pos_emb",
                              "edge_conditioning": i == 0 and getattr(models[0], "conditioning", "") == "edges", "bias": True})
    if path:
        with open(path, "w") as f:
            json.dump(out, f)
    return out
