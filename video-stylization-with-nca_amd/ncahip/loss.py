"""Training objective for ConditionedNCATrainer (reference: EncoderConditioning/loss/loss.py:17-76).

In scope here: the overflow term (loss.py:37-40), exact.  The VGG16 appearance / content terms are the
"next" row f2 of SURVEY.md section 8: torchvision and the ImageNet weights are not available offline, so the
feature extractor below is a pure-torch VGG16-features definition that loads a local `vgg16-397923af.pth`
when NCAHIP_VGG16_WEIGHTS points at one and otherwise runs with seeded random weights (timing stand-in;
loss values vs the reference are then "parity unpinned").  Appearance types: 'OT' (the reference's default: relaxed
EMD over cosine distances + first/second moment matching, appearance_loss.py:149-220), 'Gram' and 'SlW' (sliced
Wasserstein, project-sort).  The OT arithmetic itself is pinned by tests/test_trainer_logic.py against an independent
float64 evaluation of the same formulas.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
STYLE_LAYERS = (1, 6, 11, 18, 25)   # appearance_loss.py:85,114,144
CONTENT_LAYER = 19                  # content_loss.py:17 (conv4_2 pre-ReLU index in torchvision's features)


class VGG16Features(nn.Module):
    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=False)]
                c = v
        self.features = nn.Sequential(*layers)
        path = os.environ.get("NCAHIP_VGG16_WEIGHTS", "")
        if path and os.path.exists(path):
            sd = torch.load(path, weights_only=True)
            self.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.")})
            self.pinned = True
        else:
            g = torch.Generator().manual_seed(16)
            with torch.no_grad():
                for m in self.features:
                    if isinstance(m, nn.Conv2d):
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight[0].numel())) ** 0.5)
                        m.bias.zero_()
            self.pinned = False
            warnings.warn("ncahip.loss: no VGG16 weights (NCAHIP_VGG16_WEIGHTS); using seeded random features -- "
                          "appearance/content loss values are not comparable with the reference")
        for p in self.parameters():
            p.requires_grad_(False)
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406])[None, :, None, None])
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225])[None, :, None, None])

    def forward(self, x, layers):
        x = (x - self.mean) / self.std
        out, last = {}, max(layers)
        for i, m in enumerate(self.features):
            x = m(x)
            if i in layers:
                out[i] = x
            if i >= last:
                break
        return out


def _gram(f):
    b, c, h, w = f.shape
    f = f.reshape(b, c, h * w)
    return f @ f.transpose(1, 2) / (h * w)


def _sliced_wasserstein(a, b, n_proj=32):
    bsz, c = a.shape[0], a.shape[1]
    a, b = a.reshape(bsz, c, -1), b.reshape(b.shape[0], c, -1)
    proj = F.normalize(torch.randn(c, n_proj, device=a.device), dim=0)
    pa = torch.sort(torch.einsum("bcn,cp->bpn", a, proj), dim=-1)[0]
    pb = torch.sort(torch.einsum("bcn,cp->bpn", b, proj), dim=-1)[0]
    if pa.shape[-1] != pb.shape[-1]:
        pb = F.interpolate(pb, size=pa.shape[-1], mode="nearest")
    return ((pa - pb) ** 2).mean()


def _pairwise_cos(x, y):
    """appearance_loss.py:150-156: 1 - <x_i, y_j> / (|x_i| + 1e-10) / (|y_j| + 1e-10), x [N,d], y [M,d]."""
    xn = torch.sqrt((x ** 2).sum(1)).view(-1, 1)
    yn = torch.sqrt((y ** 2).sum(1)).view(1, -1)
    return 1.0 - torch.mm(x, y.t()) / (xn + 1e-10) / (yn + 1e-10)


def _relaxed_emd(x, y):
    """appearance_loss.py:158-174: max of the two mean nearest-neighbour cosine distances."""
    d = _pairwise_cos(x, y)
    return torch.max(d.min(1)[0].mean(), d.min(0)[0].mean())


def _moment_loss(x, y):
    """appearance_loss.py:176-192: mean |mu_x - mu_y| + mean |cov_x - cov_y| (unbiased covariances), x [N,d], y [M,d]."""
    mx, my = x.mean(0, keepdim=True), y.mean(0, keepdim=True)
    xc, yc = x - mx, y - my
    cx = torch.mm(xc.t(), xc) / (x.shape[0] - 1)
    cy = torch.mm(yc.t(), yc) / (y.shape[0] - 1)
    return (mx - my).abs().mean() + (cx - cy).abs().mean()


def ot_loss_single(target_feats, gen_feats, n_samples=1000):
    """appearance_loss.py:194-210 for ONE generated image: per style layer, (sub-sampled when the map is larger than
    32x32: np.random.choice(h*w, 1000, replace=False), sorted -- the same consumption of numpy's global stream) relaxed
    EMD + moment loss between the target's and the image's feature vectors."""
    loss = 0
    for t, gfeat in zip(target_feats, gen_feats):
        c, h, w = t.shape[1], t.shape[2], t.shape[3]
        tv, gv = t.reshape(c, -1), gfeat.reshape(c, -1)
        if h > 32:
            idx = torch.as_tensor(np.sort(np.random.choice(np.arange(h * w), size=n_samples, replace=False)), device=t.device)
            tv, gv = tv[:, idx], gv[:, idx]
        tv, gv = tv.t(), gv.t()          # [N, d]
        loss = loss + _relaxed_emd(tv, gv) + _moment_loss(tv, gv)
    return loss


class Loss(nn.Module):
    def __init__(self, device, content_loss_weight=1.0, overflow_loss_weight=1.0, appearance_loss_weight=1.0,
                 appearance_loss_type="OT", target_style_image=None):
        super().__init__()
        self.device = device
        self.appearance_loss_type = appearance_loss_type
        self.appearance_loss_weight = appearance_loss_weight
        self.content_loss_weight = content_loss_weight
        self.overflow_loss_weight = overflow_loss_weight
        self.loss_weights = {}
        if overflow_loss_weight != 0:
            self.loss_weights["overflow"] = overflow_loss_weight
        if appearance_loss_weight != 0:
            assert target_style_image is not None, "Target style image required to use appearance loss"
            if appearance_loss_type not in ("OT", "Gram", "SlW"):
                raise ValueError(f"ncahip.loss: unknown appearance_loss_type={appearance_loss_type!r}")
            self.loss_weights["appearance"] = appearance_loss_weight
        if content_loss_weight != 0:
            self.loss_weights["content"] = content_loss_weight
        self.vgg = VGG16Features().to(device) if (appearance_loss_weight != 0 or content_loss_weight != 0) else None
        if appearance_loss_weight != 0:
            style = torch.as_tensor(target_style_image, dtype=torch.float32, device=device)
            if style.dim() == 3:
                style = style[None]
            with torch.no_grad():
                self.style_feats = self.vgg(style, STYLE_LAYERS)

    def get_overflow_loss(self, input_dict):  # loss.py:37-40
        s = input_dict["nca_state"]
        return (s - s.clamp(-1.0, 1.0)).abs().mean()

    def forward(self, input_dict, return_summary=True):
        loss, log = 0, {}
        terms = {}
        if "overflow" in self.loss_weights:
            terms["overflow"] = self.get_overflow_loss(input_dict)
        if self.vgg is not None:
            gen = input_dict["generated_images"]
            need = set()
            if "appearance" in self.loss_weights:
                need |= set(STYLE_LAYERS)
            if "content" in self.loss_weights:
                need.add(CONTENT_LAYER)
            gf = self.vgg(gen, tuple(sorted(need)))
            if "appearance" in self.loss_weights:
                acc = 0
                if self.appearance_loss_type == "OT":    # appearance_loss.py:212-220: mean over the batch
                    tgt = [self.style_feats[l] for l in STYLE_LAYERS]
                    for b in range(gen.shape[0]):
                        acc = acc + ot_loss_single(tgt, [gf[l][b:b + 1] for l in STYLE_LAYERS])
                    acc = acc / gen.shape[0]
                for l in (STYLE_LAYERS if self.appearance_loss_type != "OT" else ()):
                    if self.appearance_loss_type == "Gram":
                        acc = acc + F.mse_loss(_gram(gf[l]), _gram(self.style_feats[l]).expand(gf[l].shape[0], -1, -1))
                    else:
                        acc = acc + _sliced_wasserstein(gf[l], self.style_feats[l].expand(gf[l].shape[0], -1, -1, -1))
                terms["appearance"] = acc
            if "content" in self.loss_weights:
                with torch.no_grad():
                    tf = self.vgg(input_dict["target_images"], (CONTENT_LAYER,))[CONTENT_LAYER]
                terms["content"] = F.mse_loss(gf[CONTENT_LAYER], tf)
        for k, v in terms.items():
            v = v * self.loss_weights[k]
            log[k] = v.detach()
            loss = loss + v
        return [loss, log if return_summary else None]
