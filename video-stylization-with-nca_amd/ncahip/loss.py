"""Training objective for ConditionedNCATrainer (reference: EncoderConditioning/loss/loss.py:17-76).

In scope here: the overflow term (loss.py:37-40), exact.  The VGG16 appearance / content terms are the
"next" row f2 of SURVEY.md section 8: torchvision and the ImageNet weights are not available offline, so the
feature extractor below is a pure-torch VGG16-features definition that loads a local `vgg16-397923af.pth`
when NCAHIP_VGG16_WEIGHTS points at one and otherwise runs with seeded random weights (timing stand-in;
loss values vs the reference are then "parity unpinned").  Appearance types: 'OT' (the reference's default: relaxed
EMD over cosine distances + first/second moment matching, appearance_loss.py:149-220), 'Gram' and 'SlW' (sliced
Wasserstein, project-sort over the image + the five style levels, appearance_loss.py:109-140).  The OT arithmetic itself is
pinned by tests/test_trainer_logic.py against an independent float64 evaluation of the same formulas; the batch is evaluated
with batched matrix products (`ot_loss_batched`) instead of the reference's per-sample Python loop (:212-220) -- same index
draws from numpy's global stream in the same order, same value up to summation order.  `feature_dtype=torch.bfloat16` runs the
VGG convolutions in bf16 on the device (BASELINE configs[2]); the loss arithmetic on the features stays fp32.
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
        dt = self.features[0].weight.dtype          # bf16 when the module was cast (.to(torch.bfloat16)): convs on bf16 MFMA
        x = x.to(dt)
        if self.features[0].weight.is_contiguous(memory_format=torch.channels_last) and not self.features[0].weight.is_contiguous():
            x = x.contiguous(memory_format=torch.channels_last)     # NHWC convolutions (Loss(channels_last=True))
        out, last = {}, max(layers)
        for i, m in enumerate(self.features):
            x = m(x)
            if i in layers:
                out[i] = x.float()
            if i >= last:
                break
        return out


def _gram(f):
    b, c, h, w = f.shape
    f = f.reshape(b, c, h * w)
    return f @ f.transpose(1, 2) / (h * w)


def _sliced_wasserstein(source, target, n_proj=32):
    """appearance_loss.py:121-136: source [b,c,n], target [1,c,m] flattened features; random unit projections drawn on the
    CPU generator (torch.randn(ch, 32)), project + sort, nearest-resample the target to n, SUM of squared differences."""
    ch, n = source.shape[-2:]
    proj = F.normalize(torch.randn(ch, n_proj), dim=0).to(source.device)
    ps = torch.einsum("bcn,cp->bpn", source, proj).sort()[0]
    pt = torch.einsum("bcn,cp->bpn", target, proj).sort()[0]
    return (ps - F.interpolate(pt, n, mode="nearest")).square().sum()


def _to_nchw(img) -> torch.Tensor:
    """appearance_loss.py:41-43 (torchvision ToTensor + unsqueeze): a PIL image or an H x W x C uint8 array becomes a
    [1,C,H,W] float tensor in [0,1]; float arrays are only transposed; a float tensor is taken as C x H x W (or N x C x H x W)
    already in network range."""
    if isinstance(img, torch.Tensor):
        t = img.float()
        return t[None] if t.dim() == 3 else t
    if not isinstance(img, np.ndarray):          # PIL.Image (or anything exposing the array interface)
        img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
    t = t.float().div(255.0) if img.dtype == np.uint8 else t.float()
    return t[None]


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


def ot_loss_batched(target_feats, gen_feats, n_samples=1000):
    """The batch mean of ot_loss_single (appearance_loss.py:212-220) without the per-sample Python loop: the sub-sampling
    indices are drawn sample by sample, layer by layer, exactly as the loop would draw them from numpy's global stream; the
    cosine-distance, nearest-neighbour and covariance products then run as batched GEMMs over all samples of a layer."""
    B = gen_feats[0].shape[0]
    idx = [[None] * len(gen_feats) for _ in range(B)]
    for b in range(B):                                     # the reference's draw order: sample-major, layer-minor
        for li, t in enumerate(target_feats):
            h, w = t.shape[2], t.shape[3]
            if h > 32:
                idx[b][li] = np.sort(np.random.choice(np.arange(h * w), size=n_samples, replace=False))
    total = 0
    for li, (t, g) in enumerate(zip(target_feats, gen_feats)):
        c = t.shape[1]
        tv, gv = t.reshape(1, c, -1), g.reshape(B, c, -1)
        if idx[0][li] is not None:
            ix = torch.as_tensor(np.stack([idx[b][li] for b in range(B)]), device=t.device)          # [B, N]
            gv = torch.gather(gv, 2, ix[:, None, :].expand(B, c, -1))
            tv = tv.expand(B, c, -1).gather(2, ix[:, None, :].expand(B, c, -1))
        else:
            tv = tv.expand(B, c, -1)
        x, y = tv.transpose(1, 2), gv.transpose(1, 2)                                               # [B, N, d]
        xn = torch.sqrt((x ** 2).sum(2))[:, :, None]
        yn = torch.sqrt((y ** 2).sum(2))[:, None, :]
        d = 1.0 - torch.bmm(x, y.transpose(1, 2)) / (xn + 1e-10) / (yn + 1e-10)
        remd = torch.maximum(d.min(2)[0].mean(1), d.min(1)[0].mean(1))                              # [B]
        mx, my = x.mean(1, keepdim=True), y.mean(1, keepdim=True)
        xc, yc = x - mx, y - my
        n = x.shape[1]
        cx = torch.bmm(xc.transpose(1, 2), xc) / (n - 1)
        cy = torch.bmm(yc.transpose(1, 2), yc) / (n - 1)
        mom = (mx - my).abs().mean(dim=(1, 2)) + (cx - cy).abs().mean(dim=(1, 2))
        total = total + (remd + mom).sum()
    return total / B


class Loss(nn.Module):
    def __init__(self, device, content_loss_weight=1.0, overflow_loss_weight=1.0, appearance_loss_weight=1.0,
                 appearance_loss_type="OT", target_style_image=None, feature_dtype=torch.float32, channels_last=False):
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
        if self.vgg is not None and feature_dtype != torch.float32:
            self.vgg.features.to(feature_dtype)
        if self.vgg is not None and channels_last:
            self.vgg.features.to(memory_format=torch.channels_last)
        if appearance_loss_weight != 0:
            self.target_style_tensor = _to_nchw(target_style_image).to(device)
            with torch.no_grad():
                self.style_feats = self.vgg(self.target_style_tensor, STYLE_LAYERS)

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
                    acc = ot_loss_batched([self.style_feats[l] for l in STYLE_LAYERS], [gf[l] for l in STYLE_LAYERS])
                elif self.appearance_loss_type == "Gram":   # :98-106
                    for l in STYLE_LAYERS:
                        acc = acc + (_gram(self.style_feats[l]) - _gram(gf[l])).square().mean()
                else:                                       # 'SlW', :138-140: the normalised image is the first "feature" level
                    flat = lambda f: f.reshape(f.shape[0], f.shape[1], -1)
                    norm = lambda im: (im - self.vgg.mean) / self.vgg.std
                    src = [flat(norm(gen))] + [flat(gf[l]) for l in STYLE_LAYERS]
                    tgt = [flat(norm(self.target_style_tensor))] + [flat(self.style_feats[l]) for l in STYLE_LAYERS]
                    acc = sum(_sliced_wasserstein(x, y) for x, y in zip(src, tgt))
                terms["appearance"] = acc
            if "content" in self.loss_weights:
                tgt_img = input_dict["target_images"]
                if tgt_img.shape[-2:] != gen.shape[-2:]:      # content_loss.py:31-32 (torchvision resize: antialiased bilinear)
                    tgt_img = F.interpolate(tgt_img, size=gen.shape[-2:], mode="bilinear", align_corners=False, antialias=True)
                with torch.no_grad():
                    tf = self.vgg(tgt_img, (CONTENT_LAYER,))[CONTENT_LAYER]
                terms["content"] = F.mse_loss(gf[CONTENT_LAYER], tf)
        for k, v in terms.items():
            v = v * self.loss_weights[k]
            log[k] = v.detach()
            loss = loss + v
        return [loss, log if return_summary else None]
