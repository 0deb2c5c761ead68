"""DyNCA / EdgeExtractor / CPE2D drop-ins (reference: ConditioneDyNCA/models/dynca.py:7-253).

Same constructor, attributes (w1, w2, cond_layer, sobel_filter_x/y, laplacian_filter, identity_filter)
and methods as the reference.  forward()/forward_nsteps()/perceive_torch() run on libncahip.so.
The conditioning map (EdgeExtractor on a constant cond_img, recomputed every step under no_grad in the
reference, dynca.py:118-124) is hoisted: computed once per forward_nsteps call.

Step (dynca.py:117-138):  y = [x | Sx*x | Sy*x | L*x | cond] ; dx = w2 relu(w1 y + b1) + b2 ;
                          x <- x + dx * floor(u + update_rate) ; rgb = 2 x[:, :c_out]
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..autograd import dynca_nsteps_autograd, hip_perceive


class EdgeExtractor(nn.Module):
    """Sobel-x, Sobel-y, Laplacian of a 1-channel image, zero padding, optional tanh (dynca.py:182-213)."""

    def __init__(self, transform):
        super().__init__()

        def frozen(taps):
            conv = nn.Conv2d(1, 1, kernel_size=3, padding=1, bias=False)
            conv.weight = nn.Parameter(torch.tensor([[taps]], dtype=torch.float32), requires_grad=False)
            return conv

        self.sobel_x = frozen([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
        self.sobel_y = frozen([[-1, -2, -1], [0, 0, 0], [1, 2, 1]])
        self.laplacian = frozen([[1, 2, 1], [2, -12, 2], [1, 2, 1]])
        self.edge_transform = nn.Tanh() if transform == "tanh" else nn.Identity()

    def forward(self, x):
        bank = torch.cat((self.sobel_x.weight, self.sobel_y.weight, self.laplacian.weight), dim=0)
        if x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad):
            return ops.edge_extractor(x, bank, isinstance(self.edge_transform, nn.Tanh))   # one HIP pass incl. the tanh
        return self.edge_transform(F.conv2d(x, bank, padding=1))  # one 1->3 channel conv instead of three + cat


class CPE2D(nn.Module):
    """Cartesian positional encoding, cached per input shape (dynca.py:216-253)."""

    def __init__(self):
        super().__init__()
        self.cached_penc = None
        self.last_tensor_shape = None

    def forward(self, tensor):
        if tensor.dim() != 4:
            raise RuntimeError("The input tensor has to be 4d!")
        if self.cached_penc is not None and self.last_tensor_shape == tensor.shape \
                and self.cached_penc.device == tensor.device:
            return self.cached_penc
        b, _, h, w = tensor.shape
        xs = torch.arange(h, device=tensor.device) / h
        ys = torch.arange(w, device=tensor.device) / w
        xs = 2.0 * (xs - 0.5 + 0.5 / h)
        ys = 2.0 * (ys - 0.5 + 0.5 / w)
        emb = torch.zeros((2, h, w), device=tensor.device, dtype=tensor.dtype)
        emb[0] = xs[:, None]
        emb[1] = ys[None, :]
        self.cached_penc = emb.unsqueeze(0).repeat(b, 1, 1, 1)
        self.last_tensor_shape = tensor.shape
        return self.cached_penc


class DyNCA(nn.Module):
    SEED_MODES = ["random", "center_on", "zeros"]

    def __init__(self, c_in, c_out, fc_dim=96, padding_mode="replicate", seed_mode="zeros", conditioning="edges",
                 edge_transform="tanh", perception_scales=[0], device=torch.device("cuda:0")):
        super().__init__()
        assert seed_mode in DyNCA.SEED_MODES
        self.c_in, self.c_out, self.fc_dim = c_in, c_out, fc_dim
        self.perception_scales = perception_scales
        self.padding_mode, self.seed_mode = padding_mode, seed_mode
        self.random_seed = 42
        self.conditioning = conditioning
        self.device = device
        self.expand = 4
        self.c_cond = 0
        if conditioning == "pos_emb":
            self.cond_layer = CPE2D()
            self.c_cond = 2
        elif conditioning == "edges":
            self.cond_layer = EdgeExtractor(edge_transform).to(device)
            self.c_cond = 3
        else:
            self.cond_layer = None
        self.w1 = nn.Conv2d(c_in * self.expand + self.c_cond, fc_dim, 1, device=device)
        nn.init.xavier_normal_(self.w1.weight, gain=0.2)
        self.w2 = nn.Conv2d(fc_dim, c_in, 1, bias=True, device=device)
        nn.init.xavier_normal_(self.w2.weight, gain=0.1)
        nn.init.zeros_(self.w2.bias)
        self.sobel_filter_x = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], device=device)
        self.sobel_filter_y = self.sobel_filter_x.T
        self.identity_filter = torch.tensor([[0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0]], device=device)
        self.laplacian_filter = torch.tensor([[1.0, 2.0, 1.0], [2.0, -12.0, 2.0], [1.0, 2.0, 1.0]], device=device)
        self.mask_rng, self.mask_seed, self._mask_step = "torch", 0, 0

    # ------------------------------------------------------------------ helpers
    def _draw(self, x, steps, update_rate=None):
        if self.mask_rng == "philox":
            return None
        b, _, h, w = x.shape
        if x.is_cuda and update_rate is not None and 0.0 <= float(update_rate) < 1.0:
            # dynca.py:131's draw per step, evaluated to floor(u + rate) right away and kept as bits (ops.draw_fire_masks)
            return ops.draw_fire_masks(b, h, w, steps, float(update_rate), "dynca", x.device)
        return torch.stack([torch.rand(b, 1, h, w, device=x.device) for _ in range(steps)])  # dynca.py:131, per step

    def _cond(self, x, cond_img):
        if self.cond_layer is None:
            return None
        if self.conditioning == "pos_emb":
            return self.cond_layer(x).float().contiguous()
        with torch.no_grad():  # dynca.py:123: no gradient into the edge map
            return self.cond_layer(cond_img).float().contiguous()

    def _multiscale(self) -> bool:
        return list(self.perception_scales) != [0]

    def _two_scale_fused(self, x) -> bool:
        """perception_scales == [0, 1] (every shipped video model; the default of fit_video_motion.py) on the fused two-scale
        kernels, forward and backward: even sizes, C <= 16, fc <= 128, fp32 states."""
        return (list(self.perception_scales) == [0, 1] and x.dtype == torch.float32
                and ops.two_scale_fused_ok(self.c_in, x.shape[2], x.shape[3], self.w1.out_channels))

    def _composed(self, x) -> bool:
        """True when the step runs as HIP stencil + device resampling + library GEMMs instead of the fused kernels:
        multi-scale perception outside what _two_scale_fused covers (other scale sets, odd sizes, C > 16, fc > 128).
        Single-scale forward and backward are fused for every C <= 32 and fc <= 1024 (BASELINE configs[4] trains at C = 32,
        fc = 256: hidden layers wider than 128 run as one launch per 128-wide slice)."""
        return self._multiscale() and not self._two_scale_fused(x)

    # ------------------------------------------------------------------ reference surface
    def perceive_torch(self, x, scale=0):
        """dynca.py:75-100: [x, Sx*x, Sy*x, L*x]; scale > 0: bilinear down by 2^scale, perceive, bilinear up.  The stencil
        is the HIP kernel (ncahip_dynca_perceive_f32); the two resamplings are torch ops on the device."""
        assert scale in [0, 1, 2, 3, 4, 5]
        x = x.float()
        if scale == 0:
            return hip_perceive(x, self.padding_mode)
        _, _, h, w = x.shape
        xs = F.interpolate(x, size=(int(h // (2 ** scale)), int(w // (2 ** scale))), mode="bilinear", align_corners=False)
        return F.interpolate(hip_perceive(xs, self.padding_mode), size=(h, w), mode="bilinear", align_corners=False)

    def perceive_multiscale(self, x, cond_mat=None):
        """dynca.py:102-115: mean over perception_scales, then the conditioning channels."""
        y = sum(self.perceive_torch(x, scale=s) for s in self.perception_scales) / len(self.perception_scales)
        return y if cond_mat is None else torch.cat([y, cond_mat], dim=1)

    def _step_multiscale(self, x, cond, update_rate, u):
        """One step with perception_scales != [0] (dynca.py:117-138).  Not fused: HIP stencil per scale + device
        resampling + the two 1x1 convolutions as library GEMMs (the fused kernel covers the
        single-scale models)."""
        y = self.perceive_multiscale(x, cond)
        dx = self.w2(F.relu(self.w1(y)))
        mask = torch.floor(u + update_rate)
        return x + dx * mask

    def _draw_one(self, x):
        b, _, h, w = x.shape
        if self.mask_rng == "philox":
            u = ops.philox_uniform(b, h, w, self.mask_seed, self._mask_step, device=x.device)
        else:
            u = torch.rand(b, 1, h, w, device=x.device)        # dynca.py:131
        self._mask_step += 1
        return u

    def forward(self, x, update_rate=0.5, return_perception=False, cond_img=None):
        cond = self._cond(x, cond_img)
        if self._composed(x):
            out = self._step_multiscale(x.float(), cond, update_rate, self._draw_one(x))
        elif self._multiscale():
            out, _ = dynca_nsteps_autograd(self, x, cond, 1, update_rate, two_scale=True)
        else:
            out, _ = dynca_nsteps_autograd(self, x, cond, 1, update_rate)
        if return_perception:
            return out, self.to_rgb(out), self.perceive_multiscale(x, cond)
        return out, self.to_rgb(out)

    def to_rgb(self, x):
        return x[:, :self.c_out, ...] * 2.0

    def seed(self, n, size=128):
        size_x, size_y = (size, size) if isinstance(size, int) else size
        c = self.c_in
        if self.seed_mode == "zeros":
            return torch.zeros(n, c, size_y, size_x, device=self.device)
        if self.seed_mode == "center_on":
            sd = torch.zeros(n, c, size_y, size_x, device=self.device)
            sd[:, :, size_y // 2, size_x // 2] = 1.0
            return sd
        np.random.seed(self.random_seed)  # 'random': reseeds every generator, as dynca.py:157-160
        torch.manual_seed(self.random_seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.random_seed)
        sd = torch.rand(1, c, size_y, size_x) - 0.5
        return sd.repeat(n, 1, 1, 1).to(self.device)

    def forward_nsteps(self, input_state, step_n, update_rate=0.5, return_middle_feature=False, cond_img=None):
        cond = self._cond(input_state, cond_img)
        if self._composed(input_state):
            x, mids = input_state.float(), []
            for _ in range(step_n):
                x = self._step_multiscale(x, cond, update_rate, self._draw_one(x))
                if return_middle_feature:
                    mids.append(self.to_rgb(x))
            return (x, self.to_rgb(x), mids) if return_middle_feature else (x, self.to_rgb(x))
        out, states = dynca_nsteps_autograd(self, input_state, cond, step_n, update_rate,
                                            want_states=return_middle_feature, two_scale=self._multiscale())
        feature = self.to_rgb(out)
        if return_middle_feature:
            return out, feature, [self.to_rgb(states[t]) for t in range(1, step_n + 1)]
        return out, feature
