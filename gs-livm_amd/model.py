"""Host-side mirror of the pieces of GaussianModel that sit either side of the rasterizer, on the fused
kernels of csrc/optimizer.hip (SURVEY.md section 8(f) "next" row 1):

  * `FusedActivations` -- the getters Get_scaling / Get_rotation / Get_opacity / Get_features
    (include/gs/gs/gaussian.cuh:40-54) as ONE autograd node instead of five Torch ops;
  * `GaussianParameters` -- the six leaf tensors with those getter names;
  * `FusedAdam` -- torch::optim::Adam as the reference configures it (src/gs/gaussian.cu:396-428: one group
    per leaf, eps 1e-15, betas (0.9, 0.999), no weight decay) stepping every group in one launch and clearing
    the gradients it consumed (step + zero_grad, src/liw/lioOptimization.cpp:1831-1832);
  * `GrowableGaussians` -- row 4: the model as capacity buffers that grow in place
    (GaussianModel::addNewPointcloud / densification_postfix / cat_tensors_to_optimizer,
    src/gs/gaussian.cu:241-313, 451-472, 524-540).
"""
import torch

from . import _capi


class FusedActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling_raw, rotation_raw, opacity_raw, features_dc, features_rest):
        scales, rot, opac, shs = _capi.activate(scaling_raw.contiguous(), rotation_raw.contiguous(),
                                                opacity_raw.contiguous(), features_dc.contiguous(),
                                                features_rest.contiguous())
        ctx.save_for_backward(rotation_raw, scales, opac)
        return scales, rot, opac, shs

    @staticmethod
    def backward(ctx, g_scales, g_rot, g_opac, g_shs):
        rotation_raw, scales, opac = ctx.saved_tensors
        z = torch.zeros_like
        g_scales = z(scales) if g_scales is None else g_scales.contiguous()
        g_rot = z(rotation_raw) if g_rot is None else g_rot.contiguous()
        g_opac = z(opac) if g_opac is None else g_opac.contiguous()
        assert g_shs is not None
        return _capi.activate_backward(rotation_raw.contiguous(), scales, opac, g_scales, g_rot, g_opac,
                                       g_shs.contiguous())


class _StashingActivations(torch.autograd.Function):
    """FusedActivations for the fused optimiser tail (FusedAdam.step_model): forward hands out the activated
    values the last step already computed (no launch) or computes them; backward does NOT run the chain rule -- it
    parks the gradients w.r.t. the activated tensors on the model, and FusedAdam.step_model applies chain rule +
    Adam + next activations in one kernel (csrc/optimizer.hip, k_model_step)."""

    @staticmethod
    def forward(ctx, model, scaling_raw, rotation_raw, opacity_raw, features_dc, features_rest):
        ctx.model = model
        cached = model._next_act
        model._next_act = None
        if cached is not None:
            return cached
        return _capi.activate(scaling_raw.contiguous(), rotation_raw.contiguous(), opacity_raw.contiguous(),
                              features_dc.contiguous(), features_rest.contiguous())

    @staticmethod
    def backward(ctx, g_scales, g_rot, g_opac, g_shs):
        m = ctx.model
        new = [g_scales, g_rot, g_opac, g_shs]
        shapes = [(m._scaling.shape), (m._rotation.shape), (m._opacity.shape),
                  (m._xyz.shape[0], 1 + m._features_rest.shape[1], 3)]
        new = [torch.zeros(sh, device=m._xyz.device) if g is None else g.contiguous() for g, sh in zip(new, shapes)]
        if m._act_grads is None:
            m._act_grads = new
        else:  # several backward passes before one step (views rendered in separate graphs): gradients add up
            m._act_grads = [a + b for a, b in zip(m._act_grads, new)]
        return None, None, None, None, None, None


class GaussianParameters(torch.nn.Module):
    """The leaf tensors of GaussianModel (include/gs/gs/gaussian.cuh:107-119) and its getters."""

    def __init__(self, xyz, features_dc, features_rest, scaling, rotation, opacity):
        super().__init__()
        P = torch.nn.Parameter
        self._xyz, self._features_dc, self._features_rest = P(xyz), P(features_dc), P(features_rest)
        self._scaling, self._rotation, self._opacity = P(scaling), P(rotation), P(opacity)
        self._init_fused_tail()

    def _init_fused_tail(self):
        self.fused_tail = False   # True: activated() parks gradients for FusedAdam.step_model (see there)
        self._next_act = None     # activated values of the current parameters, left by the last step_model
        self._act_grads = None    # gradients w.r.t. (scales, rotations, opacities, shs) parked by backward

    def activated(self):
        """(xyz, opacity [P,1], scales, rotations, shs) through one fused node."""
        if self.fused_tail:
            scales, rot, opac, shs = _StashingActivations.apply(self, self._scaling, self._rotation, self._opacity,
                                                                self._features_dc, self._features_rest)
        else:
            scales, rot, opac, shs = FusedActivations.apply(self._scaling, self._rotation, self._opacity,
                                                            self._features_dc, self._features_rest)
        return self._xyz, opac, scales, rot, shs

    # reference getter names (each call runs the fused node; use activated() to get all at once)
    def Get_xyz(self):
        return self._xyz

    def Get_max_sh_degree(self):
        m = 1 + int(self._features_rest.shape[1])  # coefficients per channel = (degree + 1)^2
        return int(round(m ** 0.5)) - 1

    def Get_opacity(self):
        return self.activated()[1]

    def Get_scaling(self):
        return self.activated()[2]

    def Get_rotation(self):
        return self.activated()[3]

    def Get_features(self):
        return self.activated()[4]

    def param_groups(self, position_lr=0.0005, feature_lr=0.001, opacity_lr=0.025, scaling_lr=0.0025,
                     rotation_lr=0.0025, spatial_lr_scale=1.0):
        """Groups and learning rates of GaussianModel::Training_setup (src/gs/gaussian.cu:396-428) with the
        defaults of config/basic_common.yaml:54-62."""
        return [
            {"params": [self._xyz], "lr": position_lr * spatial_lr_scale},
            {"params": [self._features_dc], "lr": feature_lr},
            {"params": [self._features_rest], "lr": feature_lr / 20.0},
            {"params": [self._scaling], "lr": scaling_lr * spatial_lr_scale},
            {"params": [self._rotation], "lr": rotation_lr},
            {"params": [self._opacity], "lr": opacity_lr},
        ]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay / amsgrad), all groups in ONE kernel launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-15):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._step = 0

    @torch.no_grad()
    def step(self, zero_grads=True):
        ps, gs, ms, vs, lrs = [], [], [], [], []
        betas = eps = None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None or p.numel() == 0:
                    continue
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                assert betas in (None, group["betas"]) and eps in (None, group["eps"]), \
                    "FusedAdam: one (betas, eps) for all groups, as the reference configures it"
                betas, eps = group["betas"], group["eps"]
                ps.append(p); gs.append(p.grad); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
                lrs.append(group["lr"])
        self._step += 1
        for i in range(0, len(ps), 8):
            _capi.adam_step(ps[i:i + 8], gs[i:i + 8], ms[i:i + 8], vs[i:i + 8], lrs[i:i + 8], betas[0], betas[1],
                            eps, self._step, zero_grads)

    @torch.no_grad()
    def step_model(self, model):
        """The optimiser tail of one iteration in ONE launch (k_model_step): chain rule of the activations, Adam on
        the six groups of `model` (which must be this optimiser's six groups, `model.fused_tail = True`), and the
        activated values of the updated parameters, which the next `model.activated()` hands out without a launch.
        Equivalent to FusedActivations' backward + step(); the raw-space gradients are never materialised."""
        assert model.fused_tail and model._act_grads is not None and model._xyz.grad is not None, \
            "step_model needs a backward through model.activated() with model.fused_tail = True"
        ps = [model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity]
        lrs, ms, vs = [], [], []
        betas, eps = self.defaults["betas"], self.defaults["eps"]
        for p in ps:
            if p.numel() == 0:  # e.g. features_rest at SH degree 0: nothing to step, its group may be absent
                lrs.append(0.0); ms.append(p.detach()); vs.append(p.detach())
                continue
            grp = next(g for g in self.param_groups if any(q is p for q in g["params"]))
            st = self.state[p]
            if not st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            lrs.append(grp["lr"]); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            betas, eps = grp["betas"], grp["eps"]
        self._step += 1
        g_scales, g_rot, g_opac, g_shs = model._act_grads
        model._next_act = _capi.model_step(ps, ms, vs, model._xyz.grad.contiguous(), g_scales, g_rot, g_opac, g_shs, lrs,
                                           betas[0], betas[1], eps, self._step)
        model._act_grads = None
        model._xyz.grad = None


class GrowableGaussians(GaussianParameters):
    """The six leaves and their Adam moments as CAPACITY buffers; the leaves are views of the first P rows.

    The reference grows the map every few frames by building the new rows with Torch ops and then
    `torch::cat`-ing ALL six parameter tensors and all twelve optimiser-state tensors (whole-model copies,
    src/gs/gaussian.cu:451-472, 524-540).  Here `add_new_pointcloud` initialises rows [P, P + n) in place with
    one kernel (csrc/growth.hip) and re-binds the leaves: O(n) bytes move; the moments of the new rows are the
    zeros the buffers were created with (= the reference's `zeros_like(extension_tensor)`), the step count is
    shared (= the reference keeps the old AdamParamState's step).  When the capacity is exhausted the buffers
    double (one amortised copy).
    """

    _NAMES = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")

    def __init__(self, capacity, M, device):
        torch.nn.Module.__init__(self)
        self._init_fused_tail()
        self.M, self.P, self.capacity, self.device = int(M), 0, 0, torch.device(device)
        self._buf, self._m, self._v = {}, {}, {}
        self._optimizer = None
        self._reserve(max(1, int(capacity)))
        self._bind()

    def _shapes(self):
        return {"_xyz": (3,), "_features_dc": (1, 3), "_features_rest": (self.M - 1, 3), "_scaling": (3,),
                "_rotation": (4,), "_opacity": (1,)}

    def _reserve(self, capacity):
        for name, tail in self._shapes().items():
            for store in (self._buf, self._m, self._v):
                new = torch.zeros((capacity,) + tail, dtype=torch.float32, device=self.device)
                if name in store and self.P:
                    new[:self.P].copy_(store[name][:self.P])
                store[name] = new
        self.capacity = capacity

    def _bind(self):
        """Leaves = views of the first P rows (new Parameter objects, same storage); optimiser state follows."""
        for name in self._NAMES:
            setattr(self, name, torch.nn.Parameter(self._buf[name][:self.P]))
        self._next_act = None  # the cached activations describe the old row count
        self._act_grads = None  # ... and so would gradients parked by a backward that ran before the growth
        if self._optimizer is not None:
            self._optimizer.rebind(self)

    def moments(self, name):
        return self._m[name][:self.P], self._v[name][:self.P]

    def attach(self, optimizer):
        self._optimizer = optimizer

    @torch.no_grad()
    def add_new_pointcloud(self, xyz, covs, rgbs, scale_factor=1.0):
        """GaussianModel::addNewPointcloud (src/gs/gaussian.cu:241-313): xyz [n,3], covs [n,3,3], rgbs [n,3] (0..255),
        device f32.  Returns the row range of the new Gaussians."""
        n = int(xyz.size(0))
        if n == 0:
            return self.P, self.P
        if self.P + n > self.capacity:
            self._reserve(max(2 * self.capacity, self.P + n))
        lo, hi = self.P, self.P + n
        b = self._buf
        _capi.init_gaussians(xyz.contiguous(), covs.contiguous(), rgbs.contiguous(), scale_factor, b["_xyz"][lo:hi],
                             b["_features_dc"][lo:hi], b["_features_rest"][lo:hi], b["_scaling"][lo:hi],
                             b["_rotation"][lo:hi], b["_opacity"][lo:hi])
        self.P = hi
        self._bind()
        return lo, hi


class GrowableAdam(FusedAdam):
    """FusedAdam whose moments live in the model's capacity buffers, so growing the model needs no optimiser-state
    concatenation (cat_tensors_to_optimizer, src/gs/gaussian.cu:451-472)."""

    def __init__(self, model, eps=1e-15, **lrs):
        self._lrs = lrs
        super().__init__(model.param_groups(**lrs), eps=eps)
        model.attach(self)
        self.rebind(model)

    def rebind(self, model):
        groups = model.param_groups(**self._lrs)
        for g_old, g_new in zip(self.param_groups, groups):
            g_old["params"] = g_new["params"]
        self.state.clear()
        for name in model._NAMES:
            p = getattr(model, name)
            m, v = model.moments(name)
            self.state[p] = {"exp_avg": m, "exp_avg_sq": v}
