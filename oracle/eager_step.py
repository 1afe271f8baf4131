"""ORACLE (test infrastructure / CPU baseline, never imported by the product): the "reference CPU eager path".

The reference cannot run on CPU (hard CUDA + DeepSpeed dependencies: train.py:296,299; models/base.py:230), so the
eager reference step is *defined* as SURVEY.md section 8(c) prescribes: sequential composition of the adapter's
`to_layers()` and `get_loss_fn()` in fp32, DeepSpeed's gradient-accumulation semantics restated
([3P] deepspeed==0.18.4, parity unpinned), the reference's own `clip_grad_norm_` (utils/patches.py:175-246) and
loss functions (models/base.py:418-436, models/sdxl.py:281-355,632-651) followed line by line.
"""
import torch
import torch.nn.functional as F


# ---- losses ----------------------------------------------------------------------------------------------------
def default_loss_fn(config=None):
    """models/base.py:418-436."""
    config = config or {}

    def loss_fn(output, label):
        target, mask = label
        output = output.to(torch.float32)
        target = target.to(output.device, torch.float32)
        if 'huber_delta' in config:
            loss = F.huber_loss(output, target, reduction='none', delta=config['huber_delta'])
        elif 'smooth_l1_beta' in config:
            loss = F.smooth_l1_loss(output, target, reduction='none', beta=config['smooth_l1_beta'])
        else:
            loss = F.mse_loss(output, target, reduction='none')
        if mask.numel() > 0:
            mask = mask.to(output.device, torch.float32)
            loss *= mask
        return loss.mean()
    return loss_fn


def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """[3P] diffusers DDPMScheduler(beta_schedule='scaled_linear') as configured for SDXL."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def all_snr(alphas_cumprod):
    """models/sdxl.py:281-292 prepare_scheduler_for_custom_training."""
    alpha = torch.sqrt(alphas_cumprod)
    sigma = torch.sqrt(1.0 - alphas_cumprod)
    return (alpha / sigma) ** 2


def apply_snr_weight(loss, timesteps, snr_table, gamma, v_prediction=False):
    """models/sdxl.py:333-344."""
    snr = torch.stack([snr_table[t] for t in timesteps])
    min_snr_gamma = torch.minimum(snr, torch.full_like(snr, gamma))
    if v_prediction:
        w = torch.div(min_snr_gamma, snr + 1).float().to(loss.device)
    else:
        w = torch.div(min_snr_gamma, snr).float().to(loss.device)
    return loss * w


def apply_debiased_estimation(loss, timesteps, snr_table, v_prediction=False):
    """models/sdxl.py:347-355."""
    snr_t = torch.stack([snr_table[t] for t in timesteps])
    snr_t = torch.minimum(snr_t, torch.ones_like(snr_t) * 1000)
    weight = 1 / (snr_t + 1) if v_prediction else 1 / torch.sqrt(snr_t)
    return loss * weight.to(loss.device)


def sdxl_loss_fn(snr_table=None, min_snr_gamma=None, debiased_estimation_loss=None, v_pred=False):
    """models/sdxl.py:632-651."""
    def loss_fn(output, label):
        output, timesteps = output
        target, mask = label
        output = output.to(torch.float32)
        target = target.to(output.device, torch.float32)
        loss = F.mse_loss(output, target, reduction='none')
        if mask.numel() > 0:
            mask = mask.to(output.device, torch.float32)
            loss *= mask
        loss = loss.mean([1, 2, 3])
        if min_snr_gamma is not None:
            loss = apply_snr_weight(loss, timesteps, snr_table, min_snr_gamma, v_pred)
        if debiased_estimation_loss is not None:
            loss = apply_debiased_estimation(loss, timesteps, snr_table, v_pred)
        return loss.mean()
    return loss_fn


def ddpm_add_noise(latents, noise, timesteps, alphas_cumprod):
    """[3P] diffusers DDPMScheduler.add_noise (call site models/sdxl.py:560)."""
    a = alphas_cumprod[timesteps].to(latents.dtype)
    sa = a.sqrt().view(-1, *([1] * (latents.dim() - 1)))
    sb = (1 - a).sqrt().view(-1, *([1] * (latents.dim() - 1)))
    return sa * latents + sb * noise


# ---- gradient clipping -------------------------------------------------------------------------------------------
def clip_grad_norm_(parameters, max_norm, norm_type=2, mpu=None):
    """utils/patches.py:175-246 for a single process (world size 1): per-parameter fp32 L2 norms -> stack ->
    square -> sum -> pow(1/2); clip_coef = min(1, max_norm / (total_norm + 1e-6)); grads scaled in place.
    Returns the total norm before clipping."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    all_norms = [p.grad.data.detach().float().norm(float(norm_type)) for p in parameters]
    if len(all_norms) > 0:
        total_norm = torch.stack(all_norms).square().sum().float()
    else:
        total_norm = torch.zeros(())
    total_norm = total_norm.pow(1.0 / norm_type)
    clip_coef = torch.tensor([float(max_norm)]) / (total_norm + 1e-6)
    clip_coef = torch.min(torch.tensor([1.0]), clip_coef)
    for p in parameters:
        p.grad.data.mul_(clip_coef.to(p.grad.device))
    return total_norm


# ---- the eager step -------------------------------------------------------------------------------------------------
def run_layers(layers, features):
    """[3P] PipelineModule.forward with every layer on one stage: x = layer(x) in order; a single-tensor feature
    tuple is unwrapped exactly as DeepSpeed's exec_func does."""
    x = features
    if isinstance(x, (tuple, list)) and len(x) == 1:
        x = x[0]
    for layer in layers:
        x = layer(x)
    return x


def eager_train_step(layers, loss_fn, micro_batches, optimizer=None, gradient_clipping=0.0, lr_scheduler=None, params=None):
    """One optimizer step over `micro_batches` = [(features, label), ...].

    Returns (mean loss over micro-batches, pre-clip global grad norm or None).  Loss of each micro-batch is divided
    by the number of micro-batches before backward ([3P] DeepSpeedEngine.backward scale_wrt_gas); gradients
    accumulate in the parameter dtype."""
    gas = len(micro_batches)
    total = None
    for features, label in micro_batches:
        out = run_layers(layers, features)
        loss = loss_fn(out, label)
        total = loss.detach().clone() if total is None else total + loss.detach()
        (loss / gas).backward()
    if params is None:
        params = [p for l in layers if isinstance(l, torch.nn.Module) for p in l.parameters()]
    grad_norm = None
    if gradient_clipping > 0:
        grad_norm = clip_grad_norm_(params, gradient_clipping)
    else:
        norms = [p.grad.detach().float().norm(2) for p in params if p.grad is not None]
        grad_norm = torch.stack(norms).square().sum().sqrt() if norms else None
    if optimizer is not None:
        optimizer.step()
        optimizer.zero_grad()
    if lr_scheduler is not None:
        lr_scheduler.step()
    return total / gas, grad_norm


def eager_eval(layers, loss_fn, micro_batches):
    with torch.no_grad():
        losses = [loss_fn(run_layers(layers, f), l) for f, l in micro_batches]
    return torch.stack(losses).mean()


class TorchGradKernels:
    """CPU stand-in for the two local gradient kernels (ops.grads_sumsq / ops.grads_clip_scale_) so that the
    engine's distributed clip composition can be exercised by the gloo tests.  Test infrastructure only."""

    @staticmethod
    def grads_sumsq(grads):
        return torch.stack([g.detach().float().norm(2) for g in grads]).square().sum().float()

    @staticmethod
    def grads_clip_scale_(grads, total_sumsq, max_norm):
        coef = torch.clamp(float(max_norm) / (total_sumsq.sqrt() + 1e-6), max=1.0)
        for g in grads:
            g.mul_(coef.to(g.dtype))
