"""Flow-matching scheduler of the hot path (host side, tiny): mirror of the reference's
`FlowMatchScheduler` (utils/scheduler.py:106-176) restricted to what inference uses — `set_timesteps`
(:118-141, shifted sigmas sigma' = s*sigma / (1 + (s-1)*sigma)) and `add_noise` (:159-176, argmin
lookup) — plus `get_denoising_schedule` (v2v.py:133-136).  Tables live on the session's device so no
step of the block loop synchronises with the host (release_server.py:555-560)."""
import torch


def _ops():
    from . import ops   # deferred: the schedule tables are usable without the kernel library
    return ops


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False):
        self.num_train_timesteps = num_train_timesteps
        self.shift, self.sigma_max, self.sigma_min = shift, sigma_max, sigma_min
        self.inverse_timesteps, self.extra_one_step, self.reverse_sigmas = inverse_timesteps, extra_one_step, reverse_sigmas
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False):
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        n = num_inference_steps + 1 if self.extra_one_step else num_inference_steps
        sig = torch.linspace(sigma_start, self.sigma_min, n)
        if self.extra_one_step:
            sig = sig[:-1]
        if self.inverse_timesteps:
            sig = torch.flip(sig, dims=[0])
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        if self.reverse_sigmas:
            sig = 1 - sig
        self.sigmas = sig
        self.timesteps = sig * self.num_train_timesteps

    def to(self, device):
        self.sigmas, self.timesteps = self.sigmas.to(device), self.timesteps.to(device)
        return self

    def _sigma_of(self, timestep, device):
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        self.to(device)
        idx = torch.argmin((self.timesteps.unsqueeze(0) - timestep.to(device).unsqueeze(1)).abs(), dim=1)
        return self.sigmas[idx].reshape(-1, 1, 1, 1)

    def add_noise(self, original_samples, noise, timestep):
        if (noise.is_cuda and noise.ndim == 4 and noise.dtype == torch.bfloat16 and original_samples.dtype == torch.bfloat16
                and original_samples.shape == noise.shape and timestep.ndim == 1 and timestep.dtype in _ops()._T_KIND
                and timestep.numel() == noise.shape[0]            # a 1-element timestep broadcasts in the chain below
                and self.timesteps.dtype == torch.float32 and self.sigmas.dtype == torch.float32):
            self.to(noise.device)   # one launch: lookup + blend (csrc/scheduler.hip), same roundings as the chain below
            return _ops().scheduler_step(self.timesteps, self.sigmas, x0=original_samples, noise=noise,
                                         t_next=timestep.to(noise.device).contiguous())[1]
        sigma = self._sigma_of(timestep, noise.device)
        return ((1 - sigma) * original_samples + sigma * noise).type_as(noise)

    def step(self, model_output, timestep, sample, to_final=False):
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        self.to(model_output.device)
        idx = torch.argmin((self.timesteps.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
        sigma = self.sigmas[idx].reshape(-1, 1, 1, 1)
        if to_final or bool((idx + 1 >= len(self.timesteps)).any()):
            sigma_ = 1 if (self.inverse_timesteps or self.reverse_sigmas) else 0
        else:
            sigma_ = self.sigmas[idx + 1].reshape(-1, 1, 1, 1)
        return sample + model_output * (sigma_ - sigma)


def get_denoising_schedule(timesteps, denoising_strength, steps=4):
    """v2v.py:133-136; `timesteps` is the scheduler table padded with a trailing zero."""
    lst = torch.linspace(denoising_strength * 1000, 0, steps, dtype=torch.float32, device=timesteps.device).to(torch.long)
    return timesteps[1000 - lst]
