"""ORACLE SUPPORT (test infrastructure only): cheap, deterministic stand-ins with the call contracts of the reference's
VAEDecoderWrapper / VAEEncoderWrapper / WanTextEncoder, used where a test pins the SESSION ORCHESTRATION
(release_server.py GenerationSession: which latents, noises, context frames and cache resets go where) against a golden
minted from the reference's own class.  The real VAE at 832x480 is far too slow for the CPU reference run and is pinned
separately (tests/golden/vae_decoder.pt, vae_encoder.pt); these functions are pure torch and device-agnostic so the same
code runs inside the reference session (CPU), the oracle and the native session (GPU)."""
import torch


def standin_decoder(latents, *cache):
    """Contract of VAEDecoderWrapper.forward (demo_utils/vae_block3.py:195-230): latents [1, T, 16, h, w] half ->
    (pixels [1, T', 3, 8h, 8w] float32 in [-1, 1], 55-slot cache); T' = 4T - 3 on a fresh cache, else 4T."""
    first = cache[0] is None
    z = latents.float()
    T = z.shape[1]
    rgb = torch.tanh(z[0, :, :3] * 0.5 + z[0, :, 3:6] * 0.25)                       # [T, 3, h, w]
    px = torch.nn.functional.interpolate(rgb, scale_factor=8, mode="nearest")     # [T, 3, 8h, 8w]
    px = px.repeat_interleave(4, dim=0)
    k = torch.arange(4, device=px.device, dtype=px.dtype).repeat(T).view(-1, 1, 1, 1)
    px = px * (1.0 - 0.05 * k)                                                      # the 4 sub-frames differ
    if first:
        px = px[3:]
    out = list(cache)
    out[0] = torch.ones(1, device=px.device)
    return px.unsqueeze(0), out


def standin_encoder(frames, cache, stream=False):
    """Contract of VAEEncoderWrapper.forward (vae_block3.py:122-175) for single frames: frames [1, 3, 1, H, W] half ->
    (mu [1, 16, 1, H/8, W/8], cache)."""
    x = frames.float()[0, :, 0]                                                    # [3, H, W]
    pooled = torch.nn.functional.avg_pool2d(x.unsqueeze(0), 8)[0]                   # [3, h, w]
    scale = 1.0 + 0.1 * torch.arange(16, device=x.device, dtype=x.dtype).view(16, 1, 1)
    z = pooled[torch.arange(16, device=x.device) % 3] * scale                      # [16, h, w]
    return z.view(1, 16, 1, *z.shape[1:]), cache


class StandinTextEncoder:
    """WanTextEncoder's call contract (utils/wan_wrapper.py:43-56) returning fixed embeddings."""

    def __init__(self, prompt_embeds):
        self.prompt_embeds = prompt_embeds

    def __call__(self, text_prompts=None):
        return {"prompt_embeds": self.prompt_embeds.clone()}
