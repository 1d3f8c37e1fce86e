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
    """Contract of VAEEncoderWrapper.forward (vae_block3.py:122-175): frames [1, 3, T, H, W] half -> (mu [1, 16, T', H/8, W/8],
    cache).  Fresh / non-streamed calls take 1 + 4k frames (first frame alone, then groups of 4), streamed calls 4k frames."""
    x = frames.float()[0]                                                          # [3, T, H, W]
    T = x.shape[1]
    if stream:
        assert T % 4 == 0, T
        groups = [range(4 * i, 4 * i + 4) for i in range(T // 4)]
    else:
        assert (T - 1) % 4 == 0, T
        groups = [range(0, 1)] + [range(1 + 4 * i, 5 + 4 * i) for i in range((T - 1) // 4)]
    scale = 1.0 + 0.1 * torch.arange(16, device=x.device, dtype=x.dtype).view(16, 1, 1)
    idx = torch.arange(16, device=x.device) % 3
    out = []
    for g in groups:
        m = x[:, list(g)].mean(dim=1)                                              # [3, H, W]
        pooled = torch.nn.functional.avg_pool2d(m.unsqueeze(0), 8)[0]               # [3, h, w]
        out.append(pooled[idx] * scale)
    z = torch.stack(out, dim=1)                                                    # [16, T', h, w]
    return z.unsqueeze(0), cache


class StandinVAE:
    """WanVAEWrapper.decode_to_pixel's contract (utils/wan_wrapper.py): latents [B, T, 16, h, w] -> pixels [B, T', 3, 8h, 8w]
    in [-1, 1] with T' = 4T - 3 (whole-sequence decode, no cache)."""

    def decode_to_pixel(self, latents, use_cache=False):
        px, _ = standin_decoder(latents[:1], *([None] * 55))
        return px


class StandinTextEncoder:
    """WanTextEncoder's call contract (utils/wan_wrapper.py:43-56) returning fixed embeddings; `by_prompt` maps specific
    prompt strings to other embeddings (prompt transitions)."""

    def __init__(self, prompt_embeds, by_prompt=None):
        self.prompt_embeds = prompt_embeds
        self.by_prompt = by_prompt or {}

    def __call__(self, text_prompts=None):
        e = self.by_prompt.get(text_prompts[0] if text_prompts else None, self.prompt_embeds)
        return {"prompt_embeds": e.clone()}
