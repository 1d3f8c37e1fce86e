"""Mirror of the reference's streaming block loop: `GenerationSession` (release_server.py:344-751)
restricted to the generation hot path — `init_models` (:542-560), `get_clean_context_frames`
(:563-576), `recompute_kv_cache` (:588-633), `generate_block_internal` (:636-736), the streaming
video-to-video input side `process_webcam_frames` (:489-527, frames handed in with `push_frame` instead of
the server's WebSocket queue) and `interpolate_prompt_embeds` (:459-468).  Host orchestration stays Python
on PyTorch-ROCm exactly as in the reference; every forward it issues is the native `rtv_dit_forward` / VAE
kernel path.  The web layer (FastAPI / WebSocket / JPEG encoding) is outside the hot-path scope.
"""
import types
from collections import deque

import numpy as np
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from .scheduler import FlowMatchScheduler, get_denoising_schedule
from .vae_encoder import encode_video_latent


@dataclass
class GenerateParams:
    """Fields of the reference's pydantic GenerateParams (release_server.py:315-341) used by T2V."""
    prompt: str = ""
    width: int = 832
    height: int = 480
    seed: Optional[int] = None
    strength: float = 1.0
    context_noise: float = 0.0
    keep_first_frame: bool = False
    webcam_mode: bool = False          # streaming video-to-video: incoming frames are VAE-encoded every block
    input_frames: object = None        # offline video-to-video: the decoded input video [T, 3, H, W] in [-1, 1] (the reference's
                                       # `input_video` path / URL after load_video_as_rgb, v2v.py:33-131; file decoding is out of scope)
    start_frame: object = None         # image-to-video start: PIL image or [3, H, W] tensor in [0, 1] (release_server.py:578-586)
    interp_blocks: int = -1
    kv_cache_num_frames: int = 3
    num_blocks: int = 9
    num_denoising_steps: Optional[int] = 5  # "use 4 for performance"
    timestep_shift: float = 5.0


def Models(transformer, pipeline, text_encoder=None, vae_decoder=None, vae_encoder=None):
    return types.SimpleNamespace(transformer=transformer, pipeline=pipeline, text_encoder=text_encoder,
                                 vae_decoder=vae_decoder, vae_encoder=vae_encoder)


class StaticTextEncoder:
    """Counterpart of release_server.py:125-133: returns a fixed conditioning dict (used with synthetic
    prompt embeddings — the UMT5 encoder runs once per prompt and is outside the hot path)."""

    def __init__(self, prompt_embeds):
        self.cond = {"prompt_embeds": prompt_embeds}

    def __call__(self, text_prompts=None):
        return dict(self.cond)


def resample_array(array, target_length):
    """release_server.py:59-64: resample a list to the target length by linear interpolation of indices."""
    if len(array) == target_length:
        return array
    indices = np.round(np.linspace(0, len(array) - 1, target_length)).astype(int)
    return [array[i] for i in indices]


class GenerationSession:
    def __init__(self, params: GenerateParams, models, frame_callback: Optional[Callable] = None, device="cuda"):
        self.params, self.models = params, models
        self.frame_callback = frame_callback or (lambda *a, **k: None)
        self.gpu = torch.device(device)
        self.block_idx = 0
        self.width, self.height = params.width // 8 * 8, params.height // 8 * 8
        self.latent_width, self.latent_height = self.width // 8, self.height // 8
        self.kv_cache_num_frames = params.kv_cache_num_frames
        self.num_blocks = params.num_blocks
        self.frame_context_cache = deque(maxlen=1 + (params.kv_cache_num_frames - 1) * 4)
        self.decode_vae_cache = [None] * 55
        self.encode_vae_cache = [None] * 55
        self.frame_queue = deque()                    # webcam / v2v input frames, [3, H, W] in [-1, 1] (release_server.py:470-487)
        self.interpolated_prompt_embeds = []
        self.num_frame_per_block = 3
        self.rnd = torch.Generator(self.gpu).manual_seed(params.seed if params.seed is not None else 0)
        shape = [1, self.num_blocks * self.num_frame_per_block, 16, self.latent_height, self.latent_width]
        self.all_latents = torch.zeros(shape, device=self.gpu, dtype=torch.bfloat16)
        self.noise = torch.randn(shape, device=self.gpu, dtype=torch.bfloat16, generator=self.rnd)
        self.current_start_frame = 0
        self.total_frames_sent = 0
        self.current_prompt_embeds = None
        self.resume_latents = None
        self.last_pred = None
        self.init_models(models, params)
        self.denoising_step_list = get_denoising_schedule(self.zero_padded_timesteps, params.strength,
                                                          steps=params.num_denoising_steps)
        if params.input_frames is not None:           # release_server.py:417-428
            self.setup_input_video(params.input_frames, models)
        if params.start_frame is not None:            # release_server.py:429-431
            self.setup_start_frame(params.start_frame, models)

    # release_server.py:417-428 (+ :529-540 encode_v2v)
    def setup_input_video(self, frames, models):
        """Offline video-to-video: the video's latents, noised to the first step's level with the session generator, replace
        the noise; the block count follows the video (latent frames / 3 - 1, capped by params.num_blocks)."""
        if models.vae_encoder is None:
            raise RuntimeError("input_frames needs a VAE encoder")
        s0 = self.denoising_step_list[0] / 1000
        latents, _ = encode_video_latent(models.vae_encoder, [None] * 55, frames=frames.to(self.gpu), height=self.params.height,
                                         width=self.params.width, stream=False, max_frames=None, resample_to=None)
        latents = latents[None].to(self.gpu, dtype=self.noise.dtype).movedim(1, 2)
        self.noise = (latents * (1.0 - s0) + self._randn(latents.shape) * s0).contiguous()
        self.num_blocks = min(latents.shape[1] // self.num_frame_per_block - 1, self.params.num_blocks)

    # release_server.py:578-586
    def setup_start_frame(self, image, models):
        """Image-to-video start: the image, repeated over the whole pixel context window (1 + (c-1)*4 frames), is encoded
        and becomes `resume_latents` - block 0 then recomputes the KV cache from it and generation continues behind it.
        `image`: PIL image, or the [3, H, W] tensor in [0, 1] torchvision's to_tensor would make of it."""
        if models.vae_encoder is None:
            raise RuntimeError("start_frame needs a VAE encoder")
        if not torch.is_tensor(image):
            import numpy as np
            arr = np.asarray(image.convert("RGB"))
            image = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)      # TF.to_tensor
        frame_cache_len = 1 + (self.params.kv_cache_num_frames - 1) * 4
        tensor = image.to(dtype=torch.float16).to(self.gpu).sub_(0.5).mul_(2.0)
        tensors = torch.stack([tensor] * frame_cache_len)
        latents = encode_video_latent(models.vae_encoder, [None] * 55, resample_to=16, max_frames=81, video_path_or_url=None,
                                      frames=tensors, height=self.height, width=self.width, stream=False)[0]
        self.resume_latents = latents.transpose(0, 1)[None]

    def _randn(self, shape):
        """Re-noising draw of release_server.py:692 (bf16 from the session generator); a hook so tests can
        feed the same stream to the CPU oracle."""
        return torch.randn(*shape, generator=self.rnd, device=self.gpu, dtype=torch.bfloat16)

    # release_server.py:542-560
    def init_models(self, models, params):
        pipe = models.pipeline
        # Tokens per latent frame follow the session's resolution (1560 at 832x480).  The reference hard-codes 1560
        # (causal_inference.py:35, causal_model.py:351), which at any other size leaves never-written gaps inside the attention
        # window and a wrong RoPE frame; here cache sizing, `current_start` and the block mask all use the real count, which is
        # also what makes the index-only cache reset below output-identical (every row of the window is written before it is read).
        pipe.frame_seq_length = (self.latent_height // 2) * (self.latent_width // 2)
        attn_size = params.kv_cache_num_frames + pipe.num_frame_per_block
        for block in pipe.generator.model.blocks:
            block.self_attn.local_attn_size = -1
        pipe.local_attn_size = attn_size
        pipe.zero_on_reset = False   # per-block cache reset = index reset only (output-identical, see pipeline.py)
        pipe._initialize_kv_cache(batch_size=1, dtype=torch.bfloat16, device=self.gpu)
        pipe._initialize_crossattn_cache(batch_size=1, dtype=torch.bfloat16, device=self.gpu)
        pipe.generator.model.block_mask = None
        pipe.scheduler = FlowMatchScheduler(shift=params.timestep_shift, sigma_min=0.0, extra_one_step=True)
        pipe.scheduler.set_timesteps(1000, training=True)
        st = pipe.scheduler.timesteps
        self.zero_padded_timesteps = torch.cat((st.cpu(), torch.tensor([0], dtype=torch.float32))).to(self.gpu)
        pipe.scheduler.to(self.gpu)

    # release_server.py:459-468
    def interpolate_prompt_embeds(self, models, new_prompt, interpolation_steps):
        """Blend from the current prompt embedding to the new prompt's over the next `interpolation_steps` blocks."""
        if self.current_prompt_embeds is None:
            return
        e1 = self.current_prompt_embeds
        e2 = models.text_encoder(text_prompts=[new_prompt])["prompt_embeds"].to(dtype=torch.bfloat16)
        x = torch.lerp(e1, e2, torch.linspace(0, 1, steps=interpolation_steps).unsqueeze(1).unsqueeze(2).to(e1))
        self.interpolated_prompt_embeds = list(x.chunk(interpolation_steps, dim=0))

    # release_server.py:470-487 (queue side) and :489-527
    def push_frame(self, frame):
        """Queue one input frame [3, H, W] in [-1, 1] (webcam / video-to-video mode)."""
        self.frame_queue.append(frame)

    def process_webcam_frames(self, models, idx):
        """Encode the queued input frames of this block with the streaming VAE encoder: 9 frames for block 0 (fresh
        caches: chunks 1 + 4 + 4), 12 afterwards (stream=True: 4 + 4 + 4) -> 3 latent frames.  Returns None when not
        enough frames are queued (the reference's server thread polls here)."""
        n = 9 if idx == 0 else 12
        if len(self.frame_queue) < n:
            return None
        frame_list = list(self.frame_queue)
        self.frame_queue.clear()
        frames = torch.stack(resample_array(frame_list, n)).to(self.gpu)
        latents, self.encode_vae_cache = encode_video_latent(models.vae_encoder, self.encode_vae_cache, frames=frames,
                                                             height=self.params.height, width=self.params.width,
                                                             stream=idx > 0)
        return latents

    def _randn_like(self, t):
        """torch.randn_like of release_server.py:657 (global generator there); a hook for the tests."""
        return torch.randn_like(t)

    # release_server.py:563-576
    def get_clean_context_frames(self, models):
        c = self.kv_cache_num_frames
        ctx = self.all_latents[:, :self.current_start_frame]
        if self.params.keep_first_frame or (self.block_idx - 1) * models.pipeline.num_frame_per_block < c:
            if c == 1:
                return ctx[:, :1]
            return torch.cat((ctx[:, :1], ctx[:, 1:][:, -c + 1:]), dim=1)
        tail = ctx[:, 1:][:, -c + 1:]
        if models.vae_encoder is None:
            raise RuntimeError("first-frame re-encode needs a VAE encoder (release_server.py:572-575); "
                               "set keep_first_frame=True to run without one")
        # re-encode the oldest pixel frame of the context window as a fresh first frame (release_server.py:574)
        first = encode_video_latent(models.vae_encoder, [None] * 55, resample_to=16, max_frames=81, video_path_or_url=None,
                                    frames=self.frame_context_cache[0][0].half(), height=self.height, width=self.width,
                                    stream=False)[0].transpose(0, 1)[None]          # [1, 1, 16, h, w]
        return torch.cat((first, tail), dim=1).to(self.all_latents)

    # release_server.py:588-633
    def recompute_kv_cache(self, models):
        pipe = models.pipeline
        if self.block_idx == 0:
            pipe._initialize_kv_cache(batch_size=1, dtype=torch.bfloat16, device=self.gpu)
            if self.resume_latents is not None:
                self.current_start_frame = self.resume_latents.shape[1]
                self.all_latents[:, :self.current_start_frame] = self.resume_latents
            else:
                return self.current_start_frame
        for block in pipe.generator.model.blocks:
            block.self_attn.num_frame_per_block = pipe.num_frame_per_block
        start = min(self.current_start_frame, self.params.kv_cache_num_frames)
        ctx = self.get_clean_context_frames(models)
        pipe._initialize_kv_cache(batch_size=ctx.shape[0], dtype=ctx.dtype, device=ctx.device)
        model = pipe.generator.model
        model.block_mask = model._prepare_blockwise_causal_attn_mask(
            device=str(ctx.device), num_frames=ctx.shape[1], frame_seqlen=pipe.frame_seq_length,
            num_frame_per_block=pipe.num_frame_per_block, local_attn_size=-1)
        t0 = torch.zeros([ctx.shape[0], ctx.shape[1]], device=ctx.device, dtype=torch.int64)
        # the pass only fills the KV cache, its output is discarded here as in the reference: the native model may stop behind
        # the last layer's cache write (everything after it - 2 % of the forward - is dead work)
        try:
            models.transformer(noisy_image_or_video=ctx, conditional_dict=self.conditional_dict, timestep=t0,
                               kv_cache=pipe.kv_cache1, crossattn_cache=pipe.crossattn_cache,
                               current_start=start * pipe.frame_seq_length, kv_cache_only=True)
        finally:
            model.block_mask = None
        return start

    # release_server.py:636-736
    @torch.inference_mode()
    def generate_block_internal(self, models):
        idx = self.block_idx
        if idx >= self.num_blocks:
            return None
        pipe = models.pipeline
        nfpb = pipe.num_frame_per_block
        if self.current_prompt_embeds is None:
            self.conditional_dict = models.text_encoder(text_prompts=[self.params.prompt])
            for k, v in self.conditional_dict.items():
                self.conditional_dict[k] = v.to(dtype=torch.bfloat16).contiguous()
            self.current_prompt_embeds = self.conditional_dict["prompt_embeds"]
        start = self.recompute_kv_cache(models)
        steps = self.denoising_step_list
        if self.params.webcam_mode:   # :651-657: start from the encoded input frames, noised to the first step's level
            latents = self.process_webcam_frames(models, idx)
            if latents is None:
                return None
            strength = steps[0] / 1000.0
            latents = latents[None].to(self.gpu, dtype=self.noise.dtype).movedim(1, 2)
            noisy_input = latents * (1.0 - strength) + self._randn_like(latents) * strength
        else:
            noisy_input = self.noise[:, self.current_start_frame:self.current_start_frame + nfpb]
        if self.interpolated_prompt_embeds:   # :662-666: prompt transition -> new text K/V for the cross-attention
            pipe._initialize_crossattn_cache(batch_size=1, dtype=torch.bfloat16, device=self.gpu)
            nxt = self.interpolated_prompt_embeds.pop(0)
            self.current_prompt_embeds = nxt.to(dtype=self.current_prompt_embeds.dtype,
                                                device=self.current_prompt_embeds.device)
        self.conditional_dict["prompt_embeds"] = self.current_prompt_embeds
        denoised_pred = None
        for index, current_timestep in enumerate(steps):
            timestep = torch.ones([1, nfpb], device=self.gpu, dtype=torch.int64) * current_timestep
            renoise = None
            if index < len(steps) - 1:
                # release_server.py:688-694; the draw does not depend on the forward, so it is made first and x0 +
                # add_noise run as one launch behind the model
                renoise = (self._randn((nfpb,) + tuple(noisy_input.shape[2:])),
                           steps[index + 1] * torch.ones([nfpb], device=self.gpu, dtype=torch.long))
            res = models.transformer(
                noisy_image_or_video=noisy_input, conditional_dict=self.conditional_dict, timestep=timestep,
                kv_cache=pipe.kv_cache1, crossattn_cache=pipe.crossattn_cache,
                current_start=start * pipe.frame_seq_length, renoise=renoise)
            denoised_pred = res[1]
            if renoise is not None:
                noisy_input = res[2]
        self.all_latents[:, self.current_start_frame:self.current_start_frame + nfpb] = denoised_pred
        self.last_pred = denoised_pred
        pixels = None
        if models.vae_decoder is not None:
            pixels, self.decode_vae_cache = models.vae_decoder(denoised_pred.half(), *self.decode_vae_cache)
            self.frame_context_cache.extend(pixels.split(1, dim=1))
            if idx == 0:
                pixels = pixels[:, 3:]  # the first block yields 9 frames, 3 are dropped (:722-723)
            self.most_recent_frame = pixels[:, -1:].clone()
            event = torch.cuda.Event()
            event.record()
            self.frame_callback(pixels, [], event)
            self.total_frames_sent += pixels.shape[1]
        self.current_start_frame += nfpb
        self.block_idx += 1
        self.resume_latents = None
        return pixels if pixels is not None else denoised_pred

    def generate_block(self, models=None):
        out = self.generate_block_internal(models or self.models)
        if out is None:
            raise StopIteration("all blocks generated")
        return out
