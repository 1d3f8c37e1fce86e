"""Mirror of the reference's `CausalInferencePipeline` (pipeline/causal_inference.py:9-339): same
constructor / attribute surface / `inference(...)` contract and — the part the rest of the system
leans on — the KV-cache manager `_initialize_kv_cache` / `_initialize_crossattn_cache` with the
list-of-dicts contract (:305-311, :333-338):

    kv_cache1[i]      = {"k": [B, kv_size, H, 128], "v": same, "global_end_index": int, "local_end_index": int}
    crossattn_cache[i] = {"k": [B, 512, H, 128], "v": same, "is_init": bool}

MI355X layout: every layer's K and V live in ONE arena allocation ([L, 2, B, kv_size, H, 128], sized for
288 GB HBM: the full 32760-token cache of the 14B model is 26.8 GB); the dict entries are views, so a
reset is a single memset and the native forward receives plain base pointers.
"""
import types

import torch

from .wan_wrapper import WanDiffusionWrapper


class CausalInferencePipeline:
    def __init__(self, args, device, generator: WanDiffusionWrapper = None, text_encoder=None, vae=None):
        if generator is None:
            raise ValueError("pass a WanDiffusionWrapper (weights are loaded by the caller; there is no checkpoint lookup here)")
        self.generator, self.text_encoder, self.vae = generator, text_encoder, vae
        self.device = torch.device(device)
        self.scheduler = self.generator.get_scheduler()
        self.denoising_step_list = torch.tensor(getattr(args, "denoising_step_list", [1000, 750, 500, 250]),
                                                dtype=torch.long)
        if getattr(args, "warp_denoising_step", False):  # causal_inference.py:29-32
            timesteps = torch.cat((self.scheduler.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
            self.denoising_step_list = timesteps[1000 - self.denoising_step_list]
        self.num_transformer_blocks = len(self.generator.model.blocks)
        self.frame_seq_length = 1560
        self.kv_cache1 = None
        self.crossattn_cache = None
        self.args = args
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        self.independent_first_frame = getattr(args, "independent_first_frame", False)
        self.local_attn_size = self.generator.model.local_attn_size
        self.context_noise = getattr(args, "context_noise", 0)
        self._randn_like = torch.randn_like     # re-noising draw of inference() (causal_inference.py:209); tests inject a stream
        if self.num_frame_per_block > 1:
            self.generator.model.num_frame_per_block = self.num_frame_per_block

    # ------------------------------------------------------------------ KV-cache manager
    def _initialize_kv_cache(self, batch_size, dtype, device):
        """causal_inference.py:279-314."""
        kv_cache_size = self.local_attn_size * self.frame_seq_length if self.local_attn_size != -1 else 32760
        cfg = self.generator.model.config
        model = self.generator.model
        heads = model.kv_cache_heads() if hasattr(model, "kv_cache_heads") else cfg.num_heads
        shape = [batch_size, kv_cache_size, heads, cfg.dim // cfg.num_heads]
        if self.kv_cache1 and list(self.kv_cache1[0]["k"].shape) == shape and self.kv_cache1[0]["k"].dtype == dtype:
            # Reset.  The reference zero_()s all L x 2 tensors (causal_inference.py:296-303; 7.7 GB of HBM writes per block
            # at 14B).  The forward only ever reads rows [.., local_end_index) and writes every row of that
            # window first, so with `zero_on_reset = False` (set by GenerationSession) only the indices are reset.
            if getattr(self, "zero_on_reset", True):
                self._kv_arena.zero_()
            for c in self.kv_cache1:
                c["global_end_index"] = 0
                c["local_end_index"] = 0
            return
        L = self.num_transformer_blocks
        # K and V rows interleaved per cache row ([L, B, kv, 2, H, hd]): a block of new rows of BOTH K and V is one
        # contiguous region, i.e. one all-gather message per layer under context parallelism; the dict entries
        # stay [B, kv, H, hd] views (row stride 2*H*hd), which the attention kernel reads in place.
        b, kvs, h, hd = shape
        self._kv_arena = torch.zeros([L, b, kvs, 2, h, hd], dtype=dtype, device=device)
        self.kv_cache1 = [{"k": self._kv_arena[i, :, :, 0], "v": self._kv_arena[i, :, :, 1],
                           "global_end_index": 0, "local_end_index": 0} for i in range(L)]
        self.k_shape = self.v_shape = shape

    def _initialize_crossattn_cache(self, batch_size, dtype, device):
        """causal_inference.py:316-339."""
        cfg = self.generator.model.config
        text_len = getattr(self.generator.model, "text_len", 512)
        shape = [batch_size, text_len, cfg.num_heads, cfg.dim // cfg.num_heads]
        if self.crossattn_cache and list(self.crossattn_cache[0]["k"].shape) == shape:
            self._ca_arena.zero_()
            for c in self.crossattn_cache:
                c["is_init"] = False
            return
        L = self.num_transformer_blocks
        self._ca_arena = torch.zeros([L, 2] + shape, dtype=dtype, device=device)
        self.crossattn_cache = [{"k": self._ca_arena[i, 0], "v": self._ca_arena[i, 1], "is_init": False}
                                for i in range(L)]

    # ------------------------------------------------------------------ original Self-Forcing driver
    def inference(self, noise, text_prompts, initial_latent=None, return_latents=False, profile=False,
                  low_memory=False):
        """causal_inference.py:48-277.  noise: [B, F, 16, h, w].  Returns video in [0,1] (needs `vae`) and/or
        latents.  Per block: `len(denoising_step_list)` denoise forwards, then one forward at
        `context_noise` that writes the clean K/V into the cache (:227-236)."""
        batch_size, num_frames, num_channels, height, width = noise.shape
        nfpb = self.num_frame_per_block
        if not self.independent_first_frame or initial_latent is not None:
            assert num_frames % nfpb == 0
            num_blocks = num_frames // nfpb
        else:   # a [1, n, n, ...] model generating without image conditioning (causal_inference.py:80-83)
            assert (num_frames - 1) % nfpb == 0
            num_blocks = (num_frames - 1) // nfpb
        num_input_frames = initial_latent.shape[1] if initial_latent is not None else 0
        conditional_dict = self.text_encoder(text_prompts=text_prompts)
        output = torch.zeros([batch_size, num_frames + num_input_frames, num_channels, height, width],
                             device=noise.device, dtype=noise.dtype)
        if self.kv_cache1 is None:
            self._initialize_kv_cache(batch_size, noise.dtype, noise.device)
            self._initialize_crossattn_cache(batch_size, noise.dtype, noise.device)
        else:
            for c in self.crossattn_cache:
                c["is_init"] = False
            for c in self.kv_cache1:
                c["global_end_index"] = 0
                c["local_end_index"] = 0
        events = []
        current_start_frame = 0

        def cache_context(ref):     # Step 2: clean frames at t = 0 only fill the KV cache
            timestep = torch.zeros([batch_size, ref.shape[1]], device=noise.device, dtype=torch.int64)
            self.generator(noisy_image_or_video=ref, conditional_dict=conditional_dict, timestep=timestep,
                           kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                           current_start=current_start_frame * self.frame_seq_length)

        if initial_latent is not None:
            if self.independent_first_frame:   # 1 + nfpb * k input frames (:139-154)
                assert (num_input_frames - 1) % nfpb == 0
                num_input_blocks = (num_input_frames - 1) // nfpb
                output[:, :1] = initial_latent[:, :1]
                cache_context(initial_latent[:, :1])
                current_start_frame += 1
            else:
                assert num_input_frames % nfpb == 0
                num_input_blocks = num_input_frames // nfpb
            for _ in range(num_input_blocks):
                ref = initial_latent[:, current_start_frame:current_start_frame + nfpb]
                output[:, current_start_frame:current_start_frame + nfpb] = ref
                cache_context(ref)
                current_start_frame += nfpb
        all_num_frames = [nfpb] * num_blocks
        if self.independent_first_frame and initial_latent is None:
            all_num_frames = [1] + all_num_frames
        steps = self.denoising_step_list.to(noise.device)
        for cur in all_num_frames:
            if profile:
                s = torch.cuda.Event(enable_timing=True)
                s.record()
            lo = current_start_frame - num_input_frames
            noisy_input = noise[:, lo:lo + cur]
            for index, current_timestep in enumerate(steps):
                timestep = torch.ones([batch_size, cur], device=noise.device, dtype=torch.int64) * current_timestep
                renoise = None
                if index < len(steps) - 1:
                    # the re-noising draw (causal_inference.py:209) does not depend on the forward: drawn first so that x0
                    # and add_noise are one launch (the forward itself consumes no random numbers)
                    shape = (batch_size * cur,) + tuple(noisy_input.shape[2:])
                    like = torch.empty(shape, dtype=torch.bfloat16, device=noise.device)
                    renoise = (self._randn_like(like),
                               steps[index + 1] * torch.ones([batch_size * cur], device=noise.device, dtype=torch.long))
                res = self.generator(noisy_image_or_video=noisy_input, conditional_dict=conditional_dict,
                                     timestep=timestep, kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                                     current_start=current_start_frame * self.frame_seq_length, renoise=renoise)
                denoised_pred = res[1]
                if renoise is not None:
                    noisy_input = res[2]
            output[:, current_start_frame:current_start_frame + cur] = denoised_pred
            context_timestep = torch.ones_like(timestep) * self.context_noise
            self.generator(noisy_image_or_video=denoised_pred, conditional_dict=conditional_dict,
                           timestep=context_timestep, kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                           current_start=current_start_frame * self.frame_seq_length)
            if profile:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                events.append((s, e))
            current_start_frame += cur
        if profile:
            torch.cuda.synchronize()
            self.block_times_ms = [s.elapsed_time(e) for s, e in events]
        video = None
        if self.vae is not None:
            video = (self.vae.decode_to_pixel(output, use_cache=False) * 0.5 + 0.5).clamp(0, 1)
        elif not return_latents:
            raise RuntimeError("no VAE attached: call inference(..., return_latents=True)")
        return (video, output) if return_latents else video


def make_args(**kw):
    """Tiny stand-in for the OmegaConf node the reference passes as `args`."""
    return types.SimpleNamespace(**kw)
