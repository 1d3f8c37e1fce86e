"""Mirror of the reference's `WanDiffusionWrapper` (utils/wan_wrapper.py:121-323) for the causal
inference path: `forward(noisy_image_or_video[B,F,16,h,w], conditional_dict, timestep[B,F], kv_cache,
crossattn_cache, current_start)` -> `(flow_pred, pred_x0)`, flow->x0 conversion in float64
(:181-205), scheduler binding (:148-151, :303-323)."""
import torch

from .causal_model import CausalWanModel
from .scheduler import FlowMatchScheduler


class WanDiffusionWrapper:
    def __init__(self, model: CausalWanModel, timestep_shift=8.0, is_causal=True, local_attn_size=-1, sink_size=0):
        if not is_causal:
            raise NotImplementedError("only the causal model is on the hot path")
        self.model = model
        self.uniform_timestep = not is_causal
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.seq_len = 32760

    def eval(self):
        return self

    def get_scheduler(self):
        return self.scheduler

    def _convert_flow_pred_to_x0(self, flow_pred, xt, timestep):
        dt = flow_pred.dtype
        sch = self.scheduler.to(flow_pred.device)
        fp, x, sig, ts = flow_pred.double(), xt.double(), sch.sigmas.double(), sch.timesteps.double()
        idx = torch.argmin((ts.unsqueeze(0) - timestep.to(flow_pred.device).unsqueeze(1)).abs(), dim=1)
        return (x - sig[idx].reshape(-1, 1, 1, 1) * fp).to(dt)

    def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache=None, crossattn_cache=None,
                current_start=None, cache_start=None, **unused):
        if kv_cache is None:
            raise NotImplementedError("non-cached (training / bidirectional) forwards are out of scope")
        flow = self.model(noisy_image_or_video.permute(0, 2, 1, 3, 4), t=timestep,
                          context=conditional_dict["prompt_embeds"], seq_len=self.seq_len, kv_cache=kv_cache,
                          crossattn_cache=crossattn_cache, current_start=current_start,
                          cache_start=cache_start).permute(0, 2, 1, 3, 4)
        x0 = self._convert_flow_pred_to_x0(flow.flatten(0, 1), noisy_image_or_video.flatten(0, 1),
                                           timestep.flatten(0, 1)).unflatten(0, flow.shape[:2])
        return flow, x0

    __call__ = forward
