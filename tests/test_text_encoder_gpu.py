"""Native text encoder (rtv_t5_encode behind realtime_video_amd.text_encoder.WanTextEncoder) vs the CPU oracle and the
golden minted from the reference's T5Encoder.  Tolerance: the reference computes in float32; here the
linears take bf16-rounded activations (weights are exact, accumulation / residual stream / norms / softmax float32), so the
result differs by bf16 input rounding only: rel-L2 <= 1e-2, max-abs <= 5e-2 on outputs of magnitude ~1."""
import pytest
import torch

from conftest import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _encoder(cfg, w, text_len):
    from realtime_video_amd.text_encoder import WanTextEncoder
    enc = WanTextEncoder(device=DEV, text_len=text_len, **cfg)
    enc.load_state_dict(w)
    return enc


def test_text_encoder_matches_reference_golden(golden):
    from oracle import t5_oracle as to
    gold = golden("t5_encoder.pt")
    cfg = dict(to.TINY_T5)
    w = to.make_t5_weights(cfg, seed=0)
    enc = _encoder(cfg, w, text_len=48)
    out = enc.encode_ids(gold["ids"], gold["mask"])["prompt_embeds"]
    assert out.shape == (2, 48, cfg["dim"]) and out.dtype == torch.float32
    ref = gold["prompt_embeds"]
    assert rel_l2(out.cpu(), ref) <= 1e-2 and max_abs(out.cpu(), ref) <= 5e-2
    assert float(out[0, 29:].abs().max()) == 0.0          # padding rows are exactly zero (wan_wrapper.py:52-53)


@pytest.mark.parametrize("lens", [(512, 77), (1, 33, 300)])
def test_text_encoder_full_window_matches_oracle(lens):
    """512-slot window (the production text_len): every relative distance up to +-511, multi-block key loops, a one-token
    prompt, lengths that are not multiples of 32; 4 heads, 3 layers."""
    from oracle import t5_oracle as to
    cfg = dict(to.TINY_T5, num_heads=4, dim_attn=256, num_layers=3, vocab=300)
    w = to.make_t5_weights(cfg, seed=3)
    ids, mask = to.t5_inputs(cfg, seed=11, L=512, lens=lens)
    ref = to.text_encoder_forward(w, ids, mask, cfg)["prompt_embeds"]
    enc = _encoder(cfg, w, text_len=512)
    out = enc.encode_ids(ids, mask)["prompt_embeds"].cpu()
    for b, n in enumerate(lens):
        assert rel_l2(out[b, :n], ref[b, :n]) <= 1e-2, (b, n)
        assert max_abs(out[b, :n], ref[b, :n]) <= 5e-2
        assert float(out[b, n:].abs().max() if n < 512 else 0.0) == 0.0


def test_text_encoder_api_errors():
    from oracle import t5_oracle as to
    from realtime_video_amd.text_encoder import WanTextEncoder
    cfg = dict(to.TINY_T5)
    w = to.make_t5_weights(cfg, seed=0)
    enc = _encoder(cfg, w, text_len=48)
    ids, mask = to.t5_inputs(cfg)
    with pytest.raises(RuntimeError):
        enc(["a prompt"])                                  # no tokenizer files offline
    bad = mask.clone()
    bad[0, 3] = 0
    with pytest.raises(ValueError):
        enc.encode_ids(ids, bad)
    with pytest.raises(KeyError):
        WanTextEncoder(device=DEV, text_len=48, **cfg).load_state_dict({k: v for k, v in w.items() if k != "norm.weight"})
    with pytest.raises(RuntimeError):
        WanTextEncoder(device=DEV, text_len=48, **cfg).encode_ids(ids, mask)
