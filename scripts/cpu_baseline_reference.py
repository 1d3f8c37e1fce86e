"""ORACLE SUPPORT (authoring container only: needs /root/reference) - the bench's `cpu_baseline` sample timed on the
REFERENCE ITSELF next to the oracle port, on the same cores.

bench.py's cpu_baseline leg runs on the GPU box, where the Python reference does not exist, so it times the port
(oracle/wan_oracle.attention_block, oracle/vae_oracle.decoder_wrapper_forward) and says `kind: "port"`.  This script runs the
same two samples on the upstream modules - `CausalWanAttentionBlock` (wan/modules/causal_model.py:440-492) at the 14B width
with M = 4680 query tokens over 9360 cached keys, and `VAEDecoderWrapper` (demo_utils/vae_block3.py:195-230) for one latent
frame at 480 x 832 on warm caches - and on the port, interleaved, and prints both with their ratio.  The output is committed
as profiles/r03_cpu_baseline_reference_vs_port.txt: it shows that the port times what the reference costs.

    python scripts/cpu_baseline_reference.py [--no-vae]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, vae_oracle as vo, wan_oracle as wo  # noqa: E402


def main():
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    ref = ref_shim.load()
    d, ffn, H = 5120, 13824, 40
    cfg = dict(dim=d, ffn_dim=ffn, num_heads=H, num_layers=1, freq_dim=256, text_len=512, eps=1e-6, num_frame_per_block=3)
    w = wo.make_weights(cfg, seed=0, text_dim=256)
    model = ref_shim.build_reference_model(ref, cfg, w, 256)
    blk = model.blocks[0]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4680, d, generator=g).to(torch.bfloat16)
    e = torch.randn(1, 3, 6, d, generator=g).to(torch.bfloat16) * 0.1
    ctx = torch.randn(1, 512, d, generator=g).to(torch.bfloat16)
    freqs = wo.rope_table(128)
    seq_lens = torch.tensor([4680])
    grid = torch.tensor([[3, 30, 52]])

    def caches():
        kv = wo.initialize_kv_cache(1, 1, 9360, H, 128, torch.bfloat16)[0]
        kv["k"][:, :4680].normal_(generator=g)
        kv["v"][:, :4680].normal_(generator=g)
        kv["global_end_index"] = kv["local_end_index"] = 4680
        return kv, wo.initialize_crossattn_cache(1, 1, H, 128, torch.bfloat16)[0]

    times = {"reference": [], "port": []}
    outs = {}
    with torch.inference_mode():
        for rep in range(3):
            kv, ca = caches()
            t0 = time.perf_counter()
            outs["reference"] = blk(x, e, seq_lens, grid, model.freqs, ctx, None, block_mask=None, kv_cache=kv,
                                    crossattn_cache=ca, current_start=4680)
            times["reference"].append(time.perf_counter() - t0)
            kv, ca = caches()
            t0 = time.perf_counter()
            outs["port"] = wo.attention_block(w, "blocks.0", x, e, (3, 30, 52), freqs, ctx, H, kv, ca, 4680, False)
            times["port"].append(time.perf_counter() - t0)
    rel = float((outs["port"].double() - outs["reference"].double()).norm() / outs["reference"].double().norm())
    print(f"host: {cores} cores usable, torch {torch.__version__}, threads {torch.get_num_threads()}")
    print(f"DiT layer (d={d}, ffn={ffn}, H={H}, M=4680, 9360 cached keys, bf16), runs 2-3 of 3, seconds:")
    names = {"reference": "CausalWanAttentionBlock.forward (upstream)", "port": "wan_oracle.attention_block            "}
    for k in ("reference", "port"):
        print(f"  {k:9s} {names[k]} : " + "  ".join(f"{t:.2f}" for t in times[k][1:]))
    r = min(times["port"][1:]) / min(times["reference"][1:])
    print(f"  port / reference = {r:.2f}; outputs rel-L2 {rel:.2e}")

    if "--no-vae" not in sys.argv:
        wv = vo.make_vae_weights(seed=0)
        dec = ref.vae_block3.VAEDecoderWrapper().eval()
        dec.load_state_dict(wv, strict=False)
        z = [torch.randn(1, 1, 16, 60, 104, generator=g) for _ in range(2)]
        with torch.inference_mode():
            _, c_ref = dec(z[0], *([None] * 55))
            _, c_port = vo.decoder_wrapper_forward(wv, z[0], [None] * 55)
            t0 = time.perf_counter()
            p_ref, _ = dec(z[1], *c_ref)
            t_ref = time.perf_counter() - t0
            t0 = time.perf_counter()
            p_port, _ = vo.decoder_wrapper_forward(wv, z[1], c_port)
            t_port = time.perf_counter() - t0
        print("VAE decoder, one latent frame (4 pixel frames) at 480x832 on warm caches, fp32, seconds:")
        print(f"  reference VAEDecoderWrapper.forward : {t_ref:.1f}")
        print(f"  port      decoder_wrapper_forward   : {t_port:.1f}")
        print(f"  port / reference = {t_port / t_ref:.2f}; pixels max-abs diff {float((p_ref - p_port).abs().max()):.2e}")


if __name__ == "__main__":
    main()
