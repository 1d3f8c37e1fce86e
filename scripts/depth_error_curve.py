"""Error-vs-depth curve of the native DiT forward at production width.

For L in a ladder up to the full stack, ONE denoise forward (M = 4680 query tokens at cache offset 4680 over a 9360-row window
whose first half holds earlier K/V; the L-layer models share the first L layers of one set of device weights) is run through
  (a) the native path (rtv_dit_forward),
  (b) the bf16 oracle graph (oracle/wan_oracle.py = the reference's eager arithmetic) evaluated by torch on the device,
  (c) the fp32 gold graph,
and rel-L2 / max-abs of the flow output are tabulated: ours vs oracle, ours vs gold, oracle vs gold.  Then the two-block session
case of tests/test_depth_gpu.py at full depth.  Usage (GPU box):  python scripts/depth_error_curve.py [14b|1.3b] > profiles/...txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_depth_gpu as td  # noqa: E402
from conftest import max_abs, rel_l2  # noqa: E402
from oracle import wan_oracle as wo  # noqa: E402

DEV = "cuda"


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "14b"
    full, wr_full, cfg_full = td.native_model(arch, text_dim=4096, seed=0)
    Lmax = cfg_full["num_layers"]
    H = cfg_full["num_heads"]
    ladder = [l for l in (1, 2, 4, 8, 12, 16, 24, 32, 40) if l < Lmax] + [Lmax]
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    ctx = torch.randn(64, 4096, generator=g).to(torch.bfloat16).to(DEV)
    t = torch.tensor([[713.0, 713.0, 713.0]], device=DEV)
    old_k = torch.randn(1, 4680, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    old_v = torch.randn(1, 4680, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    sd = td.reference_state_dict(full)
    sd32 = {k: v.float() for k, v in sd.items()}
    print(f"# depth error curve, {arch}: d={cfg_full['dim']} H={H} ffn={cfg_full['ffn_dim']}, one denoise forward, M=4680 at cache "
          f"offset 4680, window 9360 rows; flow output [1,3,16,60,104]")
    print("# L  rel_l2(ours,oracle_bf16)  rel_l2(ours,gold_fp32)  rel_l2(oracle_bf16,gold_fp32)  max_abs(ours,gold)  "
          "max_abs(oracle,gold)  rel_l2(K_new ours,oracle) last layer")
    for L in ladder:
        model, wr, cfg = td.native_model(arch, text_dim=4096, num_layers=L, share=full)

        def prefilled(dtype):
            kv = wo.initialize_kv_cache(L, 1, 9360, H, 128, dtype, DEV)
            for c in kv:
                c["k"][:, :4680], c["v"][:, :4680] = old_k.to(dtype), old_v.to(dtype)
                c["global_end_index"] = c["local_end_index"] = 4680
            return kv, wo.initialize_crossattn_cache(L, 1, H, 128, dtype, device=DEV)

        with torch.inference_mode():
            kvr, car = prefilled(torch.bfloat16)
            ref, _ = wo.wrapper_forward(sd, cfg, wo.FlowMatchScheduler(), lat, [ctx], t, kvr, car, 4680)
            kvg, cag = prefilled(torch.float32)
            gold, _ = wo.wrapper_forward(sd32, cfg, wo.FlowMatchScheduler(), lat.float(), [ctx.float()], t, kvg, cag, 4680,
                                         attn_fn=td.fp32_attention)
            del kvg, cag
        kv, ca = prefilled(torch.bfloat16)
        flow, _ = wr(lat, {"prompt_embeds": [ctx]}, t, kv, ca, current_start=4680)
        print(f"{L:3d}  {rel_l2(flow, ref):.3e}  {rel_l2(flow, gold):.3e}  {rel_l2(ref, gold):.3e}  {max_abs(flow, gold):.3e}  "
              f"{max_abs(ref, gold):.3e}  {rel_l2(kv[L - 1]['k'][0, 4680:], kvr[L - 1]['k'][0, 4680:]):.3e}", flush=True)
        del kv, ca, kvr, car, model, wr
        torch.cuda.empty_cache()
    del sd32, full, wr_full, sd
    torch.cuda.empty_cache()
    r = td.run_depth_case(arch, blocks=2, gold=True)
    print(f"# session, full depth ({Lmax} layers), 2 blocks (block 1 = KV-recompute + 4 denoise steps): per block")
    for b, e in enumerate(r["blocks"]):
        print(f"block {b}: " + "  ".join(f"{k}={v:.3e}" for k, v in e.items()))
    print("kv rows rel-L2 (layer: k, v): " + "  ".join(f"{l}: {a:.2e}, {b_:.2e}" for l, (a, b_) in r["kv"].items()))
    print(f"cache indices exact: {r['indices'] == r['ref_indices']}")


if __name__ == "__main__":
    main()
