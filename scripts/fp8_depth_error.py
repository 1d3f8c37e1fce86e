"""fp8 weight path (BASELINE config 5, release_server.py:179-182) at 14B width: error of one session block (four denoise forwards)
against the fp8 oracle and the bf16 oracle for a ladder of depths - the fp8 line beside profiles/r04_depth_error_curve_14b.txt
(VERDICT r04 item 1b).  Usage (GPU box):  python scripts/fp8_depth_error.py > profiles/r05_fp8_depth_error_14b.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_depth_gpu as td  # noqa: E402


def main():
    ladder = [int(a) for a in sys.argv[1:]] or [1, 2, 8, 16, 40]
    print("# fp8 e4m3 linears (per-tensor dynamic activation scale, per-tensor weight scale), 14B width (d=5120, H=40, ffn=13824), one block of")
    print("# the session loop = four denoise forwards (M=4680, window growing to 4680 rows) + scheduler steps; latents [1,3,16,60,104]")
    print("# L  rel_l2(ours_fp8, oracle_fp8)  max_abs(ours_fp8, oracle_fp8)  rel_l2(ours_fp8, oracle_bf16)  rel_l2(oracle_fp8, oracle_bf16)  "
          "rel_l2(K last layer: ours, oracle_fp8)  rel_l2(K last layer: oracle_fp8, oracle_bf16)")
    for L in ladder:
        r = td.run_fp8_depth_case(L)
        print(f"{L:3d}  {r['rel_l2_vs_fp8_oracle']:.3e}  {r['max_abs_vs_fp8_oracle']:.3e}  {r['rel_l2_vs_bf16_oracle']:.3e}  "
              f"{r['fp8_oracle_rel_l2_vs_bf16_oracle']:.3e}  {r['k_last_layer_rel_l2']:.3e}  {r['k_last_layer_fp8_oracle_vs_bf16_oracle']:.3e}", flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
