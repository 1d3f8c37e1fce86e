/* Test and measurement hooks of librtv_hip.so - NOT part of the drop-in boundary (include/rtv_hip.h is; a reference-side
 * binding never needs anything declared here).  These select between implementations of one entry point that compute the same
 * result, so that tests can hold the variants against each other (bit-identity / stated tolerance) and scripts/ can time them.
 * Each switch is one process-wide std::atomic read at launch time: flipping it while another thread launches is safe and
 * affects only which variant later launches pick; none of them changes a result beyond what the declaration states.
 *
 * Experimental kernels (csrc/gemm4.hip, the timing-only tile configurations 81-91) are compiled only into the lab build
 * (`make -C realtime_video_amd/csrc LAB=1` -> realtime_video_amd/librtv_hip_lab.so, -DRTV_LAB); the product library rejects those
 * tile configurations. */
#ifndef RTV_HIP_LAB_H
#define RTV_HIP_LAB_H
#ifdef __cplusplus
extern "C" {
#endif

/* Workgroup shape / kernel of rtv_attn_fwd.  0 = by grid size and window (default): 128-row workgroups (4 waves, lockstep
 * kernel) for launches whose 256-row grid leaves most CUs idle (< 5/8 of the device's CUs) and for windows of <= 512 keys;
 * otherwise 256-row workgroups - bf16 over one row range of >= 1024 keys on the one-wave-per-SIMD kernel (r05, attn_w4.hip:
 * 4 waves x 64 rows, asm-owned accumulation registers), ring windows (two row ranges) / f16 on the four-phase kernel (8 waves x
 * 32 rows, K / V by LDS DMA, the two wave groups one phase apart), shorter windows on the lockstep kernel.
 * 4 / 8 = force the workgroup size; 81 / 82 = 256 rows on the lockstep / four-phase schedule; 840 + v = 256 rows on the
 * one-wave-per-SIMD kernel, variant v where it applies (product build: 600 = default, 200, 0; lab build: schedule + 10 x timing experiment +
 * 100 x options, attn_w4.hip).  The lockstep, four-phase and one-wave-per-SIMD (variants without option bit 0) kernels compute
 * every row with the same arithmetic in the same order: bit-identical outputs. */
int rtv_attn_set_waves(int waves);

/* gemm8 (256x256 ping-pong GEMM): tail round as 128x256 half tiles (default 1) or as K segments with an fp32 slab reduction (0).
 * Half tiles keep the unsplit summation order (bit-identical with tile config 4). */
int rtv_gemm_set_half_tail(int on);
/* gemm8: where it removes the tail round (ffn-in at M = 4680), a ragged last row of tiles (<= 128 real rows) runs as 128 x 512 strips
 * in front of the tile grid (default 1) or stays in the grid as half-empty 256-row tiles (0).  Bit-identical per output element
 * (full K through the 128-row body); the strips also replace that shape's split-K units, whose sums are re-associated. */
int rtv_gemm_set_ragged_strips(int on);
/* gemm8: waves whose 128 rows all lie beyond M run an idle loop (default 1) or the full K loop on clamped rows (0).  Bit-identical. */
int rtv_gemm_set_skip_idle(int on);
/* VAE decoder: RMS_norm + SiLU fused into the producing 96-channel conv (default 1) or as a separate pass (0); the two differ by
 * the fp32 summation order of the 96 squares (tests: pixels within 2e-3). */
int rtv_conv_set_fuse_norm(int on);
/* Four-phase attention kernel: waves whose query rows all lie beyond Lq run an idle loop (default 1) or the full loop (0).
 * Bit-identical. */
int rtv_attn_set_skip_idle(int on);
/* VAE 3x3x3 stride-1 convolutions (and the 3x3 ones behind a nearest-2x upsampling): 0 = the implicit-GEMM gather kernel everywhere;
 * 1 (default) = 6 = the persistent one-wave-per-SIMD halo-tile kernel; 2 / 3 = the one- / two-waves-per-SIMD halo kernel with one
 * workgroup per tile.  The three halo forms are bit-identical with each other (same K order, same epilogue arithmetic); the gather
 * kernel sums a pixel's taps in another order (0.03-0.25 % of the fp16 outputs differ by one rounding).  Lab build: 4 / 5 =
 * timing-only forms of 2 (no DMA in the loop / no epilogue; garbage results). */
int rtv_conv_set_halo(int on);
/* Self-attention K / V cache write: 1 (default) = the V third of the fused QKV projection runs as a GEMM launch of its own whose output
 * matrix is the call's rows of the V cache, whenever those are one physical row range (the RoPE / cache kernel then moves a third
 * less); 0 = one fused launch and the kernel copies V out of its output, the r04 form.  Same bits in the cache wherever neither form
 * splits K (tile config 4; the default dispatch cuts K by a launch's tile count). */
int rtv_dit_set_direct_v(int on);
/* RMSNorm(q,k) + RoPE + cache-write kernel: -1 = by row count (default), 0 = one 256-thread workgroup per row, 1 = two waves per row.
 * Bit-identical: both forms sum a row's squares in ONE canonical order (four accumulators per lane column, combined, then the
 * 64-lane butterfly; tests/test_kernels_gpu.py::test_qk_norm_rope_cache_forms_are_bit_identical). */
int rtv_rope_set_wave(int mode);
/* VAE encoder, first chunk of a stream (one frame on fresh caches; the T2V path re-encodes one frame per block, release_server.py:572-575):
 * 1 (default) = every causal 3x3x3 convolution runs its LAST time tap only, as a 1x3x3 convolution of the new frame - the two cached
 * slices are the zero padding of vae.py:17-36, taps 0-17 multiply zeros; 0 = the full 27-tap launch over the zero slices.
 * Bit-identical (the surviving products accumulate in the same order), a third of the matrix work. */
int rtv_vae_set_fresh_tap_skip(int on);
/* 1 when the library was built with -DRTV_LAB (experimental kernels present), else 0. */
int rtv_lab_build(void);

#ifdef __cplusplus
}
#endif
#endif
