// Host-side internals shared by the launchers: error state, launch checks, event profiling.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/rtv_hip.h"
#include "../../include/rtv_hip_lab.h"

namespace rtv {

int set_error(int code, const char* msg);  // records message, returns code (never 0)
int check_launch(const char* what);        // hipGetLastError -> set_error

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, DEVICE): a process that drives several GPUs (the reference's
// GenerationSession.to(gpu) replica pattern, release_server.py:438-454) must set it on each of them.  One LdsAttr per kernel
// remembers the devices (bit d) it has been set on; the check is one thread-local hipGetDevice per launch.
struct LdsAttr {
  std::atomic<uint64_t> done{0};
};
int ensure_dynamic_lds(const void* kernel, int bytes, LdsAttr* state, const char* what);

// CUs of the current device (cached per device; 0 when the device cannot be queried).  Every dispatch rule that talks about
// "rounds" of workgroups uses THIS number - gemm.hip's tile choice, plan_split_k, the attention split planner - not a literal 256.
int device_num_cus();

enum ProfClass { PROF_GEMM = 0, PROF_ATTN, PROF_LN, PROF_ROPE, PROF_CONV, PROF_MISC, PROF_NCLASS };

// Brackets one kernel launch with hipEvents on its stream when profiling is enabled
// (rtv_prof_enable); otherwise free.
struct ProfScope {
  ProfScope(int cls, hipStream_t stream, double work);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

// Which kernel VARIANT a dispatch rule chose, counted per launch (rtv_dispatch_counts): bench.py records the variants that ran in
// its JSON line, so that a dispatch regression - a shape silently falling back to an older kernel - shows in the driver's numbers
// (VERDICT r05 item 8).  One relaxed atomic increment per launch.
enum DispatchKernel {
  DK_GEMM_128x128 = 0, DK_GEMM_OTHER_CFG, DK_GEMM8_256x256, DK_GEMM8M_128x256, DK_GEMM5_160x256, DK_GEMM_FP8_256x256,
  DK_ATTN_W4, DK_ATTN_FOUR_PHASE, DK_ATTN_LOCKSTEP_256ROW, DK_ATTN_LOCKSTEP_128ROW, DK_ATTN_SPLIT_COMBINE,
  DK_CONV_HALO4P, DK_CONV_HALO4, DK_CONV_HALO, DK_CONV_IGEMM, DK_COUNT
};
void note_kernel(int id);

struct GemmParams;
int launch_gemm(const GemmParams& p, int dtype, int tile_cfg, hipStream_t stream);
int launch_gemm8(const GemmParams& p, bool f16, bool split_k, hipStream_t stream);  // 256x256 ping-pong kernel (gemm8.hip)
int launch_gemm4(const GemmParams& p, bool f16, hipStream_t stream, int lab = 0);   // 256x256, one wave per SIMD, 128x128 per wave (gemm4.hip)
int launch_gemm8m(const GemmParams& p, bool f16, bool split_k, hipStream_t stream); // its 128x256 variant for few-row problems
int launch_gemm5(const GemmParams& p, bool f16, bool split_k, hipStream_t stream);  // 160x256, one wave per SIMD (gemm5.hip): token shards
// fp8 (e4m3) path, gemm_fp8.hip: dynamic per-tensor activation quantisation + 256x256 fp8 GEMM with the bf16 epilogue
int launch_quantize_fp8(const uint16_t* x, int64_t ld, int M, int d, uint8_t* q, int64_t ldq, float* scale_out,
                        unsigned* amax_scratch, hipStream_t stream);
int launch_gemm_fp8(GemmParams p, const uint8_t* A, int lda, const uint8_t* W, int ldw, const float* a_scale, float w_scale,
                    hipStream_t stream);

// elementwise.hip: rtv_qk_norm_rope_cache with the optional head-group scatter, and its inverse for the attention output
int qk_norm_rope_check(int64_t cache_row_stride, int cache_row0, int M, int d, int num_heads, int F, int gh, int gw, int start_frame,
                       int row_offset, int group_cols, int64_t q_group_stride, int64_t kv_group_stride, int ring_lo, int ring_size,
                       int ring_shift, int parts);
int qk_norm_rope_launch(const void* qkv, void* q_out, void* k_cache, void* v_cache, int64_t cache_row_stride,
                        int cache_row0, int M, int d, int num_heads, float eps, const void* wq, const void* wk,
                        const void* rope_cs, int F, int gh, int gw, int start_frame, int row_offset, int group_cols,
                        int64_t q_group_stride, int64_t kv_group_stride, int ring_lo, int ring_size, int ring_shift,
                        int parts /* 1 = q, 2 = k and v, 3 = all */, rtv_stream_t stream);
int regroup_heads(const void* in, void* out, int rows, int G, int group_cols, rtv_stream_t stream);

// vae_conv.hip: the last time tap of a 3x3x3 causal convolution as a 1x3x3 convolution of one new frame (fresh streams: the two
// cached slices are zeros), weight used in place; gamma != null: with the fused RMS_norm + SiLU epilogue (returns 1 where the layer
// has none).  Bit-identical with the full launch.
int conv3_last_tap(const void* frame, const void* w3, const void* bias, const void* gamma, const void* residual, int res_ld, void* out,
                   int out_ld, int H, int W, int Cin, int Cout, int flags, const void* zeros, hipStream_t stream);

}  // namespace rtv
