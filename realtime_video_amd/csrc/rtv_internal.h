// Host-side internals shared by the launchers: error state, launch checks, event profiling.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/rtv_hip.h"

namespace rtv {

int set_error(int code, const char* msg);  // records message, returns code (never 0)
int check_launch(const char* what);        // hipGetLastError -> set_error

enum ProfClass { PROF_GEMM = 0, PROF_ATTN, PROF_LN, PROF_ROPE, PROF_CONV, PROF_MISC, PROF_NCLASS };

// Brackets one kernel launch with hipEvents on its stream when profiling is enabled
// (rtv_prof_enable); otherwise free.
struct ProfScope {
  ProfScope(int cls, hipStream_t stream, double work);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

struct GemmParams;
int launch_gemm(const GemmParams& p, int dtype, int tile_cfg, hipStream_t stream);
int launch_gemm8(const GemmParams& p, bool f16, int variant, hipStream_t stream);  // 256x256 ping-pong variant (gemm8.hip)
int launch_gemm9(const GemmParams& p, bool f16, bool split, int dist, hipStream_t stream);
int launch_gemm10(const GemmParams& p, bool f16, int abl, hipStream_t stream);  // 256x256, 4 waves of 128x128 (gemm10.hip)  // 256x256 software-pipelined variant (gemm9.hip)

}  // namespace rtv
