// ssdk_ctx.h -- the caller-owned context of the two multi-launch entry points (ssdk_decode_nms_ctx, ssdk_run_ops_ctx).
// Everything that used to be process-global in round 1 (side stream + fork/join events of the plan executor, fork events
// and tail stream of the decode stage, the two profiling rings) lives here: the library itself is stateless apart from
// the per-thread error text, a context belongs to ONE device and is used by one host thread at a time, and two threads
// (or two devices) simply use two contexts (SURVEY.md 8b "re-entrant, stateless").  The legacy entry points without a
// context argument use a context owned by the calling thread for the current device (default_ctx()).
#pragma once
#include "ssdk_common.h"

constexpr int kSsdkProfSlots = 256;
constexpr int kSsdkMaxProfOps = 128;
// debug stamps (SSDK_TAIL_STAMPS=1): [0, 24) tail_kernel phases of workgroup 0, [24, 48) scan kernel phases of workgroup 0,
// [48, 48 + 2 * 4096) wall-clock (100 MHz) start / end of the first 4096 scan workgroups
constexpr int kSsdkStampWords = 48 + 2 * 4096;

struct ssdk_ctx {
  int device;
  // ---- decode stage (ssdk_decode_nms_ctx) ----
  hipStream_t tail_stream;      // level/NMS (or the fused tail) go here when set; scan stays on the caller's stream
  hipEvent_t tail_fork[8];
  bool tail_fork_ready;
  unsigned tail_fork_i;
  int prof_on;
  bool prof_ready;
  hipEvent_t prof_ev[kSsdkProfSlots][4];
  bool prof_fused[kSsdkProfSlots];  // the slot was recorded in stage mode (prof_on == 2): events 0 and 3 only
  long long prof_calls;
  unsigned long long* stamps;   // device, kSsdkStampWords words (SSDK_TAIL_STAMPS=1 only)
  // ---- plan executor (ssdk_run_ops_ctx) ----
  int side_lane;                // -1: environment default (SSDK_SIDE_STREAM, on), 0 off, 1 on
  hipStream_t side;
  hipEvent_t fork[32], join;
  bool side_ready;
  int op_prof, op_n;
  bool op_ev_ready;
  hipEvent_t op_ev[kSsdkMaxProfOps + 1];
  const char* op_kernel[kSsdkMaxProfOps];
};

namespace ssdk {
ssdk_ctx* default_ctx();          // the calling thread's context for the current device (created on first use), or null
int ctx_enter(ssdk_ctx* ctx);     // SSDK_OK when ctx is non-null and belongs to the current device
}  // namespace ssdk
