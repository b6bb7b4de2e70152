// ssdk_ctx.cpp -- contexts (ssdk_ctx.h) and the decode-stage driver ssdk_decode_nms[_ctx].
#include <mutex>
#include <unordered_map>

#include "ssdk_ctx.h"
#include "ssdk_decode.h"

namespace ssdk {

static ssdk_ctx* ctx_new() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    set_error("ctx: no HIP device");
    return nullptr;
  }
  ssdk_ctx* c = (ssdk_ctx*)calloc(1, sizeof(ssdk_ctx));
  if (!c) {
    set_error("ctx: out of memory");
    return nullptr;
  }
  c->device = dev;
  c->side_lane = -1;
  return c;
}

static void ctx_free(ssdk_ctx* c) {
  if (!c) return;
  if (c->tail_fork_ready)
    for (auto& e : c->tail_fork) (void)hipEventDestroy(e);
  if (c->prof_ready)
    for (int s = 0; s < kSsdkProfSlots; ++s)
      for (int i = 0; i < 4; ++i) (void)hipEventDestroy(c->prof_ev[s][i]);
  if (c->side_ready) {
    for (auto& e : c->fork) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->join);
    (void)hipStreamDestroy(c->side);
  }
  if (c->op_ev_ready)
    for (auto& e : c->op_ev) (void)hipEventDestroy(e);
  if (c->stamps) (void)hipFree(c->stamps);
  free(c);
}

// one default context per (host thread, device); freed when the thread ends
struct ThreadCtxs {
  std::unordered_map<int, ssdk_ctx*> by_device;
  ~ThreadCtxs() {
    // the HIP runtime may already be gone at thread / process exit: leak the handles, free the host memory
    for (auto& kv : by_device) free(kv.second);
  }
};
static thread_local ThreadCtxs g_thread_ctxs;

ssdk_ctx* default_ctx() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    set_error("no HIP device");
    return nullptr;
  }
  auto it = g_thread_ctxs.by_device.find(dev);
  if (it != g_thread_ctxs.by_device.end()) return it->second;
  ssdk_ctx* c = ctx_new();
  if (c) g_thread_ctxs.by_device[dev] = c;
  return c;
}

int ctx_enter(ssdk_ctx* ctx) {
  if (!ctx) {
    set_error("null context");
    return SSDK_E_BADARG;
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != ctx->device) {
    (void)hipGetLastError();
    set_error("context belongs to device %d, the current device is %d", ctx->device, dev);
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// everything enqueued on `from` so far happens before what is enqueued on `to` from now on
static int stream_fork(ssdk_ctx* c, hipStream_t from, hipStream_t to) {
  if (!c->tail_fork_ready) {
    for (auto& e : c->tail_fork)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        set_error("decode: hipEventCreate failed");
        return SSDK_E_LAUNCH;
      }
    c->tail_fork_ready = true;
  }
  hipEvent_t e = c->tail_fork[c->tail_fork_i++ & 7u];
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
    set_error("decode: stream fork failed");
    return SSDK_E_LAUNCH;
  }
  return SSDK_OK;
}

static int record(hipEvent_t e, hipStream_t s) {
  if (hipEventRecord(e, s) != hipSuccess) {
    set_error("decode: hipEventRecord failed");
    return SSDK_E_LAUNCH;
  }
  return SSDK_OK;
}

}  // namespace ssdk

using namespace ssdk;

extern "C" ssdk_ctx* ssdk_ctx_create(void) { return ctx_new(); }
extern "C" void ssdk_ctx_destroy(ssdk_ctx* ctx) { ctx_free(ctx); }

extern "C" int ssdk_ctx_set_tail_stream(ssdk_ctx* ctx, void* stream) {
  if (int rc = ctx_enter(ctx)) return rc;
  ctx->tail_stream = (hipStream_t)stream;
  return SSDK_OK;
}
extern "C" int ssdk_set_decode_tail_stream(void* stream) { return ssdk_ctx_set_tail_stream(default_ctx(), stream); }

extern "C" int ssdk_ctx_set_side_lane(ssdk_ctx* ctx, int enable) {
  if (int rc = ctx_enter(ctx)) return rc;
  ctx->side_lane = enable < 0 ? -1 : (enable ? 1 : 0);
  return SSDK_OK;
}

extern "C" int ssdk_ctx_set_profiling(ssdk_ctx* ctx, int enable) {
  if (int rc = ctx_enter(ctx)) return rc;
  if (enable && !ctx->prof_ready) {
    for (int s = 0; s < kSsdkProfSlots; ++s)
      for (int i = 0; i < 4; ++i)
        if (hipEventCreate(&ctx->prof_ev[s][i]) != hipSuccess) {
          set_error("set_profiling: hipEventCreate failed");
          return SSDK_E_LAUNCH;
        }
    ctx->prof_ready = true;
  }
  ctx->prof_on = enable == 2 ? 2 : (enable ? 1 : 0);  // 2: one interval around the whole stage, no event between its launches
  ctx->prof_calls = 0;
  return SSDK_OK;
}
extern "C" int ssdk_set_profiling(int enable) { return ssdk_ctx_set_profiling(default_ctx(), enable); }

// ms[0] = scan_kernel, ms[1] = tail_kernel (fused path) or level_kernel, ms[2] = nms_kernel (0 on the fused path) of
// the profiled call `back` calls before the most recent one.  Synchronises on that call's last event.
extern "C" int ssdk_ctx_get_timings(ssdk_ctx* ctx, int back, float* ms, int n) {
  if (int rc = ctx_enter(ctx)) return rc;
  if (!ms || n < 3 || back < 0 || back >= kSsdkProfSlots || (long long)back >= ctx->prof_calls) {
    set_error("get_timings: slot %d not recorded (%lld profiled calls, ring of %d)", back, ctx->prof_calls,
              kSsdkProfSlots);
    return SSDK_E_BADARG;
  }
  const long long slot = (ctx->prof_calls - 1 - back) % kSsdkProfSlots;
  hipEvent_t* ev = ctx->prof_ev[slot];
  if (hipEventSynchronize(ev[3]) != hipSuccess) return SSDK_E_LAUNCH;
  if (ctx->prof_fused[slot]) {  // stage mode: ms[0] = the whole stage (scan .. last launch), one interval
    ms[1] = ms[2] = 0.0f;
    return hipEventElapsedTime(&ms[0], ev[0], ev[3]) == hipSuccess ? SSDK_OK : SSDK_E_LAUNCH;
  }
  for (int i = 0; i < 3; ++i)
    if (hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) return SSDK_E_LAUNCH;
  return SSDK_OK;
}
extern "C" int ssdk_get_timings(int back, float* ms, int n) { return ssdk_ctx_get_timings(default_ctx(), back, ms, n); }

// (debug) SSDK_TAIL_STAMPS=1: shader-clock stamps of workgroup 0 at the phase boundaries of the last tail_kernel
// (out[0..5]: start, lists staged, winners decoded, sorted, walked, end) and scan_kernel (out[8..12]: start, cut found,
// streamed, selected, end; out[13] = fast-path flag << 32 | winners).  Synchronises the device.
extern "C" int ssdk_ctx_get_tail_stamps(ssdk_ctx* ctx, unsigned long long* out, int n) {
  if (int rc = ctx_enter(ctx)) return rc;
  if (!ctx->stamps || !out || n < 48 || n > kSsdkStampWords) {
    set_error("get_tail_stamps: no stamps (set SSDK_TAIL_STAMPS=1 before the first call)");
    return SSDK_E_BADARG;
  }
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, ctx->stamps, (size_t)n * sizeof(*out), hipMemcpyDeviceToHost) != hipSuccess)
    return SSDK_E_LAUNCH;
  return SSDK_OK;
}

extern "C" size_t ssdk_decode_nms_workspace_bytes(const ssdk_level* levels, int L, int B, int dtype,
                                                  int top_n_per_level, int ndetections) {
  DecodePlan pl;
  if (make_plan(levels, L, B, dtype, top_n_per_level, &pl, ndetections > 0 ? ndetections : 1)) return 0;
  const size_t dec = pl.cand_bytes + pl.cnt_bytes;
  const size_t n = (size_t)B * L * top_n_per_level;
  // the mid buffers are only used by the 3-launch path (or when the caller passes none of its own); always reserved:
  // the environment may switch paths between the query and the call
  return align256(dec) + align256(n * 4) + align256(n * 16) + align256(n * 4);
}

extern "C" int ssdk_decode_nms_ctx(ssdk_ctx* ctx, const ssdk_level* levels, int L, int B, int dtype, float threshold,
                                   int top_n_per_level, int rescore, float nms_threshold, int ndetections,
                                   int using_diou, float* out_scores, float* out_boxes, float* out_classes,
                                   float* mid_scores, float* mid_boxes, float* mid_classes, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (int rc = ctx_enter(ctx)) return rc;
  if (ndetections < 1 || ndetections > SSDK_MAX_NDET) {
    set_error("decode_nms: ndetections=%d outside [1, %d]", ndetections, SSDK_MAX_NDET);
    return SSDK_E_BADARG;
  }
  const int K = top_n_per_level;
  DecodePlan pl;
  int rc = make_plan(levels, L, B, dtype, K, &pl, ndetections);
  if (rc) return rc;
  const size_t dec = pl.cand_bytes + pl.cnt_bytes;
  const size_t n = (size_t)B * L * K;
  const size_t need = align256(dec) + align256(n * 4) + align256(n * 16) + align256(n * 4);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) {
    set_error("decode_nms: workspace too small or not 256-byte aligned (%zu < %zu)", workspace_bytes, need);
    return SSDK_E_WORKSPACE;
  }
  if ((size_t)L * K > SSDK_MAX_NMS_N) {
    set_error("decode_nms: L*top_n_per_level = %zu candidates per image (limit %d)", (size_t)L * K, SSDK_MAX_NMS_N);
    return SSDK_E_BADARG;
  }
  hipStream_t main_s = (hipStream_t)stream;
  const bool prof_any = ctx->prof_on && ctx->prof_ready;
  const bool prof = prof_any && ctx->prof_on == 1;  // per-launch events
  hipEvent_t* ev = prof_any ? ctx->prof_ev[ctx->prof_calls % kSsdkProfSlots] : nullptr;

  static const bool want_stamps = [] {
    const char* e = getenv("SSDK_TAIL_STAMPS");
    return e && atoi(e) != 0;
  }();
  if (want_stamps && !ctx->stamps) {
    if (hipMalloc((void**)&ctx->stamps, kSsdkStampWords * sizeof(unsigned long long)) != hipSuccess) {
      (void)hipGetLastError();
      ctx->stamps = nullptr;
    } else {
      (void)hipMemset(ctx->stamps, 0, kSsdkStampWords * sizeof(unsigned long long));
    }
  }
  if (prof_any && (rc = record(ev[0], main_s))) return rc;
  rc = launch_scan(levels, L, B, dtype, threshold, K, pl, workspace, dec, main_s, ctx->stamps ? ctx->stamps + 24 : nullptr);
  if (rc) return rc;
  if (prof && (rc = record(ev[1], main_s))) return rc;

  hipStream_t st2 = main_s;
  if (ctx->tail_stream && ctx->tail_stream != main_s) {  // the rest of the stage goes to the tail stream, after the scan
    rc = stream_fork(ctx, main_s, ctx->tail_stream);
    if (rc) return rc;
    st2 = ctx->tail_stream;
  }
  char* w = (char*)workspace + align256(dec);
  float* ms = mid_scores ? mid_scores : (float*)w;
  w += align256(n * 4);
  float* mb = mid_boxes ? mid_boxes : (float*)w;
  w += align256(n * 16);
  float* mc = mid_classes ? mid_classes : (float*)w;
  if (pl.fused) {  // levelsel_kernel (B x L workgroups: select + order + decode per level) + nmswalk_kernel (per image)
    u32 hbase, hsh;
    hist_window(threshold, &hbase, &hsh);
    rc = launch_levelsel(levels, L, B, dtype, K, rescore, pl.units, pl.unit_base, pl.units_per_image, workspace,
                         (const char*)workspace + pl.cand_bytes, hbase, hsh, ms, mb, mc, ctx->stamps, st2);
    if (rc) return rc;
    if (prof && (rc = record(ev[2], st2))) return rc;
    rc = launch_nmswalk(ms, mb, mc, B, L * K, nms_threshold, ndetections, using_diou, out_scores, out_boxes, out_classes,
                        ctx->stamps, st2);
    if (rc) return rc;
    if (prof && (rc = record(ev[3], st2))) return rc;
  } else {
    rc = launch_level(levels, L, B, dtype, K, rescore, pl, workspace, ms, mb, mc, st2);
    if (rc) return rc;
    if (prof && (rc = record(ev[2], st2))) return rc;
    rc = launch_nms(ms, mb, mc, B, L * K, nms_threshold, ndetections, using_diou, out_scores, out_boxes, out_classes,
                    st2);
    if (rc) return rc;
    if (prof && (rc = record(ev[3], st2))) return rc;
  }
  if (prof_any && !prof && (rc = record(ev[3], st2))) return rc;
  if (prof_any) {
    ctx->prof_fused[ctx->prof_calls % kSsdkProfSlots] = !prof;
    ++ctx->prof_calls;
  }
  return SSDK_OK;
}

extern "C" int ssdk_decode_nms(const ssdk_level* levels, int L, int B, int dtype, float threshold,
                               int top_n_per_level, int rescore, float nms_threshold, int ndetections,
                               int using_diou, float* out_scores, float* out_boxes, float* out_classes,
                               float* mid_scores, float* mid_boxes, float* mid_classes, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return ssdk_decode_nms_ctx(default_ctx(), levels, L, B, dtype, threshold, top_n_per_level, rescore, nms_threshold,
                             ndetections, using_diou, out_scores, out_boxes, out_classes, mid_scores, mid_boxes,
                             mid_classes, workspace, workspace_bytes, stream);
}
