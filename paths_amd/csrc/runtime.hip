// Error reporting + build info for libpaths_hip.so.  No global mutable state besides the thread-local
// error string; no allocation, no synchronisation (include/paths_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";
static thread_local hipEvent_t g_stop_event = nullptr;
hipEvent_t paths_take_stop_event(void) { hipEvent_t ev = g_stop_event; g_stop_event = nullptr; return ev; }

int paths_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" {
const char* paths_last_error(void) { return g_err; }
const char* paths_build_info(void) { return "paths_hip gfx950 split-operand MFMA r2"; }
// 2 (round 5): dropout masks are one 16-bit hash half per element since round 4 (thr16 = round(p 65536), scale 1 / (1 - thr16 / 65536):
// same (key, p) -> other masks than ABI 1), events of paths_event_create are device-side joins only (no system-scope fence: not for
// host waits), paths_stop_event_pending / paths_record_event / paths_clear_stop_event added.
int paths_abi_version(void) { return 2; }

// ---- stream plumbing for the launch tape (paths_amd/utils.py:TapedRecursion): the recorded launch sequence of a recursion is
// replayed as a flat list of C calls, so its cross-stream joins and zero fills are C calls too.
// An event handle for paths_stream_wait (host object, created once per join of a tape, destroyed by paths_event_destroy).
void* paths_event_create(void) {
  hipEvent_t ev = nullptr;
  // (device-side joins of streams of ONE device only: no system-scope fence when the event is recorded - the kernel that carries it as its
  // stop event would otherwise end with a release to the system.  PATHS_EVENT_FLAGS overrides, for A/B runs.)
  static const unsigned flags = getenv("PATHS_EVENT_FLAGS") ? (unsigned)strtoul(getenv("PATHS_EVENT_FLAGS"), nullptr, 0)
                                                            : (hipEventDisableTiming | hipEventDisableSystemFence);
  if (hipEventCreateWithFlags(&ev, flags) != hipSuccess) { paths_set_error(PATHS_ELAUNCH, "event_create failed"); return nullptr; }
  return ev;
}
// Destroy an event made by paths_event_create (the launch tape destroys its events when it is closed / collected).
int paths_event_destroy(void* event) {
  PATHS_REQUIRE(event != nullptr, "event_destroy: null event");
  if (hipEventDestroy(static_cast<hipEvent_t>(event)) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "event_destroy: hipEventDestroy failed");
  return PATHS_OK;
}
// dst waits for everything enqueued on src so far (hipEventRecord + hipStreamWaitEvent: no host synchronisation)
int paths_stream_wait(hipStream_t dst, hipStream_t src, void* event) {
  PATHS_REQUIRE(event != nullptr, "stream_wait: null event");
  if (hipEventRecord(static_cast<hipEvent_t>(event), src) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "stream_wait: hipEventRecord failed");
  if (hipStreamWaitEvent(dst, static_cast<hipEvent_t>(event), 0) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "stream_wait: hipStreamWaitEvent failed");
  return PATHS_OK;
}
// ---- stop events (common.h: PATHS_LAUNCH_STOP).  paths_set_stop_event(ev): the next stop-capable launch of this host thread (the
// importance / projection finish kernel, the top-K kernel) carries `ev` as its completion event; paths_flush_stop_event(src): if no
// launch took it, it is recorded on `src` the ordinary way (so a waiter never waits on an event nobody recorded);
// paths_stream_wait_event(dst, ev): dst waits for ev.
int paths_set_stop_event(void* event) {
  PATHS_REQUIRE(event != nullptr, "set_stop_event: null event");
  g_stop_event = static_cast<hipEvent_t>(event);
  return PATHS_OK;
}
int paths_flush_stop_event(hipStream_t src) {
  hipEvent_t ev = g_stop_event;
  g_stop_event = nullptr;
  if (ev != nullptr && hipEventRecord(ev, src) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "flush_stop_event: hipEventRecord failed");
  return PATHS_OK;
}
// 1 while an event armed by paths_set_stop_event has not been taken by a launch (nor flushed), else 0
int paths_stop_event_pending(void) { return g_stop_event != nullptr ? 1 : 0; }
// Disarm without recording (error paths of a tape replay: a later, unrelated stop-capable launch must not carry a stale event)
int paths_clear_stop_event(void) { g_stop_event = nullptr; return PATHS_OK; }
// hipEventRecord(event, stream): the ordinary record, for a join that must also cover launches enqueued AFTER the stop-capable kernel
int paths_record_event(void* event, hipStream_t stream) {
  PATHS_REQUIRE(event != nullptr, "record_event: null event");
  if (hipEventRecord(static_cast<hipEvent_t>(event), stream) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "record_event: hipEventRecord failed");
  return PATHS_OK;
}
int paths_stream_wait_event(hipStream_t dst, void* event) {
  PATHS_REQUIRE(event != nullptr, "stream_wait_event: null event");
  if (hipStreamWaitEvent(dst, static_cast<hipEvent_t>(event), 0) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "stream_wait_event: hipStreamWaitEvent failed");
  return PATHS_OK;
}
// A HIP stream restricted to the compute units whose bits are set in cu_mask (words x 32 bits; hipExtStreamCreateWithCUMask): the
// A/B of VERDICT r3 1c (aggregator stream on a subset of the CUs, PATHS_AGG_CU_MASK in paths_amd/utils.py).  Host object, never destroyed.
void* paths_stream_create_masked(const uint32_t* cu_mask, int words) {
  hipStream_t st = nullptr;
  if (cu_mask == nullptr || words <= 0 || hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask) != hipSuccess) {
    paths_set_error(PATHS_ELAUNCH, "stream_create_masked failed");
    return nullptr;
  }
  return st;
}
int paths_memset_zero(void* p, size_t bytes, hipStream_t stream) {
  PATHS_REQUIRE(p != nullptr && bytes > 0, "memset_zero: bad arguments");
  if (hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) return paths_set_error(PATHS_ELAUNCH, "memset_zero: hipMemsetAsync failed");
  return PATHS_OK;
}
}
