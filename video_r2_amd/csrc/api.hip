// C-ABI plumbing shared by every entry point: thread-local error string, launch check, version, device sync helpers.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vr2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vr2_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vr2_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

extern "C" const char* vr2_last_error(void) { return g_err; }

// ---- launch-time tuning knobs (kernel-variant selection for A/B measurements; defaults are the measured best)
static int g_tune[VR2_TUNE_COUNT] = {2, 0, 0, 2, 1, 0, 0, 0};   // (keys 3 / 4: decode attention tiles in flight, suffix 2 / prefix 1)
int vr2_tune_get(int key) { return (key >= 0 && key < VR2_TUNE_COUNT) ? g_tune[key] : 0; }
extern "C" int vr2_tune_set(int key, int value) {
  if (key < 0 || key >= VR2_TUNE_COUNT) { vr2_set_error("vr2_tune_set: unknown key %d", key); return -1; }
  g_tune[key] = value;
  return 0;
}
extern "C" int vr2_version(void) { return 100; }  // 0.1.0

// ---- HIP event timing on an explicit stream (bench.py measures kernels on the engine's stream)
extern "C" int vr2_event_create(void** ev) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) { vr2_set_error("vr2_event_create failed"); return -2; }
  *ev = (void*)e;
  return 0;
}
extern "C" int vr2_event_record(void* ev, void* stream) {
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { vr2_set_error("vr2_event_record failed"); return -2; }
  return 0;
}
extern "C" int vr2_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    vr2_set_error("vr2_event_elapsed_ms failed");
    return -2;
  }
  return 0;
}
extern "C" int vr2_event_destroy(void* ev) { hipEventDestroy((hipEvent_t)ev); return 0; }

// ---- hipGraph capture of the launch-bound decode step (one graph = one token for all G sequences)
extern "C" int vr2_graph_begin(void* stream) {
  if (hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    vr2_set_error("vr2_graph_begin: hipStreamBeginCapture failed: %s", hipGetErrorString(hipGetLastError()));
    return -2;
  }
  return 0;
}
extern "C" int vr2_graph_end(void* stream, void** exec_out) {
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || g == nullptr) {
    vr2_set_error("vr2_graph_end: hipStreamEndCapture failed: %s", hipGetErrorString(hipGetLastError()));
    return -2;
  }
  hipGraphExec_t ex = nullptr;
  const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) { vr2_set_error("vr2_graph_end: hipGraphInstantiate failed: %s", hipGetErrorString(e)); return -2; }
  *exec_out = (void*)ex;
  return 0;
}
extern "C" int vr2_graph_launch(void* exec, void* stream) {
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
  if (e != hipSuccess) { vr2_set_error("vr2_graph_launch failed: %s", hipGetErrorString(e)); return -2; }
  return 0;
}
extern "C" int vr2_graph_destroy(void* exec) { hipGraphExecDestroy((hipGraphExec_t)exec); return 0; }
