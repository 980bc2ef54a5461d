// runtime.hip — error plumbing, device info and HIP-graph capture for libimm_hip.so.
// The graph entry points replace the role of TF's static-graph session.run
// (/root/reference/imm/train/cnn_train_multi.py:459): a whole training-step launch sequence is
// captured once on the caller's stream and replayed without host work.
#include "common.h"

thread_local char imm_err_buf[512] = "";

int imm_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(imm_err_buf, sizeof(imm_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int imm_abi_version(void) { return IMM_ABI_VERSION; }
extern "C" const char* imm_last_error(void) { return imm_err_buf; }

extern "C" int imm_device_info(int32_t* out2) {
  IMM_REQUIRE(out2, "device_info: null");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  out2[0] = p.multiProcessorCount;
  int arch = 0;
  const char* n = p.gcnArchName;  // "gfx950:sramecc+:xnack-"
  if (n[0] == 'g' && n[1] == 'f' && n[2] == 'x')
    for (const char* c = n + 3; *c && *c != ':'; ++c) {
      if (*c >= '0' && *c <= '9') arch = arch * 16 + (*c - '0');
      else if (*c >= 'a' && *c <= 'f') arch = arch * 16 + (*c - 'a' + 10);
    }
  out2[1] = arch;  // hex digits of the gfx name: gfx950 -> 0x950
  return 0;
}

extern "C" int imm_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int imm_graph_end(void* stream, void** graph_exec_out) {
  IMM_REQUIRE(graph_exec_out, "graph_end: null out");
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)ge;
  return 0;
}

extern "C" int imm_graph_launch(void* graph_exec, void* stream) {
  IMM_REQUIRE(graph_exec, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipGraphLaunch: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int imm_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return 0;
}
