// Error plumbing, device info, HIP graph / event helpers of the C ABI.
#include "icaf_common.h"
#include <cstring>

namespace icaf {
std::string& last_error() {
    static thread_local std::string e;
    return e;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
// Probe knobs of the library (icaf.h: icaf_set_option).  The C side reads NO environment variable: the host's one options object
// (icafusion_amd/options.py) pushes them when the library is loaded.
LibOptions g_opt;
}  // namespace icaf

using namespace icaf;

extern "C" const char* icaf_last_error(void) { return last_error().c_str(); }

extern "C" int icaf_set_option(const char* name, int value) {
    if (!name) return fail(ICAF_ERR_ARG, "icaf_set_option: null name");
    if (!strcmp(name, "detect_elementwise")) g_opt.detect_elementwise = value;
    else if (!strcmp(name, "attn_qsplit")) g_opt.attn_qsplit = value;
    else if (!strcmp(name, "sppf_vpb")) g_opt.sppf_vpb = value;
    else return fail(ICAF_ERR_ARG, "icaf_set_option: unknown option '%s' (detect_elementwise, attn_qsplit, sppf_vpb)", name);
    return ICAF_OK;
}
extern "C" int icaf_version(void) { return 100; }

extern "C" int icaf_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    int dev = 0;
    ICAF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    ICAF_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return ICAF_OK;
}

extern "C" int icaf_graph_begin(icaf_stream_t s) {
    ICAF_HIP(hipStreamBeginCapture(S(s), hipStreamCaptureModeThreadLocal));
    return ICAF_OK;
}
extern "C" int icaf_graph_end(icaf_stream_t s, void** graph_exec) {
    hipGraph_t graph = nullptr;
    ICAF_HIP(hipStreamEndCapture(S(s), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(ICAF_ERR_HIP, "hipGraphInstantiate -> %s", hipGetErrorString(e));
    *graph_exec = (void*)exec;
    return ICAF_OK;
}
extern "C" int icaf_graph_launch(void* graph_exec, icaf_stream_t s) {
    ICAF_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, S(s)));
    return ICAF_OK;
}
extern "C" int icaf_graph_destroy(void* graph_exec) {
    if (graph_exec) ICAF_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return ICAF_OK;
}
extern "C" int icaf_event_create(void** ev) {
    hipEvent_t e;
    ICAF_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return ICAF_OK;
}
extern "C" int icaf_event_record(void* ev, icaf_stream_t s) {
    ICAF_HIP(hipEventRecord((hipEvent_t)ev, S(s)));
    return ICAF_OK;
}
extern "C" int icaf_stream_wait_event(icaf_stream_t s, void* ev) {
    ICAF_HIP(hipStreamWaitEvent(S(s), (hipEvent_t)ev, 0));
    return ICAF_OK;
}
extern "C" int icaf_event_elapsed_ms(void* start, void* stop, float* ms) {
    ICAF_HIP(hipEventSynchronize((hipEvent_t)stop));
    ICAF_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ICAF_OK;
}
extern "C" int icaf_event_destroy(void* ev) {
    if (ev) ICAF_HIP(hipEventDestroy((hipEvent_t)ev));
    return ICAF_OK;
}
extern "C" int icaf_stream_sync(icaf_stream_t s) {
    ICAF_HIP(hipStreamSynchronize(S(s)));
    return ICAF_OK;
}
