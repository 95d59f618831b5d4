// Developer probe: where the ~80-240 ms of a fresh process's tl_create go — the HIP runtime's own start-up, call by call — and what
// leaving the process costs (the runtime's teardown at exit vs _exit).  Links libamdhip64 directly; no teeline code.
//   g++ -O2 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/probes/create_cost_probe.cpp -o /tmp/create_cost_probe -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
//   /tmp/create_cost_probe [quick]        quick: leave through _exit(0) after printing (time the process from outside)
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>

int main(int argc, char **argv)
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    auto lap = [&]() { const auto n = clk::now(); const double ms = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return ms; };
    int count = 0;
    (void)hipGetDeviceCount(&count);
    const double t_count = lap();
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const double t_prop = lap();
    int v = 0;
    (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    const double t_attr = lap();
    (void)hipSetDevice(0);
    const double t_set = lap();
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const double t_stream = lap();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double t_ev = lap();
    void *p = nullptr;
    (void)hipMalloc(&p, 1 << 20);
    const double t_malloc = lap();
    (void)hipStreamSynchronize(s);
    const double t_sync = lap();
    std::printf("{\"devices\": %d, \"arch\": \"%s\", \"hipGetDeviceCount_ms\": %.3f, \"hipGetDeviceProperties_ms\": %.3f, \"hipDeviceGetAttribute_ms\": %.3f, "
                "\"hipSetDevice_ms\": %.3f, \"hipStreamCreate_ms\": %.3f, \"two_hipEventCreate_ms\": %.3f, \"first_hipMalloc_ms\": %.3f, \"hipStreamSynchronize_ms\": %.3f}\n",
                count, prop.gcnArchName, t_count, t_prop, t_attr, t_set, t_stream, t_ev, t_malloc, t_sync);
    std::fflush(stdout);
    if (argc > 1 && !std::strcmp(argv[1], "quick")) _exit(0);
    return 0;
}
