// bao_timers.cpp -- the two timers the reference's demo and driver time themselves with (interface: basic/bao_basic_cuda.h:63-90, declared
// here in include/bao_basic_cuda.h, which the drop-in class header includes as the reference's does).  One stopwatch over two clocks:
//   DeviceClock  a HIP event pair on the null stream: device time in milliseconds (bao_timer_gpu)
//   HostClock    CLOCK_MONOTONIC with a device synchronisation before every reading: wall time in seconds (bao_timer_gpu_cpu)
// and one reporting routine; the four printed lines keep the reference's wording (basic/bao_basic_cuda.cpp:69-122), since programs
// written against it -- its own main.cpp, built unmodified by oracle/Makefile -- are compared by their output.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>

#include "../../include/bao_basic_cuda.h"

namespace {
enum Report { kPerFrame, kFramesPerSecond };
struct Wording { const char* per_frame; const char* rate; double units_per_second; };
const Wording kDeviceWording = {"Running time (%s) is: %5.4f ms.\n", "Running time (%s) is: %5.2f fps.\n", 1.0e3};
const Wording kHostWording = {"Running time (%s) is: %5.5f Seconds.\n", "Running time (%s) is: %5.5f frame per second.\n", 1.0};

// elapsed: in the clock's own unit (ms or s); the value printed is also the value returned
double report(const Wording& w, Report what, double elapsed, const char* disp, int nr_frame)
{
    const double v = (what == kPerFrame) ? elapsed / nr_frame : (double)nr_frame / (elapsed / w.units_per_second);
    printf(what == kPerFrame ? w.per_frame : w.rate, disp, v);
    return v;
}

struct DeviceClock {
    static void open(void*& a, void*& b)
    {
        hipEvent_t e[2] = {nullptr, nullptr};
        for (hipEvent_t& x : e) (void)hipEventCreate(&x);
        a = e[0]; b = e[1];
    }
    static void close(void* a, void* b)
    {
        for (void* x : {a, b})
            if (x) (void)hipEventDestroy((hipEvent_t)x);
    }
    static void mark(void* ev) { (void)hipEventRecord((hipEvent_t)ev, 0); }
    static double between(void* a, void* b)          // ms; waits for b
    {
        float ms = 0.0f;
        (void)hipEventSynchronize((hipEvent_t)b);
        (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b);
        return ms;
    }
};

struct HostClock {
    static timeval now()                              // the device has finished everything issued so far
    {
        (void)hipDeviceSynchronize();
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        timeval v;
        v.tv_sec = t.tv_sec; v.tv_usec = t.tv_nsec / 1000;
        return v;
    }
    static double between(const timeval& a, const timeval& b) { return (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_usec - a.tv_usec) * 1.0e-6; }
};
}  // namespace

bao_timer_gpu::bao_timer_gpu() : m_start(nullptr), m_stop(nullptr) { DeviceClock::open(m_start, m_stop); }
bao_timer_gpu::~bao_timer_gpu() { DeviceClock::close(m_start, m_stop); }
void bao_timer_gpu::start() { DeviceClock::mark(m_start); }
double bao_timer_gpu::stop() { DeviceClock::mark(m_stop); return DeviceClock::between(m_start, m_stop); }
double bao_timer_gpu::time_display(const char* disp, int nr_frame) { return report(kDeviceWording, kPerFrame, stop(), disp, nr_frame); }
double bao_timer_gpu::fps_display(const char* disp, int nr_frame) { return report(kDeviceWording, kFramesPerSecond, stop(), disp, nr_frame); }

void bao_timer_gpu_cpu::start() { timerStart = HostClock::now(); }
double bao_timer_gpu_cpu::stop() { return HostClock::between(timerStart, HostClock::now()); }
double bao_timer_gpu_cpu::time_display(const char* disp, int nr_frame) { return report(kHostWording, kPerFrame, stop(), disp, nr_frame); }
double bao_timer_gpu_cpu::fps_display(const char* disp, int nr_frame) { return report(kHostWording, kFramesPerSecond, stop(), disp, nr_frame); }
