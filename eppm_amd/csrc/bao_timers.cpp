// bao_timers.cpp -- the reference's two timers (basic/bao_basic_cuda.cpp:37-122) on the HIP runtime; declared in
// include/bao_basic_cuda.h, which the drop-in class header includes as the reference's does.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/bao_basic_cuda.h"

bao_timer_gpu::bao_timer_gpu() : m_start(nullptr), m_stop(nullptr)
{
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    m_start = a; m_stop = b;
}

bao_timer_gpu::~bao_timer_gpu()
{
    if (m_start) (void)hipEventDestroy((hipEvent_t)m_start);
    if (m_stop) (void)hipEventDestroy((hipEvent_t)m_stop);
}

void bao_timer_gpu::start() { (void)hipEventRecord((hipEvent_t)m_start, 0); }

double bao_timer_gpu::stop()
{
    float elapsed = 0.0f;
    (void)hipEventRecord((hipEvent_t)m_stop, 0);
    (void)hipEventSynchronize((hipEvent_t)m_stop);
    (void)hipEventElapsedTime(&elapsed, (hipEvent_t)m_start, (hipEvent_t)m_stop);
    return elapsed;
}

double bao_timer_gpu::time_display(const char* disp, int nr_frame)
{
    const double ms = stop() / nr_frame;
    printf("Running time (%s) is: %5.4f ms.\n", disp, ms);
    return ms;
}

double bao_timer_gpu::fps_display(const char* disp, int nr_frame)
{
    const double fps = (double)nr_frame / (stop() * 1.0e-3f);
    printf("Running time (%s) is: %5.2f fps.\n", disp, fps);
    return fps;
}

void bao_timer_gpu_cpu::start()
{
    (void)hipDeviceSynchronize();
    gettimeofday(&timerStart, NULL);
}

double bao_timer_gpu_cpu::stop()
{
    (void)hipDeviceSynchronize();
    struct timeval timerStop, timerElapsed;
    gettimeofday(&timerStop, NULL);
    timersub(&timerStop, &timerStart, &timerElapsed);
    return timerElapsed.tv_sec + timerElapsed.tv_usec / 1000000.0;
}

double bao_timer_gpu_cpu::time_display(const char* disp, int nr_frame)
{
    const double sec = stop() / nr_frame;
    printf("Running time (%s) is: %5.5f Seconds.\n", disp, sec);
    return sec;
}

double bao_timer_gpu_cpu::fps_display(const char* disp, int nr_frame)
{
    const double fps = (double)nr_frame / stop();
    printf("Running time (%s) is: %5.5f frame per second.\n", disp, fps);
    return fps;
}
