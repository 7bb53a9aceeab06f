/*
 * Drop-in for the part of the reference's basic/bao_basic_cuda.h that its public class header pulls in
 * (bao_flow_patchmatch_multiscale_cuda.h:31) and that callers of the class use: the two timers the demo
 * and the driver time themselves with (basic/bao_basic_cuda.h:63-90, implementation
 * basic/bao_basic_cuda.cpp:37-122).  Plain C++: no GPU runtime header is needed to include it, so the
 * reference's main.cpp compiles unchanged against include/ with any host compiler
 * (oracle/Makefile, target runeppm_ref).
 *
 *   bao_timer_gpu      device time between start() and stop() on the null stream (hipEvent pair), in ms
 *   bao_timer_gpu_cpu  wall time with a device synchronisation on both sides (monotonic clock), in seconds
 *
 * The device-memory templates of the reference header (bao_cuda_alloc / bao_cuda_copy_*, :96-253) are
 * private plumbing of its driver; the C ABI of eppm.h offers the equivalents (eppm_malloc_device, ...).
 */
#ifndef _BAO_BASIC_CUDA_H_
#define _BAO_BASIC_CUDA_H_

#include <sys/time.h>

#ifndef __max
#define __max(a,b) (((a) > (b)) ? (a) : (b))
#endif
#ifndef __min
#define __min(a,b) (((a) < (b)) ? (a) : (b))
#endif

class bao_timer_gpu
{
public:
    bao_timer_gpu();
    ~bao_timer_gpu();
    void start();
    double stop();                                              // ms since start()
    double time_display(const char *disp="",int nr_frame=1);    // prints "Running time (%s) is: %5.4f ms."
    double fps_display(const char *disp="",int nr_frame=1);
private:
    void* m_start;      // hipEvent_t
    void* m_stop;
};

class bao_timer_gpu_cpu //synchronize between cpu and gpu time
{
public:
    void start();
    double stop();                                              // seconds since start()
    double time_display(const char* disp="", int nr_frame=1);   // prints "Running time (%s) is: %5.5f Seconds."
    double fps_display(const char* disp="", int nr_frame=1);
private:
    struct timeval timerStart;
};

#endif
