/*
 * Drop-in replacement of the reference's public class (bao_flow_patchmatch_multiscale_cuda.h:33-44),
 * implemented on the MI355X-native C ABI of include/eppm.h.  Public names, signatures and argument
 * meaning are the reference's; the private section (CUDA vector-type buffer tables, :46-98 there) is
 * replaced by one opaque handle.
 *
 * Memory layout expected from the caller (basic/bao_basic.h:105-174): img[y][x][c] row-pointer tables
 * over R,G,B bytes, disp[y][x] row-pointer tables over floats; rows need not be contiguous.
 *
 * Differences from the reference, all on the error path: nothing here calls exit() or getchar();
 * a failed call prints one line to stderr and leaves the outputs untouched; init() may be called
 * again (the previous buffers are released first; the reference leaks them).
 */
#ifndef _BAO_FLOW_PATCHMATCH_MULTISCALE_CUDA_H_
#define _BAO_FLOW_PATCHMATCH_MULTISCALE_CUDA_H_

#include <stddef.h>
#include "bao_basic_cuda.h"   /* as the reference's header does (:31): callers get bao_timer_gpu / bao_timer_gpu_cpu through it */

struct eppm_ctx;

class bao_flow_patchmatch_multiscale_cuda
{
public:
    bao_flow_patchmatch_multiscale_cuda();
    ~bao_flow_patchmatch_multiscale_cuda();

public:
    //interface (bao_flow_patchmatch_multiscale_cuda.h:40-44)
    void init(int h,int w);
    void init(unsigned char***img1,unsigned char***img2,int h,int w);
    bool set_data(unsigned char***img1,unsigned char***img2); //always true in the reference (driver .cpp:159-168); false here only on error
    void compute_flow(float**disp1_x,float**disp1_y,unsigned char***color_flow=NULL);

    // additions (not in the reference): device selection and run-time values of the defs.h constants, both
    // before init(); access to the C handle
    void set_device(int device) { m_device = device; }
    // name = a field of eppm_params (include/eppm.h): "patch_r", "num_iter", "search_range", "num_guess",
    // "seg_len", "wmf_iters", "seed", "propagation", "levels"; or "pin_caller_buffers" (0/1, default 0): register the contiguous
    // image and flow blocks passed to set_data / compute_flow for DMA (eppm_host_register) the first time they are seen, so that no
    // host copy remains between the caller's memory and the GPU -- the caller then keeps those blocks allocated until the object
    // is destroyed or init() is called again.  false: unknown name.
    bool set_option(const char* name, long long value);
    eppm_ctx* handle() const { return m_ctx; }

private:
    void _destroy();

private:
    int m_h;
    int m_w;
    int m_device;
    eppm_ctx* m_ctx;
    void* m_params;           // eppm_params*
    unsigned char* m_stage;   // contiguous RGB staging for row-pointer inputs that are not one block (allocated when needed)
    float* m_u;               // ... and for flow planes whose rows are not contiguous
    float* m_v;
    void* m_priv;             // caller blocks seen so far (verified layouts, DMA registrations)
};

#endif
