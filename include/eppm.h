/*
 * eppm.h -- C ABI of the MI355X-native EPPM optical-flow engine (libeppm_hip.so).
 *
 * Drop-in boundary for the hot path of linchaobao/EPPM: image pair -> dense flow.
 * Two layers are exported:
 *
 *  (1) the context API (eppm_*): what a host program or an FFI binding uses.  It replaces
 *      class bao_flow_patchmatch_multiscale_cuda (bao_flow_patchmatch_multiscale_cuda.h:33-44):
 *      eppm_create      <- init(h,w)                                   driver .cpp:112-157
 *      eppm_set_images  <- set_data(img1,img2)                         driver .cpp:159-168
 *      eppm_compute     <- compute_flow(disp1_x,disp1_y)               driver .cpp:217-306
 *      eppm_destroy     <- ~bao_flow_patchmatch_multiscale_cuda()      driver .cpp:170-209
 *      The C++ class itself is kept, source compatible, in
 *      include/bao_flow_patchmatch_multiscale_cuda.h on top of this ABI.
 *
 *  (2) the nine live stage launchers of the reference's link-level ABI, with the reference's
 *      names, argument order and meaning (driver .cpp:40-62): see "stage launchers" below.
 *
 * Conventions: plain pointers and sizes only; every function returns an eppm_status
 * (0 = OK) except the reference-signature launchers, which are void like the originals
 * and record their status for eppm_last_error().  Nothing in this library calls exit().
 * One context = one device + one stream; contexts are independent (one per host thread or
 * per GPU); a single context is not thread-safe.
 */
#ifndef EPPM_H_
#define EPPM_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    EPPM_OK = 0,
    EPPM_ERR_ARG = 1,        /* bad argument (NULL, non-positive size, unsupported parameter) */
    EPPM_ERR_HIP = 2,        /* a HIP runtime call failed; see eppm_last_error() */
    EPPM_ERR_STATE = 3,      /* call order: compute before set_images, ... */
    EPPM_ERR_NOMEM = 4
} eppm_status;

/* Pixel / vector element layouts (identical to CUDA's uchar4 / short2 / float2). */
typedef struct { uint8_t x, y, z, w; } eppm_uchar4;
typedef struct { int16_t x, y; } eppm_short2;
typedef struct { float x, y; } eppm_float2;

/* Tunables.  The reference fixes them at compile time (defs.h:31-76); defaults are those values. */
typedef struct {
    int patch_r;        /* PATCH_R 9 (odd, <= 31)                         defs.h:44 */
    int num_iter;       /* NUM_ITER 10                                    defs.h:45 */
    int search_range;   /* SEARCH_RANGE 30                                defs.h:36 */
    int num_guess;      /* NUM_RAND_GUESS 6 (<= 8)                        defs.h:38 */
    int seg_len;        /* PROP_SEG_LENGTH 10              bao_pmflow_kernel.cu:979 */
    int wmf_iters;      /* 20                                         driver .cpp:239 */
    unsigned long long seed; /* 1234                        bao_pmflow_kernel.cu:68 */
    int propagation;    /* 0: segmented scan-line sweeps, baoSegPropagate (live, bao_pmflow_kernel.cu:1812)
                           1: jump flood, baoJumpPropagate (steps 32..1, :800-857; disabled in the reference :1813)
                           2: 4-neighbour propagation, 10x baoParallelPropagate (:720-795; disabled at :1804-1809) */
    int levels;         /* PYR_MAX_DEPTH 3 (1..8): pyramid depth; PatchMatch runs at level levels-1   defs.h:31 */
} eppm_params;

typedef struct eppm_ctx eppm_ctx;

/* ----------------------------------------------------------------------------------------
 * context API
 * -------------------------------------------------------------------------------------- */
int  eppm_default_params(eppm_params* p);
/* Allocates every device buffer for an h x w pair (3-level pyramid, PYR_MAX_DEPTH defs.h:31). */
int  eppm_create(eppm_ctx** out, int h, int w, int device, const eppm_params* params /* NULL = defaults */);
int  eppm_destroy(eppm_ctx* ctx);
/* Use an existing hipStream_t (passed as void*) instead of the context's own stream. */
int  eppm_set_stream(eppm_ctx* ctx, void* hip_stream);

/* Host images: h rows of w RGB triplets, row_stride bytes apart (>= 3*w).  RGB->RGBA, H2D,
 * prefilter, pyramid, census (set_data + _prepare_data, driver .cpp:159-168,212-215). */
int  eppm_set_images(eppm_ctx* ctx, const uint8_t* rgb1, const uint8_t* rgb2, size_t row_stride);
/* Caller memory pinned for DMA.  eppm_set_images / eppm_batch_set_images read an image that lies inside a registered block where
 * it is (no staging copy), and eppm_compute / eppm_compute_begin_into / the batch forms write flow planes that lie inside
 * registered blocks directly; anything else goes through the context's pinned staging buffers (one host copy each way), with
 * identical results.  Register the buffers a program reuses from pair to pair once (hipHostRegister underneath: the cost of a
 * registration is that of pinning the pages, paid once); eppm_host_alloc returns pinned memory that counts as registered.
 * eppm_set_images returns when the images have been read (set_data is a synchronous cudaMemcpy in the reference,
 * driver .cpp:165-166).  Registrations are counted: registering a block (or a range inside a registered block) again adds an owner,
 * eppm_host_unregister removes one, and the pages are unpinned when the last owner unregisters -- after the transfers other contexts
 * have in flight on the block completed (an unregister under one's own pending eppm_compute_begin_into fails with EPPM_ERR_STATE
 * after a bounded wait instead of deadlocking).  Thread-safe. */
int  eppm_host_register(void* p, size_t bytes);
int  eppm_host_unregister(void* p);
int  eppm_host_is_registered(const void* p, size_t bytes);   /* 1 / 0 */
int  eppm_host_alloc(void** p, size_t bytes);
int  eppm_host_free(void* p);

/* Device-resident RGBA (uchar4, alpha ignored/0) images, pitch in bytes: copies them into the context (device to device,
 * in the order of the context's stream) and runs prepare.  The planes must be complete before the call (or produced on
 * that stream) and stay valid until that copy has run (eppm_synchronize, or any later synchronous call on this
 * context); they are never read in place by a kernel. */
int  eppm_set_images_device(eppm_ctx* ctx, const void* d_rgba1, const void* d_rgba2, size_t pitch);

/* ----------------------------------------------------------------------------------------
 * batches of independent pairs of one size (BASELINE.json configs[2]: many pairs per GPU; SURVEY 7 "batch dimension
 * in the kernel grid").  A batch context holds `npairs` pairs; every kernel launch of the path covers all active pairs
 * (the quarter-resolution stages of ONE pair cannot fill 256 CUs, those of 8 pairs can).  Each pair's result is
 * bit-identical to what a single-pair context computes for it.  The single-pair calls above and below work on a batch
 * context too (they address pair 0 and make it the only active pair).
 * -------------------------------------------------------------------------------------- */
int  eppm_create_batch(eppm_ctx** out, int h, int w, int device, const eppm_params* params, int npairs);
int  eppm_batch_size(const eppm_ctx* ctx);
/* set_data for pairs 0..n-1 (n <= npairs; n becomes the number of active pairs): arrays of n host RGB image pointers */
int  eppm_batch_set_images(eppm_ctx* ctx, int n, const uint8_t* const* rgb1, const uint8_t* const* rgb2, size_t row_stride);
/* the same from device-resident RGBA planes (arrays of n device pointers); copied in stream order, never read in place */
int  eppm_batch_set_images_device(eppm_ctx* ctx, int n, const void* const* d_rgba1, const void* const* d_rgba2, size_t pitch);
/* compute_flow for every active pair.  u[k], v[k]: h*w floats of pair k (host); synchronous */
int  eppm_batch_compute(eppm_ctx* ctx, float* const* u, float* const* v);
/* asynchronous; d_flows: NULL, or n device pointers (NULL entries allowed) receiving the interleaved float2 flows */
int  eppm_batch_compute_device(eppm_ctx* ctx, void* const* d_flows);
/* second half of eppm_compute_begin / eppm_batch_compute_begin_into for every active pair */
int  eppm_batch_compute_end(eppm_ctx* ctx, float* const* u, float* const* v);
/* eppm_compute_begin_into for every active pair */
int  eppm_batch_compute_begin_into(eppm_ctx* ctx, float* const* u, float* const* v);
/* eppm_get_plane of pair `pair` */
int  eppm_batch_get_plane(eppm_ctx* ctx, int pair, const char* name, int level, void* dst, size_t dst_bytes);

/* compute_flow (driver .cpp:217-306).  u, v: h*w floats each (host). Synchronous. */
int  eppm_compute(eppm_ctx* ctx, float* u, float* v);
/* The same in two halves, for a host thread that keeps several contexts in flight (PCIe copies of one pair overlap
 * the kernels of another): begin enqueues the path and the device-to-host copy and returns at once; end waits for
 * this context's stream and writes u, v.  eppm_set_images on a context does NOT drain its stream: the staging buffers are
 * double-buffered with an event each, images in registered memory are read by DMA and the call returns once that DMA is done; u, v
 * handed to eppm_compute_begin_into must stay allocated until eppm_compute_end has returned. */
int  eppm_compute_begin(eppm_ctx* ctx);
int  eppm_compute_end(eppm_ctx* ctx, float* u, float* v);
/* eppm_compute_begin with the destination named up front: planes in registered memory (eppm_host_register) are written by the
 * copy engine directly and eppm_compute_end(ctx, u, v) with the same pointers only waits.  eppm_compute = this + eppm_compute_end. */
int  eppm_compute_begin_into(eppm_ctx* ctx, float* u, float* v);
/* Optional colour-coded flow of the last eppm_compute* (compute_flow's color_flow argument, driver .cpp:308-314):
 * Middlebury colour wheel on the device flow (basic/bao_basic_cuda.cuh:776-845), h rows of w R,G,B triplets,
 * row_stride bytes apart.  The reference calls it with max_disp (20,20).  On a batch context: pair 0. */
int  eppm_compute_color(eppm_ctx* ctx, uint8_t* rgb, size_t row_stride, float max_disp_x, float max_disp_y);
/* Same, asynchronous on the context's stream; the interleaved float2 flow stays in HBM.
 * d_flow may be NULL (result kept in the context; fetch with eppm_get_plane("flow",0)). */
int  eppm_compute_device(eppm_ctx* ctx, void* d_flow);
int  eppm_synchronize(eppm_ctx* ctx);

/* Geometry of the pyramid (bao_pyr_init_dim, basic/bao_basic.h:196-211). */
int  eppm_num_levels(const eppm_ctx* ctx);
int  eppm_level_dims(const eppm_ctx* ctx, int level, int* h, int* w);

/* Copy an internal plane to the host, tightly packed (row = w elements).  Names:
 *  "img1","img2" (uchar4), "census1","census2" (u8), "nnf1","nnf2" (short2), "cost1","cost2" (f32),
 *  "flow" (float2).  Valid after the stage that produces it has run. */
int  eppm_get_plane(eppm_ctx* ctx, const char* name, int level, void* dst, size_t dst_bytes);

/* Per-stage device times in ms (hipEvent pairs on the context's stream), one entry per stage per
 * call since the last eppm_clear_stage_times (names repeat across calls; prepare entries first).
 * names[i] points to static strings.  Returns the number of entries written (<= max). */
int  eppm_stage_times(eppm_ctx* ctx, const char** names, float* ms, int max);
int  eppm_clear_stage_times(eppm_ctx* ctx);
/* 0: no events (default); 1: an event pair around every stage; 2: only around the dominant kernel (the candidate
 * refine, entries "c2f_refine_L<l>").  Events come from a per-context pool: none is created in a steady-state step. */
int  eppm_enable_stage_timing(eppm_ctx* ctx, int on);

const char* eppm_last_error(void);
const char* eppm_version(void);

/* ----------------------------------------------------------------------------------------
 * device-memory plumbing for callers without a HIP runtime binding (tests, FFI hosts)
 * -------------------------------------------------------------------------------------- */
int  eppm_device_count(int* n);
int  eppm_set_device(int device);
int  eppm_malloc_device(void** p, size_t bytes);
int  eppm_malloc_pitched(void** p, size_t* pitch, size_t width_bytes, size_t rows);  /* cudaMallocPitch analogue */
int  eppm_free_device(void* p);
int  eppm_memcpy_h2d(void* dst, const void* src, size_t bytes);
int  eppm_memcpy_d2h(void* dst, const void* src, size_t bytes);
int  eppm_memcpy2d_h2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows);
int  eppm_memcpy2d_d2h(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows);
int  eppm_memset_device(void* p, int value, size_t bytes);
int  eppm_device_synchronize(void);
/* PCI address of a device, lower case as sysfs spells it ("0000:c1:00.0"; hipDeviceGetPCIBusId); buf: at least 13 bytes */
int  eppm_device_pci_bus_id(int device, char* buf, size_t len);
/* Multi-GPU hosts (one host thread per GPU, one context each: pair i -> GPU i mod N): binds the CALLING thread, and the threads it
 * creates afterwards, to the CPUs of the NUMA node the device's PCIe slot belongs to (sysfs), within the CPUs the thread may use now.
 * *numa_node = -1, *ncpus = 0 and nothing bound when the topology is not visible; either pointer may be NULL. */
int  eppm_bind_thread_to_device(int device, int* numa_node, int* ncpus);
/* free / total memory of the current device (sizing the number of contexts in flight; leak checks) */
int  eppm_device_mem_info(size_t* free_bytes, size_t* total_bytes);
/* A destroyed context's slab and pinned staging buffers are kept (a few blocks, bounded in bytes) for the next context of the same size:
 * allocation is inside the window the reference's demo times.  This gives them back to the runtime. */
int  eppm_release_cached_memory(void);
/* Stream used by the reference-signature launchers below (default: the null stream). */
int  eppm_set_launcher_stream(void* hip_stream);
/* Parameters used by the reference-signature launchers (default: defs.h values). */
int  eppm_set_launcher_params(const eppm_params* p);
/* Status of the most recent void launcher below (they cannot return one). */
int  eppm_launcher_status(void);

/* ----------------------------------------------------------------------------------------
 * stage launchers: the reference's live extern "C" ABI (driver .cpp:40-62).
 * All pointers are DEVICE pointers, all *_pitch are in BYTES, argument order is (w,h).
 * Pyramid tables (T**, int*, size_t*) are HOST arrays of device pointers
 * (basic/bao_basic_cuda.h:209-229).  uchar4/short2/float2 are the layouts declared above.
 * -------------------------------------------------------------------------------------- */
/* bao_pmflow_refine_kernel.cu:1060-1071 */
void baoCudaPatchMatchMultiscalePrepare(eppm_uchar4** pImgPyr1, eppm_uchar4** pImgPyr2, unsigned char** pCensusPyr1,
        unsigned char** pCensusPyr2, eppm_uchar4** pTempPyr1, eppm_uchar4** pTempPyr2, int* arrH, int* arrW,
        size_t* arrPitchUchar4, size_t* arrPitchUchar1, int nLevels, eppm_uchar4* d_img1, eppm_uchar4* d_img2, int h, int w);
/* bao_pmflow_census_kernel.cu:93-112 */
void baoCudaCensusTransform(unsigned char* d_census1, unsigned char* d_census2, eppm_uchar4* d_img1, eppm_uchar4* d_img2,
        int w, int h, size_t img_pitch, size_t census_pitch);
/* bao_pmflow_kernel.cu:1760-1826 */
void baoCudaPatchMatch(eppm_short2* d_disp_vec, float* d_cost, eppm_uchar4* d_img1, eppm_uchar4* d_img2,
        unsigned char* d_census1, unsigned char* d_census2, int w, int h, size_t img_pitch, size_t cost_pitch,
        size_t disp_pitch, size_t census_pitch);
/* bao_pmflow_refine_kernel.cu:78-92 */
void baoCudaLeftRightCheck(eppm_short2* d_disp_vec, float* d_cost, eppm_short2* d_disp_vec2, float* d_cost2,
        int w, int h, size_t cost_pitch, size_t disp_pitch);
/* bao_pmflow_refine_kernel.cu:185-193 */
void baoCudaOutlierRemoval(eppm_short2* d_disp_vec, float* d_cost, int w, int h, size_t cost_pitch, size_t disp_pitch);
/* bao_pmflow_refine_kernel.cu:261-286 */
void baoCudaWeightedMedianFilter(eppm_short2* d_disp_vec, float* d_cost, eppm_uchar4* d_img, int w, int h,
        size_t img_pitch, size_t cost_pitch, size_t disp_pitch, int num_iter, bool is_only_occlusion);
/* bao_pmflow_refine_kernel.cu:373-390 */
void baoCudaFillHole(eppm_short2* d_disp_vec, float* d_cost, eppm_uchar4* d_img, int w, int h,
        size_t img_pitch, size_t cost_pitch, size_t disp_pitch);
/* bao_pmflow_refine_kernel.cu:724-734 */
void baoCudaNNF2Flow(eppm_float2* d_flow, eppm_short2* d_disp_vec, int w, int h, size_t disp_pitch, size_t flow_pitch);
/* bao_pmflow_refine_kernel.cu:1076-1087 */
void baoCudaBLF_C2F(eppm_float2** pFlowPyr, eppm_uchar4** pImgPyr1, eppm_uchar4** pImgPyr2, unsigned char** pCensusPyr1,
        unsigned char** pCensusPyr2, eppm_float2** pTempPyr1, eppm_float2** pTempPyr2, int* arrH, int* arrW,
        size_t* arrPitchUchar4, size_t* arrPitchUchar1, int nLayerIdx);
/* bao_pmflow_kernel.cu:2042-2069 */
void baoCudaBLFCostFilterRefine(eppm_float2* d_flow_vec, eppm_uchar4* d_img1, eppm_uchar4* d_img2, unsigned char* d_census1,
        unsigned char* d_census2, int w, int h, size_t img_pitch, size_t census_pitch);
/* bao_pmflow_refine_kernel.cu:801-826 */
void baoCudaFlowSmoothing(eppm_float2* d_flow, eppm_uchar4* d_img, int w, int h, size_t img_pitch, size_t flow_pitch);
/* basic/bao_basic_cuda.cuh:839-845 (float2 form): d_rgba h*w uchar4 {R,G,B,0}, d_flow h*w float2, both unpitched.
 * The library also exports the C++-linkage symbol the reference's driver declares at :64,
 *   void bao_cuda_convert_flow_to_colorshow(uchar4*, float2*, int h, int w, float max_disp_x, float max_disp_y)
 * (HIP vector types; a C header cannot declare it). */
int  eppm_flow_to_color(eppm_uchar4* d_rgba, const eppm_float2* d_flow, int h, int w, float max_disp_x, float max_disp_y);

/* ----------------------------------------------------------------------------------------
 * sub-stage entry points of PatchMatch (for parity tests at kernel granularity).  They mirror
 * the reference's inner launchers baoGenerateRandomField / baoComputeCostField / baoSegPropagate /
 * baoRandomSearch (bao_pmflow_kernel.cu:153-165, 689-696, 1167-1181, 1588-1594), with the
 * texture bindings and the global RNG state made explicit arguments.
 * rng: opaque device buffer from eppm_pm_rng_create (one XORWOW stream per 16x16 block).
 * PRECONDITION of the three propagate entry points AND of eppm_pm_random_search: d_cost[p] is the patch cost of d_nnf[p] (as eppm_pm_cost_field,
 * a propagate or a search leaves it).  A candidate equal to the pixel's stored match is rejected without being
 * evaluated -- it would reproduce the stored cost bit for bit, and the reference's strict `<` rejects it too; with a
 * cost plane that is NOT consistent with the NNF the reference would re-evaluate and could lower the cost, these would not
 * (a random guess equal to the stored match likewise sits its evaluation out).
 * -------------------------------------------------------------------------------------- */
typedef struct eppm_pm_rng eppm_pm_rng;
int  eppm_pm_rng_create(eppm_pm_rng** out, int w, int h, const eppm_params* p);
int  eppm_pm_rng_reset(eppm_pm_rng* rng);        /* back to curand_init(seed, block_id, 0) */
int  eppm_pm_rng_destroy(eppm_pm_rng* rng);
/* host copy of the per-block XORWOW state at the current stream position: 6 x uint32 per block (v[5], d) */
int  eppm_pm_rng_block_states(eppm_pm_rng* rng, uint32_t* dst, size_t dst_words);
int  eppm_pm_gen_rand_field(eppm_pm_rng* rng, eppm_short2* d_nnf, int w, int h, size_t disp_pitch);
int  eppm_pm_cost_field(float* d_cost, const eppm_short2* d_nnf, const eppm_uchar4* d_img1, const eppm_uchar4* d_img2,
        const unsigned char* d_census1, const unsigned char* d_census2, int w, int h, size_t img_pitch,
        size_t cost_pitch, size_t disp_pitch, size_t census_pitch);
/* dir: 0 row fwd, 1 col fwd, 2 row rev, 3 col rev; dir < 0: all four in the reference's order */
int  eppm_pm_seg_propagate(float* d_cost, eppm_short2* d_nnf, const eppm_uchar4* d_img1, const eppm_uchar4* d_img2,
        const unsigned char* d_census1, const unsigned char* d_census2, int w, int h, size_t img_pitch,
        size_t cost_pitch, size_t disp_pitch, size_t census_pitch, int dir);
/* baoJumpPropagate (bao_pmflow_kernel.cu:843-857): six Jacobi launches with step 32,16,8,4,2,1 */
int  eppm_pm_jump_propagate(float* d_cost, eppm_short2* d_nnf, const eppm_uchar4* d_img1, const eppm_uchar4* d_img2,
        const unsigned char* d_census1, const unsigned char* d_census2, int w, int h, size_t img_pitch,
        size_t cost_pitch, size_t disp_pitch, size_t census_pitch);
/* baoParallelPropagate (bao_pmflow_kernel.cu:720-795): ONE Jacobi launch of the 4-neighbour propagation */
int  eppm_pm_parallel_propagate(float* d_cost, eppm_short2* d_nnf, const eppm_uchar4* d_img1, const eppm_uchar4* d_img2,
        const unsigned char* d_census1, const unsigned char* d_census2, int w, int h, size_t img_pitch,
        size_t cost_pitch, size_t disp_pitch, size_t census_pitch);
int  eppm_pm_random_search(eppm_pm_rng* rng, float* d_cost, eppm_short2* d_nnf, const eppm_uchar4* d_img1,
        const eppm_uchar4* d_img2, const unsigned char* d_census1, const unsigned char* d_census2, int w, int h,
        size_t img_pitch, size_t cost_pitch, size_t disp_pitch, size_t census_pitch);
/* basic/bao_basic_cuda.cuh:437-481 and :565-615 (uchar4), :511-537 (float2) */
int  eppm_gauss_filter_rgba(eppm_uchar4* d_out, const eppm_uchar4* d_in, size_t pitch, int h, int w, float sigma, int radius);
int  eppm_resize_rgba(eppm_uchar4* d_out, size_t out_pitch, int outH, int outW, const eppm_uchar4* d_in, size_t in_pitch,
        int h, int w, float ratio);
int  eppm_resize_flow(eppm_float2* d_out, int outH, int outW, const eppm_float2* d_in, int h, int w, float ratio);
/* ----------------------------------------------------------------------------------------
 * file formats used by the reference's CLI (main.cpp:56-69)
 * -------------------------------------------------------------------------------------- */
/* P6/P5 reader tolerant of '#' comment lines (basic/bao_basic.cpp:137-218). image: h*w*3 bytes. */
int  eppm_load_ppm(const char* filename, uint8_t* image, int h, int w, int* channels);
int  eppm_ppm_size(const char* filename, int* h, int* w);
/* Middlebury .flo: "PIEH", int32 w, int32 h, interleaved f32 (u,v) rows (flowIO.cpp:122-163). */
int  eppm_save_flo(const char* filename, const float* u, const float* v, int h, int w);
int  eppm_load_flo(const char* filename, float* u, float* v, int h, int w);
int  eppm_flo_size(const char* filename, int* h, int* w);
/* EPE / AAE with the reference's validity rule (basic/bao_flow_tools.cpp:64-111); _border: `border` pixels on every side left out. */
int  eppm_flow_error(const float* u, const float* v, const float* gt_u, const float* gt_v, int h, int w, float* epe, float* aae);
int  eppm_flow_error_border(const float* u, const float* v, const float* gt_u, const float* gt_v, int h, int w, int border, float* epe, float* aae);
/* Fraction of the pixels with known ground truth whose end-point error exceeds error_thresh; error_map: h*w bytes (255 there) or
 * NULL (bao_calc_flow_error_percentage, basic/bao_flow_tools.cpp:114-141). */
int  eppm_flow_error_percentage(const float* u, const float* v, const float* gt_u, const float* gt_v, int h, int w, int error_thresh,
                                uint8_t* error_map, float* fraction);
/* Both components clamped to [-|cutoff|, |cutoff|]; unknown vectors pass through unless cut_invalid (bao_flow_cutoff, :166-197). */
int  eppm_flow_cutoff(float* u_out, float* v_out, const float* u, const float* v, int h, int w, int cutoff, int cut_invalid);
/* Host colour coding scaled by the field's largest known radius, unknown vectors black; rgb: h*w*3 bytes R,G,B
 * (bao_convert_flow_to_colorshow, :200-231, on Middlebury's computeColor, 3rdparty/middlebury/colorcode.cpp:30-85).  Where the
 * reference is undefined this is defined: a field with no motion or no known vector (largest radius 0: 0/0 there, then
 * colorwheel[(int)NaN]) is scaled by 1 -- a static scene is white --, and a vector with a NaN component is black. */
int  eppm_flow_to_color_host(uint8_t* rgb, const float* u, const float* v, int h, int w);

#ifdef __cplusplus
}
#endif
#endif /* EPPM_H_ */
