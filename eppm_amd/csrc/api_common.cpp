// api_common.cpp -- errors, version, parameters, host-side look-up tables and pyramid geometry of the C ABI (include/eppm.h).
//
// The driver behind the ABI follows bao_flow_patchmatch_multiscale_cuda.cpp: init :112-157, set_data :159-168,
// _prepare_data :212-215, compute_flow :217-306 (context.cpp).  Dead work of the reference is not reproduced: the
// level-1/0 weighted-median calls on never-initialised planes (driver :281), the debug D2H
// of the level-2 flow (:265-270) and the per-call RNG cudaMalloc (kernel.cu:1767).
#include "api_internal.h"

using namespace eppm;

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int set_err(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* eppm_last_error(void) { return g_err; }
#ifdef EPPM_TOL
extern "C" const char* eppm_version(void) { return "eppm-hip 0.3 (gfx950, tolerance arithmetic: integer-domain tables in the patch term, not bit-identical to the oracle)"; }
#else
extern "C" const char* eppm_version(void) { return "eppm-hip 0.3 (gfx950)"; }
#endif

extern "C" int eppm_default_params(eppm_params* p)
{
    if (!p) return set_err(EPPM_ERR_ARG, "eppm_default_params: NULL");
    p->patch_r = 9; p->num_iter = 10; p->search_range = 30; p->num_guess = 6;
    p->seg_len = 10; p->wmf_iters = 20; p->seed = 1234ULL; p->propagation = 0; p->levels = kNumLevels;
    return EPPM_OK;
}

int check_params(const eppm_params& p)
{
    if (p.patch_r < 1 || p.patch_r + 1 > kMaxS) return set_err(EPPM_ERR_ARG, "patch_r %d out of range [1,%d]", p.patch_r, kMaxS - 1);
    if (p.num_iter < 0 || p.wmf_iters < 0) return set_err(EPPM_ERR_ARG, "negative iteration count");
    if (p.num_guess < 1 || p.num_guess > 8) return set_err(EPPM_ERR_ARG, "num_guess %d out of range [1,8]", p.num_guess);
    if (p.seg_len < 2) return set_err(EPPM_ERR_ARG, "seg_len %d < 2", p.seg_len);
    if (p.search_range < 1) return set_err(EPPM_ERR_ARG, "search_range %d < 1", p.search_range);
    if (p.levels < 1 || p.levels > kMaxLevels) return set_err(EPPM_ERR_ARG, "levels %d out of range [1,%d]", p.levels, kMaxLevels);
    if (p.propagation < 0 || p.propagation > 2) return set_err(EPPM_ERR_ARG, "propagation %d: 0 (segmented sweeps), 1 (jump flood) or 2 (4-neighbour)", p.propagation);
    return EPPM_OK;
}

// LUTs, host side (kernel.cu:670-687; refine :270-275, :811-816).  gs[0..R] then cn[0..8].
void host_pm_lut(int R, std::vector<float>& v)
{
    v.resize(R + 1 + 9);
    const float sig_s = 0.5f * R;   // PM_SIG_S, defs.h:47
    for (int i = 0; i <= R; i++) v[i] = expf(-(i * i) / (sig_s * sig_s));
    for (int i = 0; i <= 8; i++) v[R + 1 + i] = 1 - expf(-float(i * i) / (0.3f * 8 * 0.3f * 8));
#ifdef EPPM_TOL
    // the tolerance library's integer-domain tables (eppm_device.cuh: make_texel): td[k] = 1 - exp(-(k/255)^2/s), ta[k] = exp(-(k/255)^2/s),
    // s = LAMBDA_AD^2 = PM_SIG_R^2 as the float product the reference forms (defs.h:48,51), everything else in double
    v.resize(R + 1 + 9 + 512);
    const double s = double(0.1f * 0.1f);
    for (int k = 0; k < 256; k++) {
        const double d = double(k) / 255.0, e = exp(-(d * d) / s);
        v[R + 10 + k] = float(1.0 - e);
        v[R + 10 + 256 + k] = float(e * 16777216.0);          // 2^24 per factor of a weight (eppm_device.cuh: kTolWeightBias): exp(-100) stays a normal float
    }
#endif
}
void host_wmf_lut(std::vector<float>& v)
{
    v.resize(kWmfRadius + 1);
    const float s = kWmfRadius * 1.0f;
    for (int i = 0; i <= kWmfRadius; i++) v[i] = expf(-float(i * i) / (s * s));
}
void host_blf_lut(std::vector<float>& v)
{
    v.resize(kBlfRadius + 1);
    for (int i = 0; i <= kBlfRadius; i++) v[i] = expf(-float(i * i) / float(5 * 5));
}

// bao_pyr_init_dim (maxDepth overload), basic/bao_basic.h:196-211; BAO_FLOAT is double (:56)
int pyr_init_dim(int* arrH, int* arrW, int h, int w, int maxDepth, double ratio)
{
    int n = maxDepth <= 0 ? 1 : maxDepth;
    arrH[0] = h; arrW[0] = w;
    for (int i = 1; i < n; i++) {
        arrH[i] = int(double(h) * pow(ratio, i));
        arrW[i] = int(double(w) * pow(ratio, i));
    }
    return n;
}
int upload_lut(float** dst, const std::vector<float>& v)
{
    HIPCHK(hipMalloc(dst, v.size() * sizeof(float)));
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return EPPM_OK;
}

// ---------------------------------------------------------------------------------------------------
// DeltaTab (eppm_device.cuh): f(d) by table for d = the L-inf distance of two unorm8 texels
// ---------------------------------------------------------------------------------------------------
const DeltaIndex& delta_index()
{
    static DeltaIndex D;
    static std::once_flag once;
    std::call_once(once, [] {
        float g[256];
        for (int k = 0; k < 256; k++) g[k] = (float)k / 255.0f;            // unorm8 (cudaReadModeNormalizedFloat)
        uint32_t lo[256], hi[256];
        for (int k = 0; k < 256; k++) { lo[k] = 0xffffffffu; hi[k] = 0u; }
        for (int a = 0; a < 256; a++)
            for (int b = 0; b < 256; b++) {
                const float d = fabsf(g[a] - g[b]);
                uint32_t bits;
                memcpy(&bits, &d, 4);
                const int kd = abs(a - b);
                if (bits < lo[kd]) lo[kd] = bits;
                if (bits > hi[kd]) hi[kd] = bits;
            }
        uint32_t off = 0;
        for (int kd = 0; kd < 256; kd++) {
            D.t1[kd] = (int32_t)(off * 4u - (lo[kd] << 2));                  // entry of d: (bits(d) << 2) + t1[kd], modulo 2^32
            for (uint32_t bits = lo[kd]; bits <= hi[kd]; bits++) {
                float d;
                memcpy(&d, &bits, 4);
                D.dval.push_back(d);
            }
            off += hi[kd] - lo[kd] + 1;
        }
    });
    return D;
}

int upload_lut_delta(float** dst, const std::vector<float>& head, int which)
{
    const DeltaIndex& D = delta_index();
    if ((int)D.dval.size() > kDeltaSlots) return set_err(EPPM_ERR_STATE, "delta table needs %zu slots, the kernels hold %d", D.dval.size(), kDeltaSlots);
    std::vector<float> v(head);
    v.resize(head.size() + 256 + kDeltaSlots, 0.0f);
    memcpy(&v[head.size()], D.t1, sizeof(D.t1));
    memcpy(&v[head.size() + 256], D.dval.data(), D.dval.size() * sizeof(float));
    CHK(upload_lut(dst, v));
    launch_delta_values(*dst + head.size() + 256, (int)D.dval.size(), which, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return EPPM_OK;
}

// the three look-up tables as the kernels read them
int upload_pm_lut(float** dst, int R)
{
    std::vector<float> v;
    host_pm_lut(R, v);
#ifdef EPPM_TOL
    return upload_lut(dst, v);                  // gs, cn, td, ta: the tolerance library's patch term reads no DeltaTab
#else
    return upload_lut_delta(dst, v, 0);         // gs, cn, then 1 - exp(-d^2 / LAMBDA_AD^2) by table
#endif
}
int upload_wmf_lut(float** dst)
{
    std::vector<float> v;
    host_wmf_lut(v);
    return upload_lut_delta(dst, v, 1);         // g[0..4], then exp(-d^2 / WMF_SIG_R^2) by table
}
int upload_blf_lut(float** dst)
{
    std::vector<float> v;
    host_blf_lut(v);
    return upload_lut_delta(dst, v, 1);         // g[0..10], then exp(-d^2 / POSTPROC_BLF_SIG_R^2) by table (the same constant)
}
