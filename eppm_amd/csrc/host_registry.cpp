// host_registry.cpp -- eppm_host_register / _unregister / _is_registered / _alloc / _free: host_registry.h bound to the HIP runtime.
#include "api_internal.h"

using namespace eppm;

static int pin_hip(void* p, size_t bytes) { return (int)hipHostRegister(p, bytes, hipHostRegisterPortable); }
static int unpin_hip(void* p) { return (int)hipHostUnregister(p); }
static HostRegistry g_host_registry(pin_hip, unpin_hip);
HostRegistry* eppm::default_host_registry() { return &g_host_registry; }

extern "C" int eppm_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return set_err(EPPM_ERR_ARG, "eppm_host_register: NULL or empty block");
    int hip = 0;
    switch (g_host_registry.add(p, bytes, &hip)) {
        case HostRegistry::kOk: return EPPM_OK;
        case HostRegistry::kBeingFreed: return set_err(EPPM_ERR_STATE, "eppm_host_register: the block is being released by eppm_host_free");
        default:
            (void)hipGetLastError();
            return set_err(EPPM_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString((hipError_t)hip));
    }
}
extern "C" int eppm_host_unregister(void* p)
{
    int hip = 0;
    switch (g_host_registry.remove(p, &hip)) {
        case HostRegistry::kOk: return EPPM_OK;
        case HostRegistry::kNotRegistered: return set_err(EPPM_ERR_ARG, "eppm_host_unregister: not a block registered with eppm_host_register");
        case HostRegistry::kOwnedBlock: return set_err(EPPM_ERR_ARG, "eppm_host_unregister: a block from eppm_host_alloc is released with eppm_host_free");
        case HostRegistry::kBeingUnregistered: return set_err(EPPM_ERR_ARG, "eppm_host_unregister: the block's last registration is being given up by another call");
        case HostRegistry::kBusy: return set_err(EPPM_ERR_STATE, "eppm_host_unregister: a transfer is still in flight on the block (eppm_compute_end pending?)");
        default:
            (void)hipGetLastError();
            return set_err(EPPM_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString((hipError_t)hip));
    }
}
extern "C" int eppm_host_is_registered(const void* p, size_t bytes) { return g_host_registry.registered(p, bytes) ? 1 : 0; }
extern "C" int eppm_host_alloc(void** p, size_t bytes)
{
    if (!p || !bytes) return set_err(EPPM_ERR_ARG, "eppm_host_alloc: NULL or empty block");
    HIPCHK(hipHostMalloc(p, bytes, hipHostMallocPortable));
    g_host_registry.add_owned(*p, bytes);
    return EPPM_OK;
}
extern "C" int eppm_host_free(void* p)
{
    if (!p) return EPPM_OK;
    switch (g_host_registry.remove_owned(p)) {
        case HostRegistry::kOk: break;
        case HostRegistry::kBusy: return set_err(EPPM_ERR_STATE, "eppm_host_free: a transfer is still in flight on the block (eppm_compute_end pending?)");
        default: return set_err(EPPM_ERR_ARG, "eppm_host_free: not a block from eppm_host_alloc");
    }
    HIPCHK(hipHostFree(p));
    return EPPM_OK;
}
