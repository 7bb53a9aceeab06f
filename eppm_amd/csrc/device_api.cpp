// device_api.cpp -- device-memory plumbing of the C ABI (what a host program in another language needs beside the context calls).
#include "api_internal.h"

using namespace eppm;

// ---------------------------------------------------------------------------------------------------
// device-memory plumbing
// ---------------------------------------------------------------------------------------------------
extern "C" int eppm_device_count(int* n) { HIPCHK(hipGetDeviceCount(n)); return EPPM_OK; }
extern "C" int eppm_set_device(int d) { HIPCHK(hipSetDevice(d)); return EPPM_OK; }
extern "C" int eppm_malloc_device(void** p, size_t bytes) { HIPCHK(hipMalloc(p, bytes ? bytes : 1)); return EPPM_OK; }
extern "C" int eppm_malloc_pitched(void** p, size_t* pitch, size_t width_bytes, size_t rows) { HIPCHK(hipMallocPitch(p, pitch, width_bytes, rows)); return EPPM_OK; }
extern "C" int eppm_free_device(void* p) { HIPCHK(hipFree(p)); return EPPM_OK; }
extern "C" int eppm_memcpy_h2d(void* d, const void* s, size_t n) { HIPCHK(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return EPPM_OK; }
extern "C" int eppm_memcpy_d2h(void* d, const void* s, size_t n) { HIPCHK(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return EPPM_OK; }
extern "C" int eppm_memcpy2d_h2d(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t rows) { HIPCHK(hipMemcpy2D(d, dp, s, sp, wb, rows, hipMemcpyHostToDevice)); return EPPM_OK; }
extern "C" int eppm_memcpy2d_d2h(void* d, size_t dp, const void* s, size_t sp, size_t wb, size_t rows) { HIPCHK(hipMemcpy2D(d, dp, s, sp, wb, rows, hipMemcpyDeviceToHost)); return EPPM_OK; }
extern "C" int eppm_memset_device(void* p, int v, size_t n) { HIPCHK(hipMemset(p, v, n)); return EPPM_OK; }
extern "C" int eppm_device_synchronize(void) { HIPCHK(hipDeviceSynchronize()); return EPPM_OK; }
// PCI address of a device ("0000:c1:00.0", hipDeviceGetPCIBusId) and, from sysfs, the NUMA node its slot hangs off
extern "C" int eppm_device_pci_bus_id(int device, char* buf, size_t len)
{
    if (!buf || len < 13) return set_err(EPPM_ERR_ARG, "eppm_device_pci_bus_id: buffer of at least 13 bytes");
    HIPCHK(hipDeviceGetPCIBusId(buf, (int)len, device));
    for (char* q = buf; *q; q++) *q = (char)tolower((unsigned char)*q);          // sysfs spells the address in lower case
    return EPPM_OK;
}
// One host thread per GPU: binds the CALLING thread (and the threads it creates afterwards) to the CPUs of the NUMA node the
// device's PCIe slot belongs to, intersected with the CPUs the thread may run on now -- staging copies, the DMA descriptors and the
// launch path then stay on the socket next to the GPU.  numa_node = -1 / ncpus = 0 and no binding when sysfs does not say (a container
// without the topology, a single-node host): never an error.
extern "C" int eppm_bind_thread_to_device(int device, int* numa_node, int* ncpus)
{
    if (numa_node) *numa_node = -1;
    if (ncpus) *ncpus = 0;
    char bdf[32] = "";
    CHK(eppm_device_pci_bus_id(device, bdf, sizeof bdf));
    char path[128];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return EPPM_OK;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    cpu_set_t want, have;
    CPU_ZERO(&want);
    if (FILE* f = fopen(path, "r")) {          // "0-31,128-159"
        int a = 0, b = 0;
        for (;;) {
            if (fscanf(f, "%d", &a) != 1) break;
            b = a;
            int ch = fgetc(f);
            if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
            for (int k = a; k <= b && k < CPU_SETSIZE; k++) CPU_SET(k, &want);
            if (ch != ',') break;
        }
        fclose(f);
    }
    if (sched_getaffinity(0, sizeof have, &have) != 0) return EPPM_OK;
    CPU_AND(&want, &want, &have);
    const int n = CPU_COUNT(&want);
    if (n < 1 || sched_setaffinity(0, sizeof want, &want) != 0) return EPPM_OK;
    if (numa_node) *numa_node = node;
    if (ncpus) *ncpus = n;
    return EPPM_OK;
}
extern "C" int eppm_device_mem_info(size_t* free_bytes, size_t* total_bytes)
{
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return EPPM_OK;
}
