/* LD_PRELOAD shim for tests/test_gpu_letterbox_no_host_sync.py: records, in call order, which HIP runtime entry points the process
 * makes - kernel launches, the calls that make the host WAIT for the device, and copies towards the host - and forwards each to the real
 * runtime (dlsym RTLD_NEXT).  Test infrastructure only: nothing in the product links or loads it.
 *   vdf_trace_reset()            forget everything recorded so far
 *   vdf_trace_count()            events since the reset
 *   vdf_trace_get(i)             event i: one of the VDF_TRACE_* codes below */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <stddef.h>
#include <stdint.h>

enum {
    VDF_TRACE_LAUNCH = 1,        /* hipLaunchKernel / hipModuleLaunchKernel / hipExtModuleLaunchKernel */
    VDF_TRACE_STREAM_SYNC = 2,   /* hipStreamSynchronize */
    VDF_TRACE_EVENT_SYNC = 3,    /* hipEventSynchronize */
    VDF_TRACE_DEVICE_SYNC = 4,   /* hipDeviceSynchronize */
    VDF_TRACE_MEMCPY_SYNC = 5,   /* hipMemcpy / hipMemcpyDtoH (blocking) */
    VDF_TRACE_MEMCPY_D2H = 6,    /* hipMemcpyAsync towards the host (kind DeviceToHost or Default) */
    VDF_TRACE_MEMCPY_OTHER = 7,  /* hipMemcpyAsync, any other direction */
    VDF_TRACE_MEMSET = 8,        /* hipMemsetAsync */
    VDF_TRACE_STREAM_QUERY = 9,  /* hipStreamQuery / hipEventQuery (polling is waiting too) */
};

#define CAP 65536
static int g_log[CAP];
static volatile int g_n = 0;
static void note(int code)
{
    int at = __sync_fetch_and_add(&g_n, 1);
    if (at < CAP) g_log[at] = code;
}
void vdf_trace_reset(void) { g_n = 0; }
int vdf_trace_count(void) { return g_n < CAP ? g_n : CAP; }
int vdf_trace_get(int i) { return i >= 0 && i < CAP && i < g_n ? g_log[i] : 0; }

typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef struct { uint32_t x, y, z; } dim3_t;

/* The real runtime: the next definition in the global scope, or - Python loads torch's bundled libamdhip64 into a LOCAL scope that
 * RTLD_NEXT does not search - the handle of whichever libamdhip64 the process has already mapped. */
static int find_hip(struct dl_phdr_info *info, size_t size, void *out)
{
    (void)size;
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) {
        *(void **)out = dlopen(info->dlpi_name, RTLD_NOLOAD | RTLD_LAZY);
        return *(void **)out != NULL;
    }
    return 0;
}
static void *real_sym(const char *name)
{
    void *p = dlsym(RTLD_NEXT, name);
    if (!p) {
        static void *hip = NULL;
        if (!hip) dl_iterate_phdr(find_hip, &hip);
        if (hip) p = dlsym(hip, name);
    }
    return p;
}

#define REAL(name, ret, ...)                               \
    typedef ret (*fn_t)(__VA_ARGS__);                      \
    static fn_t real = NULL;                               \
    if (!real) real = (fn_t)real_sym(#name);               \
    if (!real) return 1 /* hipErrorInvalidValue */

hipError_t hipLaunchKernel(const void *f, dim3_t grid, dim3_t block, void **args, size_t shmem, hipStream_t s)
{
    REAL(hipLaunchKernel, hipError_t, const void *, dim3_t, dim3_t, void **, size_t, hipStream_t);
    note(VDF_TRACE_LAUNCH);
    return real(f, grid, block, args, shmem, s);
}
hipError_t hipModuleLaunchKernel(void *f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shmem,
                                 hipStream_t s, void **params, void **extra)
{
    REAL(hipModuleLaunchKernel, hipError_t, void *, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void **, void **);
    note(VDF_TRACE_LAUNCH);
    return real(f, gx, gy, gz, bx, by, bz, shmem, s, params, extra);
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    REAL(hipStreamSynchronize, hipError_t, hipStream_t);
    note(VDF_TRACE_STREAM_SYNC);
    return real(s);
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    REAL(hipEventSynchronize, hipError_t, hipEvent_t);
    note(VDF_TRACE_EVENT_SYNC);
    return real(e);
}
hipError_t hipDeviceSynchronize(void)
{
    REAL(hipDeviceSynchronize, hipError_t, void);
    note(VDF_TRACE_DEVICE_SYNC);
    return real();
}
hipError_t hipStreamQuery(hipStream_t s)
{
    REAL(hipStreamQuery, hipError_t, hipStream_t);
    note(VDF_TRACE_STREAM_QUERY);
    return real(s);
}
hipError_t hipEventQuery(hipEvent_t e)
{
    REAL(hipEventQuery, hipError_t, hipEvent_t);
    note(VDF_TRACE_STREAM_QUERY);
    return real(e);
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, int kind)
{
    REAL(hipMemcpy, hipError_t, void *, const void *, size_t, int);
    note(VDF_TRACE_MEMCPY_SYNC);
    return real(dst, src, n, kind);
}
hipError_t hipMemcpyDtoH(void *dst, void *src, size_t n)
{
    REAL(hipMemcpyDtoH, hipError_t, void *, void *, size_t);
    note(VDF_TRACE_MEMCPY_SYNC);
    return real(dst, src, n);
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, hipStream_t s)
{
    REAL(hipMemcpyAsync, hipError_t, void *, const void *, size_t, int, hipStream_t);
    note(kind == 2 /* hipMemcpyDeviceToHost */ || kind == 4 /* hipMemcpyDefault */ ? VDF_TRACE_MEMCPY_D2H : VDF_TRACE_MEMCPY_OTHER);
    return real(dst, src, n, kind, s);
}
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s)
{
    REAL(hipMemsetAsync, hipError_t, void *, int, size_t, hipStream_t);
    note(VDF_TRACE_MEMSET);
    return real(dst, v, n, s);
}
