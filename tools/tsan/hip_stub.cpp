// Stand-in for the HIP runtime, for the ThreadSanitizer build of the library's HOST concurrency code
// (tests/test_host_tsan.py): `devices` are plain host memory, streams run synchronously, events are empty.
// Only what csrc/api.hip calls.  Never linked into the product.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

static thread_local int t_dev = 0;
static std::atomic<long> g_live_allocs{0};
static int stub_ndev() { const char *v = getenv("EIP_STUB_NDEV"); return v ? atoi(v) : 2; }

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = stub_ndev(); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= stub_ndev()) return hipErrorInvalidDevice; t_dev = d; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (!*p) return hipErrorOutOfMemory; g_live_allocs++; return hipSuccess; }
hipError_t hipFree(void *p) { if (p) { free(p); g_live_allocs--; } return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { memset(a, 0, sizeof *a); a->type = hipMemoryTypeHost; return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *, dim3 *, size_t *, hipStream_t *) { return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipUnregisterFatBinary(void **) {}
long eip_stub_live_allocs(void) { return g_live_allocs.load(); }
}
