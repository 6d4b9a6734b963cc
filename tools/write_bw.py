"""HBM write ceiling probe: time hipMemsetD32Async / hipMemcpyDtoD on 1.5 GB buffers (driver kernels)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
n = 1536 * 1024 * 1024
a, b = ctypes.c_void_p(), ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(a), ctypes.c_size_t(n)) == 0
assert hip.hipMalloc(ctypes.byref(b), ctypes.c_size_t(n)) == 0
def t(fn, reps=10):
    fn(); hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    hip.hipDeviceSynchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: hip.hipMemsetD32Async(a, 0x3f000000, ctypes.c_size_t(n // 4), None))
print("memset 1.5 GiB: %.3f ms = %.2f TB/s written" % (dt * 1e3, n / dt / 1e12))
dt = t(lambda: hip.hipMemcpyAsync(b, a, ctypes.c_size_t(n), 3, None))
print("copy   1.5 GiB: %.3f ms = %.2f TB/s read + %.2f TB/s written" % (dt * 1e3, n / dt / 1e12, n / dt / 1e12))
