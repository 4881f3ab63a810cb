"""Raw hipEvent_t handles through ctypes (libamdhip64), for timing a kernel on the
stream it is launched on and for the vsv_profile_next_decode hook."""
import ctypes as C

_hip = None


def _lib():
    global _hip
    if _hip is None:
        import torch  # noqa: F401  (load torch's HIP runtime first; the SONAME below then resolves to it)
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    return _hip


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        rc = _lib().hipEventCreate(C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"hipEventCreate failed: {rc}")

    def record(self, stream_handle):
        rc = _lib().hipEventRecord(self.h, C.c_void_p(stream_handle))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed: {rc}")

    def synchronize(self):
        _lib().hipEventSynchronize(self.h)

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        rc = _lib().hipEventElapsedTime(C.byref(ms), self.h, end.h)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed: {rc}")
        return float(ms.value)

    def __del__(self):
        try:
            if self.h:
                _lib().hipEventDestroy(self.h)
        except Exception:
            pass
