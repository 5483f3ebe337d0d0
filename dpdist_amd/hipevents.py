"""Stream-ordering events without the system-scope fence (ctypes on the HIP runtime that torch already loaded).

`torch.cuda.Event.record()` in the middle of a stream is a barrier packet with a SYSTEM-scope release on this runtime: the
compute stream idles ~20 us at every cross-stream hop (tools/event_cost.py).  An event created with
`hipEventDisableSystemFence` orders work between two streams of the SAME device with an agent-scope release only, which is all
a kernel on another stream of this GPU (a local RCCL kernel, a prefetched front end, a parallel branch of the backward) needs.
Such an event must NOT be used to publish device memory to the host or to another GPU.
"""
import ctypes

import torch

hipEventDisableTiming = 0x2
hipEventDisableSystemFence = 0x20000000
_hip = None


def _rt():
    global _hip
    if _hip is None:
        if not torch.cuda.is_available():
            raise RuntimeError("dpdist_amd.hipevents needs a GPU (HIP runtime)")
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = ctypes.CDLL(name)        # already mapped by torch: dlopen returns the same handle
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
    return _hip


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError_t %d" % (what, rc))


def _raw(stream):
    return ctypes.c_void_p((torch.cuda.current_stream() if stream is None else stream).cuda_stream)


class LightEvent:
    """Device-local ordering event: hipEventDisableTiming | hipEventDisableSystemFence (system_fence=True: plain)."""

    def __init__(self, system_fence=False):
        self._ev = ctypes.c_void_p()
        flags = hipEventDisableTiming | (0 if system_fence else hipEventDisableSystemFence)
        _check(_rt().hipEventCreateWithFlags(ctypes.byref(self._ev), flags), "hipEventCreateWithFlags")

    def record(self, stream=None):
        _check(_rt().hipEventRecord(self._ev, _raw(stream)), "hipEventRecord")

    def wait(self, stream=None):
        """make `stream` (default: current) wait for the work recorded in this event"""
        _check(_rt().hipStreamWaitEvent(_raw(stream), self._ev, 0), "hipStreamWaitEvent")

    def synchronize(self):
        _check(_rt().hipEventSynchronize(self._ev), "hipEventSynchronize")

    def __del__(self):
        try:
            if self._ev and _hip is not None:
                _hip.hipEventDestroy(self._ev)
        except Exception:
            pass
