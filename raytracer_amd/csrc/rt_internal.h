/*
 * rt_internal.h - what the translation units of librt_hip.so share besides the device layout
 * (rt_device.h): the error message behind rt_hip_last_error() and the per-device set-up hooks.
 * Not part of the C-ABI.
 */
#pragma once

/* printf-style; the message rt_hip_last_error() returns on the calling thread (rt_api.hip) */
void rt_set_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

/* rt_tone.hip: resolve the device address of the static scratch block of rt_hip_tone_static for the
 * current device (once per device; later calls return the cached result).  0 or a negative hipError_t. */
int rt_tone_static_init(void);

/* rt_jit.cpp: the fused trace kernels compiled at run time (hiprtc) for the shape of a small scene; the handles are
 * hipFunction_t of a module loaded on the current device */
struct DevScene;
struct RtJitKernels { void *primary, *bounce, *subtree; };
const RtJitKernels *rt_jit_get(const DevScene &S, bool fast, bool start, bool wait);
int rt_jit_status(const DevScene &S, bool fast, char *log, size_t log_bytes, double *seconds);
