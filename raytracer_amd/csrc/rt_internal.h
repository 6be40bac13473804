/*
 * rt_internal.h - what the translation units of librt_hip.so share besides the device layout
 * (rt_device.h): the error message behind rt_hip_last_error().
 * Not part of the C-ABI.
 */
#pragma once

/* printf-style; the message rt_hip_last_error() returns on the calling thread (rt_api.hip) */
void rt_set_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
