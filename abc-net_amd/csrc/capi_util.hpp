// Host-side helpers for the C-ABI launchers: error text + launch check.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/abcnet_hip.h"

int abc_fail(int code, const char* msg);     // records msg, returns code
int abc_check_launch(const char* what);      // hipGetLastError -> ABC_ELAUNCH
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per (kernel, device) --
// `done` is the caller's per-kernel bitmask over device ordinals -- and report a refusal instead of failing later at launch
int abc_allow_lds(const void* fn, int bytes, unsigned long long* done);
// bytes per element of an abc_dtype
inline int abc_dsize(int dtype) { return dtype == ABC_F32 ? 4 : (dtype == ABC_BF16 ? 2 : 1); }
