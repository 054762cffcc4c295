// Host-side helpers for the C-ABI launchers: error text + launch check.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/abcnet_hip.h"

int abc_fail(int code, const char* msg);     // records msg, returns code
int abc_check_launch(const char* what);      // hipGetLastError -> ABC_ELAUNCH
