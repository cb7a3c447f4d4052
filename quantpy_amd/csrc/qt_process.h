// Process-tomography device state and kernels (reference quantpy/tomography/process.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qt {

struct ProcessState {
  void release() {}
};

}  // namespace qt
