// tests/devemu/emu_runtime.cpp — state of the lane-serial grid emulation (see hip/hip_runtime.h). TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>
namespace hipemu {
thread_local dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
thread_local hipError_t last_error = hipSuccess;
}
