// Diagnosis aid: a one-thread kernel that records the order in which it ran (ticket from a device counter) and the device clock.
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(int* counter, long long* slots, int i) {
  int t = atomicAdd(counter, 1);
  slots[2 * i] = t;
  slots[2 * i + 1] = (long long)wall_clock64();
}
extern "C" int dbg_stamp(int* counter, long long* slots, int i, hipStream_t stream) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream, counter, slots, i);
  return (int)hipGetLastError();
}
