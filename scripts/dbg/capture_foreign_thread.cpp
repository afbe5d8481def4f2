// Diagnostic (not product code): which HIP calls made from ANOTHER thread fail while a stream capture is in progress, per
// capture mode.  torch's ProcessGroupNCCL watchdog / heartbeat threads make such calls (event query / destroy / record,
// stream query) and wrap each in C10_HIP_CHECK: an error there is an uncaught exception in that thread -> abort().
// Build: hipcc --offload-arch=gfx950 -O1 -pthread scripts/dbg/capture_foreign_thread.cpp -o scripts/dbg/capture_foreign_thread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <unistd.h>
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
__global__ void slow(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0000001f + 1e-9f; p[threadIdx.x] = v; }
static const char* modeName(int m) { return m == hipStreamCaptureModeGlobal ? "global" : m == hipStreamCaptureModeThreadLocal ? "thread_local" : "relaxed"; }
#define TRY(x) do { hipError_t e_ = (x); printf("    %-58s -> %d %s\n", #x, (int)e_, e_ == hipSuccess ? "" : hipGetErrorString(e_)); (void)hipGetLastError(); } while (0)

int main() {
  float *p, *q; hipMalloc(&p, 4096); hipMalloc(&q, 4096); hipMemset(p, 0, 4096); hipMemset(q, 0, 4096);
  hipStream_t cap, other; hipStreamCreateWithFlags(&cap, hipStreamNonBlocking); hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
  for (int mode : {(int)hipStreamCaptureModeThreadLocal, (int)hipStreamCaptureModeRelaxed, (int)hipStreamCaptureModeGlobal}) {
    printf("capture mode %s\n", modeName(mode)); fflush(stdout);
    hipEvent_t done_ev, busy_ev, tmp; hipEventCreateWithFlags(&done_ev, hipEventDisableTiming); hipEventCreateWithFlags(&busy_ev, hipEventDisableTiming);
    hipEventRecord(done_ev, other); hipStreamSynchronize(other);
    slow<<<1, 64, 0, other>>>(q, 200000000); hipEventRecord(busy_ev, other);      // an event that is NOT ready yet
    hipError_t rc = hipStreamBeginCapture(cap, (hipStreamCaptureMode)mode);
    printf("  begin -> %d\n", (int)rc);
    k<<<1, 64, 0, cap>>>(p);
    std::thread th([&] {
      printf("  calls from a second thread during the capture:\n");
      TRY(hipEventQuery(done_ev));
      TRY(hipEventQuery(busy_ev));
      TRY(hipStreamQuery(other));
      TRY(hipEventCreateWithFlags(&tmp, hipEventDisableTiming));
      TRY(hipEventRecord(tmp, other));
      TRY(hipStreamWaitEvent(other, tmp, 0));
      TRY(hipEventDestroy(tmp));
      TRY(hipEventSynchronize(done_ev));
      TRY(hipGetLastError());
      float* m = nullptr;
      TRY(hipMalloc(&m, 1 << 20));
      TRY(hipMemsetAsync(q + 512, 0, 256, other));
      if (m) TRY(hipFree(m));
      fflush(stdout);
    });
    th.join();
    k<<<1, 64, 0, cap>>>(p);
    hipGraph_t g = nullptr; rc = hipStreamEndCapture(cap, &g);
    printf("  end capture -> %d %s\n", (int)rc, rc == hipSuccess ? "" : hipGetErrorString(rc)); (void)hipGetLastError();
    hipDeviceSynchronize();
    if (g) hipGraphDestroy(g);
    fflush(stdout);
  }
  return 0;
}
