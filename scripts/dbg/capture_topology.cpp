// Diagnostic (not product code): which capture topologies around a hand-made graph-external event-record node survive
// hipStreamEndCapture on this runtime.  One topology per process (argv[1]) so a host-side crash in one does not hide the
// others.  Build: hipcc --offload-arch=gfx950 -O1 scripts/dbg/capture_topology.cpp -o scripts/dbg/capture_topology
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %d (%s)\n", #x, (int)e_, hipGetErrorString(e_)); fflush(stdout); } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }

static void info(const char* tag, hipStream_t s) {
  hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t n = 0;
  hipError_t rc = hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n);
  printf("  [%s] rc=%d status=%d id=%llu graph=%p ndeps=%zu:", tag, (int)rc, (int)st, id, (void*)g, n);
  for (size_t i = 0; i < n; ++i) printf(" %p", (void*)deps[i]);
  printf("\n"); fflush(stdout);
}
// how=0: node + SetCaptureDependencies (product code today); how=1: node left as a leaf (frontier untouched);
// how=2: hipEventRecordWithFlags(External); how=3: node + AddCaptureDependencies
static void ext_record(hipEvent_t ev, hipStream_t s, int how) {
  if (how == 2) { CK(hipEventRecordWithFlags(ev, s, hipEventRecordExternal)); return; }
  hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t n = 0;
  CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &n));
  hipGraphNode_t node;
  CK(hipGraphAddEventRecordNode(&node, g, deps, n, ev));
  printf("  ext node %p with %zu deps\n", (void*)node, n);
  if (how == 0) CK(hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies));
  if (how == 3) CK(hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamAddCaptureDependencies));
}
static void fork_to(hipStream_t from, hipStream_t to) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventRecord(e, from)); CK(hipStreamWaitEvent(to, e, 0)); }

int main(int argc, char** argv) {
  int topo = argc > 1 ? atoi(argv[1]) : 0, how = argc > 2 ? atoi(argv[2]) : 0;
  printf("topology %d, external-record form %d\n", topo, how); fflush(stdout);
  float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  hipStream_t mainS, side; CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t ext[3]; for (auto& e : ext) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipStream_t outside; CK(hipStreamCreateWithFlags(&outside, hipStreamNonBlocking));
  CK(hipStreamBeginCapture(mainS, hipStreamCaptureModeThreadLocal));
  if (topo == 3 || topo == 4) {
    // what round 1's first data-parallel capture did at every bucket: cur.wait_stream(st) for EVERY auxiliary stream,
    // also for one that had not joined the capture (torch: event.record(st) eagerly, then cur.wait_event(event))
    if (topo == 4) k<<<1, 64, 0, outside>>>(p + 256);        // ... with eager work pending on it
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, outside));
    info("outside stream", outside);
    printf("  capturing stream waits on an event recorded OUTSIDE the capture\n"); fflush(stdout);
    CK(hipStreamWaitEvent(mainS, e, 0));
    info("main after foreign wait", mainS);
  }
  if (topo == 6 || topo == 7) {
    // two NON-origin streams that wait on each other: A waits on B's event and B on A's.  hip::Stream::EndCapture()
    // recurses over parallelCaptureStreams_ (every non-origin waiter is appended to the waited stream's list), so mutual
    // membership never terminates.  topo 7: the same dependencies added by hand instead (no hipStreamWaitEvent).
    hipStream_t A = side, B = outside;
    k<<<1, 64, 0, mainS>>>(p);
    fork_to(mainS, A); fork_to(mainS, B);
    k<<<1, 64, 0, A>>>(p + 64); k<<<1, 64, 0, B>>>(p + 128);
    auto join = [&](hipStream_t waiter, hipStream_t waited) {
      if (topo == 6) { fork_to(waited, waiter); return; }
      hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t n = 0;
      CK(hipStreamGetCaptureInfo_v2(waited, &st, &id, &g, &deps, &n));
      CK(hipStreamUpdateCaptureDependencies(waiter, const_cast<hipGraphNode_t*>(deps), n, hipStreamAddCaptureDependencies));
    };
    join(B, A); k<<<1, 64, 0, B>>>(p + 128);
    join(A, B); k<<<1, 64, 0, A>>>(p + 64);
    fork_to(A, mainS); fork_to(B, mainS);
    k<<<1, 64, 0, mainS>>>(p);
    info("main before end", mainS);
  } else
  for (int b = 0; b < 3; ++b) {
    k<<<1, 64, 0, mainS>>>(p);
    fork_to(mainS, side);                       // side.wait_stream(main)
    k<<<1, 64, 0, side>>>(p + 64);
    if (topo == 0) {                            // edges only INTO the side branch; event node on the side branch
      ext_record(ext[b], side, how);
      info("side after ext", side);
    } else {                                    // join side -> main at every bucket, event node on main
      fork_to(side, mainS);
      info("main after join", mainS);
      ext_record(ext[b], mainS, how);
      info("main after ext", mainS);
      if (topo == 2) { k<<<1, 64, 0, mainS>>>(p + 128); info("main after a kernel behind ext", mainS); }
    }
  }
  if (topo == 0) fork_to(side, mainS);          // the one rejoin at the end
  hipGraph_t g = nullptr;
  printf("  ending capture...\n"); fflush(stdout);
  hipError_t rc = hipStreamEndCapture(mainS, &g);
  printf("  hipStreamEndCapture -> %d (%s) graph=%p\n", (int)rc, hipGetErrorString(rc), (void*)g); fflush(stdout);
  if (rc == hipSuccess && g) {
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphExec_t ex; rc = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    printf("  nodes=%zu instantiate -> %d\n", nn, (int)rc);
    if (rc == hipSuccess) { CK(hipGraphLaunch(ex, mainS)); CK(hipStreamSynchronize(mainS)); for (auto& e : ext) printf("  ext event query -> %d\n", (int)hipEventQuery(e)); }
    float h[192]; CK(hipMemcpy(h, p, sizeof h, hipMemcpyDeviceToHost)); printf("  p[0]=%g p[64]=%g p[128]=%g\n", h[0], h[64], h[128]);
  }
  printf("topology %d form %d: done\n", topo, how);
  return 0;
}
