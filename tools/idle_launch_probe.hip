// Where does a multi-millisecond first-launch delay come from?  Times a trivial kernel launch
// (launch + sync) after: nothing, host idle periods, a hipMalloc, a hipMalloc + hipMemset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
__global__ void k_touch(unsigned* p) { p[threadIdx.x] = threadIdx.x; }
static double now_ms() { struct timeval t; gettimeofday(&t, 0); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }
static unsigned* d;
static void probe(const char* what) {
  double a = now_ms();
  hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, 0, d);
  (void)hipDeviceSynchronize();
  printf("%-40s launch+sync %.3f ms\n", what, now_ms() - a);
}
int main() {
  (void)hipMalloc((void**)&d, 4096);
  probe("first launch of the process");
  probe("back to back");
  for (int ms : {1, 10, 50, 200, 1000}) { usleep(ms * 1000); char b[64]; snprintf(b, 64, "after %d ms host idle", ms); probe(b); }
  void* big = nullptr;
  (void)hipMalloc(&big, 8 << 20); probe("after hipMalloc 8 MB");
  (void)hipMemset(big, 0, 8 << 20); probe("after hipMemset 8 MB");
  void* big2 = nullptr;
  (void)hipMalloc(&big2, 256 << 20); probe("after hipMalloc 256 MB");
  (void)hipFree(big2); probe("after hipFree 256 MB");
  void* h = malloc(32 << 20); memset(h, 1, 32 << 20);
  (void)hipMemcpy(big, h, 8 << 20, hipMemcpyHostToDevice); probe("after pageable H2D 8 MB");
  (void)hipMemcpy(h, big, 8 << 20, hipMemcpyDeviceToHost); probe("after pageable D2H 8 MB");
  return 0;
}
