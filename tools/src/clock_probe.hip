// Developer probe: the rate of wall_clock64() (the clock the bounded device-side waits count in) against the host's clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void read_clock(unsigned long long *out) { *out = wall_clock64(); }
int main() {
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  unsigned long long *d = nullptr, a = 0, b = 0;
  hipMalloc(&d, 8);
  read_clock<<<1, 1>>>(d); hipMemcpy(&a, d, 8, hipMemcpyDeviceToHost);
  const auto t0 = std::chrono::steady_clock::now();
  std::this_thread::sleep_for(std::chrono::milliseconds(500));
  read_clock<<<1, 1>>>(d); hipMemcpy(&b, d, 8, hipMemcpyDeviceToHost);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("hipDeviceAttributeWallClockRate = %d kHz; measured %.1f MHz (%llu ticks in %.3f s)\n", rate_khz, (double)(b - a) / dt / 1e6, b - a, dt);
  return 0;
}
