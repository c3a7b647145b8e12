// Kernel launch helper with optional per-kernel hipEvent timing (mgs_profile_*).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>

namespace mgs {

bool profile_on();
void profile_push(const char* name, hipEvent_t a, hipEvent_t b);

// Launch status of THIS library's kernels on the calling thread.  hipGetLastError() is sticky per thread
// and shared with every other HIP user in the process (PyTorch, RCCL): an error left behind by someone
// else must not be reported as a failed launch of ours, and a failed launch of ours is reported with the
// kernel's name and HIP's message (stderr) instead of a bare status.
hipError_t& launch_error_slot();
inline void note_launch(const char* name, hipError_t e) {
  if (e != hipSuccess && launch_error_slot() == hipSuccess) {
    launch_error_slot() = e;
    fprintf(stderr, "monogs_raster: launch of %s failed: %s\n", name, hipGetErrorString(e));
  }
}
// status of the launches since the last call (and reset): what the C-ABI entry points return
inline bool launches_ok() {
  const bool ok = launch_error_slot() == hipSuccess;
  launch_error_slot() = hipSuccess;
  return ok;
}

// An error another HIP user of this thread (PyTorch, RCCL) left pending is not ours to return - but it is not
// ours to swallow silently either: hipGetLastError() clears it, so it is named on stderr (the first few times).
inline void drain_foreign_error() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return;
  static std::atomic<int> said{0};
  if (said.fetch_add(1) < 4)
    fprintf(stderr, "monogs_raster: a HIP error was already pending on this thread before a launch of ours "
                    "(left by another HIP user, cleared here): %s\n", hipGetErrorString(e));
}

// hipMemsetAsync / hipMemcpyAsync of the entry points go through the same per-thread slot and message
inline bool hip_ok(const char* what, hipError_t e) {
  note_launch(what, e);
  return e == hipSuccess;
}

template <typename K, typename... A>
inline void launch_smem(const char* name, K kernel, dim3 grid, dim3 block, size_t smem,
                        hipStream_t st, A... args) {
  drain_foreign_error();
  if (profile_on()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, st);
    hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
    note_launch(name, hipGetLastError());
    (void)hipEventRecord(b, st);
    profile_push(name, a, b);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
    note_launch(name, hipGetLastError());
  }
}

template <typename K, typename... A>
inline void launch(const char* name, K kernel, dim3 grid, dim3 block, hipStream_t st, A... args) {
  launch_smem(name, kernel, grid, block, 0, st, args...);
}

}  // namespace mgs
