// Kernel launch helper with optional per-kernel hipEvent timing (mgs_profile_*).
#pragma once
#include <hip/hip_runtime.h>
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

template <typename K, typename... A>
inline void launch_smem(const char* name, K kernel, dim3 grid, dim3 block, size_t smem,
                        hipStream_t st, A... args) {
  (void)hipGetLastError();        // whatever is pending on this thread is not ours
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
